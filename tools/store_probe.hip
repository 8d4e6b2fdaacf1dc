// Write-stream rate by workgroup shape: 512-thread workgroups each storing `per_thread` x 16 B per thread (contiguous 1 KiB per
// wave instruction, like bin_scatter's write-out), non-temporal or plain, with a dynamic-LDS reservation that caps the number of
// resident workgroups per CU like the real kernel's staging buffer does. tools/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(512) void k_store(u32x4 *out, int per_thread) {
    extern __shared__ unsigned char s_pad[];
    u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    u32x4 *p = out + (size_t)blockIdx.x * 512 * per_thread + threadIdx.x;
    for (int k = 0; k < per_thread; ++k) {
        if (NT) __builtin_nontemporal_store(v, p + (size_t)k * 512);
        else p[(size_t)k * 512] = v;
    }
    if (per_thread < 0) s_pad[threadIdx.x] = 1;
}
struct Big { unsigned a[600]; };   // ~2.4 KB of kernel arguments, like LevelTable + BinPlan by value
template <bool NT>
__global__ __launch_bounds__(512) void k_store2d(Big big, u32x4 *out, int per_thread) {
    extern __shared__ unsigned char s_pad[];
    __shared__ unsigned s_x[516];
    u32x4 v = {threadIdx.x, blockIdx.x, big.a[blockIdx.y], 4u};
    u32x4 *p = out + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 512 * per_thread + threadIdx.x;
    for (int k = 0; k < per_thread; ++k) {
        if (NT) __builtin_nontemporal_store(v, p + (size_t)k * 512);
        else p[(size_t)k * 512] = v;
    }
    if (per_thread < 0) { s_pad[threadIdx.x] = 1; s_x[threadIdx.x] = 2; }
}
int main() {
    const size_t bytes = (size_t)1 << 30;
    u32x4 *out; CK(hipMalloc(&out, bytes));
    CK(hipFuncSetAttribute((const void *)k_store<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CK(hipFuncSetAttribute((const void *)k_store<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int nt = 1; nt >= 0; --nt) for (int per_thread : {8, 32, 128}) for (int lds : {0, 36 * 1024, 72 * 1024, 144 * 1024}) {
        const int blocks = (int)(bytes / ((size_t)512 * per_thread * 16));
        auto launch = [&]() {
            if (nt) hipLaunchKernelGGL(k_store<true>, dim3(blocks), dim3(512), lds, 0, out, per_thread);
            else hipLaunchKernelGGL(k_store<false>, dim3(blocks), dim3(512), lds, 0, out, per_thread);
        };
        for (int w = 0; w < 2; ++w) launch();
        CK(hipEventRecord(a, 0));
        for (int it = 0; it < 5; ++it) launch();
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%s stores, %3d x 16 B per thread (%5d KiB per workgroup), dynamic LDS %3d KiB: %7.1f GB/s\n", nt ? "nt   " : "plain",
               per_thread, 512 * per_thread * 16 / 1024, lds / 1024, bytes * 5 / (ms * 1e-3) / 1e9);
    }
    CK(hipFuncSetAttribute((const void *)k_store2d<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    Big big{};
    for (int rep = 0; rep < 2; ++rep) for (int ny : {15, 1}) {
        const int nx = ny == 15 ? 1024 : 15360;
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_store2d<true>, dim3(nx, ny), dim3(512), 70 * 1024, 0, big, out, 8);
        CK(hipEventRecord(a, 0));
        for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(k_store2d<true>, dim3(nx, ny), dim3(512), 70 * 1024, 0, big, out, 8);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("scatter-shaped: grid (%d, %d) x 512 threads, 8 x 16 B nt stores each, 70 KiB LDS, 2.4 KB kernarg: %7.1f us per launch = %7.1f GB/s\n",
               nx, ny, ms * 1000 / 5, (double)nx * ny * 65536 * 5 / (ms * 1e-3) / 1e9);
    }
    return 0;
}
