// Read-stream rate by workgroup shape (the consume pass's): workgroups of 1024 threads with a 128 KiB (or 64 KiB) LDS reservation,
// each streaming a contiguous 1 MiB chunk of 16-byte items, UN loads in flight per thread, non-temporal or plain. tools/load_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <bool NT, int UN>
__global__ __launch_bounds__(1024) void k_load(const u32x4 *in, unsigned *out, int items_per_block) {
    extern __shared__ unsigned s_pad[];
    const u32x4 *p = in + (size_t)blockIdx.x * items_per_block;
    unsigned acc = 0;
    for (int p0 = threadIdx.x; p0 < items_per_block; p0 += 1024 * UN) {
        u32x4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int q = p0 + u * 1024;
            if (q < items_per_block) v[u] = NT ? __builtin_nontemporal_load(p + q) : p[q];
            else v[u] = u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345u) { s_pad[threadIdx.x] = acc; out[blockIdx.x] = s_pad[0]; }
}
int main() {
    const size_t bytes = (size_t)1 << 30;
    u32x4 *in; unsigned *out; CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, 1 << 20)); CK(hipMemset(in, 1, bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](auto kern, const char *name, int lds, int chunk_kib) -> int {
        CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        const int items = chunk_kib * 1024 / 16, blocks = (int)(bytes / ((size_t)chunk_kib * 1024));
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), lds, 0, in, out, items);
        CK(hipEventRecord(a, 0));
        for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), lds, 0, in, out, items);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-22s LDS %3d KiB, %4d KiB per workgroup: %7.1f GB/s\n", name, lds / 1024, chunk_kib, bytes * 5 / (ms * 1e-3) / 1e9);
        return 0;
    };
    for (int lds : {128 * 1024, 64 * 1024, 0}) for (int chunk : {1024, 256}) {
        if (run(k_load<true, 8>, "nt, 8 in flight", lds, chunk)) return 1;
        if (run(k_load<false, 8>, "plain, 8 in flight", lds, chunk)) return 1;
        if (run(k_load<true, 16>, "nt, 16 in flight", lds, chunk)) return 1;
        if (run(k_load<true, 4>, "nt, 4 in flight", lds, chunk)) return 1;
    }
    return 0;
}
