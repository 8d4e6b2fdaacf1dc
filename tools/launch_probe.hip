// Cost of launching workgroups that exit at once, by threads / dynamic LDS per workgroup (MI355X): tools/launch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
extern "C" __global__ __launch_bounds__(1024) void k_empty(const unsigned *limit, float *sink) {
    extern __shared__ double s_acc[];
    if (blockIdx.x >= limit[0]) return;
    s_acc[threadIdx.x] = 1.0;
    __syncthreads();
    sink[0] = (float)s_acc[0];
}
extern "C" __global__ __launch_bounds__(1024) void k_zero_flush(const unsigned *limit, float *out, int rows) {
    extern __shared__ double s_acc[];
    if (blockIdx.x >= limit[0]) return;
    for (int e = threadIdx.x; e < rows * 2; e += 1024) s_acc[e] = 0.0;
    __syncthreads();
    for (int e = threadIdx.x; e < rows * 2; e += 1024) out[(size_t)blockIdx.x * rows * 2 + e] = (float)s_acc[e];
}
int main() {
    unsigned *limit; float *sink, *out;
    CK(hipMalloc(&limit, 4)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&out, (size_t)2048 * 8192 * 2 * 4));
    CK(hipFuncSetAttribute((const void *)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void *)k_zero_flush, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    unsigned zero = 0, all = 1u << 30;
    for (int threads : {256, 1024}) for (int lds : {0, 65536, 131072}) for (int blocks : {256, 1293, 4096}) {
        CK(hipMemcpy(limit, &zero, 4, hipMemcpyHostToDevice));
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(threads), lds, 0, limit, sink);
        CK(hipEventRecord(a, 0));
        for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(threads), lds, 0, limit, sink);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("empty workgroups: threads %4d  lds %6d  blocks %5d : %7.1f us per launch\n", threads, lds, blocks, ms * 1000 / 20);
    }
    for (int blocks : {640, 1280}) {
        CK(hipMemcpy(limit, &all, 4, hipMemcpyHostToDevice));
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_zero_flush, dim3(blocks), dim3(1024), 131072, 0, limit, out, 8192);
        CK(hipEventRecord(a, 0));
        for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(k_zero_flush, dim3(blocks), dim3(1024), 131072, 0, limit, out, 8192);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("zero + flush of a 128 KiB image (8192 rows x 2 floats out): blocks %5d : %7.1f us per launch\n", blocks, ms * 1000 / 20);
    }
    return 0;
}
