// tools/microbench2.hip -- second round: LDS atomic flavours (f32/u32/u64/f64, random vs conflict-free),
// L1-resident gathers, and sorted-neighbour gathers. Build like microbench.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

enum { L_F32, L_U32, L_U64, L_F64, L_F32_RTN, L_PLAIN_RMW };

// PATTERN 0: random rows; 1: conflict-free (lane -> its own bank, row varies per iteration uniformly per wave)
template <int KIND, int PATTERN>
__global__ __launch_bounds__(1024) void k_lds(int rows, int iters, float *__restrict__ sink) {
    extern __shared__ unsigned long long lds64[];
    float *ldsf = reinterpret_cast<float *>(lds64);
    for (int i = threadIdx.x; i < rows * 2; i += blockDim.x) ldsf[i] = 0.f;
    __syncthreads();
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t mask = rows - 1;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        uint32_t h = mix(tid * 0x9E3779B9u + it * 0x85EBCA6Bu + 1u);
        uint32_t hw = mix((tid >> 6) * 0x9E3779B9u + it * 0x85EBCA6Bu + 1u);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t r;
            if (PATTERN == 0) r = mix(h + k * 0xC2B2AE35u) & mask;
            else r = ((mix(hw + k * 0xC2B2AE35u) & mask) & ~63u) | lane;  // 64 consecutive rows
            if constexpr (KIND == L_F32) atomicAdd(&ldsf[2 * r], 1.0f);
            else if constexpr (KIND == L_U32) atomicAdd(reinterpret_cast<unsigned int *>(ldsf) + 2 * r, 1u);
            else if constexpr (KIND == L_U64) atomicAdd(&lds64[r], 1ull);
            else if constexpr (KIND == L_F64) atomicAdd(reinterpret_cast<double *>(lds64) + r, 1.0);
            else if constexpr (KIND == L_F32_RTN) acc += atomicAdd(&ldsf[2 * r], 1.0f);
            else if constexpr (KIND == L_PLAIN_RMW) { float2 v = reinterpret_cast<float2 *>(ldsf)[r]; v.x += 1.f; v.y += 1.f; reinterpret_cast<float2 *>(ldsf)[r] = v; }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < rows * 2; i += blockDim.x) acc += ldsf[i];
    if (acc == 123.456f) sink[0] = acc;
}

// gathers: MODE 0 random rows in table; MODE 1 "sorted": lanes of a wave read rows within a window of W rows
template <int MODE>
__global__ __launch_bounds__(256) void k_gather(const float2 *__restrict__ table, uint32_t rowmask, uint32_t window,
                                                int iters, float *__restrict__ sink) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        uint32_t h = mix(tid * 0x9E3779B9u + it * 0x85EBCA6Bu + 1u);
        uint32_t hw = mix((tid >> 6) * 0x9E3779B9u + it * 0x85EBCA6Bu + 1u);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t r;
            if (MODE == 0) r = mix(h + k * 0xC2B2AE35u) & rowmask;
            else r = ((mix(hw + k * 0xC2B2AE35u) & rowmask) & ~(window - 1)) | (mix(h + k) & (window - 1));
            float2 v = table[r];
            acc += v.x + v.y;
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <typename Fn> static float time_ms(Fn fn, int reps = 3) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    fn();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a)); fn(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        best = std::min(best, ms);
    }
    CK(hipGetLastError());
    return best;
}

template <int KIND, int PATTERN> static void run_lds(const char *name, int rows, float *sink) {
    const int blocks = 1024, iters = 16;
    size_t shmem = (size_t)rows * 8;
    float ms = time_ms([&] { hipLaunchKernelGGL((k_lds<KIND, PATTERN>), dim3(blocks), dim3(1024), shmem, 0, rows, iters, sink); });
    double ops = (double)blocks * 1024 * iters * 8;
    printf("lds %-14s %-13s rows=%5d  %8.3f ms  %8.2f Gops/s  (%.2f ops/clk/CU @2.4GHz)\n", name, PATTERN ? "conflict-free" : "random", rows, ms,
           ops / ms / 1e6, ops / ms / 1e6 / (256 * 2.4));
    fflush(stdout);
}

int main() {
    float2 *table; float *sink;
    const size_t big = (size_t)256 << 20;
    CK(hipMalloc(&table, big)); CK(hipMalloc(&sink, 64)); CK(hipMemset(table, 0, big));
    printf("== LDS atomics ==\n");
    for (int rows : {4096, 16384}) {
        run_lds<L_F32, 0>("add_f32", rows, sink);
        run_lds<L_F32, 1>("add_f32", rows, sink);
        run_lds<L_F32_RTN, 0>("add_rtn_f32", rows, sink);
        run_lds<L_U32, 0>("add_u32", rows, sink);
        run_lds<L_U32, 1>("add_u32", rows, sink);
        run_lds<L_U64, 0>("add_u64", rows, sink);
        run_lds<L_U64, 1>("add_u64", rows, sink);
        run_lds<L_F64, 0>("add_f64", rows, sink);
        run_lds<L_PLAIN_RMW, 0>("plain rmw 8B", rows, sink);
        run_lds<L_PLAIN_RMW, 1>("plain rmw 8B", rows, sink);
    }
    printf("== gathers, 8-byte rows ==\n");
    for (size_t s : {(size_t)8 << 10, (size_t)16 << 10, (size_t)32 << 10, (size_t)64 << 10, (size_t)256 << 10, (size_t)2 << 20, (size_t)4 << 20, (size_t)8 << 20, (size_t)16 << 20, (size_t)32 << 20}) {
        const int blocks = 8192, iters = 8;
        uint32_t rows = (uint32_t)(s / 8);
        float ms = time_ms([&] { hipLaunchKernelGGL(k_gather<0>, dim3(blocks), dim3(256), 0, 0, table, rows - 1, 1u, iters, sink); });
        double ops = (double)blocks * 256 * iters * 8;
        printf("gather random      table %9.3f MB  %8.3f ms  %8.2f Gops/s\n", s / 1048576.0, ms, ops / ms / 1e6);
    }
    for (uint32_t window : {16u, 64u, 256u, 1024u}) {
        for (size_t s : {(size_t)4 << 20, (size_t)64 << 20}) {
            const int blocks = 8192, iters = 8;
            uint32_t rows = (uint32_t)(s / 8);
            float ms = time_ms([&] { hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(256), 0, 0, table, rows - 1, window, iters, sink); });
            double ops = (double)blocks * 256 * iters * 8;
            printf("gather wave-window %5u rows  table %6.1f MB  %8.3f ms  %8.2f Gops/s\n", window, s / 1048576.0, ms, ops / ms / 1e6);
        }
    }
    return 0;
}
