// tools/microbench.hip -- MI355X micro-measurements that size the hash-grid kernels' design choices:
// random 8-byte row gathers, scattered float atomics (global, by scope, by type), LDS atomics, streams.
// Build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <string>
#include <vector>

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

enum Mode {
    GATHER8, GATHER16, GATHER8_PAIRHASH,
    ATOM_F32, ATOM_F32_PAIR, ATOM_F32_PAIR_LANES, ATOM_U32, ATOM_U64, ATOM_F64, ATOM_F32_WG, ATOM_F32_WAVE,
    ATOM_F32_COALESCED, RMW_PLAIN, STORE8, ATOM_F32_RTN
};

// rows: number of 8-byte rows in the table (power of two). Each thread performs `iters` x 8 operations.
template <int MODE>
__global__ __launch_bounds__(256) void k_random(float2 *__restrict__ table, uint32_t rowmask, int iters,
                                                float *__restrict__ sink) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        uint32_t h = mix(tid * 0x9E3779B9u + it * 0x85EBCA6Bu + 1u);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t r = mix(h + k * 0xC2B2AE35u) & rowmask;
            if constexpr (MODE == GATHER8) {
                float2 v = table[r];
                acc += v.x + v.y;
            } else if constexpr (MODE == GATHER16) {
                float4 v = reinterpret_cast<const float4 *>(table)[r >> 1];
                acc += v.x + v.y + v.z + v.w;
            } else if constexpr (MODE == GATHER8_PAIRHASH) {
                // corners x, x+1 of a hashed level: rows r and r^1 (same 16 B) half of the time
                uint32_t r2 = (k & 1) ? (r ^ 1u) : r;
                float2 v = table[r2];
                acc += v.x + v.y;
            } else if constexpr (MODE == ATOM_F32) {
                unsafeAtomicAdd(reinterpret_cast<float *>(table) + 2 * r, 1.0f);
            } else if constexpr (MODE == ATOM_F32_PAIR) {
                unsafeAtomicAdd(reinterpret_cast<float *>(table) + 2 * r, 1.0f);
                unsafeAtomicAdd(reinterpret_cast<float *>(table) + 2 * r + 1, 1.0f);
            } else if constexpr (MODE == ATOM_F32_PAIR_LANES) {
                // lanes 2m, 2m+1 add the two features of the SAME row in one wave instruction
                uint32_t rr = mix(mix((tid >> 1) * 0x9E3779B9u + it * 0x85EBCA6Bu + 1u) + k * 0xC2B2AE35u) & rowmask;
                unsafeAtomicAdd(reinterpret_cast<float *>(table) + 2 * rr + (tid & 1), 1.0f);
            } else if constexpr (MODE == ATOM_U32) {
                atomicAdd(reinterpret_cast<unsigned int *>(table) + 2 * r, 1u);
            } else if constexpr (MODE == ATOM_U64) {
                atomicAdd(reinterpret_cast<unsigned long long *>(table) + r, 1ull);
            } else if constexpr (MODE == ATOM_F64) {
                unsafeAtomicAdd(reinterpret_cast<double *>(table) + r, 1.0);
            } else if constexpr (MODE == ATOM_F32_WG) {
                __hip_atomic_fetch_add(reinterpret_cast<float *>(table) + 2 * r, 1.0f, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if constexpr (MODE == ATOM_F32_WAVE) {
                __hip_atomic_fetch_add(reinterpret_cast<float *>(table) + 2 * r, 1.0f, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WAVEFRONT);
            } else if constexpr (MODE == ATOM_F32_COALESCED) {
                uint32_t base = (mix(h + k) & rowmask) & ~63u;  // wave-uniform? no: per-lane; make it per wave:
                base = (mix((tid >> 6) * 0x9E3779B9u + it * 131u + k) & rowmask) & ~31u;
                unsafeAtomicAdd(reinterpret_cast<float *>(table) + 2 * base + (tid & 63), 1.0f);
            } else if constexpr (MODE == RMW_PLAIN) {
                float2 v = table[r];
                v.x += 1.f; v.y += 1.f;
                table[r] = v;
            } else if constexpr (MODE == STORE8) {
                table[r] = make_float2(1.f, 2.f);
            } else if constexpr (MODE == ATOM_F32_RTN) {
                acc += atomicAdd(reinterpret_cast<float *>(table) + 2 * r, 1.0f);
            }
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}

// LDS atomics: each block owns an LDS table of `lds_rows` 8-byte rows; random ds_add_f32.
template <int PAIR>
__global__ __launch_bounds__(1024) void k_lds_atomic(int lds_rows, int iters, float *__restrict__ sink) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < lds_rows * 2; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t mask = lds_rows - 1;
    for (int it = 0; it < iters; ++it) {
        uint32_t h = mix(tid * 0x9E3779B9u + it * 0x85EBCA6Bu + 1u);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t r = mix(h + k * 0xC2B2AE35u) & mask;
            if (PAIR) {
                atomicAdd(&lds[2 * r], 1.0f);
                atomicAdd(&lds[2 * r + 1], 1.0f);
            } else {
                atomicAdd(&lds[2 * r], 1.0f);
            }
        }
    }
    __syncthreads();
    float a = 0.f;
    for (int i = threadIdx.x; i < lds_rows * 2; i += blockDim.x) a += lds[i];
    if (a == 123.456f) sink[0] = a;
}

__global__ __launch_bounds__(256) void k_copy(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_read(const float4 *__restrict__ src, size_t n, float *sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float a = 0.f;
    for (; i < n; i += stride) { float4 v = src[i]; a += v.x + v.y + v.z + v.w; }
    if (a == 123.456f) sink[0] = a;
}
__global__ __launch_bounds__(256) void k_write(float4 *__restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

template <typename Fn> static float time_ms(Fn fn, int reps = 3) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    fn();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a));
        fn();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        best = std::min(best, ms);
    }
    CK(hipGetLastError());
    return best;
}

template <int MODE> static void bench_random(const char *name, float2 *table, size_t table_bytes, float *sink,
                                              int opsPerK = 1, int blocks = 8192, int iters = 8) {
    const uint32_t rows = (uint32_t)(table_bytes / 8);
    float ms = time_ms([&] { hipLaunchKernelGGL(k_random<MODE>, dim3(blocks), dim3(256), 0, 0, table, rows - 1, iters, sink); });
    double ops = (double)blocks * 256 * iters * 8 * opsPerK;
    printf("%-28s table %8.2f MB  %8.3f ms  %8.2f Gops/s  (%7.1f GB/s at 8 B/op)\n", name, table_bytes / 1048576.0, ms,
           ops / ms / 1e6, ops * 8 / ms / 1e6);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d MHz  L2=%d KB\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000,
           prop.l2CacheSize / 1024);
    const size_t big = (size_t)1 << 30;
    float2 *table; float *sink; float4 *buf2;
    CK(hipMalloc(&table, big)); CK(hipMalloc(&buf2, big)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(table, 0, big)); CK(hipMemset(buf2, 0, big));

    const size_t sizes[] = {(size_t)128 << 10, (size_t)1 << 20, (size_t)4 << 20, (size_t)48 << 20 /*pow2 below*/, (size_t)64 << 20, (size_t)512 << 20};
    printf("\n== random gathers (each op = one 8-byte row unless noted) ==\n");
    for (size_t s : sizes) {
        if (s == ((size_t)48 << 20)) continue;
        bench_random<GATHER8>("gather 8B", table, s, sink);
        bench_random<GATHER16>("gather 16B (rows/2)", table, s, sink);
        bench_random<GATHER8_PAIRHASH>("gather 8B pair-hash", table, s, sink);
    }
    printf("\n== scattered read-modify-write ==\n");
    for (size_t s : {(size_t)4 << 20, (size_t)64 << 20}) {
        bench_random<ATOM_F32>("atomic f32 (1/lane)", table, s, sink, 1, 2048, 4);
        bench_random<ATOM_F32_PAIR>("atomic f32 pair (2/thread)", table, s, sink, 1, 2048, 4);
        bench_random<ATOM_F32_PAIR_LANES>("atomic f32 pair (lane pair)", table, s, sink, 1, 2048, 4);
        bench_random<ATOM_F32_RTN>("atomic f32 returning", table, s, sink, 1, 2048, 4);
        bench_random<ATOM_U32>("atomic u32", table, s, sink, 1, 2048, 4);
        bench_random<ATOM_U64>("atomic u64", table, s, sink, 1, 2048, 4);
        bench_random<ATOM_F64>("atomic f64", table, s, sink, 1, 2048, 4);
        bench_random<ATOM_F32_WG>("atomic f32 scope=workgroup", table, s, sink, 1, 2048, 4);
        bench_random<ATOM_F32_WAVE>("atomic f32 scope=wavefront", table, s, sink, 1, 2048, 4);
        bench_random<ATOM_F32_COALESCED>("atomic f32 coalesced 256B", table, s, sink, 1, 2048, 4);
        bench_random<RMW_PLAIN>("plain load+add+store 8B", table, s, sink, 1, 8192, 8);
        bench_random<STORE8>("plain store 8B", table, s, sink, 1, 8192, 8);
    }
    printf("\n== LDS atomics (per-block private table) ==\n");
    for (int rows : {4096, 16384}) {
        for (int pair = 0; pair < 2; ++pair) {
            const int blocks = 1024, iters = 16;
            size_t shmem = (size_t)rows * 8;
            float ms = time_ms([&] {
                if (pair) hipLaunchKernelGGL(k_lds_atomic<1>, dim3(blocks), dim3(1024), shmem, 0, rows, iters, sink);
                else hipLaunchKernelGGL(k_lds_atomic<0>, dim3(blocks), dim3(1024), shmem, 0, rows, iters, sink);
            });
            double ops = (double)blocks * 1024 * iters * 8 * (pair ? 2 : 1);
            printf("lds atomic f32 %s rows=%5d (%3zu KB)  %8.3f ms  %8.2f Gadds/s\n", pair ? "pair" : "single", rows, shmem >> 10, ms, ops / ms / 1e6);
        }
    }
    printf("\n== streams ==\n");
    for (size_t s : {(size_t)64 << 20, (size_t)1 << 30}) {
        size_t n = s / 16;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(4096), dim3(256), 0, 0, (const float4 *)table, buf2, n); });
        printf("copy  %6zu MB: %7.3f ms  %7.1f GB/s (r+w)\n", s >> 20, ms, 2.0 * s / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, (const float4 *)table, n, sink); });
        printf("read  %6zu MB: %7.3f ms  %7.1f GB/s\n", s >> 20, ms, 1.0 * s / ms / 1e6);
        ms = time_ms([&] { hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, 0, buf2, n); });
        printf("write %6zu MB: %7.3f ms  %7.1f GB/s\n", s >> 20, ms, 1.0 * s / ms / 1e6);
    }
    return 0;
}
