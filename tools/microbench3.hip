// tools/microbench3.hip -- round 5: what a scattered RUN of items costs the memory system, and 12- vs 16-byte lane accesses.
//   runs_write: every wave writes `run_bytes` contiguous bytes at a pseudo-random run slot of a large array (each slot once):
//               the write pattern of the binned backward's scatter pass without its ALU / LDS work.
//   runs_read:  the same slots read back (the consume pass's pattern is sequential; this is the gather counterpart).
//   lane12 / lane16: streaming copy with 12-byte (dwordx3) or 16-byte (dwordx4) accesses per lane.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/microbench3 tools/microbench3.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));

// slot permutation: slot = (run * ODD) mod nslots (nslots a power of two) -> every slot exactly once, neighbours far apart
template <bool NT>
__global__ __launch_bounds__(512) void runs_write(unsigned char *__restrict__ buf, uint32_t nslots_mask, uint32_t run_bytes,
                                                  uint32_t misalign, uint32_t runs_per_wave) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    for (uint32_t k = 0; k < runs_per_wave; ++k) {
        const uint32_t run = wave * runs_per_wave + k;
        const uint32_t slot = (run * 2654435761u) & nslots_mask;
        unsigned char *p = buf + (size_t)slot * run_bytes + misalign;
        const u32x4 v = {run, lane, k, 1u};
        for (uint32_t o = lane * 16u; o < run_bytes; o += 1024u) {
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p + o));
            else *reinterpret_cast<u32x4 *>(p + o) = v;
        }
    }
}

__global__ __launch_bounds__(512) void runs_read(const unsigned char *__restrict__ buf, uint32_t nslots_mask, uint32_t run_bytes,
                                                 uint32_t misalign, uint32_t runs_per_wave, uint32_t *__restrict__ sink) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    uint32_t acc = 0;
    for (uint32_t k = 0; k < runs_per_wave; ++k) {
        const uint32_t run = wave * runs_per_wave + k;
        const uint32_t slot = (run * 2654435761u) & nslots_mask;
        const unsigned char *p = buf + (size_t)slot * run_bytes + misalign;
        for (uint32_t o = lane * 16u; o < run_bytes; o += 1024u) {
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p + o));
            acc += v[0] ^ v[3];
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// streaming with W-byte lane accesses (W = 12: dwordx3, W = 16: dwordx4), grid-stride, UN accesses in flight
template <int W, bool NT> __global__ __launch_bounds__(512) void lane_write(unsigned char *__restrict__ buf, size_t n_items) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += stride) {
        if constexpr (W == 16) {
            const u32x4 v = {(uint32_t)i, 1u, 2u, 3u};
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(buf + i * 16));
            else *reinterpret_cast<u32x4 *>(buf + i * 16) = v;
        } else {
            const u32x3 v = {(uint32_t)i, 1u, 2u};
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x3 *>(buf + i * 12));
            else *reinterpret_cast<u32x3 *>(buf + i * 12) = v;
        }
    }
}
template <int W, bool NT> __global__ __launch_bounds__(1024) void lane_read(const unsigned char *__restrict__ buf, size_t n_items,
                                                                           uint32_t *__restrict__ sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < n_items; i += 8 * stride) {
        if constexpr (W == 16) {
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(buf + (i + u * stride) * 16))
                          : *reinterpret_cast<const u32x4 *>(buf + (i + u * stride) * 16);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u][0] ^ v[u][3];
        } else {
            u32x3 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x3 *>(buf + (i + u * stride) * 12))
                          : *reinterpret_cast<const u32x3 *>(buf + (i + u * stride) * 12);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u][0] ^ v[u][2];
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <typename Fn> static float time_ms(Fn fn, int reps = 5) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    fn();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a));
        fn();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    const size_t cap = (size_t)1 << 30;
    unsigned char *buf;
    uint32_t *sink;
    CK(hipMalloc(&buf, cap + 4096));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, cap + 4096));
    printf("== scattered runs: each wave writes / reads run_bytes contiguous bytes at a random slot (16-byte lanes) ==\n");
    for (size_t total : {(size_t)1 << 30}) {
        for (uint32_t run_bytes : {512u, 1024u, 2048u, 768u, 1536u}) {
            for (uint32_t mis : {0u, 16u, 32u, 64u, 48u}) {
                uint32_t nslots = 1; while ((size_t)nslots * 2 * run_bytes <= total) nslots *= 2;   // power of two (the permutation needs it)
                const uint32_t rpw = 4;
                const uint32_t waves = nslots / rpw;
                const uint32_t blocks = waves / 8;
                const float w_nt = time_ms([&] { hipLaunchKernelGGL((runs_write<true>), dim3(blocks), dim3(512), 0, 0, buf, nslots - 1, run_bytes, mis, rpw); });
                const float w_pl = time_ms([&] { hipLaunchKernelGGL((runs_write<false>), dim3(blocks), dim3(512), 0, 0, buf, nslots - 1, run_bytes, mis, rpw); });
                const float r_nt = time_ms([&] { hipLaunchKernelGGL(runs_read, dim3(blocks), dim3(512), 0, 0, buf, nslots - 1, run_bytes, mis, rpw, sink); });
                printf("array %5zu MB run %6u B misalign %2u: write nt %7.1f GB/s (%6.2f Gruns/s)  write plain %7.1f GB/s  read nt %7.1f GB/s\n",
                       total >> 20, run_bytes, mis, (double)nslots * run_bytes / w_nt * 1e-6, nslots / w_nt * 1e-6, (double)nslots * run_bytes / w_pl * 1e-6, (double)nslots * run_bytes / r_nt * 1e-6);
            }
        }
    }
    printf("== streaming, 12- vs 16-byte lane accesses (768 MB) ==\n");
    {
        const size_t bytes = (size_t)768 << 20;
        const float w16 = time_ms([&] { hipLaunchKernelGGL((lane_write<16, true>), dim3(4096), dim3(512), 0, 0, buf, bytes / 16); });
        const float w12 = time_ms([&] { hipLaunchKernelGGL((lane_write<12, true>), dim3(4096), dim3(512), 0, 0, buf, bytes / 12); });
        const float w16p = time_ms([&] { hipLaunchKernelGGL((lane_write<16, false>), dim3(4096), dim3(512), 0, 0, buf, bytes / 16); });
        const float w12p = time_ms([&] { hipLaunchKernelGGL((lane_write<12, false>), dim3(4096), dim3(512), 0, 0, buf, bytes / 12); });
        const float r16 = time_ms([&] { hipLaunchKernelGGL((lane_read<16, true>), dim3(512), dim3(1024), 0, 0, buf, bytes / 16, sink); });
        const float r12 = time_ms([&] { hipLaunchKernelGGL((lane_read<12, true>), dim3(512), dim3(1024), 0, 0, buf, bytes / 12, sink); });
        const float r16p = time_ms([&] { hipLaunchKernelGGL((lane_read<16, false>), dim3(512), dim3(1024), 0, 0, buf, bytes / 16, sink); });
        const float r12p = time_ms([&] { hipLaunchKernelGGL((lane_read<12, false>), dim3(512), dim3(1024), 0, 0, buf, bytes / 12, sink); });
        printf("write nt    16 B: %7.1f GB/s   12 B: %7.1f GB/s\n", bytes / w16 * 1e-6, bytes / w12 * 1e-6);
        printf("write plain 16 B: %7.1f GB/s   12 B: %7.1f GB/s\n", bytes / w16p * 1e-6, bytes / w12p * 1e-6);
        printf("read  nt    16 B: %7.1f GB/s   12 B: %7.1f GB/s\n", bytes / r16 * 1e-6, bytes / r12 * 1e-6);
        printf("read  plain 16 B: %7.1f GB/s   12 B: %7.1f GB/s\n", bytes / r16p * 1e-6, bytes / r12p * 1e-6);
    }
    return 0;
}
