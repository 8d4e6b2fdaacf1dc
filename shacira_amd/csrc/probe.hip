// probe.hip -- streaming-rate probes (diagnostics for bench.py's `measured_stream_rates`; not on the product path).
//
// What this chip's memory system sustains on plain streams with the access shape the hash-grid kernels use: 16 bytes per
// lane, 8 accesses in flight per lane, non-temporal. torch's own reductions / copies are not that shape (torch.sum reads at
// 3.9 TB/s where this read probe reaches 6.6 TB/s), so quoting them as "measured peak" flattered every fraction.
#include "internal.h"

namespace shacira {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// kind 0: read only (xor-folded into one word per workgroup so the loads cannot be dropped); 1: write only; 2: copy a -> b
template <int KIND>
__global__ __launch_bounds__(256) void stream_probe_kernel(const u32x4 *__restrict__ a, u32x4 *__restrict__ b, size_t nvec,
                                                           uint32_t *__restrict__ sink) {
    constexpr int U = 8;
    const size_t stride = (size_t)gridDim.x * 256 * U;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x; base < nvec; base += stride) {
        u32x4 v[U];
        if constexpr (KIND != 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t e = base + (size_t)u * 256;
                v[u] = __builtin_nontemporal_load(a + (e < nvec ? e : nvec - 1));
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t e = base + (size_t)u * 256;
            if constexpr (KIND == 0) acc ^= v[u];
            if constexpr (KIND == 1) {
                const u32x4 w = {(uint32_t)e, 1u, 2u, 3u};
                if (e < nvec) __builtin_nontemporal_store(w, b + e);
            }
            if constexpr (KIND == 2) {
                if (e < nvec) __builtin_nontemporal_store(v[u], b + e);
            }
        }
    }
    if constexpr (KIND == 0) {
        const uint32_t x = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
        if (x == 0x9E3779B1u && sink) sink[0] = x;   // practically never: keeps the loads alive
    }
}

hipError_t stream_probe_launch(int kind, const void *a, void *b, size_t bytes, uint32_t *sink, hipStream_t s) {
    const size_t nvec = bytes / 16;
    if (nvec == 0) return hipSuccess;
    size_t blocks = (nvec + 256 * 8 - 1) / (256 * 8);
    if (blocks > 256 * 16) blocks = 256 * 16;   // 16 workgroups per CU, grid-stride
    const u32x4 *pa = static_cast<const u32x4 *>(a);
    u32x4 *pb = static_cast<u32x4 *>(b);
    if (kind == 0) hipLaunchKernelGGL(stream_probe_kernel<0>, dim3((uint32_t)blocks), dim3(256), 0, s, pa, pb, nvec, sink);
    else if (kind == 1) hipLaunchKernelGGL(stream_probe_kernel<1>, dim3((uint32_t)blocks), dim3(256), 0, s, pa, pb, nvec, sink);
    else hipLaunchKernelGGL(stream_probe_kernel<2>, dim3((uint32_t)blocks), dim3(256), 0, s, pa, pb, nvec, sink);
    return hipGetLastError();
}

}  // namespace shacira
