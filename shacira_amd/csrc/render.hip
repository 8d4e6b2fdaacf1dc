// render.hip -- the steps either side of the hash-grid path in the NeRF pipeline (gfx950; SURVEY.md section 8 "next" f2):
// sample generation on a dense occupancy grid and volume integration over variable-length packs.
//
// Reference: the reference does these through un-vendored kaolin 0.13 CUDA (`spc_render.exponential_integration`,
// `sum_reduce`, `unbatched_raytrace`, `unbatched_query`) driven by its own Python
// (wisp/tracers/packed_rf_tracer.py:109-151, wisp/accelstructs/octree_as.py:171-290). kaolin's exponential_integration is
// a chain of five ATen/kaolin kernels with a TODO "this should be a fused kernel"; here it is one.
//
// A pack = the consecutive samples of one ray, [pack_start[r], pack_start[r+1]). One 64-lane wave walks one pack in
// chunks of 64 samples: loads are coalesced, the running optical depth is a wave scan plus a carry.
#include "internal.h"

namespace shacira {

constexpr int kPackWaves = 4;   // packs per 256-thread workgroup

__device__ __forceinline__ float wave_incl_scan(float v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float n = __shfl_up(v, off, 64);
        if (lane >= off) v += n;
    }
    return v;
}
__device__ __forceinline__ float wave_incl_scan_rev(float v, int lane) {   // suffix sums: lane i gets sum_{k >= i}
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float n = __shfl_down(v, off, 64);
        if (lane + off < 64) v += n;
    }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ray_feats[r, c] = sum_i w_i feats[i, c];  w_i = exp(-sum_{j<i} tau_j) * (1 - exp(-tau_i))   (weights out)
template <int C>
__global__ __launch_bounds__(64 * kPackWaves) void pack_integrate_fwd_kernel(const float *__restrict__ feats,
                                                                             const float *__restrict__ tau,
                                                                             const int64_t *__restrict__ pack_start,
                                                                             float *__restrict__ ray_feats,
                                                                             float *__restrict__ weights, int64_t R,
                                                                             int channels) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kPackWaves + (threadIdx.x >> 6);
    if (r >= R) return;
    const int nc = C > 0 ? C : channels;
    const int64_t begin = pack_start[r], end = pack_start[r + 1];
    float carry = 0.0f;
    float acc[C > 0 ? C : 16];
#pragma unroll
    for (int c = 0; c < (C > 0 ? C : 16); ++c) acc[c] = 0.0f;
    for (int64_t base = begin; base < end; base += 64) {
        const int64_t i = base + lane;
        const bool live = i < end;
        const float t = live ? tau[i] : 0.0f;
        const float incl = wave_incl_scan(t, lane);
        const float excl = carry + (incl - t);
        const float w = live ? expf(-excl) * (1.0f - expf(-t)) : 0.0f;
        if (live) {
            weights[i] = w;
#pragma unroll
            for (int c = 0; c < (C > 0 ? C : 16); ++c)
                if (c < nc) acc[c] = fmaf(w, feats[i * nc + c], acc[c]);
        }
        carry += __shfl(incl, 63, 64);
    }
#pragma unroll
    for (int c = 0; c < (C > 0 ? C : 16); ++c) {
        if (c < nc) {
            const float s = wave_sum(acc[c]);
            if (lane == 0) ray_feats[r * nc + c] = s;
        }
    }
}

// Backward. G_i = gw_i + sum_c gray[r, c] f[i, c];  g_f[i, c] = w_i gray[r, c];
// g_tau[i] = G_i exp(-(excl_i + tau_i)) - sum_{k > i} G_k w_k.   Pass 1 (forward) leaves excl_i in g_tau; pass 2
// walks the chunks backwards so the suffix sum is accumulated in order.
template <int C>
__global__ __launch_bounds__(64 * kPackWaves) void pack_integrate_bwd_kernel(
    const float *__restrict__ feats, const float *__restrict__ tau, const int64_t *__restrict__ pack_start,
    const float *__restrict__ g_ray, const float *__restrict__ g_w, float *__restrict__ g_feats,
    float *__restrict__ g_tau, int64_t R, int channels) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kPackWaves + (threadIdx.x >> 6);
    if (r >= R) return;
    const int nc = C > 0 ? C : channels;
    const int64_t begin = pack_start[r], end = pack_start[r + 1];
    if (end <= begin) return;
    float gr[C > 0 ? C : 16];
#pragma unroll
    for (int c = 0; c < (C > 0 ? C : 16); ++c) gr[c] = (c < nc) ? g_ray[r * nc + c] : 0.0f;
    float carry = 0.0f;
    for (int64_t base = begin; base < end; base += 64) {
        const int64_t i = base + lane;
        const bool live = i < end;
        const float t = live ? tau[i] : 0.0f;
        const float incl = wave_incl_scan(t, lane);
        if (live) g_tau[i] = carry + (incl - t);
        carry += __shfl(incl, 63, 64);
    }
    const int64_t nchunks = (end - begin + 63) / 64;
    float suffix = 0.0f;   // sum of G_k w_k over the chunks after this one
    for (int64_t ch = nchunks - 1; ch >= 0; --ch) {
        const int64_t i = begin + ch * 64 + lane;
        const bool live = i < end;
        float gw_term = 0.0f, e_term = 0.0f, w = 0.0f;
        if (live) {
            const float t = tau[i];
            const float excl = g_tau[i];
            const float T = expf(-excl), et = expf(-t);
            w = T * (1.0f - et);
            float G = g_w ? g_w[i] : 0.0f;
#pragma unroll
            for (int c = 0; c < (C > 0 ? C : 16); ++c) {
                if (c < nc) {
                    G = fmaf(gr[c], feats[i * nc + c], G);
                    g_feats[i * nc + c] = w * gr[c];
                }
            }
            gw_term = G * w;
            e_term = G * T * et;
        }
        const float incl_rev = wave_incl_scan_rev(gw_term, lane);   // sum_{k >= i} within the chunk
        if (live) g_tau[i] = e_term - (suffix + (incl_rev - gw_term));
        suffix += __shfl(incl_rev, 0, 64);
    }
}

// out[r, c] = sum_i x[i, c]          (kaolin sum_reduce)
__global__ __launch_bounds__(64 * kPackWaves) void pack_sum_kernel(const float *__restrict__ x,
                                                                   const int64_t *__restrict__ pack_start,
                                                                   float *__restrict__ out, int64_t R, int nc) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kPackWaves + (threadIdx.x >> 6);
    if (r >= R) return;
    const int64_t begin = pack_start[r] * nc, end = pack_start[r + 1] * nc;
    // lanes walk the flat [samples * nc] range; lane l only ever meets channels (l + 64 k) % nc
    for (int c = 0; c < nc; ++c) {
        float acc = 0.0f;
        for (int64_t e = begin + c + (int64_t)lane * nc; e < end; e += (int64_t)64 * nc) acc += x[e];
        acc = wave_sum(acc);
        if (lane == 0) out[r * nc + c] = acc;
    }
}

// out[i, c] = per_pack[r(i), c]       (gradient of sum_reduce)
__global__ __launch_bounds__(64 * kPackWaves) void pack_broadcast_kernel(const float *__restrict__ per_pack,
                                                                         const int64_t *__restrict__ pack_start,
                                                                         float *__restrict__ out, int64_t R, int nc) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kPackWaves + (threadIdx.x >> 6);
    if (r >= R) return;
    const int64_t begin = pack_start[r] * nc, end = pack_start[r + 1] * nc;
    for (int64_t e = begin + lane; e < end; e += 64) out[e] = per_pack[r * nc + (int)((e - begin) % nc)];
}

// ---------------------------------------------------------------------------------------------- sample generation
__device__ __forceinline__ int quantize_axis(float x, int G) {
    // kaolin's float query: cell = floor(res * (x + 1) / 2); a cell outside [0, res) does not exist (spc identify returns
    // -1 out of bounds), so a point outside the cube -- or a NaN -- belongs to no cell: -1
    const float v = floorf((float)G * (x + 1.0f) / 2.0f);
    return (v >= 0.0f && v < (float)G) ? (int)v : -1;
}

__device__ __forceinline__ bool occupied(const uint8_t *__restrict__ occ, int G, int x, int y, int z) {
    return occ[((size_t)x * G + y) * G + z] != 0;
}

// 'ray' marcher (octree_as.py:257-290): sample j of ray r sits at depth (lin[j] + jitter[r, j] / ns) * (far - near) +
// near and survives when its cell is occupied. EMIT = false: counts[r] = survivors. EMIT = true: writes them at
// offsets[r]... in sample order. One wave per ray, lanes = samples.
template <bool EMIT>
__global__ __launch_bounds__(64 * kPackWaves) void raymarch_ray_kernel(
    int64_t num_rays, int ns, const float *__restrict__ origins, const float *__restrict__ dirs, float dist_min,
    float dist_max, const float *__restrict__ lin, const float *__restrict__ jitter, const uint8_t *__restrict__ occ,
    int G, int32_t *__restrict__ counts, const int64_t *__restrict__ offsets, int64_t *__restrict__ ridx,
    float *__restrict__ samples, float *__restrict__ depth_out, float *__restrict__ deltas,
    uint8_t *__restrict__ boundary, int64_t capacity) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * kPackWaves + (threadIdx.x >> 6);
    if (r >= num_rays) return;
    const float ox = origins[r * 3], oy = origins[r * 3 + 1], oz = origins[r * 3 + 2];
    const float dx = dirs[r * 3], dy = dirs[r * 3 + 1], dz = dirs[r * 3 + 2];
    const float span = dist_max - dist_min;
    int64_t out = EMIT ? offsets[r] : 0;
    const int64_t first = out;
    int total = 0;
    for (int base = 0; base < ns; base += 64) {
        const int j = base + lane;
        const bool live = j < ns;
        float depth = 0.0f, prev = 0.0f, px = 0.0f, py = 0.0f, pz = 0.0f;
        bool keep = false;
        if (live) {
            depth = (lin[j] + jitter[r * ns + j] / (float)ns) * span + dist_min;
            prev = (j == 0) ? dist_min : (lin[j - 1] + jitter[r * ns + j - 1] / (float)ns) * span + dist_min;
            px = fmaf(dx, depth, ox);
            py = fmaf(dy, depth, oy);
            pz = fmaf(dz, depth, oz);
            const int qx = quantize_axis(px, G), qy = quantize_axis(py, G), qz = quantize_axis(pz, G);
            keep = qx >= 0 && qy >= 0 && qz >= 0 && occupied(occ, G, qx, qy, qz);
        }
        const unsigned long long m = __ballot(keep);
        if constexpr (EMIT) {
            const int64_t k = out + __popcll(m & ((1ull << lane) - 1ull));
            // (capacity: rows of the caller's fixed-size buffers -- a step captured into a HIP graph cannot size them from
            // the count; survivors beyond it are dropped, the caller clamps its pack offsets the same way)
            if (keep && k < capacity) {
                ridx[k] = r;
                samples[k * 3] = px;
                samples[k * 3 + 1] = py;
                samples[k * 3 + 2] = pz;
                depth_out[k] = depth;
                deltas[k] = depth - prev;
                boundary[k] = (k == first) ? 1 : 0;
            }
            out += __popcll(m);
        } else {
            total += __popcll(m);
        }
    }
    if constexpr (!EMIT)
        if (lane == 0) counts[r] = total;
}

// Ray / occupied-cell intersections on the dense grid, ordered by depth (3-D DDA, Amanatides & Woo); thread per ray.
// EMIT = false: counts[r]; EMIT = true: ridx / pidx (Morton index of the cell, x most significant) / depth [K, 2]
// (entry clipped to 0 for rays starting inside the volume).
__device__ __forceinline__ uint32_t morton_x_major(uint32_t x, uint32_t y, uint32_t z, int level) {
    uint32_t m = 0;
    for (int b = 0; b < level; ++b)
        m |= (((x >> b) & 1u) << (3 * b + 2)) | (((y >> b) & 1u) << (3 * b + 1)) | (((z >> b) & 1u) << (3 * b));
    return m;
}

template <bool EMIT>
__global__ __launch_bounds__(256) void raytrace_dense_kernel(int64_t num_rays, const float *__restrict__ origins,
                                                             const float *__restrict__ dirs,
                                                             const uint8_t *__restrict__ occ, int level,
                                                             int32_t *__restrict__ counts,
                                                             const int64_t *__restrict__ offsets,
                                                             int32_t *__restrict__ ridx, int32_t *__restrict__ pidx,
                                                             float *__restrict__ depth) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= num_rays) return;
    const int G = 1 << level;
    const float o[3] = {origins[r * 3], origins[r * 3 + 1], origins[r * 3 + 2]};
    const float d[3] = {dirs[r * 3], dirs[r * 3 + 1], dirs[r * 3 + 2]};
    // slab test against [-1, 1]^3
    float tn = -INFINITY, tf = INFINITY;
    bool miss = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (d[a] == 0.0f) {
            if (o[a] < -1.0f || o[a] > 1.0f) miss = true;
        } else {
            const float inv = 1.0f / d[a];
            const float t0 = (-1.0f - o[a]) * inv, t1 = (1.0f - o[a]) * inv;
            tn = fmaxf(tn, fminf(t0, t1));
            tf = fminf(tf, fmaxf(t0, t1));
        }
    }
    int32_t n = 0;
    int64_t out = EMIT ? offsets[r] : 0;
    if (!miss && tf > fmaxf(tn, 0.0f)) {
        float t = fmaxf(tn, 0.0f);
        const float cell = 2.0f / (float)G;
        int c[3], step[3];
        float tmax[3], tdelta[3];
        const float tin = t;   // a start cell that is off by one rounding is skipped by the walk (zero-length stay)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float p = fmaf(d[a], tin, o[a]);
            int q = (int)floorf((p + 1.0f) * 0.5f * (float)G);
            q = q < 0 ? 0 : (q > G - 1 ? G - 1 : q);
            c[a] = q;
            if (d[a] > 0.0f) {
                step[a] = 1;
                tdelta[a] = cell / d[a];
                tmax[a] = ((-1.0f + (float)(q + 1) * cell) - o[a]) / d[a];
            } else if (d[a] < 0.0f) {
                step[a] = -1;
                tdelta[a] = -cell / d[a];
                tmax[a] = ((-1.0f + (float)q * cell) - o[a]) / d[a];
            } else {
                step[a] = 0;
                tdelta[a] = INFINITY;
                tmax[a] = INFINITY;
            }
        }
        for (int it = 0; it < 3 * G + 3; ++it) {
            const int a = (tmax[0] <= tmax[1]) ? ((tmax[0] <= tmax[2]) ? 0 : 2) : ((tmax[1] <= tmax[2]) ? 1 : 2);
            const float texit = fminf(tmax[a], tf);
            if (texit > t && occupied(occ, G, c[0], c[1], c[2])) {
                if constexpr (EMIT) {
                    ridx[out] = (int32_t)r;
                    pidx[out] = (int32_t)morton_x_major((uint32_t)c[0], (uint32_t)c[1], (uint32_t)c[2], level);
                    depth[out * 2] = t;
                    depth[out * 2 + 1] = texit;
                    ++out;
                }
                ++n;
            }
            t = fmaxf(t, texit);
            if (tmax[a] >= tf) break;
            c[a] += step[a];
            if (c[a] < 0 || c[a] >= G) break;
            tmax[a] += tdelta[a];
        }
    }
    if constexpr (!EMIT) counts[r] = n;
}

// ------------------------------------------------------------------------------------------------------ host side
static inline uint32_t pack_blocks(int64_t R) { return (uint32_t)((R + kPackWaves - 1) / kPackWaves); }

hipError_t pack_integrate_launch(bool bwd, int64_t R, int C, const float *feats, const float *tau,
                                 const int64_t *pack_start, float *ray_feats, float *weights, const float *g_ray,
                                 const float *g_w, float *g_feats, float *g_tau, hipStream_t s) {
    if (R == 0) return hipSuccess;
    const dim3 grid(pack_blocks(R)), block(64 * kPackWaves);
#define SHACIRA_PACK(CC)                                                                                           \
    do {                                                                                                           \
        if (!bwd)                                                                                                  \
            hipLaunchKernelGGL((pack_integrate_fwd_kernel<CC>), grid, block, 0, s, feats, tau, pack_start, ray_feats, \
                               weights, R, C);                                                                     \
        else                                                                                                       \
            hipLaunchKernelGGL((pack_integrate_bwd_kernel<CC>), grid, block, 0, s, feats, tau, pack_start, g_ray,  \
                               g_w, g_feats, g_tau, R, C);                                                         \
    } while (0)
    switch (C) {
        case 1: SHACIRA_PACK(1); break;
        case 3: SHACIRA_PACK(3); break;
        case 4: SHACIRA_PACK(4); break;
        default: SHACIRA_PACK(0); break;
    }
#undef SHACIRA_PACK
    return hipGetLastError();
}

hipError_t pack_sum_launch(bool broadcast, int64_t R, int C, const float *in, const int64_t *pack_start, float *out,
                           hipStream_t s) {
    if (R == 0) return hipSuccess;
    if (!broadcast)
        hipLaunchKernelGGL(pack_sum_kernel, dim3(pack_blocks(R)), dim3(64 * kPackWaves), 0, s, in, pack_start, out, R, C);
    else
        hipLaunchKernelGGL(pack_broadcast_kernel, dim3(pack_blocks(R)), dim3(64 * kPackWaves), 0, s, in, pack_start, out,
                           R, C);
    return hipGetLastError();
}

hipError_t raymarch_ray_launch(bool emit, int64_t num_rays, int ns, const float *origins, const float *dirs,
                               float dist_min, float dist_max, const float *lin, const float *jitter,
                               const uint8_t *occ, int level, int32_t *counts, const int64_t *offsets, int64_t *ridx,
                               float *samples, float *depth, float *deltas, uint8_t *boundary, int64_t capacity,
                               hipStream_t s) {
    if (num_rays == 0) return hipSuccess;
    const dim3 grid(pack_blocks(num_rays)), block(64 * kPackWaves);
    if (!emit)
        hipLaunchKernelGGL(raymarch_ray_kernel<false>, grid, block, 0, s, num_rays, ns, origins, dirs, dist_min, dist_max,
                           lin, jitter, occ, 1 << level, counts, offsets, ridx, samples, depth, deltas, boundary, capacity);
    else
        hipLaunchKernelGGL(raymarch_ray_kernel<true>, grid, block, 0, s, num_rays, ns, origins, dirs, dist_min, dist_max,
                           lin, jitter, occ, 1 << level, counts, offsets, ridx, samples, depth, deltas, boundary, capacity);
    return hipGetLastError();
}

hipError_t raytrace_dense_launch(bool emit, int64_t num_rays, const float *origins, const float *dirs,
                                 const uint8_t *occ, int level, int32_t *counts, const int64_t *offsets, int32_t *ridx,
                                 int32_t *pidx, float *depth, hipStream_t s) {
    if (num_rays == 0) return hipSuccess;
    const dim3 grid((uint32_t)((num_rays + 255) / 256)), block(256);
    if (!emit)
        hipLaunchKernelGGL(raytrace_dense_kernel<false>, grid, block, 0, s, num_rays, origins, dirs, occ, level, counts,
                           offsets, ridx, pidx, depth);
    else
        hipLaunchKernelGGL(raytrace_dense_kernel<true>, grid, block, 0, s, num_rays, origins, dirs, occ, level, counts,
                           offsets, ridx, pidx, depth);
    return hipGetLastError();
}

}  // namespace shacira
