// tools/unit_stream_bench.hip -- what a persistent, one-workgroup-per-CU consumer of contiguous "units" can stream, as a function
// of the unit size and of how its loads are pipelined (round 4: the consume pass reads nerf_lego's one-round units at 1.5 TB/s).
//   hipcc -O3 --offload-arch=gfx950 tools/unit_stream_bench.hip -o tools/unit_stream_bench && tools/unit_stream_bench
// Each unit = U bytes; 1024 threads; a round = 8 x 16-byte loads per thread (128 KiB per workgroup); after the unit's last
// round: LDS "image" zeroing (128 KiB of ds_write) + barrier + a 64 KiB store "flush", like consume_unit.
// mode 0: load -> wait -> next (as shipped)      mode 1: next round's / next unit's loads issued before the current wait
// mode 2: like 0 but TWO workgroups per CU (64 KiB LDS each, 512 threads)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int THREADS, bool NT>
__device__ __forceinline__ void load_round(const u32x4 *p, size_t idx, size_t lim, u32x4 (&v)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const size_t q = idx + (size_t)u * THREADS;
        v[u] = u32x4{0, 0, 0, 0};
        if (q < lim) v[u] = NT ? __builtin_nontemporal_load(p + q) : p[q];
    }
}

template <int THREADS, int MODE, bool NT, bool TAIL>
__global__ __launch_bounds__(THREADS) void stream_units(const u32x4 *__restrict__ buf, size_t unit_vecs, unsigned n_units,
                                                        unsigned *__restrict__ counter, float *__restrict__ flush,
                                                        unsigned *__restrict__ sink, int lds_bytes) {
    extern __shared__ double s_img[];
    __shared__ unsigned s_unit[2];
    unsigned acc = 0;
    if (threadIdx.x == 0) { s_unit[0] = atomicAdd(counter, 1u); s_unit[1] = atomicAdd(counter, 1u); }
    __syncthreads();
    unsigned cur = s_unit[0], nxt = s_unit[1];
    u32x4 a[8], b[8];
    if (MODE == 1 && cur < n_units) load_round<THREADS, NT>(buf, (size_t)cur * unit_vecs + threadIdx.x, (size_t)(cur + 1) * unit_vecs, a);
    while (cur < n_units) {
        const size_t base = (size_t)cur * unit_vecs, lim = base + unit_vecs;
        const size_t stride = (size_t)THREADS * 8;
        if (MODE != 1) {
            for (size_t p = base + threadIdx.x; p < lim; p += stride) {
                load_round<THREADS, NT>(buf, p, lim, a);
                if (TAIL && p == base + threadIdx.x) {   // zero the image inside the first round's latency, as consume_unit does
                    for (int e = threadIdx.x; e < lds_bytes / 8; e += THREADS) s_img[e] = 0.0;
                    __syncthreads();
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += a[u][0] ^ a[u][1] ^ a[u][2] ^ a[u][3];
            }
        } else {
            // software pipeline across rounds AND units: the loads of the next round (or of the next unit's first round) are
            // in flight while the current one is consumed
            bool first = true;
            for (size_t p = base + threadIdx.x; p < lim; p += stride) {
                const size_t pn = p + stride;
                if (pn < lim) load_round<THREADS, NT>(buf, pn, lim, b);
                else if (nxt < n_units) load_round<THREADS, NT>(buf, (size_t)nxt * unit_vecs + threadIdx.x, (size_t)(nxt + 1) * unit_vecs, b);
                if (TAIL && first) {
                    for (int e = threadIdx.x; e < lds_bytes / 8; e += THREADS) s_img[e] = 0.0;
                    __syncthreads();
                    first = false;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += a[u][0] ^ a[u][1] ^ a[u][2] ^ a[u][3];
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] = b[u];
            }
        }
        if (TAIL) {
            __syncthreads();
            // flush: lds_bytes / 2 of fp32 out (plain 16-byte stores)
            float *dst = flush + (size_t)cur * (lds_bytes / 8);
            for (int e = threadIdx.x * 4; e < lds_bytes / 8; e += THREADS * 4) {
                float4 o = {(float)s_img[e], (float)s_img[e + 1], (float)s_img[e + 2], (float)s_img[e + 3]};
                *reinterpret_cast<float4 *>(dst + e) = o;
            }
        }
        if (threadIdx.x == 0) { s_unit[0] = nxt; s_unit[1] = atomicAdd(counter, 1u); }
        __syncthreads();
        cur = s_unit[0];
        nxt = s_unit[1];
        __syncthreads();
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// 24-byte items read as the consume pass reads them: one 16-byte + one 8-byte load per item and lane (both instructions of
// a wave touch the same twelve 128-byte lines)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <bool NT, int SPLIT>
__global__ __launch_bounds__(1024) void stream_items24(const char *__restrict__ buf, size_t unit_items, unsigned n_units,
                                                       unsigned *__restrict__ counter, float *__restrict__ flush,
                                                       unsigned *__restrict__ sink, int lds_bytes) {
    extern __shared__ double s_img[];
    __shared__ unsigned s_unit[2];
    unsigned acc = 0;
    if (threadIdx.x == 0) { s_unit[0] = atomicAdd(counter, 1u); s_unit[1] = atomicAdd(counter, 1u); }
    __syncthreads();
    unsigned cur = s_unit[0], nxt = s_unit[1];
    while (cur < n_units) {
        const size_t base = (size_t)cur * unit_items, lim = base + unit_items;
        for (size_t p = base + threadIdx.x; p < lim; p += 8192) {
            u32x4 a[8];
            u32x2 b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const size_t q = p + (size_t)u * 1024;
                a[u] = u32x4{0, 0, 0, 0};
                b[u] = u32x2{0, 0};
                if (q < lim) {
                    if (SPLIT == 0) {           // as shipped: 16 + 8 bytes of item q
                        const char *it = buf + q * 24;
                        a[u] = NT ? __builtin_nontemporal_load((const u32x4 *)it) : *(const u32x4 *)it;
                        b[u] = NT ? __builtin_nontemporal_load((const u32x2 *)(it + 16)) : *(const u32x2 *)(it + 16);
                    } else {                    // the same bytes as contiguous 16-byte chunks: 1.5 per item
                        const size_t w0 = (q - threadIdx.x % 64) * 24;   // byte offset of the wave's 64 items
                        const char *c0 = buf + w0 + (threadIdx.x % 64) * 16;
                        a[u] = NT ? __builtin_nontemporal_load((const u32x4 *)c0) : *(const u32x4 *)c0;
                        if (threadIdx.x % 64 < 32) {
                            const u32x4 t = NT ? __builtin_nontemporal_load((const u32x4 *)(c0 + 1024)) : *(const u32x4 *)(c0 + 1024);
                            b[u] = u32x2{t[0] ^ t[2], t[1] ^ t[3]};
                        }
                    }
                }
            }
            if (p == base + threadIdx.x) {
                for (int e = threadIdx.x; e < lds_bytes / 8; e += 1024) s_img[e] = 0.0;
                __syncthreads();
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += a[u][0] ^ a[u][1] ^ a[u][2] ^ a[u][3] ^ b[u][0] ^ b[u][1];
        }
        __syncthreads();
        float *dst = flush + (size_t)cur * (lds_bytes / 8);
        for (int e = threadIdx.x * 4; e < lds_bytes / 8; e += 1024 * 4) {
            float4 o = {(float)s_img[e], (float)s_img[e + 1], (float)s_img[e + 2], (float)s_img[e + 3]};
            *reinterpret_cast<float4 *>(dst + e) = o;
        }
        if (threadIdx.x == 0) { s_unit[0] = nxt; s_unit[1] = atomicAdd(counter, 1u); }
        __syncthreads();
        cur = s_unit[0];
        nxt = s_unit[1];
        __syncthreads();
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <bool NT, int SPLIT>
static float run24(const char *buf, size_t total_bytes, size_t unit_items, unsigned *counter, float *flush, unsigned *sink) {
    const unsigned n_units = (unsigned)(total_bytes / (unit_items * 24));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto k = stream_items24<NT, SPLIT>;
    CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10));
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        CHECK(hipMemset(counter, 0, 4));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(256), dim3(1024), 128 << 10, 0, buf, unit_items, n_units, counter, flush, sink, 128 << 10);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (it && ms < best) best = ms;
    }
    return best;
}

template <int THREADS, int MODE, bool NT, bool TAIL>
static float run(const u32x4 *buf, size_t total_bytes, size_t unit_bytes, int grid, int lds_bytes, unsigned *counter, float *flush,
                 unsigned *sink) {
    const unsigned n_units = (unsigned)(total_bytes / unit_bytes);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto k = stream_units<THREADS, MODE, NT, TAIL>;
    CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        CHECK(hipMemset(counter, 0, 4));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(grid), dim3(THREADS), lds_bytes, 0, buf, unit_bytes / 16, n_units, counter, flush, sink, lds_bytes);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (it && ms < best) best = ms;
    }
    return best;
}

int main() {
    const size_t total = (size_t)960 << 20;
    u32x4 *buf;
    unsigned *counter, *sink;
    float *flush;
    CHECK(hipMalloc(&buf, total));
    CHECK(hipMemset(buf, 1, total));
    CHECK(hipMalloc(&counter, 64));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMalloc(&flush, (size_t)1 << 30));
    const size_t units[] = {96u << 10, 128u << 10, 192u << 10, 256u << 10, 384u << 10, 512u << 10, 1u << 20, 4u << 20};
    printf("unit KiB | as shipped (1 wg/CU, 128 KiB image): nt, plain | no zero/flush tail: nt | pipelined (mode 1): nt, plain | "
           "2 wg/CU x 512 thr, 64 KiB: nt   [GB/s]\n");
    for (size_t U : units) {
        const double gb = (double)(total / U * U) / 1e9;
        float t0 = run<1024, 0, true, true>(buf, total, U, 256, 128 << 10, counter, flush, sink);
        float t1 = run<1024, 0, false, true>(buf, total, U, 256, 128 << 10, counter, flush, sink);
        float t2 = run<1024, 0, true, false>(buf, total, U, 256, 128 << 10, counter, flush, sink);
        float t3 = run<1024, 1, true, true>(buf, total, U, 256, 128 << 10, counter, flush, sink);
        float t4 = run<1024, 1, false, true>(buf, total, U, 256, 128 << 10, counter, flush, sink);
        float t5 = run<512, 0, true, true>(buf, total, U, 512, 64 << 10, counter, flush, sink);
        printf("%8zu | %7.0f %7.0f | %7.0f | %7.0f %7.0f | %7.0f\n", U >> 10, gb / t0 * 1e3, gb / t1 * 1e3, gb / t2 * 1e3, gb / t3 * 1e3,
               gb / t4 * 1e3, gb / t5 * 1e3);
    }
    printf("\n24-byte items (one round = 8 items per thread), GB/s: unit items | 16+8 per lane: nt, plain | contiguous 16-byte chunks: nt, plain\n");
    for (size_t items : {8192u, 16384u, 65536u}) {
        const double gb = (double)(total / (items * 24) * (items * 24)) / 1e9;
        float a = run24<true, 0>((const char *)buf, total, items, counter, flush, sink);
        float b = run24<false, 0>((const char *)buf, total, items, counter, flush, sink);
        float c = run24<true, 1>((const char *)buf, total, items, counter, flush, sink);
        float d = run24<false, 1>((const char *)buf, total, items, counter, flush, sink);
        printf("%8zu | %7.0f %7.0f | %7.0f %7.0f\n", items, gb / a * 1e3, gb / b * 1e3, gb / c * 1e3, gb / d * 1e3);
    }
    return 0;
}
