// symbols.hip -- integer symbol statistics of the rounded latents (gfx950) + the host-side range coder.
//
// Reference: LatentGrid.size (wisp/models/grids/latent_grid.py:138-174) rounds every latent channel, runs
// torch.unique(return_counts=True) (a device-wide sort, once per epoch and per channel) and turns the counts into an
// entropy estimate or, with torchac, into an arithmetic-coded byte stream. Here:
//   symbol_range      per channel min / max of (int)rint(latent)            one streaming pass, int32 atomics
//   symbol_histogram  per channel counts over [lo, lo + nbins)              one streaming pass, LDS-private bins
//   range coder       static-model byte-wise range coder on the HOST (the reference also codes on the CPU:
//                     `cdf.detach().cpu()`); carry-propagating 40-bit range / 16-bit frequency scale.
// Integer work: results must be exact.
#include <vector>

#include "internal.h"

namespace shacira {

constexpr int kSymMaxLD = 16;
constexpr int kSymThreads = 256;
constexpr int kSymLdsBins = 12288;   // 48 KiB of private uint32 bins per workgroup

__device__ __forceinline__ int32_t to_symbol(float v) {
    // torch.round (half to even) then .long(); clamped so that lo/hi arithmetic cannot overflow. NaN -> 0.
    const float r = rintf(v);
    const float c = fminf(fmaxf(r, -1073741824.0f), 1073741824.0f);
    return (c == c) ? (int32_t)c : 0;
}

__global__ void symbol_range_init_kernel(int32_t *__restrict__ minmax, int ld) {
    const int c = threadIdx.x;
    if (c < ld) {
        minmax[2 * c] = INT32_MAX;
        minmax[2 * c + 1] = INT32_MIN;
    }
}

__global__ __launch_bounds__(kSymThreads) void symbol_range_kernel(const float *__restrict__ latent, int64_t rows,
                                                                   int ld, int32_t *__restrict__ minmax) {
    __shared__ int32_t s_mm[2 * kSymMaxLD];
    if (threadIdx.x < 2 * kSymMaxLD) s_mm[threadIdx.x] = (threadIdx.x & 1) ? INT32_MIN : INT32_MAX;
    __syncthreads();
    // flat walk: consecutive threads read consecutive floats; a thread's channel changes by (stride % ld) per step
    const int64_t total = rows * ld;
    const int64_t stride = (int64_t)gridDim.x * kSymThreads;
    int32_t mn[kSymMaxLD], mx[kSymMaxLD];
#pragma unroll
    for (int c = 0; c < kSymMaxLD; ++c) {
        mn[c] = INT32_MAX;
        mx[c] = INT32_MIN;
    }
    for (int64_t e = (int64_t)blockIdx.x * kSymThreads + threadIdx.x; e < total; e += stride) {
        const int c = (int)(e % ld);
        const int32_t s = to_symbol(latent[e]);
#pragma unroll
        for (int k = 0; k < kSymMaxLD; ++k)   // register-resident select instead of a dynamically indexed array
            if (k == c) {
                mn[k] = s < mn[k] ? s : mn[k];
                mx[k] = s > mx[k] ? s : mx[k];
            }
    }
#pragma unroll
    for (int c = 0; c < kSymMaxLD; ++c) {
        if (c < ld && mn[c] <= mx[c]) {
            atomicMin(&s_mm[2 * c], mn[c]);
            atomicMax(&s_mm[2 * c + 1], mx[c]);
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * ld) {
        const int32_t v = s_mm[threadIdx.x];
        if (threadIdx.x & 1) { if (v != INT32_MIN) atomicMax(&minmax[threadIdx.x], v); }
        else                 { if (v != INT32_MAX) atomicMin(&minmax[threadIdx.x], v); }
    }
}

// counts[c][s - lo[c]] += 1 ; symbols outside [lo, lo + nbins) are ignored (cannot happen with lo/nbins from
// symbol_range). LDS-private bins when ld * nbins fits, flushed with one global atomic per non-empty bin.
template <bool LDS>
__global__ __launch_bounds__(kSymThreads) void symbol_histogram_kernel(const float *__restrict__ latent, int64_t rows,
                                                                       int ld, const int32_t *__restrict__ minmax,
                                                                       int nbins,
                                                                       unsigned long long *__restrict__ counts) {
    extern __shared__ uint32_t s_bins[];
    __shared__ int32_t s_lo[kSymMaxLD];
    if ((int)threadIdx.x < ld) s_lo[threadIdx.x] = minmax[2 * threadIdx.x];
    const int nb_all = ld * nbins;
    if constexpr (LDS)
        for (int k = threadIdx.x; k < nb_all; k += kSymThreads) s_bins[k] = 0;
    __syncthreads();
    const int64_t total = rows * ld;
    const int64_t stride = (int64_t)gridDim.x * kSymThreads;
    for (int64_t e = (int64_t)blockIdx.x * kSymThreads + threadIdx.x; e < total; e += stride) {
        const int c = (int)(e % ld);
        const int64_t b = (int64_t)to_symbol(latent[e]) - (int64_t)s_lo[c];
        if (b < 0 || b >= nbins) continue;
        if constexpr (LDS) atomicAdd(&s_bins[c * nbins + (int)b], 1u);
        else atomicAdd(&counts[(size_t)c * nbins + (size_t)b], 1ull);
    }
    if constexpr (LDS) {
        __syncthreads();
        for (int k = threadIdx.x; k < nb_all; k += kSymThreads) {
            const uint32_t v = s_bins[k];
            if (v) atomicAdd(&counts[k], (unsigned long long)v);
        }
    }
}

static uint32_t stream_blocks(int64_t total) {
    int64_t b = (total + kSymThreads * 8 - 1) / (kSymThreads * 8);
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (uint32_t)b;
}

bool symbols_supported(int ld) { return ld >= 1 && ld <= kSymMaxLD; }

hipError_t symbol_range_launch(const float *latent, int64_t rows, int ld, int32_t *minmax, hipStream_t s) {
    hipLaunchKernelGGL(symbol_range_init_kernel, dim3(1), dim3(64), 0, s, minmax, ld);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || rows == 0) return e;
    hipLaunchKernelGGL(symbol_range_kernel, dim3(stream_blocks(rows * ld)), dim3(kSymThreads), 0, s, latent, rows, ld,
                       minmax);
    return hipGetLastError();
}

hipError_t symbol_histogram_launch(const float *latent, int64_t rows, int ld, const int32_t *minmax, int nbins,
                                   uint64_t *counts, hipStream_t s) {
    hipError_t e = hipMemsetAsync(counts, 0, (size_t)ld * nbins * sizeof(uint64_t), s);
    if (e != hipSuccess || rows == 0) return e;
    const uint32_t blocks = stream_blocks(rows * ld);
    auto *c = reinterpret_cast<unsigned long long *>(counts);
    if ((int64_t)ld * nbins <= kSymLdsBins)
        hipLaunchKernelGGL(symbol_histogram_kernel<true>, dim3(blocks), dim3(kSymThreads),
                           (size_t)ld * nbins * sizeof(uint32_t), s, latent, rows, ld, minmax, nbins, c);
    else
        hipLaunchKernelGGL(symbol_histogram_kernel<false>, dim3(blocks), dim3(kSymThreads), 0, s, latent, rows, ld,
                           minmax, nbins, c);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ range coder (host)
// Byte-wise range coder with carry propagation (the construction used by LZMA's rc, generalised from binary to
// multi-symbol models): 40-bit range (renormalised byte-wise above 2^32, so range / 2^16 keeps >= 16 significant bits
// and the truncation loss stays below 3e-5 bit per symbol -- it matters for near-deterministic channels), 41-bit low
// with carry, frequencies scaled to a total of 2^16 (every coded symbol has freq >= 1). Static model: the caller
// supplies the frequency table, which also goes into the container.
constexpr uint32_t kRcTotalBits = 16;
constexpr uint64_t kRcRangeInit = (1ull << 40) - 1;
constexpr uint64_t kRcTop = 1ull << 32;
constexpr uint64_t kRcMask40 = (1ull << 40) - 1;

int rc_check_model(const uint32_t *freq, int nsym) {
    if (!freq || nsym < 1 || nsym > (1 << kRcTotalBits)) return SHACIRA_EINVAL;
    uint64_t sum = 0;
    for (int k = 0; k < nsym; ++k) sum += freq[k];
    return sum == (1u << kRcTotalBits) ? 0 : SHACIRA_EINVAL;
}

size_t rc_encode_bound(int64_t n) { return (size_t)n * 2 + 16; }  // >= 16 bits per symbol is the worst case (freq 1)

int rc_encode(const int32_t *sym, int64_t n, const uint32_t *freq, int nsym, uint8_t *out, size_t cap, size_t *len) {
    int rc = rc_check_model(freq, nsym);
    if (rc) return rc;
    if (n < 0 || (n > 0 && !sym) || !out || !len) return SHACIRA_EINVAL;
    std::vector<uint32_t> cum((size_t)nsym + 1, 0);
    for (int k = 0; k < nsym; ++k) cum[k + 1] = cum[k] + freq[k];
    uint64_t low = 0;
    uint64_t range = kRcRangeInit;
    uint8_t cache = 0;
    uint64_t cache_size = 1;
    size_t pos = 0;
    bool overflow = false;
    auto put = [&](uint8_t b) {
        if (pos < cap) out[pos] = b; else overflow = true;
        ++pos;
    };
    auto shift_low = [&]() {
        if ((low & kRcMask40) < (0xFFull << 32) || (low >> 40) != 0) {
            const uint8_t carry = (uint8_t)(low >> 40);
            uint8_t temp = cache;
            do {
                put((uint8_t)(temp + carry));
                temp = 0xFF;
            } while (--cache_size != 0);
            cache = (uint8_t)((low >> 32) & 0xFF);
        }
        ++cache_size;
        low = (low & 0xFFFFFFFFull) << 8;
    };
    for (int64_t i = 0; i < n; ++i) {
        const int32_t s = sym[i];
        if (s < 0 || s >= nsym || freq[s] == 0) return SHACIRA_EINVAL;   // symbol without code space
        const uint64_t r = range >> kRcTotalBits;
        low += r * cum[s];
        range = r * freq[s];
        while (range < kRcTop) {
            range <<= 8;
            shift_low();
        }
    }
    for (int k = 0; k < 6; ++k) shift_low();
    *len = pos;
    return overflow ? SHACIRA_EWORKSPACE : 0;
}

int rc_decode(const uint8_t *in, size_t len, const uint32_t *freq, int nsym, int64_t n, int32_t *sym) {
    int rc = rc_check_model(freq, nsym);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!sym || !in))) return SHACIRA_EINVAL;
    std::vector<uint32_t> cum((size_t)nsym + 1, 0);
    for (int k = 0; k < nsym; ++k) cum[k + 1] = cum[k] + freq[k];
    std::vector<uint16_t> lookup((size_t)1 << kRcTotalBits);
    if (nsym > 65536) return SHACIRA_EINVAL;
    for (int k = 0; k < nsym; ++k)
        for (uint32_t v = cum[k]; v < cum[k + 1]; ++v) lookup[v] = (uint16_t)k;
    size_t pos = 0;
    auto next = [&]() -> uint64_t { return pos < len ? in[pos++] : (pos++, 0u); };
    uint64_t code = 0, range = kRcRangeInit;
    (void)next();   // the encoder's first byte is its initial cache (always 0)
    for (int k = 0; k < 5; ++k) code = (code << 8) | next();
    for (int64_t i = 0; i < n; ++i) {
        const uint64_t r = range >> kRcTotalBits;
        uint64_t v = code / r;
        if (v >= (1u << kRcTotalBits)) v = (1u << kRcTotalBits) - 1;
        const uint32_t s = lookup[v];
        code -= r * cum[s];
        range = r * freq[s];
        while (range < kRcTop) {
            code = (code << 8) | next();
            range <<= 8;
        }
        sym[i] = (int32_t)s;
    }
    return pos <= len + 5 ? 0 : SHACIRA_EINVAL;   // ran past the stream: truncated / wrong model
}

}  // namespace shacira
