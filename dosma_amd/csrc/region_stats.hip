// region_stats.hip -- per-region count / mean / std / median of a quantitative map, for gfx950.
//
// the masked reductions behind
//     /root/reference/dosma/core/quant_vals.py:145-229  (QuantitativeValue.to_metrics:
//         np.nanmean / np.nanstd / np.nanmedian / count of map[label_map == key] for every label, then "total")
// which the reference evaluates with one boolean mask, one fancy-index copy and three numpy reductions per label over
// the whole volume.  Here the map and the label map are read once per pass; every pass handles all regions at once.
//
//   One streaming kernel per radix digit of the order-preserving 64-bit image of the double (exact radix selection of the two
//   middle order statistics of every region: histograms per (region, statistic) in LDS, merged with atomics; a small kernel
//   between passes picks the bin that holds the wanted rank and extends the key prefix -- no host round trip):
//     pass 1  digit 1 + count and sum per region                         -> mean, wanted ranks
//     pass 2  digit 2 + sum of squared deviations from the mean          -> std (numpy's two-pass nanstd)
//     pass 3+ the remaining digits
//   Digits are 11 bits wide when the histograms of all regions fit LDS (<= 6 regions incl. "total": 6 passes for a float64
//   map, 4 for a float32 map, whose low 29 key bits are fixed by the sign), 8 bits otherwise (8 / 5 passes).  Round 1 ran
//   2 moment passes + 8 one-byte selection passes whatever the map.
//
// Streaming: (sizeof(value) + sizeof(label)) bytes per voxel per pass.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstddef>
#include <cstdint>

#include "qmri.h"
#include "qmri_internal.h"

namespace qmri {

namespace {

constexpr int kMaxR = QMRI_MAX_REGIONS;
constexpr int kMaxBlocks = 2048;  // grid cap of the histogram kernels

struct StatsK {
    const void *values;
    const void *labels;  // nullptr: single region "total"
    int l_kind;          // 0 int32, 1 uint8, 2 int16
    long long N;
    int f64;
    int nkeys;  // labelled regions; region nkeys is "total"
    int use_bounds, closed;
    double lo, hi;
    int keys[kMaxR];
};

// per-call device state
struct StatsState {
    unsigned long long count[kMaxR];
    double sum[kMaxR];
    double mean[kMaxR];
    double ssd[kMaxR];
    unsigned long long prefix[2 * kMaxR];  // key prefix found so far, per (region, statistic)
    unsigned long long rank[2 * kMaxR];    // remaining rank inside the prefix
    unsigned int hist[2 * kMaxR][2048];  // (rows of 256 with 8-bit digits)
    // per-block partial sums of the two moment passes: added in a FIXED order by select_pick_kernel (atomicAdd of doubles -- rounds
    // 1-4 -- made the last bits of mean and standard deviation depend on the order the blocks finished in)
    double part[kMaxBlocks][kMaxR];
};

__device__ __forceinline__ double load_value(const StatsK &K, long long i) {
    return K.f64 ? static_cast<const double *>(K.values)[i] : (double)static_cast<const float *>(K.values)[i];
}

// usable voxel (finite, inside the bounds) -> region index of its label (or -1) and membership of "total"
__device__ __forceinline__ bool classify(const StatsK &K, long long i, double v, int &region, bool &total) {
    region = -1;
    total = false;
    if (!isfinite(v)) return false;
    if (K.use_bounds) {
        const bool lo_ok = (K.closed & 1) ? v >= K.lo : v > K.lo;
        const bool hi_ok = (K.closed & 2) ? v <= K.hi : v < K.hi;
        if (!(lo_ok && hi_ok)) return false;
    }
    if (!K.labels) {
        total = true;
        return true;
    }
    const int l = K.l_kind == 0 ? static_cast<const int *>(K.labels)[i]
                : K.l_kind == 1 ? (int)static_cast<const unsigned char *>(K.labels)[i]
                                : (int)static_cast<const short *>(K.labels)[i];
    if (l <= 0) {
        // the reference's label_map == key also matches keys <= 0 on voxels the bounds did not zero; "total" is label > 0
        for (int r = 0; r < K.nkeys; ++r)
            if (K.keys[r] == l && l != 0) region = r;
        return region >= 0;
    }
    total = true;
    for (int r = 0; r < K.nkeys; ++r)
        if (K.keys[r] == l) region = r;
    return true;
}

__device__ __forceinline__ unsigned long long order_key(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);  // ascending doubles <-> ascending unsigned keys
}

__device__ __forceinline__ double key_value(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

// One pass: histogram of digit `pass` (BITS wide, most significant first) of the keys that match the prefix found so far;
// MOMENT 1 adds count + sum (first pass, no prefix yet), MOMENT 2 the squared deviations from the mean (second pass).
template <int BITS, int MOMENT>
__global__ __launch_bounds__(BITS == 11 ? 1024 : 256) void select_hist_kernel(const StatsK K, StatsState *S, int pass, int shift, int width) {
    extern __shared__ unsigned int s_h[];  // [2 * nreg][1 << BITS]
    __shared__ unsigned long long s_prefix[2 * kMaxR];
    __shared__ double s_a[kMaxR], s_mean[kMaxR];
    __shared__ double s_w[16][kMaxR];  // per-wave partial sums (added in wave order: no floating-point atomics)
    __shared__ unsigned long long s_n[kMaxR];
    constexpr int NB = 1 << BITS;
    const int nreg = K.nkeys + 1;
    for (int i = threadIdx.x; i < 2 * nreg * NB; i += blockDim.x) s_h[i] = 0u;
    if (threadIdx.x < 2 * kMaxR) s_prefix[threadIdx.x] = S->prefix[threadIdx.x];
    if (threadIdx.x < kMaxR) {
        s_a[threadIdx.x] = 0.0;
        s_n[threadIdx.x] = 0ull;
        s_mean[threadIdx.x] = MOMENT == 2 ? S->mean[threadIdx.x] : 0.0;
    }
    __syncthreads();
    // bits above this digit (all of them compared with the prefix); the first pass has none
    const unsigned long long himask = pass == 0 ? 0ull : (~0ull << (shift + width));
    const unsigned int dmask = (1u << width) - 1u;  // (the last digit may be narrower than BITS)
    double acc[kMaxR];
    unsigned int cnt[kMaxR];
    if (MOMENT) {
#pragma unroll
        for (int r = 0; r < kMaxR; ++r) {
            acc[r] = 0.0;
            cnt[r] = 0u;
        }
    }
    auto consume = [&](long long i, double v) {
        int region;
        bool total;
        if (!classify(K, i, v, region, total)) return;
        if (MOMENT) {
#pragma unroll
            for (int r = 0; r < kMaxR; ++r) {
                const bool in = (r == region) || (total && r == K.nkeys);
                if (in) {
                    if (MOMENT == 1) {
                        acc[r] += v;
                        cnt[r] += 1u;
                    } else {
                        const double d = v - s_mean[r];
                        acc[r] += d * d;
                    }
                }
            }
        }
        const unsigned long long key = order_key(v);
        const unsigned int digit = (unsigned int)(key >> shift) & dmask;
        for (int which = 0; which < 2; ++which) {
            const int r = which == 0 ? region : (total ? K.nkeys : -1);
            if (r < 0) continue;
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if ((key & himask) == s_prefix[2 * r + q]) atomicAdd(&s_h[(2 * r + q) * NB + digit], 1u);
        }
    };
    // four independent value loads in flight per thread (the histograms cap the resident waves; the stream is latency-bound)
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < K.N; i += 4 * stride) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = load_value(K, i + u * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u) consume(i + u * stride, v[u]);
    }
    for (; i < K.N; i += stride) consume(i, load_value(K, i));
    if (MOMENT) {
#pragma unroll
        for (int r = 0; r < kMaxR; ++r) {
            if (r < nreg) {
                double a = acc[r];
                unsigned int c = cnt[r];
                for (int off = 32; off > 0; off >>= 1) {
                    a += __shfl_down(a, off);
                    c += __shfl_down(c, off);
                }
                if ((threadIdx.x & 63) == 0) {
                    s_w[threadIdx.x >> 6][r] = a;
                    if (MOMENT == 1) atomicAdd(&s_n[r], (unsigned long long)c);
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * nreg * NB; i += blockDim.x) {
        const unsigned int c = s_h[i];
        if (c) atomicAdd(&S->hist[i / NB][i % NB], c);
    }
    if (MOMENT && threadIdx.x < nreg) {
        double a = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) a += s_w[w][threadIdx.x];
        S->part[blockIdx.x][threadIdx.x] = a;  // (this block's share: select_pick_kernel adds the blocks' shares in block order)
        if (MOMENT == 1) atomicAdd(&S->count[threadIdx.x], s_n[threadIdx.x]);
    }
}

// One block per (region, statistic): [first pass: mean and the wanted rank,] the bin that holds the rank (block-wide scan of
// the histogram row), prefix extended by the digit, row cleared for the next pass.
template <int BITS>
__global__ __launch_bounds__(256) void select_pick_kernel(StatsState *S, int pass, int shift, int nblocks) {
    constexpr int NB = 1 << BITS, PER = NB / 256;
    __shared__ unsigned long long s_scan[256];
    __shared__ unsigned long long s_rank;
    __shared__ double s_tree[256];
    const int q = blockIdx.x, r = q >> 1, t = threadIdx.x;
    if (pass < 2) {
        // the moment of this pass: the histogram blocks' partial sums in a fixed order (thread t: blocks t, t + 256, ...; then a fixed
        // tree).  Both blocks of a region compute it (the odd one needs the count only, but keeps the barriers uniform); the even one stores
        double a = 0.0;
        for (int b = t; b < nblocks; b += 256) a += S->part[b][r];
        s_tree[t] = a;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (t < o) s_tree[t] += s_tree[t + o];
            __syncthreads();
        }
        if (t == 0 && (q & 1) == 0) {
            if (pass == 0) S->sum[r] = s_tree[0];
            else S->ssd[r] = s_tree[0];
        }
        __syncthreads();
    }
    if (t == 0) {
        if (pass == 0) {
            const unsigned long long n = S->count[r];
            if ((q & 1) == 0) S->mean[r] = n ? s_tree[0] / (double)n : NAN;
            // numpy.median: mean of elements (n-1)//2 and n//2 of the sorted values
            s_rank = n ? ((q & 1) ? n / 2 : (n - 1) / 2) : 0ull;
        } else {
            s_rank = S->rank[q];
        }
    }
    unsigned int c[PER];
    unsigned long long mine = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        c[k] = S->hist[q][t * PER + k];
        mine += c[k];
        S->hist[q][t * PER + k] = 0u;
    }
    s_scan[t] = mine;
    __syncthreads();
    // exclusive prefix of the per-thread sums (256 values: a serial scan by thread 0 is 256 LDS reads)
    if (t == 0) {
        unsigned long long run = 0;
        for (int k = 0; k < 256; ++k) {
            const unsigned long long v = s_scan[k];
            s_scan[k] = run;
            run += v;
        }
    }
    __syncthreads();
    const unsigned long long rank = s_rank, before = s_scan[t];
    const bool hit = rank >= before && rank < before + mine;
    const bool beyond = t == 255 && rank >= before + mine;  // an empty region: ends in the last bin (its result is NaN anyway)
    if (hit || beyond) {
        unsigned long long left = rank - before;
        int d = 0;
        if (beyond) {
            d = PER - 1;
            left = 0;
        } else {
            for (; d < PER - 1; ++d) {
                if (left < c[d]) break;
                left -= c[d];
            }
        }
        S->rank[q] = left;
        const unsigned long long p = pass == 0 ? 0ull : S->prefix[q];
        S->prefix[q] = p | ((unsigned long long)(t * PER + d) << shift);
    }
}

// skipped_bits: the low key bits no pass looked at (float32 maps: fixed by the sign -- zeros for a positive value's key,
// ones for a negative one's, whose key is the complement)
__global__ void finish_kernel(const StatsState *S, int nreg, int f64, int skipped_bits, double *out) {
    const int r = threadIdx.x;
    if (r >= nreg) return;
    const unsigned long long n = S->count[r];
    out[4 * r + 0] = (double)n;
    out[4 * r + 1] = n ? S->mean[r] : NAN;
    out[4 * r + 2] = n ? sqrt(S->ssd[r] / (double)n) : NAN;
    unsigned long long k0 = S->prefix[2 * r], k1 = S->prefix[2 * r + 1];
    if (skipped_bits) {
        const unsigned long long low = (1ull << skipped_bits) - 1ull;
        if (!(k0 >> 63)) k0 |= low;  // key without the top bit = complemented negative value
        if (!(k1 >> 63)) k1 |= low;
    }
    // numpy.median of a float32 map averages the two middle elements in float32
    double med = 0.5 * (key_value(k0) + key_value(k1));
    if (!f64) med = (double)(float)med;
    out[4 * r + 3] = n ? med : NAN;
}

}  // namespace

// values / labels: device pointers; out_dev: device [nkeys + 1][4]; state: device scratch of region_stats_state_bytes()
size_t region_stats_state_bytes() { return sizeof(StatsState); }

hipError_t region_stats_launch(const void *values, int f64, const void *labels, int l_kind, long long N, int nkeys,
                               const int *keys,
                               int use_bounds, double lo, double hi, int closed, void *state, double *out_dev,
                               int num_cu, hipStream_t stream) {
    if (nkeys < 0 || nkeys > kMaxR - 1) return hipErrorInvalidValue;
    StatsK K;
    K.values = values;
    K.labels = labels;
    K.l_kind = l_kind;
    K.N = N;
    K.f64 = f64;
    K.nkeys = labels ? nkeys : 0;
    K.use_bounds = use_bounds;
    K.closed = closed;
    K.lo = lo;
    K.hi = hi;
    for (int r = 0; r < kMaxR; ++r) K.keys[r] = (labels && r < nkeys) ? keys[r] : 0;
    StatsState *S = static_cast<StatsState *>(state);
    const int nreg = K.nkeys + 1;
    long long blocks = (N + 256 * 8 - 1) / (256 * 8);
    if (blocks > (long long)num_cu * 8) blocks = (long long)num_cu * 8;
    if (blocks > kMaxBlocks) blocks = kMaxBlocks;
    if (blocks < 1) blocks = 1;
    // (the per-block partial sums `part` are written by every block before select_pick_kernel reads [0, nblocks): not cleared)
    hipError_t e = hipMemsetAsync(S, 0, offsetof(StatsState, part), stream);
    if (e != hipSuccess) return e;
    (void)hipGetLastError();
    // digit width: 11 bits when the LDS histograms of all (region, statistic) pairs fit next to the other kernels' share
    const bool wide = 2 * nreg * 2048 * 4 <= 96 * 1024;
    const int bits = wide ? 11 : 8;
    const size_t lds = (size_t)2 * nreg * (1u << bits) * 4;
    // float32 maps: key bits 28..0 are fixed by the sign; passes that would only look at them are skipped
    const int low_fixed = f64 ? 0 : 29;
    int shift = 64, pass = 0, skipped = 0;
    if (wide) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(select_hist_kernel<11, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(select_hist_kernel<11, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(select_hist_kernel<11, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    while (shift > 0) {
        const int width = shift < bits ? shift : bits;  // (64 = 5 * 11 + 9: the last 11-bit digit is 9 bits wide, its top bins stay empty)
        shift -= width;
        if (shift + width <= low_fixed && pass >= 2) {  // nothing but fixed bits left (the two moment passes always run)
            skipped = shift + width;
            break;
        }
        const dim3 g(wide ? (unsigned)(blocks < 2LL * num_cu ? blocks : 2LL * num_cu) : (unsigned)blocks), b(wide ? 1024 : 256);
        if (wide) {
            if (pass == 0) hipLaunchKernelGGL((select_hist_kernel<11, 1>), g, b, lds, stream, K, S, pass, shift, width);
            else if (pass == 1) hipLaunchKernelGGL((select_hist_kernel<11, 2>), g, b, lds, stream, K, S, pass, shift, width);
            else hipLaunchKernelGGL((select_hist_kernel<11, 0>), g, b, lds, stream, K, S, pass, shift, width);
            hipLaunchKernelGGL(select_pick_kernel<11>, dim3(2 * nreg), dim3(256), 0, stream, S, pass, shift, (int)g.x);
        } else {
            if (pass == 0) hipLaunchKernelGGL((select_hist_kernel<8, 1>), g, b, lds, stream, K, S, pass, shift, width);
            else if (pass == 1) hipLaunchKernelGGL((select_hist_kernel<8, 2>), g, b, lds, stream, K, S, pass, shift, width);
            else hipLaunchKernelGGL((select_hist_kernel<8, 0>), g, b, lds, stream, K, S, pass, shift, width);
            hipLaunchKernelGGL(select_pick_kernel<8>, dim3(2 * nreg), dim3(256), 0, stream, S, pass, shift, (int)g.x);
        }
        ++pass;
    }
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(64), 0, stream, S, nreg, f64, skipped, out_dev);
    return hipGetLastError();
}

}  // namespace qmri
