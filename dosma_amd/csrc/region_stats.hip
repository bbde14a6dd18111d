// region_stats.hip -- per-region count / mean / std / median of a quantitative map, for gfx950.
//
// the masked reductions behind
//     /root/reference/dosma/core/quant_vals.py:145-229  (QuantitativeValue.to_metrics:
//         np.nanmean / np.nanstd / np.nanmedian / count of map[label_map == key] for every label, then "total")
// which the reference evaluates with one boolean mask, one fancy-index copy and three numpy reductions per label over
// the whole volume.  Here the map and the label map are read once per pass; every pass handles all regions at once.
//
//   pass 1  count and sum per region                      -> mean
//   pass 2  sum of squared deviations from the mean       -> std (numpy's two-pass nanstd)
//   8 selection passes (one byte of the key each, most significant first): exact radix selection of the two middle
//           order statistics of every region on the order-preserving 64-bit image of the double; histograms of 256
//           bins per (region, statistic) in LDS, merged with atomics; a one-block kernel between passes picks the
//           bin that holds the wanted rank and extends the key prefix -- no host round trip.
//
// Streaming: (sizeof(value) + sizeof(label)) bytes per voxel per pass, 10 passes.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "qmri.h"
#include "qmri_internal.h"

namespace qmri {

namespace {

constexpr int kMaxR = QMRI_MAX_REGIONS;

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
    unsigned int hist[2 * kMaxR][256];
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

template <int PASS>  // 1: count + sum, 2: squared deviations
__global__ __launch_bounds__(256) void moments_kernel(const StatsK K, StatsState *S) {
    __shared__ double s_a[kMaxR], s_mean[kMaxR];
    __shared__ unsigned long long s_n[kMaxR];
    const int nreg = K.nkeys + 1;
    if (threadIdx.x < kMaxR) {
        s_a[threadIdx.x] = 0.0;
        s_n[threadIdx.x] = 0ull;
        s_mean[threadIdx.x] = PASS == 2 ? S->mean[threadIdx.x] : 0.0;
    }
    __syncthreads();
    double acc[kMaxR];
    unsigned int cnt[kMaxR];
#pragma unroll
    for (int r = 0; r < kMaxR; ++r) {
        acc[r] = 0.0;
        cnt[r] = 0u;
    }
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < K.N; i += stride) {
        const double v = load_value(K, i);
        int region;
        bool total;
        if (!classify(K, i, v, region, total)) continue;
#pragma unroll
        for (int r = 0; r < kMaxR; ++r) {
            const bool in = (r == region) || (total && r == K.nkeys);
            if (in) {
                if (PASS == 1) {
                    acc[r] += v;
                    cnt[r] += 1u;
                } else {
                    const double d = v - s_mean[r];
                    acc[r] += d * d;
                }
            }
        }
    }
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
                atomicAdd(&s_a[r], a);
                if (PASS == 1) atomicAdd(&s_n[r], (unsigned long long)c);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < nreg) {
        if (PASS == 1) {
            atomicAdd(&S->sum[threadIdx.x], s_a[threadIdx.x]);
            atomicAdd(&S->count[threadIdx.x], s_n[threadIdx.x]);
        } else {
            atomicAdd(&S->ssd[threadIdx.x], s_a[threadIdx.x]);
        }
    }
}

// after pass 1: means and the two wanted ranks of every region
__global__ void after_moments_kernel(StatsState *S, int nreg) {
    const int r = threadIdx.x;
    if (r >= nreg) return;
    const unsigned long long n = S->count[r];
    S->mean[r] = n ? S->sum[r] / (double)n : NAN;
    S->prefix[2 * r] = S->prefix[2 * r + 1] = 0ull;
    S->rank[2 * r] = n ? (n - 1) / 2 : 0ull;  // numpy.median: mean of elements (n-1)//2 and n//2 of the sorted values
    S->rank[2 * r + 1] = n / 2;
}

// one radix-selection pass: byte `pass` (0 = most significant) of the keys that match the prefix found so far
__global__ __launch_bounds__(256) void select_hist_kernel(const StatsK K, StatsState *S, int pass) {
    __shared__ unsigned int s_h[2 * kMaxR][256];
    __shared__ unsigned long long s_prefix[2 * kMaxR];
    const int nreg = K.nkeys + 1;
    for (int i = threadIdx.x; i < 2 * kMaxR * 256; i += blockDim.x) (&s_h[0][0])[i] = 0u;
    if (threadIdx.x < 2 * kMaxR) s_prefix[threadIdx.x] = S->prefix[threadIdx.x];
    __syncthreads();
    const int shift = 56 - 8 * pass;
    const unsigned long long himask = pass == 0 ? 0ull : (~0ull << (shift + 8));
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < K.N; i += stride) {
        const double v = load_value(K, i);
        int region;
        bool total;
        if (!classify(K, i, v, region, total)) continue;
        const unsigned long long key = order_key(v);
        const unsigned int digit = (unsigned int)(key >> shift) & 255u;
        for (int which = 0; which < 2; ++which) {
            const int r = which == 0 ? region : (total ? K.nkeys : -1);
            if (r < 0) continue;
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if ((key & himask) == s_prefix[2 * r + q]) atomicAdd(&s_h[2 * r + q][digit], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * nreg * 256; i += blockDim.x) {
        const unsigned int c = (&s_h[0][0])[i];
        if (c) atomicAdd(&(&S->hist[0][0])[i], c);
    }
}

// pick the bin holding the wanted rank, extend the prefix, clear the histograms for the next pass
__global__ void select_pick_kernel(StatsState *S, int nreg, int pass) {
    const int q = threadIdx.x;  // (region, statistic)
    if (q < 2 * nreg) {
        const int shift = 56 - 8 * pass;
        unsigned long long rank = S->rank[q];
        unsigned int d = 0;
        for (; d < 255; ++d) {
            const unsigned int c = S->hist[q][d];
            if (rank < c) break;
            rank -= c;
        }
        S->rank[q] = rank;
        S->prefix[q] |= (unsigned long long)d << shift;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * kMaxR * 256; i += blockDim.x) (&S->hist[0][0])[i] = 0u;
}

__global__ void finish_kernel(const StatsState *S, int nreg, int f64, double *out) {
    const int r = threadIdx.x;
    if (r >= nreg) return;
    const unsigned long long n = S->count[r];
    out[4 * r + 0] = (double)n;
    out[4 * r + 1] = n ? S->mean[r] : NAN;
    out[4 * r + 2] = n ? sqrt(S->ssd[r] / (double)n) : NAN;
    // numpy.median of a float32 map averages the two middle elements in float32
    double med = 0.5 * (key_value(S->prefix[2 * r]) + key_value(S->prefix[2 * r + 1]));
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
    if (blocks < 1) blocks = 1;
    hipError_t e = hipMemsetAsync(S, 0, sizeof(StatsState), stream);
    if (e != hipSuccess) return e;
    (void)hipGetLastError();
    hipLaunchKernelGGL(moments_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, K, S);
    hipLaunchKernelGGL(after_moments_kernel, dim3(1), dim3(64), 0, stream, S, nreg);
    hipLaunchKernelGGL(moments_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, stream, K, S);
    for (int pass = 0; pass < 8; ++pass) {
        hipLaunchKernelGGL(select_hist_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, K, S, pass);
        hipLaunchKernelGGL(select_pick_kernel, dim3(1), dim3(256), 0, stream, S, nreg, pass);
    }
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(64), 0, stream, S, nreg, f64, out_dev);
    return hipGetLastError();
}

}  // namespace qmri
