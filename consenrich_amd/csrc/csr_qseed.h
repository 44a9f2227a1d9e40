// csr_qseed.h -- SURVEY 8(f) rank 4: the initial process-noise (Q0) seed on the device-resident matrices
// (reference: src/consenrich/cconsenrich.pyx:1441-1797 `cEstimateSameTrackProcessNoiseTransitions`, :1800-1902
// `cEstimatePooledProcessNoiseTransitions`; caller core.py:3621-3780 `_estimateInitialProcessNoiseFromData`).
//
// The reference converts both (m, n) matrices to float64 and builds an (m, n) activity mask on the host before its
// natives look at <= 32 000 evenly spaced transitions.  Here the matrices already live in HBM (float32, the batch's
// layout); the kernels touch only the sampled columns:
//   k_qs_count        one thread per scanned transition: active same-track pairs, first validation error
//   k_qs_scan         exclusive prefix of the pair counts (one workgroup) -> pair ordinals in the reference's scan order
//   k_qs_sample       one thread per precision-sample slot: the pair with the reference's ordinal (pyx:1431-1438)
//   k_qs_transitions  one thread per scanned transition: Huber location of the per-track differences / levels
//                     (pyx:1347-1393; medians by rank selection, no sort), sampling variance, effective pair count
//   k_qs_pooled       fallback (pyx:1845-1871): per-bin precision-weighted mean / variance
//   k_qs_hist         last-resort fallback (core.py:3705-3719): radix histogram for the exact median of the active
//                     observation variances (order statistics of the float32 variances; obsVar is monotone in them)
// All arithmetic is fp64 in the reference's order (IEEE division, no contraction): the tracks are bit-identical.
// The float64 source (`src64`) serves the reference-shaped natives, which take float64 matrices + a uint8 mask.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace csr {

struct QsArgs {
    int64_t m, n, stride;           // n bins, row stride in elements
    int src64;                      // 0: float32 data / munc (+ pad, activity derived like core.py:2989-3004); 1: float64 + mask
    const float *data, *munc;
    double pad, maskedHalf;         // maskedHalf = 0.5 * float32(1e30)
    const double *data64, *obs64;
    const uint8_t *act;
    int64_t scanCount, maxT;
    int capped;
    int32_t *cnt;                   // [scanCount] active pairs per scanned transition
    unsigned long long *firstErr;   // min over pairs of (ordinal << 8 | code), ~0 = none
    int64_t *prefix;                // [scanCount] exclusive prefix of cnt
    int64_t *pairCount;             // device scalar
    int64_t nPairs, sampleCount;    // host copies for k_qs_sample
    double *raw;                    // [sampleCount]
    double cap;
    double *deltas, *svar, *weights, *sig;   // [scanCount]
    int32_t *cappedCnt;             // [scanCount] pairs whose raw precision exceeds the cap
    double *work;                   // [3 * m * scanCount] per-thread columns (difference, level, precision)
    double *pooledMean, *pooledVar; // [n]
};

__device__ __forceinline__ int64_t qs_sample_index(int64_t i, int64_t items, int64_t samples) {   // pyx:1431-1438
    return (int64_t)floor((((double)i + 0.5) * (double)items) / (double)samples);
}
__device__ __forceinline__ bool qs_finite(double v) { return fabs(v) <= 1.7976931348623157e308; }   // false for NaN

__device__ __forceinline__ bool qs_active(const QsArgs &a, int64_t j, int64_t k) {
    const int64_t i = j * a.stride + k;
    if (a.src64) return a.act[i] != 0;
    const double z = (double)a.data[i], v = (double)a.munc[i], o = v + a.pad;            // core.py:2998-3003
    return qs_finite(z) && qs_finite(v) && v < a.maskedHalf && qs_finite(o) && o > 0.0;
}
__device__ __forceinline__ double qs_z(const QsArgs &a, int64_t j, int64_t k) {
    const int64_t i = j * a.stride + k;
    return a.src64 ? a.data64[i] : (double)a.data[i];
}
__device__ __forceinline__ double qs_obs(const QsArgs &a, int64_t j, int64_t k) {
    const int64_t i = j * a.stride + k;
    if (a.src64) return a.obs64[i];
    const double o = (double)a.munc[i] + a.pad;                                           // core.py:3647-3653
    return o > 1.0e-12 ? o : 1.0e-12;
}
__device__ __forceinline__ int64_t qs_column(const QsArgs &a, int64_t si) {
    return a.capped ? qs_sample_index(si, a.maxT, a.scanCount) : si;
}

// pyx:1574-1598 / 1627-1655: pair counts + the validation of every active pair; the error the reference would raise
// is the one of the first offending pair in its scan order
__global__ __launch_bounds__(256) void k_qs_count(QsArgs a) {
    const int64_t si = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (si >= a.scanCount) return;
    const int64_t k = qs_column(a, si);
    int c = 0;
    for (int64_t j = 0; j < a.m; ++j) {
        if (!(qs_active(a, j, k) && qs_active(a, j, k + 1))) continue;
        const double dl = qs_z(a, j, k), dr = qs_z(a, j, k + 1), ol = qs_obs(a, j, k), orr = qs_obs(a, j, k + 1);
        int code = 0;
        if (!qs_finite(dl) || !qs_finite(dr)) code = 1;
        else if (!qs_finite(ol) || !qs_finite(orr) || ol <= 0.0 || orr <= 0.0) code = 2;
        else {
            const double diff = dr - dl, rd = ol + orr;
            if (!qs_finite(diff) || !qs_finite(rd) || rd <= 0.0) code = 3;
            else {
                const double rp = 1.0 / rd;
                if (!qs_finite(rp) || rp <= 0.0) code = 4;
            }
        }
        if (code) atomicMin(a.firstErr, ((unsigned long long)(si * a.m + j) << 8) | (unsigned long long)code);
        ++c;
    }
    a.cnt[si] = c;
}

// exclusive prefix of cnt (one workgroup of 1024 threads; thread t owns a contiguous chunk)
__global__ __launch_bounds__(1024) void k_qs_scan(QsArgs a) {
    __shared__ int64_t part[1024];
    const int t = threadIdx.x;
    const int64_t chunk = (a.scanCount + 1023) / 1024;
    const int64_t lo = (int64_t)t * chunk, hi = (lo + chunk < a.scanCount) ? lo + chunk : a.scanCount;
    int64_t s = 0;
    for (int64_t i = lo; i < hi; ++i) s += a.cnt[i];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int64_t v = (t >= off) ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int64_t run = part[t] - s;
    for (int64_t i = lo; i < hi; ++i) {
        a.prefix[i] = run;
        run += a.cnt[i];
    }
    if (t == 1023) *a.pairCount = part[1023];
}

// pyx:1603-1626 (capped: every slot takes the pair whose ordinal is the slot's midpoint index) / 1654 (all pairs)
__global__ __launch_bounds__(256) void k_qs_sample(QsArgs a) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= a.sampleCount) return;
    const int64_t target = a.capped ? qs_sample_index(s, a.nPairs, a.sampleCount) : s;
    int64_t lo = 0, hi = a.scanCount;                    // last si with prefix[si] <= target
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (a.prefix[mid] <= target) lo = mid;
        else hi = mid;
    }
    const int64_t k = qs_column(a, lo);
    int64_t r = target - a.prefix[lo];
    double out = 0.0;
    for (int64_t j = 0; j < a.m; ++j) {
        if (!(qs_active(a, j, k) && qs_active(a, j, k + 1))) continue;
        if (r == 0) {
            out = 1.0 / (qs_obs(a, j, k) + qs_obs(a, j, k + 1));
            break;
        }
        --r;
    }
    a.raw[s] = out;
}

// order statistic `rank` of f(v_0..v_{n-1}) by counting (ties broken by index): no scratch, O(n^2) for n = tracks
template <class F>
__device__ __forceinline__ double qs_select(int n, int rank, F f) {
    for (int i = 0; i < n; ++i) {
        const double vi = f(i);
        int r = 0;
        for (int j = 0; j < n; ++j) {
            const double vj = f(j);
            r += (vj < vi) || (vj == vi && j < i);
        }
        if (r == rank) return vi;
    }
    return f(0);
}
template <class F>
__device__ __forceinline__ double qs_median(int n, F f) {                                  // pyx:1257-1291 at q = 0.5
    const double pos = 0.5 * (double)(n - 1);
    const int lo = (int)floor(pos);
    int hi = lo + 1;
    if (hi >= n) hi = n - 1;
    const double frac = pos - (double)lo;
    const double lowVal = qs_select(n, lo, f);
    if (hi == lo) return lowVal;
    const double highVal = qs_select(n, hi, f);
    return lowVal + frac * (highVal - lowVal);
}
// pyx:1347-1393
__device__ __forceinline__ double qs_robust_location(const double *v, const double *w, int64_t st, int n) {
    if (n == 1) return v[0];
    double loc = qs_median(n, [&](int i) { return v[(int64_t)i * st]; });
    const double loc0 = loc;
    const double scale = 1.4826 * qs_median(n, [&](int i) { return fabs(v[(int64_t)i * st] - loc0); });
    if (scale <= 1.0e-12) return loc;
    const double c = 1.345;
    for (int it = 0; it < 4; ++it) {
        double denom = 0.0, numer = 0.0;
        for (int i = 0; i < n; ++i) {
            const double x = v[(int64_t)i * st];
            const double resid = x - loc;
            double huber = (c * scale) / fmax(fabs(resid), 1.0e-12);
            if (huber > 1.0) huber = 1.0;
            const double eff = w[(int64_t)i * st] * huber;
            denom += eff;
            numer += eff * x;
        }
        if (denom <= 0.0) break;
        const double nextLoc = numer / denom;
        if (fabs(nextLoc - loc) <= 1.0e-10 * fmax(1.0, fabs(loc))) {
            loc = nextLoc;
            break;
        }
        loc = nextLoc;
    }
    return loc;
}

// pyx:1683-1731.  LDS = true: 64 threads per workgroup, the three per-thread columns (difference, level, precision) live
// in LDS ([element][thread], 3 * m * 512 bytes, m <= 64); otherwise in a global work area (any m).
template <bool LDS>
__global__ __launch_bounds__(LDS ? 64 : 256) void k_qs_transitions(QsArgs a) {
    extern __shared__ double qs_lds[];
    constexpr int TPB = LDS ? 64 : 256;
    const int64_t si = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (si >= a.scanCount) return;
    const int64_t k = qs_column(a, si);
    const int64_t st = LDS ? 64 : a.scanCount;           // column layout: element i of this thread at [i * st]
    double *ld = LDS ? qs_lds + threadIdx.x : a.work + si;
    double *ll = ld + a.m * st, *lp = ll + a.m * st;
    int cnt = 0, cappedPairs = 0;
    for (int64_t j = 0; j < a.m; ++j) {
        if (!(qs_active(a, j, k) && qs_active(a, j, k + 1))) continue;
        const double ol = qs_obs(a, j, k), orr = qs_obs(a, j, k + 1), zl = qs_z(a, j, k), zr = qs_z(a, j, k + 1);
        const double rawp = 1.0 / (ol + orr);
        double prec = rawp;
        if (a.cap > 0.0 && rawp > a.cap) {
            ++cappedPairs;
            prec = a.cap;
        }
        const double rd = ol + orr;
        ld[(int64_t)cnt * st] = zr - zl;
        ll[(int64_t)cnt * st] = (orr / rd) * zl + (ol / rd) * zr;
        lp[(int64_t)cnt * st] = prec;
        ++cnt;
    }
    a.cnt[si] = cnt;
    a.cappedCnt[si] = cappedPairs;
    if (cnt <= 0) return;
    const double loc = qs_robust_location(ld, lp, st, cnt);
    const double lev = qs_robust_location(ll, lp, st, cnt);
    double sumP = 0.0, sumP2 = 0.0;
    for (int j = 0; j < cnt; ++j) {
        const double pj = lp[(int64_t)j * st];
        sumP += pj;
        sumP2 += pj * pj;
    }
    double eff = sumP2 > 0.0 ? (sumP * sumP) / sumP2 : 1.0;
    if (eff < 1.0) eff = 1.0;
    a.deltas[si] = loc;
    a.svar[si] = 1.0 / sumP;
    a.weights[si] = eff;
    a.sig[si] = lev;
}

// pyx:1845-1871: pooled mean / variance per bin (NaN where no active track); error = first offending bin
__global__ __launch_bounds__(256) void k_qs_pooled(QsArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    double wsum = 0.0, zsum = 0.0;
    for (int64_t j = 0; j < a.m; ++j) {
        if (!qs_active(a, j, i)) continue;
        const double v = qs_z(a, j, i), o = qs_obs(a, j, i);
        if (!qs_finite(v) || !qs_finite(o) || o <= 0.0) {
            atomicMin(a.firstErr, ((unsigned long long)i << 8) | 6ull);
            continue;
        }
        const double w = 1.0 / o;
        wsum += w;
        zsum += v * w;
    }
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    a.pooledMean[i] = wsum > 0.0 ? zsum / wsum : nan;
    a.pooledVar[i] = wsum > 0.0 ? 1.0 / wsum : nan;
}

// order-preserving key of a float32
__device__ __forceinline__ unsigned qs_key(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// one radix pass of the exact-median selection over the ACTIVE float32 variances of a chain: histogram of the digit at
// `shift` among the keys whose higher digits equal `prefix` (mask = those digits)
__global__ __launch_bounds__(256) void k_qs_hist(QsArgs a, int shift, unsigned mask, unsigned prefix,
                                                 unsigned long long *hist) {
    __shared__ unsigned lh[256];
    lh[threadIdx.x] = 0;
    __syncthreads();
    const int64_t total = a.m * a.n;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t j = e / a.n, k = e - j * a.n;
        if (!qs_active(a, j, k)) continue;
        const unsigned key = qs_key(a.munc[j * a.stride + k]);
        if ((key & mask) == prefix) atomicAdd(&lh[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)lh[threadIdx.x]);
}

}  // namespace csr
