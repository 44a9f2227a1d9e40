// csr_gain.h -- the final forward gain summary of `runConsenrich`'s run diagnostics (core.py:7671-7731) on the device.
// Per replicate j of one chain: g_k = max(Pf00_k, 0) clip(lambda_k) / max(munc_jk + pad, 1e-12) in float64 (the reference's
// expression, IEEE division), over the finite g: count, mean, standard deviation (two passes, like np.std) and the SIX order
// statistics around the 25 / 50 / 75 % positions (the host interpolates them exactly as NumPy does).  The reference sorts m rows
// of n float64 values on the host (0.2 s of a 0.65-s chr1 x 32 call, round 5); here: a byte-wise radix SELECT -- 8 passes over
// the row, each a 256-bin histogram of the keys that still match a target's prefix -- exact, no sort, no (m, n) float64 matrix.
// Included by csr_lib.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace csr {

constexpr int GAIN_T = 6;           // order statistics per row: (lo, hi) of q = 0.25, 0.5, 0.75
constexpr int GAIN_OUT = 9;         // per row: count, mean, sd, then the six order statistics
constexpr int GAIN_ITEMS = 16;      // elements per thread and pass

struct GainArgs {
    const float *Pf;                // reference layout, `comps` floats per bin, P00 first
    const float *lam;               // reference layout or nullptr
    const float *munc;              // (m, Npad)
    int64_t off, len, Npad;
    int comps, m, nwg;
    double pad, lamLo, lamHi;
    double *partA, *partB;          // [m][nwg] partial sums / counts
    unsigned int *hist;             // [m][GAIN_T][256]
    unsigned long long *prefix;     // [m][GAIN_T]
    long long *rank;                // [m][GAIN_T], -1 = unused
    double *out;                    // [m][GAIN_OUT]
};

__device__ __forceinline__ double gain_value(const GainArgs &a, int row, int64_t k) {
    // np.maximum / np.clip propagate NaN: a NaN input gives a NaN gain, which the finite filter drops
    const double x = (double)a.Pf[(a.off + k) * a.comps];
    const double p00 = (x > 0.0 || x != x) ? x : 0.0;
    double prec = 1.0;
    if (a.lam) {
        const double l = (double)a.lam[a.off + k];
        prec = l != l ? l : (l < a.lamLo ? a.lamLo : (l > a.lamHi ? a.lamHi : l));
    }
    const double v = (double)a.munc[(int64_t)row * a.Npad + a.off + k] + a.pad;
    const double var = (v > 1.0e-12 || v != v) ? v : 1.0e-12;
    return (p00 * prec) / var;
}
__device__ __forceinline__ unsigned long long gain_key(double g) {     // monotone in g over all finite doubles
    const unsigned long long b = (unsigned long long)__double_as_longlong(g);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double gain_unkey(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// phase 0: count and sum of the finite gains; phase 1: sum of squared deviations from the mean (out[row][1])
__global__ __launch_bounds__(256) void k_gain_moments(GainArgs a, int phase) {
    __shared__ double sA[4], sB[4];
    const int row = blockIdx.y, w = blockIdx.x, t = threadIdx.x;
    const int64_t chunk = (int64_t)256 * GAIN_ITEMS;
    const int64_t k0 = (int64_t)w * chunk;
    const double mean = phase ? a.out[row * GAIN_OUT + 1] : 0.0;
    double s = 0.0, cnt = 0.0;
#pragma unroll 4
    for (int u = 0; u < GAIN_ITEMS; ++u) {
        const int64_t k = k0 + (int64_t)u * 256 + t;
        if (k < a.len) {
            const double g = gain_value(a, row, k);
            if (isfinite(g)) {
                if (phase) { const double d = g - mean; s += d * d; }
                else { s += g; cnt += 1.0; }
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); cnt += __shfl_xor(cnt, o); }
    if ((t & 63) == 0) { sA[t >> 6] = s; sB[t >> 6] = cnt; }
    __syncthreads();
    if (t == 0) {
        a.partA[(int64_t)row * a.nwg + w] = (sA[0] + sA[1]) + (sA[2] + sA[3]);
        a.partB[(int64_t)row * a.nwg + w] = (sB[0] + sB[1]) + (sB[2] + sB[3]);
    }
}

// one wavefront per row folds the workgroup records in a fixed order; phase 0 also sets up the selection (ranks, empty prefixes)
__global__ __launch_bounds__(64) void k_gain_fold(GainArgs a, int phase) {
    const int row = blockIdx.x, lane = threadIdx.x;
    double s = 0.0, cnt = 0.0;
    for (int i = lane; i < a.nwg; i += 64) { s += a.partA[(int64_t)row * a.nwg + i]; cnt += a.partB[(int64_t)row * a.nwg + i]; }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); cnt += __shfl_xor(cnt, o); }
    if (lane != 0) return;
    double *out = a.out + row * GAIN_OUT;
    if (phase == 0) {
        out[0] = cnt;
        out[1] = cnt > 0.0 ? s / cnt : __longlong_as_double(0x7ff8000000000000ll);
        const long long n = (long long)cnt;
        const double qs[3] = {0.25, 0.5, 0.75};
        for (int i = 0; i < 3; ++i) {
            long long lo = -1, hi = -1;
            if (n > 0) {
                const double pos = (double)(n - 1) * qs[i];         // NumPy's virtual index of the 'linear' method
                lo = (long long)floor(pos);
                hi = lo + 1 < n ? lo + 1 : n - 1;
            }
            a.rank[row * GAIN_T + 2 * i] = lo; a.rank[row * GAIN_T + 2 * i + 1] = hi;
            a.prefix[row * GAIN_T + 2 * i] = 0ull; a.prefix[row * GAIN_T + 2 * i + 1] = 0ull;
        }
    } else {
        out[2] = out[0] > 0.0 ? sqrt(s / out[0]) : __longlong_as_double(0x7ff8000000000000ll);
    }
}

// targets with equal prefixes share one histogram: grp[t] = the first target with t's prefix (or -1: unused)
__device__ __forceinline__ void gain_groups(const GainArgs &a, int row, int *grp) {
    for (int t = 0; t < GAIN_T; ++t) {
        grp[t] = -1;
        if (a.rank[row * GAIN_T + t] < 0) continue;
        grp[t] = t;
        for (int u = 0; u < t; ++u)
            if (grp[u] == u && a.prefix[row * GAIN_T + u] == a.prefix[row * GAIN_T + t]) { grp[t] = u; break; }
    }
}

// pass p (0 .. 7): histogram of byte 7 - p of the keys whose higher bytes equal a target's prefix
__global__ __launch_bounds__(256) void k_gain_hist(GainArgs a, int pass) {
    __shared__ unsigned int h[GAIN_T][256];
    __shared__ int grp[GAIN_T];
    __shared__ unsigned long long pre[GAIN_T];
    const int row = blockIdx.y, w = blockIdx.x, t = threadIdx.x;
    for (int i = t; i < GAIN_T * 256; i += 256) (&h[0][0])[i] = 0u;
    if (t == 0) {
        gain_groups(a, row, grp);
        for (int q = 0; q < GAIN_T; ++q) pre[q] = a.prefix[row * GAIN_T + q];
    }
    __syncthreads();
    const int shift = 56 - 8 * pass;
    const int64_t k0 = (int64_t)w * 256 * GAIN_ITEMS;
#pragma unroll 4
    for (int u = 0; u < GAIN_ITEMS; ++u) {
        const int64_t k = k0 + (int64_t)u * 256 + t;
        if (k >= a.len) continue;
        const double g = gain_value(a, row, k);
        if (!isfinite(g)) continue;
        const unsigned long long key = gain_key(g);
        const unsigned long long hi = pass == 0 ? 0ull : key >> (shift + 8);
        const unsigned int digit = (unsigned int)(key >> shift) & 255u;
#pragma unroll
        for (int q = 0; q < GAIN_T; ++q)
            if (grp[q] == q && hi == pre[q]) atomicAdd(&h[q][digit], 1u);
    }
    __syncthreads();
    for (int i = t; i < GAIN_T * 256; i += 256) {
        const unsigned int v = (&h[0][0])[i];
        if (v) atomicAdd(a.hist + (int64_t)row * GAIN_T * 256 + i, v);
    }
}

// one wavefront per row: the digit of every target from its group's histogram; the last pass writes the order statistics
__global__ __launch_bounds__(64) void k_gain_pick(GainArgs a, int pass) {
    __shared__ int grp[GAIN_T];
    const int row = blockIdx.x, lane = threadIdx.x;
    if (lane == 0) gain_groups(a, row, grp);
    __syncthreads();
    unsigned long long newPre[GAIN_T];
    long long newRank[GAIN_T];
    for (int q = 0; q < GAIN_T; ++q) {
        newPre[q] = 0ull; newRank[q] = -1;
        if (grp[q] < 0) continue;
        const unsigned int *hh = a.hist + ((int64_t)row * GAIN_T + grp[q]) * 256;
        // lane l holds digits 4 l .. 4 l + 3; exclusive prefix over the wavefront
        const unsigned int c0 = hh[4 * lane], c1 = hh[4 * lane + 1], c2 = hh[4 * lane + 2], c3 = hh[4 * lane + 3];
        long long own = (long long)c0 + c1 + c2 + c3, incl = own;
        for (int o = 1; o < 64; o <<= 1) {
            const long long v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        const long long before = incl - own;
        const long long r = a.rank[row * GAIN_T + q];
        int digit = -1;
        long long base = 0;
        if (r >= before && r < incl) {
            long long cum = before;
            const unsigned int cs[4] = {c0, c1, c2, c3};
            for (int j = 0; j < 4; ++j) {
                if (r < cum + (long long)cs[j]) { digit = 4 * lane + j; base = cum; break; }
                cum += (long long)cs[j];
            }
        }
        const unsigned long long bal = __ballot(digit >= 0);
        const int src = bal ? (int)__ffsll((long long)bal) - 1 : 0;
        digit = __shfl(digit, src);
        base = __shfl(base, src);
        newPre[q] = (a.prefix[row * GAIN_T + q] << 8) | (unsigned long long)(unsigned int)(digit < 0 ? 0 : digit);
        newRank[q] = r - base;
    }
    __syncthreads();
    if (lane == 0)
        for (int q = 0; q < GAIN_T; ++q) {
            a.prefix[row * GAIN_T + q] = newPre[q];
            if (grp[q] >= 0) a.rank[row * GAIN_T + q] = newRank[q];
            if (pass == 7) a.out[row * GAIN_OUT + 3 + q] = grp[q] >= 0 ? gain_unkey(newPre[q]) : __longlong_as_double(0x7ff8000000000000ll);
        }
    for (int i = lane; i < GAIN_T * 256; i += 64) a.hist[(int64_t)row * GAIN_T * 256 + i] = 0u;
}

}  // namespace csr
