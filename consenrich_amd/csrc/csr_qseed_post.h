// csr_qseed_post.h -- SURVEY 8(f) rank 4, second half: the bounded tail of the initial process-noise (Q0) seed on the device.
//
// What the reference computes (cconsenrich.pyx:1257-1438 helpers, 1998-2146 `cQSeedPosteriorFromTransitions`): two linear
// quantiles of the precision sample; then, from <= 2048 selected transitions (delta, sampling variance, weight): five WEIGHTED
// quantiles (centre, robust scale, median sampling variance, 0.9-quantile of the excess squares, median weight), a 64-point
// log-spaced grid of candidate levels, a Student-t log-likelihood summed over the transitions for every grid point, a
// log-normal prior, and the mode / median / 5 % / 95 % points of the normalised grid posterior.
//
// How it is done here (one workgroup of 1024 threads per chromosome, every chromosome of the batch in one launch):
//   * order statistics of the precision sample by BITWISE BISECTION on the 64-bit patterns (positive doubles order like their
//     bit patterns): 63 counting passes per requested rank, no sort, no copy -- exact, so the interpolated quantiles carry the
//     reference's bits (k_qs_select);
//   * weighted quantiles by a BITONIC SORT of (value, index) pairs -- index as tie-break = the stable order of the reference's
//     mergesort argsort -- in a scratch region, followed by one thread walking the cumulative weights in sorted order (the walk
//     is <= 2048 additions and must round like the reference's to land on the same interpolation cell);
//   * the grid posterior with one thread per grid point walking the transitions in order (same summation order as the
//     reference: the only numerical difference to it is the device's log / log1p / exp / lgamma against glibc's, <= 1 ulp per
//     call -- posterior summaries agree to ~1e-13, the fixtures' gate is 1e-12).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace csr {

struct QpCfg {
    double q_floor, q_cap, robust_t_nu, q_seed_prior_level, prior_log_sd, default_t_nu;
    int64_t min_transitions, grid_size;
};
struct QpJob {
    int64_t n;                  // transitions
    int64_t P;                  // power of two >= n (sort width)
    const double *d, *s2, *w;   // deltas, sampling variances, weights (device)
    double *key, *work;         // scratch: P doubles (sort keys), n doubles (derived values)
    int *ord;                   // scratch: P ints (sort permutation)
    double *logPost;            // scratch: grid_size doubles (+ grid_size for the grid, + grid_size for the posterior)
    csr_qseed_post *out;
    int *status;                // 0 ok; 1 deltas not finite; 2 bad sampling variance; 3 bad weight; 4 nonfinite score; 5 normalisation
};

// ---- exact order statistics by bitwise bisection -------------------------------------------------------------------------
struct QsSelJob {
    const double *v;            // positive finite values
    int64_t n;
    int64_t rank[4];            // requested 0-based ranks (-1: unused)
    double *out;                // 4 doubles
};
__global__ __launch_bounds__(1024) void k_qs_select(const QsSelJob *jobs) {
    const QsSelJob jb = jobs[blockIdx.x];
    __shared__ unsigned long long cnt[32];
    __shared__ unsigned long long ans;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int q = 0; q < 4; ++q) {
        if (jb.rank[q] < 0 || jb.n <= 0) continue;
        if (threadIdx.x == 0) ans = 0ull;
        __syncthreads();
        for (int bit = 62; bit >= 0; --bit) {
            const unsigned long long trial = ans | ((1ull << bit) - 1ull);       // largest pattern with this bit clear
            unsigned long long c = 0;
            for (int64_t i = threadIdx.x; i < jb.n; i += 1024)
                c += ((unsigned long long)__double_as_longlong(jb.v[i]) <= trial) ? 1ull : 0ull;
            for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
            if (lane == 0) cnt[wv] = c;
            __syncthreads();
            if (threadIdx.x == 0) {
                unsigned long long t = 0;
                for (int k = 0; k < 16; ++k) t += cnt[k];
                if (t <= (unsigned long long)jb.rank[q]) ans |= (1ull << bit);   // the rank-th value has this bit set
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) jb.out[q] = __longlong_as_double((long long)ans);
        __syncthreads();
    }
}

// ---- bitonic sort of (key, ord) in a scratch region, ascending, ties by ord ------------------------------------------------
__device__ __forceinline__ bool qp_less(double ka, int oa, double kb, int ob) { return ka < kb || (ka == kb && oa < ob); }
__device__ void qp_sort(double *key, int *ord, int64_t P) {
    for (int64_t k = 2; k <= P; k <<= 1) {
        for (int64_t j = k >> 1; j > 0; j >>= 1) {
            for (int64_t t = threadIdx.x; t < P; t += blockDim.x) {
                const int64_t x = t ^ j;
                if (x > t) {
                    const double ka = key[t], kb = key[x];
                    const int oa = ord[t], ob = ord[x];
                    const bool up = (t & k) == 0;
                    const bool swap = up ? qp_less(kb, ob, ka, oa) : qp_less(ka, oa, kb, ob);
                    if (swap) { key[t] = kb; key[x] = ka; ord[t] = ob; ord[x] = oa; }
                }
            }
            __syncthreads();
        }
    }
}

// weighted quantile of values val(i) with weights w[i]: sort, then ONE thread walks the cumulative weights (pyx:1294-1344)
template <class F>
__device__ double qp_weighted_quantile(const QpJob &jb, F val, double quantile, double *shared_result) {
    for (int64_t i = threadIdx.x; i < jb.P; i += blockDim.x) {
        jb.key[i] = i < jb.n ? val(i) : INFINITY;
        jb.ord[i] = (int)i;
    }
    __syncthreads();
    qp_sort(jb.key, jb.ord, jb.P);
    if (threadIdx.x == 0) {
        double total = 0.0;
        for (int64_t i = 0; i < jb.n; ++i) total += jb.w[jb.ord[i]];
        double res = NAN;
        if (total > 0.0) {
            const double target = quantile <= 0.0 ? 0.0 : (quantile >= 1.0 ? total : quantile * total);
            double cum = 0.0, prevCum = 0.0, prevValue = 0.0;
            res = jb.key[jb.n - 1];
            for (int64_t i = 0; i < jb.n; ++i) {
                const double v = jb.key[i];
                cum += jb.w[jb.ord[i]];
                if (target <= cum) {
                    const double denom = cum - prevCum;
                    res = (i == 0 || denom <= 0.0) ? v : prevValue + ((target - prevCum) / denom) * (v - prevValue);
                    break;
                }
                prevCum = cum;
                prevValue = v;
            }
        }
        *shared_result = res;
    }
    __syncthreads();
    const double r = *shared_result;
    __syncthreads();
    return r;
}

__device__ double qp_cdf_quantile(const double *grid, const double *post, int64_t G, double prob) {   // pyx:1396-1428
    const double target = prob <= 0.0 ? 0.0 : (prob >= 1.0 ? 1.0 : prob);
    double cum = 0.0, prevCum = 0.0;
    for (int64_t i = 0; i < G; ++i) {
        cum += post[i];
        if (target <= cum) {
            if (i == 0) return grid[0];
            const double denom = cum - prevCum;
            if (denom <= 0.0) return grid[i];
            return grid[i - 1] + ((target - prevCum) / denom) * (grid[i] - grid[i - 1]);
        }
        prevCum = cum;
    }
    return grid[G - 1];
}

__global__ __launch_bounds__(1024) void k_qs_posterior(const QpJob *jobs, QpCfg cf) {
    const QpJob jb = jobs[blockIdx.x];
    __shared__ double sh[8];
    __shared__ int bad;
    csr_qseed_post o;
    const int64_t n = jb.n;
    if (threadIdx.x == 0) {
        bad = 0;
        // validation and effective count in the reference's (sequential) order: n <= a few thousand
        double sumW = 0.0, sumW2 = 0.0;
        for (int64_t i = 0; i < n && !bad; ++i) {
            if (!isfinite(jb.d[i])) bad = 1;
            else if (!isfinite(jb.s2[i]) || jb.s2[i] < 0.0) bad = 2;
            else if (!isfinite(jb.w[i]) || jb.w[i] <= 0.0) bad = 3;
            sumW += jb.w[i];
            sumW2 += jb.w[i] * jb.w[i];
        }
        sh[0] = sumW2 > 0.0 ? (sumW * sumW) / sumW2 : 0.0;
    }
    __syncthreads();
    const int isBad = bad;
    const double eff = sh[0];
    __syncthreads();
    if (threadIdx.x == 0) {
        *jb.status = isBad;
        csr_qseed_post z;
        z.transition_count = n; z.ok = 0; z.reserved = 0; z.effective_transition_count = eff;
        z.median_sampling_variance = 0.0; z.prior_level = 0.0; z.posterior_mode = 0.0; z.posterior_median = 0.0;
        z.posterior_q05 = 0.0; z.posterior_q95 = 0.0; z.transition_q90 = 0.0;
        *jb.out = z;
    }
    if (isBad || n < cf.min_transitions || eff < (double)cf.min_transitions) return;

    const double qFloor = cf.q_floor, qCap = cf.q_cap;
    const double *d = jb.d, *s2 = jb.s2, *w = jb.w;
    const double center = qp_weighted_quantile(jb, [=](int64_t i) { return d[i]; }, 0.5, &sh[1]);
    const double mad = qp_weighted_quantile(jb, [=](int64_t i) { return fabs(d[i] - center); }, 0.5, &sh[1]);
    const double robustScale = 1.4826 * mad;
    const double medianS2 = qp_weighted_quantile(jb, [=](int64_t i) { return s2[i]; }, 0.5, &sh[1]);
    double qPrior = robustScale * robustScale - medianS2;
    if (qPrior < qFloor) qPrior = qFloor;
    if (qPrior < cf.q_seed_prior_level) qPrior = cf.q_seed_prior_level;
    const double q90 = qp_weighted_quantile(jb, [=](int64_t i) { double c2 = d[i] * d[i] - s2[i]; return c2 < 0.0 ? 0.0 : c2; }, 0.9, &sh[1]);
    double medianWeight = qp_weighted_quantile(jb, [=](int64_t i) { return w[i]; }, 0.5, &sh[1]);
    if (medianWeight < 2.2250738585072014e-308) medianWeight = 2.2250738585072014e-308;

    double *grid = jb.logPost + cf.grid_size, *post = grid + cf.grid_size;
    if (threadIdx.x == 0) {
        double maxDeltaSq = 0.0;
        for (int64_t i = 0; i < n; ++i) { const double c2 = d[i] * d[i]; if (c2 > maxDeltaSq) maxDeltaSq = c2; }
        const double lower = qFloor;
        double upper;
        if (isfinite(qCap)) upper = fmax(qCap, lower);
        else {
            upper = lower * 10.0;
            const double cands[5] = {qPrior * 1.0e4, q90 * 100.0, medianS2 * 100.0, maxDeltaSq * 10.0, lower * 1.0e6};
            for (int k = 0; k < 5; ++k)
                if (cands[k] > upper && cands[k] > lower) upper = cands[k];
        }
        const int64_t G = (upper <= lower * (1.0 + 1.0e-10)) ? 1 : cf.grid_size;
        if (G == 1) grid[0] = lower;
        else {
            const double logLower = log(lower), logUpper = log(upper);
            const double step = (logUpper - logLower) / (double)(G - 1);
            for (int64_t g = 0; g < G; ++g) grid[g] = exp(logLower + step * (double)g);
        }
        double nu = cf.robust_t_nu;
        if (!isfinite(nu) || nu <= 0.0) nu = cf.default_t_nu;
        if (nu < 4.0) nu = 4.0;
        sh[2] = (double)G;
        sh[3] = nu;
        sh[4] = log(fmax(qPrior, lower));                       // logPriorCenter
        sh[5] = lgamma((nu + 1.0) * 0.5) - lgamma(nu * 0.5) - 0.5 * (log(nu) + log(3.14159265358979323846264338327950288));
    }
    __syncthreads();
    const int64_t G = (int64_t)sh[2];
    const double nu = sh[3], logPriorCenter = sh[4], logNorm = sh[5];
    const double logPriorSd = fmax(cf.prior_log_sd, 1.0e-6);
    // one thread per grid point, transitions in order
    if ((int64_t)threadIdx.x < G) {
        const double q = grid[threadIdx.x];
        double logLikeSum = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            double var = q + s2[i];
            if (var < 2.2250738585072014e-308) var = 2.2250738585072014e-308;
            double wn = w[i] / medianWeight;
            if (wn < 0.25) wn = 0.25;
            else if (wn > 4.0) wn = 4.0;
            logLikeSum += wn * (logNorm - 0.5 * log(var) - 0.5 * (nu + 1.0) * log1p((d[i] * d[i]) / (nu * var)));
        }
        const double lq = (log(q) - logPriorCenter) / logPriorSd;
        jb.logPost[threadIdx.x] = logLikeSum + (-0.5 * lq * lq);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double maxLogPost = -INFINITY;
        int64_t modeIndex = 0;
        int st = 0;
        for (int64_t g = 0; g < G; ++g) {
            const double lp = jb.logPost[g];
            if (!isfinite(lp)) { st = 4; break; }
            if (lp > maxLogPost) { maxLogPost = lp; modeIndex = g; }
        }
        if (!st) {
            double total = 0.0;
            for (int64_t g = 0; g < G; ++g) { post[g] = exp(jb.logPost[g] - maxLogPost); total += post[g]; }
            if (!isfinite(total) || total <= 0.0) st = 5;
            else {
                for (int64_t g = 0; g < G; ++g) post[g] = post[g] / total;
                o.transition_count = n; o.ok = 1; o.reserved = 0; o.effective_transition_count = eff;
                o.median_sampling_variance = medianS2;
                o.prior_level = qPrior;
                o.posterior_mode = grid[modeIndex];
                o.posterior_median = qp_cdf_quantile(grid, post, G, 0.5);
                o.posterior_q05 = qp_cdf_quantile(grid, post, G, 0.05);
                o.posterior_q95 = qp_cdf_quantile(grid, post, G, 0.95);
                o.transition_q90 = q90;
                *jb.out = o;
            }
        }
        *jb.status = st;
    }
}

}  // namespace csr
