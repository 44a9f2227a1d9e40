/* qseed_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU restatement; never linked or imported by the product path).
 *
 * the natives behind the initial process-noise (Q0) seed,
 *   cEstimateSameTrackProcessNoiseTransitions   cconsenrich.pyx:1441-1797
 *   cEstimatePooledProcessNoiseTransitions      cconsenrich.pyx:1800-1902
 *   cQSeedPosteriorFromTransitions              cconsenrich.pyx:1905-2146
 * with their helpers (_linearQuantileCopyF64 pyx:1257-1291, _weightedQuantileInterpolatedF64 pyx:1294-1344,
 * _robustLocationF64 pyx:1347-1393, _cdfQuantileF64 pyx:1396-1428, _qSeedSampleIndex pyx:1431-1438).
 * Pinned bit-for-bit against the compiled reference by tests/golden/make_golden.py (qseed_*.npz).
 * Same floating-point contract as the reference build: no FMA contraction, libm log / exp / log1p / lgamma.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "consenrich_oracle.h"

static int cmp_f64(const void *a, const void *b) {
    const double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* pyx:1257-1291: linear-interpolated quantile of a copy (the reference selects with a three-way nth_element; any exact
 * selection yields the same two order statistics) */
static double linear_quantile_copy(const double *values, int64_t n, double q) {
    if (n <= 0) return NAN;
    double pos;
    if (q <= 0.0) pos = 0.0;
    else if (q >= 1.0) pos = (double)(n - 1);
    else pos = q * (double)(n - 1);
    const int64_t lo = (int64_t)floor(pos);
    int64_t hi = lo + 1;
    if (hi >= n) hi = n - 1;
    const double frac = pos - (double)lo;
    double *buf = (double *)malloc((size_t)n * sizeof(double));
    memcpy(buf, values, (size_t)n * sizeof(double));
    qsort(buf, (size_t)n, sizeof(double), cmp_f64);
    const double lowVal = buf[lo], highVal = buf[hi];
    free(buf);
    if (hi == lo) return lowVal;
    return lowVal + frac * (highVal - lowVal);
}

/* pyx:1347-1393 */
static double robust_location(const double *values, const double *weights, int64_t n) {
    if (n <= 0) return NAN;
    if (n == 1) return values[0];
    double loc = linear_quantile_copy(values, n, 0.5);
    double *absDev = (double *)malloc((size_t)n * sizeof(double));
    for (int64_t i = 0; i < n; ++i) absDev[i] = fabs(values[i] - loc);
    const double scale = 1.4826 * linear_quantile_copy(absDev, n, 0.5);
    free(absDev);
    if (scale <= 1.0e-12) return loc;
    const double c = 1.345;
    for (int it = 0; it < 4; ++it) {
        double denom = 0.0, numer = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            const double resid = values[i] - loc;
            double huber = (c * scale) / fmax(fabs(resid), 1.0e-12);
            if (huber > 1.0) huber = 1.0;
            const double eff = weights[i] * huber;
            denom += eff;
            numer += eff * values[i];
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

/* pyx:1431-1438 */
static int64_t sample_index(int64_t sampleIndex, int64_t itemCount, int64_t sampleCount) {
    return (int64_t)floor((((double)sampleIndex + 0.5) * (double)itemCount) / (double)sampleCount);
}

/* stable argsort (np.argsort(kind="mergesort")): bottom-up merge sort of an index array */
static void stable_argsort(const double *v, int64_t n, int64_t *idx) {
    int64_t *tmp = (int64_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
    for (int64_t i = 0; i < n; ++i) idx[i] = i;
    for (int64_t w = 1; w < n; w *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            const int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int64_t a = lo, b = mid, o = lo;
            while (a < mid && b < hi) tmp[o++] = (v[idx[b]] < v[idx[a]]) ? idx[b++] : idx[a++];
            while (a < mid) tmp[o++] = idx[a++];
            while (b < hi) tmp[o++] = idx[b++];
        }
        memcpy(idx, tmp, (size_t)n * sizeof(int64_t));
    }
    free(tmp);
}

/* pyx:1294-1344 */
static double weighted_quantile_interp(const double *values, const double *weights, int64_t n, double quantile) {
    if (n <= 0) return NAN;
    int64_t *order = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    stable_argsort(values, n, order);
    double total = 0.0;
    for (int64_t i = 0; i < n; ++i) total += weights[order[i]];
    double out = NAN;
    if (total > 0.0) {
        double target;
        if (quantile <= 0.0) target = 0.0;
        else if (quantile >= 1.0) target = total;
        else target = quantile * total;
        double cum = 0.0, prevCum = 0.0, prevValue = 0.0;
        out = values[order[n - 1]];
        for (int64_t i = 0; i < n; ++i) {
            const double v = values[order[i]];
            cum += weights[order[i]];
            if (target <= cum) {
                if (i == 0) out = v;
                else {
                    const double denom = cum - prevCum;
                    out = denom <= 0.0 ? v : prevValue + ((target - prevCum) / denom) * (v - prevValue);
                }
                break;
            }
            prevCum = cum;
            prevValue = v;
        }
    }
    free(order);
    return out;
}

/* pyx:1396-1428 */
static double cdf_quantile(const double *grid, const double *post, int64_t n, double prob) {
    if (n <= 0) return NAN;
    const double target = prob <= 0.0 ? 0.0 : (prob >= 1.0 ? 1.0 : prob);
    double cum = 0.0, prevCum = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        cum += post[i];
        if (target <= cum) {
            if (i == 0) return grid[0];
            const double denom = cum - prevCum;
            if (denom <= 0.0) return grid[i];
            return grid[i - 1] + ((target - prevCum) / denom) * (grid[i] - grid[i - 1]);
        }
        prevCum = cum;
    }
    return grid[n - 1];
}

/* active pair check with the reference's validation (pyx:1579-1598 / 1632-1651); returns an error code or 0 */
static int check_pair(double dl, double dr, double ol, double orr) {
    if (!isfinite(dl) || !isfinite(dr)) return COR_QSEED_ERR_DATA;
    if (!isfinite(ol) || !isfinite(orr) || ol <= 0.0 || orr <= 0.0) return COR_QSEED_ERR_OBSVAR;
    const double diff = dr - dl, rd = ol + orr;
    if (!isfinite(diff) || !isfinite(rd) || rd <= 0.0) return COR_QSEED_ERR_TRANSITION;
    const double rp = 1.0 / rd;
    if (!isfinite(rp) || rp <= 0.0) return COR_QSEED_ERR_PRECISION;
    return 0;
}

/* pyx:1441-1797.  Outputs need capacity scanCount = min(n-1, maxTransitionSamples > 0 ? maxTransitionSamples : n-1).
 * Returns the number of transitions written (>= 0) or a negative COR_QSEED_ERR_* code. */
int64_t cor_qseed_same_track(int64_t m, int64_t n, const double *data, const double *obs, const uint8_t *active,
                             double capQuantile, double capMultiplier, int64_t maxTransitionSamples,
                             int64_t precisionSampleCap, int64_t signalPanelSize, double *deltas, double *svar,
                             double *weights, cor_qseed_diag *dg) {
    memset(dg, 0, sizeof(*dg));
    dg->precisionCap = NAN;
    dg->transitionSampleFraction = 1.0;
    if (n < 2 || m <= 0) return 0;
    const int64_t maxT = n - 1;
    int64_t scanCount = maxT;
    int capped = 0;
    if (maxTransitionSamples > 0 && maxTransitionSamples < maxT) {
        capped = 1;
        scanCount = maxTransitionSamples;
        dg->transitionSampleFraction = (double)scanCount / (double)maxT;
        if (precisionSampleCap <= 0) return -COR_QSEED_ERR_SAMPLECAP;
    }
    dg->cappedMode = capped;
    dg->scanCount = scanCount;
    int64_t pairCount = 0, sampleCount = 0, cappedPairs = 0;
    double *raw = NULL;
    if (capped) { /* pyx:1574-1626 */
        for (int64_t si = 0; si < scanCount; ++si) {
            const int64_t k = sample_index(si, maxT, scanCount);
            for (int64_t j = 0; j < m; ++j)
                if (active[j * n + k] && active[j * n + k + 1]) {
                    const int e = check_pair(data[j * n + k], data[j * n + k + 1], obs[j * n + k], obs[j * n + k + 1]);
                    if (e) return -e;
                    ++pairCount;
                }
        }
        sampleCount = pairCount < precisionSampleCap ? pairCount : precisionSampleCap;
        if (sampleCount > 0) {
            raw = (double *)malloc((size_t)sampleCount * sizeof(double));
            int64_t ordinal = 0, slot = 0, target = sample_index(0, pairCount, sampleCount);
            for (int64_t si = 0; si < scanCount; ++si) {
                const int64_t k = sample_index(si, maxT, scanCount);
                for (int64_t j = 0; j < m; ++j)
                    if (active[j * n + k] && active[j * n + k + 1]) {
                        if (ordinal == target && slot < sampleCount) {
                            raw[slot++] = 1.0 / (obs[j * n + k] + obs[j * n + k + 1]);
                            if (slot < sampleCount) target = sample_index(slot, pairCount, sampleCount);
                        }
                        ++ordinal;
                    }
            }
        }
    } else { /* pyx:1627-1657 */
        raw = (double *)malloc((size_t)(m * maxT) * sizeof(double));
        for (int64_t k = 0; k < maxT; ++k)
            for (int64_t j = 0; j < m; ++j)
                if (active[j * n + k] && active[j * n + k + 1]) {
                    const int e = check_pair(data[j * n + k], data[j * n + k + 1], obs[j * n + k], obs[j * n + k + 1]);
                    if (e) { free(raw); return -e; }
                    raw[pairCount++] = 1.0 / (obs[j * n + k] + obs[j * n + k + 1]);
                }
        sampleCount = pairCount;
    }
    dg->pairCount = pairCount;
    dg->sampledPairCount = pairCount;
    dg->precisionSampleCount = sampleCount;
    double cap = NAN;
    if (sampleCount > 0) { /* pyx:1658-1668 */
        const double med = linear_quantile_copy(raw, sampleCount, 0.5);
        const double qp = linear_quantile_copy(raw, sampleCount, capQuantile);
        cap = fmin(qp, capMultiplier * med);
        if (cap > 0.0 && !capped) {
            for (int64_t j = 0; j < pairCount; ++j)
                if (raw[j] > cap) ++cappedPairs;
            dg->precisionCapFraction = (double)cappedPairs / (double)pairCount;
        }
    }
    free(raw);
    dg->precisionCap = cap;
    double *sig = (double *)malloc((size_t)scanCount * sizeof(double));
    double *ld = (double *)malloc((size_t)m * 3 * sizeof(double)), *ll = ld + m, *lp = ll + m;
    int64_t outCount = 0;
    for (int64_t si = 0; si < scanCount; ++si) { /* pyx:1683-1731 */
        const int64_t k = capped ? sample_index(si, maxT, scanCount) : si;
        int64_t cnt = 0;
        for (int64_t j = 0; j < m; ++j)
            if (active[j * n + k] && active[j * n + k + 1]) {
                const double ol = obs[j * n + k], orr = obs[j * n + k + 1];
                const double rawp = 1.0 / (ol + orr);
                if (capped && cap > 0.0 && rawp > cap) ++cappedPairs;
                double prec = rawp;
                if (cap > 0.0 && prec > cap) prec = cap;
                ld[cnt] = data[j * n + k + 1] - data[j * n + k];
                const double rd = ol + orr;
                ll[cnt] = (orr / rd) * data[j * n + k] + (ol / rd) * data[j * n + k + 1];
                lp[cnt] = prec;
                ++cnt;
            }
        if (cnt <= 0) continue;
        const double loc = robust_location(ld, lp, cnt);
        const double lev = robust_location(ll, lp, cnt);
        double sumP = 0.0, sumP2 = 0.0;
        for (int64_t j = 0; j < cnt; ++j) {
            sumP += lp[j];
            sumP2 += lp[j] * lp[j];
        }
        deltas[outCount] = loc;
        svar[outCount] = 1.0 / sumP;
        double eff = sumP2 > 0.0 ? (sumP * sumP) / sumP2 : 1.0;
        if (eff < 1.0) eff = 1.0;
        weights[outCount] = eff;
        sig[outCount] = lev;
        ++outCount;
    }
    free(ld);
    dg->candidateTransitionCount = outCount;
    dg->selectedTransitionCount = outCount;
    if (signalPanelSize > 0 && outCount > signalPanelSize) { /* pyx:1734-1768 */
        int64_t *order = (int64_t *)malloc((size_t)outCount * sizeof(int64_t));
        stable_argsort(sig, outCount, order);
        double *tmp = (double *)malloc((size_t)signalPanelSize * 3 * sizeof(double));
        for (int64_t pi = 0; pi < signalPanelSize; ++pi) {
            const int64_t ci = order[sample_index(pi, outCount, signalPanelSize)];
            tmp[pi] = deltas[ci];
            tmp[signalPanelSize + pi] = svar[ci];
            tmp[2 * signalPanelSize + pi] = weights[ci];
        }
        memcpy(deltas, tmp, (size_t)signalPanelSize * sizeof(double));
        memcpy(svar, tmp + signalPanelSize, (size_t)signalPanelSize * sizeof(double));
        memcpy(weights, tmp + 2 * signalPanelSize, (size_t)signalPanelSize * sizeof(double));
        free(tmp);
        free(order);
        outCount = signalPanelSize;
        dg->selectedTransitionCount = outCount;
    }
    free(sig);
    if (capped && pairCount > 0) dg->precisionCapFraction = (double)cappedPairs / (double)pairCount;
    return outCount;
}

/* pyx:1800-1902.  Outputs need capacity n-1.  Returns the count or a negative error code. */
int64_t cor_qseed_pooled(int64_t m, int64_t n, const double *data, const double *obs, const uint8_t *active,
                         double *deltas, double *svar, double *weights) {
    if (n < 2 || m <= 0) return 0;
    double *pm = (double *)malloc((size_t)n * 2 * sizeof(double)), *pv = pm + n;
    for (int64_t i = 0; i < n; ++i) {
        double wsum = 0.0, zsum = 0.0;
        for (int64_t j = 0; j < m; ++j)
            if (active[j * n + i]) {
                const double v = data[j * n + i], o = obs[j * n + i];
                if (!isfinite(v) || !isfinite(o) || o <= 0.0) { free(pm); return -COR_QSEED_ERR_POOLED; }
                const double w = 1.0 / o;
                wsum += w;
                zsum += v * w;
            }
        if (wsum > 0.0) { pm[i] = zsum / wsum; pv[i] = 1.0 / wsum; }
        else { pm[i] = NAN; pv[i] = NAN; }
    }
    int64_t out = 0;
    for (int64_t i = 0; i < n - 1; ++i)
        if (isfinite(pm[i]) && isfinite(pm[i + 1]) && isfinite(pv[i]) && isfinite(pv[i + 1])) {
            const double s2 = pv[i] + pv[i + 1];
            deltas[out] = pm[i + 1] - pm[i];
            svar[out] = s2;
            weights[out] = s2 > 0.0 ? 1.0 / fmax(s2, DBL_MIN) : 1.0;
            ++out;
        }
    free(pm);
    return out;
}

/* pyx:1905-2146 (argument validation of pyx:1977-1996 is done by the Python wrapper).  Returns 0, or a negative
 * COR_QSEED_ERR_* code for the data-dependent errors raised inside the loops. */
int cor_qseed_posterior(int64_t count, const double *deltas, const double *s2, const double *weights, double qFloor,
                        double qCap, double robustTNu, double qSeedPriorLevel, int64_t minTransitions,
                        double priorLogSd, double defaultTNu, int64_t gridSize, cor_qseed_post *out) {
    memset(out, 0, sizeof(*out));
    double sumW = 0.0, sumW2 = 0.0;
    for (int64_t i = 0; i < count; ++i) { /* pyx:2001-2010 */
        if (!isfinite(deltas[i])) return -COR_QSEED_ERR_POST_DELTA;
        if (!isfinite(s2[i]) || s2[i] < 0.0) return -COR_QSEED_ERR_POST_S2;
        if (!isfinite(weights[i]) || weights[i] <= 0.0) return -COR_QSEED_ERR_POST_W;
        sumW += weights[i];
        sumW2 += weights[i] * weights[i];
    }
    double eff = 0.0;
    if (sumW2 > 0.0) eff = (sumW * sumW) / sumW2;
    out->transitionCount = count;
    out->effectiveTransitionCount = eff;
    if (count < minTransitions || eff < (double)minTransitions) return 0; /* ok stays 0 */
    const double center = weighted_quantile_interp(deltas, weights, count, 0.5);
    double *work = (double *)malloc((size_t)count * sizeof(double));
    for (int64_t i = 0; i < count; ++i) work[i] = fabs(deltas[i] - center);
    const double robustScale = 1.4826 * weighted_quantile_interp(work, weights, count, 0.5);
    const double medianS2 = weighted_quantile_interp(s2, weights, count, 0.5);
    double qPrior = robustScale * robustScale - medianS2;
    if (qPrior < qFloor) qPrior = qFloor;
    if (qPrior < qSeedPriorLevel) qPrior = qSeedPriorLevel;
    double maxDeltaSq = 0.0;
    for (int64_t i = 0; i < count; ++i) {
        double cand = deltas[i] * deltas[i];
        if (cand > maxDeltaSq) maxDeltaSq = cand;
        cand -= s2[i];
        if (cand < 0.0) cand = 0.0;
        work[i] = cand;
    }
    const double q90 = weighted_quantile_interp(work, weights, count, 0.9);
    free(work);
    const double lower = qFloor;
    double upper;
    if (isfinite(qCap)) upper = fmax(qCap, lower);
    else { /* pyx:2040-2056 */
        upper = lower * 10.0;
        const double cands[5] = {qPrior * 1.0e4, q90 * 100.0, medianS2 * 100.0, maxDeltaSq * 10.0, lower * 1.0e6};
        for (int c = 0; c < 5; ++c)
            if (cands[c] > upper && cands[c] > lower) upper = cands[c];
    }
    const int64_t G = (upper <= lower * (1.0 + 1.0e-10)) ? 1 : gridSize;
    double *grid = (double *)malloc((size_t)G * 3 * sizeof(double)), *logPost = grid + G, *post = logPost + G;
    if (G == 1) grid[0] = lower;
    else {
        const double logLower = log(lower), logUpper = log(upper);
        const double step = (logUpper - logLower) / (double)(G - 1);
        for (int64_t g = 0; g < G; ++g) grid[g] = exp(logLower + step * (double)g);
    }
    double nu = robustTNu;
    if (!isfinite(nu) || nu <= 0.0) nu = defaultTNu;
    if (nu < 4.0) nu = 4.0;
    double medianWeight = weighted_quantile_interp(weights, weights, count, 0.5);
    if (medianWeight < DBL_MIN) medianWeight = DBL_MIN;
    const double logPriorCenter = log(fmax(qPrior, lower));
    const double logPriorSd = fmax(priorLogSd, 1.0e-6);
    const double logNorm = lgamma((nu + 1.0) * 0.5) - lgamma(nu * 0.5) - 0.5 * (log(nu) + log(3.14159265358979323846));
    double maxLogPost = -INFINITY;
    int64_t modeIndex = 0;
    for (int64_t g = 0; g < G; ++g) { /* pyx:2087-2116 */
        const double q = grid[g];
        double logLikeSum = 0.0;
        for (int64_t i = 0; i < count; ++i) {
            double var = q + s2[i];
            if (var < DBL_MIN) var = DBL_MIN;
            double wn = weights[i] / medianWeight;
            if (wn < 0.25) wn = 0.25;
            else if (wn > 4.0) wn = 4.0;
            logLikeSum += wn * (logNorm - 0.5 * log(var) - 0.5 * (nu + 1.0) * log1p((deltas[i] * deltas[i]) / (nu * var)));
        }
        const double logPrior = -0.5 * ((log(q) - logPriorCenter) / logPriorSd) * ((log(q) - logPriorCenter) / logPriorSd);
        const double lp = logLikeSum + logPrior;
        if (!isfinite(lp)) { free(grid); return -COR_QSEED_ERR_POST_SCORE; }
        logPost[g] = lp;
        if (lp > maxLogPost) { maxLogPost = lp; modeIndex = g; }
    }
    double total = 0.0;
    for (int64_t g = 0; g < G; ++g) { post[g] = exp(logPost[g] - maxLogPost); total += post[g]; }
    if (!isfinite(total) || total <= 0.0) { free(grid); return -COR_QSEED_ERR_POST_NORM; }
    for (int64_t g = 0; g < G; ++g) post[g] = post[g] / total;
    out->ok = 1;
    out->medianSamplingVariance = medianS2;
    out->priorLevel = qPrior;
    out->posteriorModeLevel = grid[modeIndex];
    out->posteriorMedianLevel = cdf_quantile(grid, post, G, 0.5);
    out->posteriorQ05Level = cdf_quantile(grid, post, G, 0.05);
    out->posteriorQ95Level = cdf_quantile(grid, post, G, 0.95);
    out->transitionQ90 = q90;
    free(grid);
    return 0;
}
