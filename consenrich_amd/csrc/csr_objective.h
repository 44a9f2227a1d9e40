// csr_objective.h -- SURVEY a12: the penalised objective of the reference's outer stop rule (core.py:4418-4538,
// `_scorePenalizedObjective`; stability test `_recordOuterObjective`, core.py:4750-4830).  The forward NLL term is the
// forward pass itself (csr_batch_forward_masked); the kernels here reduce the remaining terms per chain, in fp64, from
// device-resident tracks:
//   robust precision penalties   0.5 nu sum (x - log x), x = max(lambda, tiny) / max(kappa[1:], tiny)     core.py:3161-3179
//   roughness penalties          sum d1^2, sum d2^2 of the float64 background                             core.py:3182-3204
//   negative part                sum min(background, 0)^2                                                 core.py:4450-4463
//   effective observation count  #(isfinite(munc) & munc < 0.5e30)                                        core.py:2981-2986
// and the float64 weight track  sum_j clip(lambda) / max(munc + pad, 1e-8)  (core.py:4493-4511) whose median over the
// positive entries scales the negative-part penalty (selected with the background update's bitwise order-statistic
// passes).  Reductions use the background update's per-wavefront records (no atomics, fixed fold order).
#pragma once
#include "csr_background.h"

namespace csr {

struct ObjArgs {
    const float *lamNat, *kapNat, *bg;   // natural-layout tracks (lambda / kappa may be null: term is zero)
    int useLambdaPenalty, useKappaPenalty, useLambdaWeights;
    double pad, wMin, wMax, maskedHalf;
    double *w64;                         // [Npad] float64 weight track (or null)
    double *part;                        // NW x 6 partial records
    double *chainOut;                    // per chain x 6: obs, proc, d1^2, d2^2, neg^2, count
};

// one pass over the bins of every chain: weight track + the six per-chain sums (per-wavefront partial records)
__global__ __launch_bounds__(256) void k_obj_wave(Prm p, BgBatch a, ObjArgs o) {
    const int lane = threadIdx.x & 63;
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wv >= a.NW) return;
    const int c = a.waveChain[wv];
    const int64_t off = a.chainOff[c], len = a.chainLen[c];
    const double tiny = 2.2250738585072014e-308;
    double r[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int G = a.waveG0[wv]; G < a.waveG1[wv]; ++G) {
        const int64_t g = ((int64_t)G << 6) + lane;
        const int64_t k = g - off;
        if (k >= len) {
            if (o.w64 && g < p.Npad) o.w64[g] = 0.0;
            continue;
        }
        double lamClip = 1.0;
        if (o.lamNat) {
            const double lam = (double)o.lamNat[g];
            if (o.useLambdaPenalty) {
                const double v = lam > tiny ? lam : tiny;
                r[0] += v - log(v);
            }
            lamClip = lam < o.wMin ? o.wMin : (lam > o.wMax ? o.wMax : lam);
        }
        if (o.useKappaPenalty && o.kapNat && (k >= 1 || len == 1)) {
            const double kv = (double)o.kapNat[g];
            const double v = kv > tiny ? kv : tiny;
            r[1] += v - log(v);
        }
        if (o.bg) {
            const double b0 = (double)o.bg[g];
            if (k >= 1) {
                const double b1 = (double)o.bg[g - 1];
                const double d1 = b0 - b1;
                r[2] += d1 * d1;
                if (k >= 2) {
                    const double d2 = d1 - (b1 - (double)o.bg[g - 2]);      // np.diff(n=2): difference of differences
                    r[3] += d2 * d2;
                }
            }
            const double mn = b0 < 0.0 ? b0 : 0.0;
            r[4] += mn * mn;
        }
        double ws = 0.0, cnt = 0.0;
        for (int j = 0; j < p.m; ++j) {
            const double v = (double)p.munc[(int64_t)j * p.Npad + g];
            if (fabs(v) <= 1.7976931348623157e308 && v < o.maskedHalf) cnt += 1.0;
            double den = v + o.pad;
            den = den < 1.0e-8 ? 1.0e-8 : den;                              // np.maximum: a NaN variance stays NaN
            double inv = 1.0 / den;
            if (o.useLambdaWeights) inv *= lamClip;
            ws += inv;
        }
        r[5] += cnt;
        if (o.w64) o.w64[g] = ws;
    }
    for (int s = 32; s > 0; s >>= 1)
#pragma unroll
        for (int i = 0; i < 6; ++i) r[i] += __shfl_xor(r[i], s);
    if (lane == 0) {
        double *rec = o.part + 6 * (int64_t)wv;
#pragma unroll
        for (int i = 0; i < 6; ++i) rec[i] = r[i];
    }
}

__global__ __launch_bounds__(64) void k_obj_fold(BgBatch a, ObjArgs o) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const int w0 = a.chainWave0[c], nw = a.chainWaveN[c];
    double r[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int i = lane; i < nw; i += 64) {
        const double *rec = o.part + 6 * (int64_t)(w0 + i);
#pragma unroll
        for (int q = 0; q < 6; ++q) r[q] += rec[q];
    }
    for (int s = 32; s > 0; s >>= 1)
#pragma unroll
        for (int q = 0; q < 6; ++q) r[q] += __shfl_xor(r[q], s);
    if (lane == 0)
#pragma unroll
        for (int q = 0; q < 6; ++q) o.chainOut[6 * c + q] = r[q];
}

}  // namespace csr
