/*
 * consenrich_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, single thread, fp64 arithmetic on f32 storage) of the Consenrich
 * Kalman forward filter / RTS smoother / fixed-background ECM hot path.  It exists to CHECK the
 * HIP product path; it is never imported, linked or executed by consenrich_amd/ itself.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Parity is PINNED: validated bit-for-bit (store arrays) / <=1e-12 rel (scalar sums) against the real reference
 * extension compiled into oracle/_ref (oracle/Makefile `ref`; tests/golden/make_golden.py) and against the golden
 * vectors committed under tests/golden/.
 *
 * Every function cites the reference lines (src/consenrich/cconsenrich.pyx, "pyx") it restates.
 */
#ifndef CONSENRICH_ORACLE_H
#define CONSENRICH_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cor_model {
    int32_t state_dim;          /* 2 = levelTrend (pyx:291), 1 = level (pyx:538) */
    double F[4];                /* row-major 2x2 transition, ignored for state_dim 1 */
    double Q0[4];               /* row-major base process noise; level uses Q0[0] */
    double state_init;          /* already a float32 value widened to double (pyx:6593) */
    double state_covar_init;
    double pad;                 /* (double)(float)pad (pyx:6595) */
    double w_min, w_max;        /* lambda clamp (pyx:433) */
    double k_min, k_max;        /* kappa clamp (pyx:395) */
    double apn_min_q, apn_max_q, apn_thresh, apn_scale, apn_pc; /* pyx:510-527 */
} cor_model;

typedef struct cor_fwd_io {
    int64_t m, n;
    const float *data;          /* (m,n) C-order */
    const float *munc;          /* (m,n) C-order */
    const int32_t *block_map;   /* n */
    int64_t block_count;
    const float *lambda;        /* n or NULL  -> useLambda (pyx:6449) */
    const float *kappa;         /* n or NULL  -> useProcPrec (pyx:6453) */
    const float *qscale;        /* n or NULL  -> useProcessQScale (pyx:6452) */
    int32_t use_apn;            /* after the qDiagBase<=1e-12 veto (pyx:6575) */
    int32_t return_nll;
    int32_t store_nll_in_d;
    float *D;                   /* n, always written */
    float *xf;                  /* (n,d) or NULL -> doStore (pyx:6444) */
    float *Pf;                  /* (n,d,d) */
    float *pnoise;              /* (n,d,d); entry k-1 written at step k */
} cor_fwd_io;

typedef struct cor_fwd_out {
    double sum_d;
    double sum_nll;
    int64_t invalid_block_index;   /* -1 if none (pyx:389-392) */
} cor_fwd_out;

/* pyx:291-529 (trend) / pyx:538-707 (level) */
void cor_forward(const cor_model *mdl, const cor_fwd_io *io, cor_fwd_out *out);

/* pyx:6740-6848 (trend) / pyx:7116-7148 (level).  lag_rows = number of rows the lagCov buffer holds. */
void cor_backward(const cor_model *mdl, int64_t m, int64_t n, const float *data,
                  const float *xf, const float *Pf, const float *pnoise,
                  float *xs, float *Ps, float *lag, int64_t lag_rows, float *resid);

typedef struct cor_ecm_cfg {
    int64_t max_iters;          /* ECM_fixedBackgroundIters */
    int64_t inner_iters;        /* t_innerIters */
    double rtol;                /* (double)(float) ECM_fixedBackgroundRtol */
    double nu;                  /* (double)(float) ECM_robustTNu */
    int32_t use_lambda;         /* ECM_useObsPrecisionReweighting */
    int32_t use_kappa;          /* ECM_useProcessPrecisionReweighting && (!APN || qscale) (pyx:7912) */
    int32_t use_apn;
} cor_ecm_cfg;

typedef struct cor_ecm_out {
    int64_t iters_done;
    double final_nll;
    double initial_nll;
    double abs_rel_change;
    double rel_improvement;
    int64_t stable_iters;
    int64_t nll_increase_count;
    int32_t converged;
    int32_t skipped;            /* n <= 5 fallback (pyx:7998) */
    int32_t has_initial_nll;
    int64_t invalid_block_index;
} cor_ecm_out;

/* pyx:7660-8442 (trend) / pyx:7153-7657 (level).  lambda/kappa: n, in/out (warm start already clipped by
 * the caller), may be NULL when the corresponding re-weighting is off.  Work buffers are caller-owned. */
void cor_ecm(const cor_model *mdl, const cor_ecm_cfg *cfg,
             int64_t m, int64_t n, const float *data, const float *munc,
             const int32_t *block_map, int64_t block_count, const float *qscale,
             float *lambda, float *kappa,
             float *xf, float *Pf, float *pnoise,
             float *xs, float *Ps, float *lag, float *resid,
             float *D_scratch, double *nll_path /* max_iters or NULL */,
             cor_ecm_out *out);

/* pyx:710-815 (trend, d=2) / pyx:818-863 (level, d=1); float64 inputs */
void cor_transition_sums(int32_t state_dim, int64_t n, const double *xs, const double *Ps,
                         const double *lag, const double *F, double *sum_level, double *sum_trend,
                         int64_t *count);

/* ---- background update natives -------------------------------------------------------------- */
/* pyx:9700-9724 `cbackgroundWeightedStatsWithSupport`: per-interval weight = sum_j invVar[j,i] and
 * rhs = sum_j invVar[j,i]*resid[j,i] in fp64 over float32 (m,n) C-order matrices; returns the count of weight > 0. */
int64_t cor_background_stats(int64_t m, int64_t n, const float *resid, const float *inv_var, double *weight,
                             double *rhs);
/* pyx:944-1096 `csolveZeroCenteredBackground`: (diag(w) + lam_first D1'D1 + lam D2'D2) x = rhs by pentadiagonal LDL'
 * with pivot floor 1e-12, optional zero-sum Lagrange correction.  Returns -1 if no pivot was modified, otherwise the
 * index of the first modified pivot (its value in *bad_value); `out` is filled in both cases (the reference raises). */
int64_t cor_solve_background(int64_t n, const double *weight, const double *rhs, double lam, int zero_center,
                             double lam_first, double *out, double *bad_value);

/* ---- delete-block calibration natives (src/consenrich/cuncertainty.pyx, "unc") -------------- */
/* unc:97-157 `cobservationTotalInformation`: total[i] = sum over active cells of lambda_i / (munc[j,i] + pad), with the
 * exchangeable-correlation correction (unc:37-57) when rho > 0.  munc is float32 (munc_f64 = 0) or float64. */
void cor_total_information(int64_t m, int64_t n, const void *munc, int munc_f64, const uint8_t *active,
                           const double *lambda /* or NULL */, double pad, double rho, double *total);
/* unc:160-305 `cmakeFoldMaskAndInformation` (after its argument validation): mask (m,n) uint8, kept / heldout / h (n),
 * nominal (n) or NULL.  reps is (block_count, slots) row-major. */
void cor_fold_mask_information(int64_t m, int64_t n, int64_t block_len, int64_t fold, const int32_t *block_fold,
                               const int64_t *reps_count, const int64_t *reps, int64_t slots, const void *munc,
                               int munc_f64, const uint8_t *active, const double *total, const double *lambda, double pad,
                               double rho, uint8_t *mask, double *kept, double *heldout, double *h, double *nominal);

/* ---- initial process-noise (Q0) seed natives (qseed_oracle.c) -------------------------------- */
enum {
    COR_QSEED_ERR_DATA = 1,        /* "active matrixData values must be finite"             pyx:1580 */
    COR_QSEED_ERR_OBSVAR = 2,      /* "active obsVar values must be positive finite"        pyx:1589 */
    COR_QSEED_ERR_TRANSITION = 3,  /* "active transition values must be finite"             pyx:1593 */
    COR_QSEED_ERR_PRECISION = 4,   /* "active transition precision must be positive finite" pyx:1596 */
    COR_QSEED_ERR_SAMPLECAP = 5,   /* "precisionSampleCap must be positive"                 pyx:1572 */
    COR_QSEED_ERR_POOLED = 6,      /* "active pooled observations must be finite with positive variance" pyx:1862 */
    COR_QSEED_ERR_POST_DELTA = 7,  /* "deltas must be finite"                               pyx:2003 */
    COR_QSEED_ERR_POST_S2 = 8,     /* "samplingVariances must be nonnegative finite"        pyx:2005 */
    COR_QSEED_ERR_POST_W = 9,      /* "transitionWeights must be positive finite"           pyx:2007 */
    COR_QSEED_ERR_POST_SCORE = 10, /* "q seed posterior produced a nonfinite score"         pyx:2112 */
    COR_QSEED_ERR_POST_NORM = 11   /* "q seed posterior normalization failed"               pyx:2121 */
};
typedef struct cor_qseed_diag {
    int64_t pairCount, sampledPairCount, precisionSampleCount, scanCount, candidateTransitionCount,
        selectedTransitionCount;
    int32_t cappedMode, pad_;
    double precisionCap, precisionCapFraction, transitionSampleFraction;
} cor_qseed_diag;
typedef struct cor_qseed_post {
    int64_t transitionCount;
    int32_t ok, pad_;
    double effectiveTransitionCount, medianSamplingVariance, priorLevel, posteriorModeLevel, posteriorMedianLevel,
        posteriorQ05Level, posteriorQ95Level, transitionQ90;
} cor_qseed_post;
/* pyx:1441-1797 `cEstimateSameTrackProcessNoiseTransitions` on float64 (m,n) matrices + uint8 activity mask */
int64_t cor_qseed_same_track(int64_t m, int64_t n, const double *data, const double *obs, const uint8_t *active,
                             double capQuantile, double capMultiplier, int64_t maxTransitionSamples,
                             int64_t precisionSampleCap, int64_t signalPanelSize, double *deltas, double *svar,
                             double *weights, cor_qseed_diag *dg);
/* pyx:1800-1902 `cEstimatePooledProcessNoiseTransitions` */
int64_t cor_qseed_pooled(int64_t m, int64_t n, const double *data, const double *obs, const uint8_t *active,
                         double *deltas, double *svar, double *weights);
/* pyx:1905-2146 `cQSeedPosteriorFromTransitions` */
int cor_qseed_posterior(int64_t count, const double *deltas, const double *s2, const double *weights, double qFloor,
                        double qCap, double robustTNu, double qSeedPriorLevel, int64_t minTransitions,
                        double priorLogSd, double defaultTNu, int64_t gridSize, cor_qseed_post *out);

#ifdef __cplusplus
}
#endif
#endif
