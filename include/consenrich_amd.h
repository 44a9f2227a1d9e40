/*
 * consenrich_amd.h -- C ABI of the MI355X (gfx950) implementation of Consenrich's estimator hot path.
 *
 * The reference has no FFI/plugin registry: its hot path sits behind the Python module-attribute interface
 * `consenrich.cconsenrich.<name>` (core.py:30, looked up at call time: core.py:3286-3290, 4231-4321, 4351-4413,
 * 3451, 3483).  This library is what a replacement of those callables binds to: every entry point below names the
 * reference callable it replaces ("pyx" = src/consenrich/cconsenrich.pyx).  Plain C: pointers + sizes, no torch,
 * no exceptions.  All functions return 0 on success, nonzero on failure; csr_last_error() gives the message
 * (thread-local).  Host buffers are caller-owned, C-contiguous, and never retained across calls; device buffers
 * are owned by the library (one context per GPU/process).
 *
 * Two API levels:
 *   (1) reference-shaped single-chain calls on HOST buffers  (csr_forward_pass, csr_backward_pass,
 *       csr_fixed_background_ecm, csr_expected_transition_residual_sums) -- drop-in for the Cython callables;
 *   (2) a device-resident multi-chain BATCH (csr_batch_*) -- many chromosomes/contigs per launch, inputs kept in
 *       HBM across sweeps; used by the genome driver, bench.py and the multi-GPU sharding (one context per rank).
 */
#ifndef CONSENRICH_AMD_H
#define CONSENRICH_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSR_ABI_VERSION 6

/* ---- model / flags ------------------------------------------------------------------------------------- */

typedef struct csr_model {
    int32_t state_dim;          /* 2 = levelTrend (pyx:291-529), 1 = level (pyx:538-707) */
    int32_t reserved_;
    double F[4];                /* row-major transition matrix (float32 values widened; pyx:6566-6569) */
    double Q0[4];               /* row-major base process noise (pyx:6570-6573); level uses Q0[0] */
    double state_init;          /* (double)(float) stateInit        (pyx:6593) */
    double state_covar_init;    /* (double)(float) stateCovarInit   (pyx:6594) */
    double pad;                 /* (double)(float) pad              (pyx:6595) */
    double w_min, w_max;        /* observation precision (lambda) clamp, pyx:433 */
    double k_min, k_max;        /* process precision (kappa) clamp, pyx:395 */
    double apn_min_q, apn_max_q, apn_thresh, apn_scale, apn_pc; /* adaptive process noise, pyx:510-527 */
} csr_model;

enum {
    CSR_USE_LAMBDA   = 1u << 0, /* useLambda        (pyx:6449) */
    CSR_USE_KAPPA    = 1u << 1, /* useProcPrec      (pyx:6453) */
    CSR_USE_QSCALE   = 1u << 2, /* useProcessQScale (pyx:6452) */
    CSR_USE_APN      = 1u << 3, /* ECM_useAPN after the qDiagBase veto (pyx:6575); sequential fallback path */
    CSR_RETURN_NLL   = 1u << 4, /* returnNLL */
    CSR_NLL_IN_D     = 1u << 5  /* storeNLLInD */
};

/* ---- (1) reference-shaped single-chain entry points (host buffers) ---------------------------------------- */

typedef struct csr_fwd_io {
    int64_t m, n;               /* trackCount, intervalCount */
    const float *data;          /* (m,n) C-order  matrixData */
    const float *munc;          /* (m,n) C-order  matrixPluginMuncInit */
    const float *lambda;        /* n or NULL */
    const float *kappa;         /* n or NULL */
    const float *qscale;        /* n or NULL */
    uint32_t flags;             /* CSR_USE_* / CSR_RETURN_NLL / CSR_NLL_IN_D; USE_x requires the pointer */
    uint32_t reserved_;
    float *D;                   /* n, always written (vectorD) */
    float *xf;                  /* (n,d)   or NULL: all three or none (doStore, pyx:6444) */
    float *Pf;                  /* (n,d,d) */
    float *pnoise;              /* (n,d,d): rows 0..n-2 written (pyx:504-508) */
} csr_fwd_io;

typedef struct csr_fwd_out {
    double sum_d;               /* sum of float32 D[k]  -> phiHat = sum_d/n (pyx:6627) */
    double sum_nll;             /* sumNLL (pyx:468) */
} csr_fwd_out;

/* replaces cforwardPass (pyx:6393-6632) and cforwardPassLevel (pyx:6853-7049).  intervalToBlockMap is validated by
 * the host wrapper (its only effect in the reference is the range check pyx:389-392). */
int csr_forward_pass(const csr_model *mdl, const csr_fwd_io *io, csr_fwd_out *out);

/* replaces cbackwardPass (pyx:6635-6850) and cbackwardPassLevel (pyx:7052-7150).
 * lag_rows = rows available in `lag` (>= max(n-1,1)); resid is (n,m). */
int csr_backward_pass(const csr_model *mdl, int64_t m, int64_t n, const float *data,
                      const float *xf, const float *Pf, const float *pnoise,
                      float *xs, float *Ps, float *lag, int64_t lag_rows, float *resid);

typedef struct csr_ecm_cfg {
    int64_t max_iters;          /* ECM_fixedBackgroundIters */
    int64_t inner_iters;        /* t_innerIters */
    double rtol;                /* (double)(float) ECM_fixedBackgroundRtol */
    double nu;                  /* (double)(float) ECM_robustTNu */
    int32_t use_lambda;         /* ECM_useObsPrecisionReweighting */
    int32_t use_kappa;          /* ECM_useProcessPrecisionReweighting && (!APN || qscale)  (pyx:7912) */
    int32_t use_apn;
    int32_t reserved_;
} csr_ecm_cfg;

typedef struct csr_ecm_out {
    int64_t iters_done;
    double final_nll;
    double initial_nll;
    double abs_rel_change;
    double rel_improvement;
    int64_t stable_iters;
    int64_t nll_increase_count;
    int32_t converged;
    int32_t skipped;            /* n <= 5 filter+smoother fallback (pyx:7998-8129) */
    int32_t has_initial_nll;
    int32_t reserved_;
} csr_ecm_out;

/* replaces cfixedBackgroundECM (pyx:7660-8442) and cfixedBackgroundECMLevel (pyx:7153-7657).
 * lambda / kappa: n, in/out (warm start already clipped by the caller, pyx:7899-7923), NULL when disabled.
 * nll_path: max_iters doubles or NULL (per-iteration NLL, the reference's optimization_path). */
int csr_fixed_background_ecm(const csr_model *mdl, const csr_ecm_cfg *cfg, int64_t m, int64_t n,
                             const float *data, const float *munc, const float *qscale,
                             float *lambda, float *kappa,
                             float *xs, float *Ps, float *lag, float *resid,
                             double *nll_path, csr_ecm_out *out);

/* replaces cExpectedTransitionResidualSums (pyx:710-815) and ...Level (pyx:818-863); float64 host inputs. */
int csr_expected_transition_residual_sums(int32_t state_dim, int64_t n, const double *xs, const double *Ps,
                                          const double *lag, const double *F,
                                          double *sum_level, double *sum_trend, int64_t *count);

/* ---- (2) device-resident multi-chain batch ---------------------------------------------------------------- */

typedef struct csr_ctx csr_ctx;

int csr_device_count(void);
csr_ctx *csr_create(int device_ordinal);    /* NULL on failure */
void csr_destroy(csr_ctx *ctx);
const char *csr_last_error(void);
int csr_abi_version(void);
/* "abi <n> src <sha256 of the library's sources, 16 hex digits> <compiler flags>": lets a caller (bench.py, __graft_entry__)
 * record WHICH build ran -- a prebuilt library that travelled with the tree is distinguishable from a rebuild of newer sources. */
const char *csr_build_id(void);

/* Speculative-block tuning: block_len (multiple of 32; 0 = keep / choose from the batch size) and warm-up lengths in
 * BINS (rounded up to a multiple of 16; negative = keep) for the forward covariance chain, the forward state chain and
 * the backward chain.  Results never depend on these beyond the documented validation tolerance. */
int csr_set_tuning(csr_ctx *ctx, int32_t block_len, int32_t warm_p, int32_t warm_x, int32_t warm_b);

/* Carry validation of the forward STATE chain.  0 (the default of every context, and of the default context the
 * reference-shaped single-chain entry points use, ctx == NULL): a speculative block is accepted only if its carry-in is
 * bit-equal to its predecessor's carry-out, so results == the sequential recursion bit for bit, reproducible and independent
 * of block length / warm-up.  This is the only mode that holds the 1e-5 parity gate through the ECM loop on ill-conditioned
 * data: there the reference's own arithmetic moves by several times the gate when a few inputs move by one float32 ulp
 * (tests/test_hard_data.py).  k > 0 opts into the throughput mode: a carry is also accepted when
 * |dx0| <= k and |F01||dx1| <= k float32 ulps of max(|level|, 1) -- ~1e-7 relative on the state track PER PASS, ~3 x faster;
 * the contract is per pass, not through an ECM loop.  CONSENRICH_AMD_XTOL_ULPS overrides the default of both. */
int csr_set_validation(csr_ctx *ctx, int32_t x_tol_ulps);

/* Describe a batch: n_chains independent chains (chromosomes) of chain_len[c] bins, m samples each. (Re)allocates. */
int csr_batch_configure(csr_ctx *ctx, const csr_model *mdl, int64_t m, int32_t n_chains,
                        const int64_t *chain_len);
/* Replace the model parameters (same state_dim) without reallocating or re-uploading. */
int csr_batch_set_model(csr_ctx *ctx, const csr_model *mdl);
/* Per-chain base process noise: q = n_chains x 4 doubles (row-major Q0 per chain, float32 values widened) or NULL = the
 * model's Q0 for every chain.  The reference seeds Q0 per chromosome (core.py:5667, csr_batch_qseed); F and the other
 * model parameters stay batch-wide. */
int csr_batch_set_chain_q(csr_ctx *ctx, const double *q);
/* H2D one chain's (m,n_c) C-order matrices. */
int csr_batch_upload(csr_ctx *ctx, int32_t chain, const float *data, const float *munc);
/* H2D optional per-bin multipliers (any may be NULL = leave as is).  After csr_batch_configure all multipliers are 1. */
int csr_batch_upload_multipliers(csr_ctx *ctx, int32_t chain, const float *lambda, const float *kappa,
                                 const float *qscale);
/* Fill every chain with the SURVEY 8(d) synthetic recipe directly in HBM (counter-based RNG). */
int csr_batch_synthesize(csr_ctx *ctx, uint64_t seed);
/* D2H of one chain's resident inputs, (m, n) float32 each (either pointer may be NULL). */
int csr_batch_download_inputs(csr_ctx *ctx, int32_t chain, float *data, float *munc);

/* a1: per-bin sufficient statistics of (data, munc, pad); must precede forward/ECM after any upload. */
int csr_batch_stats(csr_ctx *ctx);
/* a2-a5 over all chains.  sum_d / sum_nll: n_chains doubles each (may be NULL). */
int csr_batch_forward(csr_ctx *ctx, uint32_t flags, double *sum_d, double *sum_nll);
/* a6-a7 over all chains (uses the forward results resident on the device). */
int csr_batch_backward(csr_ctx *ctx);
/* csr_batch_forward + csr_batch_backward as ONE pipeline (core.py:4207 `_runForwardBackward` calls the two back to back):
 * a single host synchronisation; the NIS/NLL epilogue runs on a side stream concurrently with the smoother chain. */
int csr_batch_forward_backward(csr_ctx *ctx, uint32_t flags, double *sum_d, double *sum_nll);
/* One pass of the hot path in one call: csr_batch_stats + csr_batch_forward_backward +
 * csr_batch_export(what) (what = 0: none) + csr_batch_sums (both pointers NULL: none, validation stays pending).
 * Same kernels and results as the separate calls.  Knowing `what` in advance lets the bit-exact mode (constant process
 * noise, >= 2 chains) start the smoother / residuals of every chain as soon as THAT chain's filtered state stands, while the
 * state chain of slower chains is still running (csr_run_stats.tail_groups counts the groups). */
int csr_batch_step(csr_ctx *ctx, uint32_t flags, uint32_t what, double *sum_d, double *sum_nll);
/* The FORWARD FILTER ALONE in one call (cforwardPass without cbackwardPass, pyx:6393-6632 -- BASELINE config 2): csr_batch_stats +
 * csr_batch_forward + csr_batch_export(what & CSR_EXPORT_FORWARD) + csr_batch_sums.  Same kernels and results as the separate
 * calls; no smoother, no residuals. */
int csr_batch_step_forward(csr_ctx *ctx, uint32_t flags, uint32_t what, double *sum_d, double *sum_nll);
/* Per-chain sumD / sumNLL of the resident forward pass (n_chains doubles each, either may be NULL).  Lets a caller
 * queue csr_batch_export behind csr_batch_forward_backward(…, NULL, NULL) and synchronise once, here. */
int csr_batch_sums(csr_ctx *ctx, double *sum_d, double *sum_nll);
/* a8-a9 over all chains in lock-step; out: n_chains entries; nll_path: n_chains*max_iters or NULL. */
int csr_batch_ecm(csr_ctx *ctx, const csr_ecm_cfg *cfg, uint32_t flags, csr_ecm_out *out, double *nll_path);
/* Same, restricted to the chains with chain_mask[c] != 0 (NULL: all); the others keep the resident results of their last
 * fit untouched (out[c].skipped = 2) -- chromosomes of a batch stop their outer background/ECM alternation independently. */
int csr_batch_ecm_masked(csr_ctx *ctx, const csr_ecm_cfg *cfg, uint32_t flags, const unsigned char *chain_mask,
                         csr_ecm_out *out, double *nll_path);

enum { /* arrays, reference (natural) layout */
    CSR_ARR_D = 0,      /* (n)      float32 */
    CSR_ARR_XF,         /* (n,d)    */
    CSR_ARR_PF,         /* (n,d,d)  */
    CSR_ARR_PNOISE,     /* (n,d,d), rows 0..n-2 valid */
    CSR_ARR_XS,         /* (n,d)    */
    CSR_ARR_PS,         /* (n,d,d)  */
    CSR_ARR_LAG,        /* (n,d,d), rows 0..n-2 valid */
    CSR_ARR_RESID,      /* (n,m)    */
    CSR_ARR_LAMBDA,     /* (n)      */
    CSR_ARR_KAPPA,      /* (n)      */
    CSR_ARR_QSCALE,     /* (n)      process-noise scale (exported with CSR_EXPORT_MULT) */
    CSR_ARR_SUMGAIN0,   /* (n)      diagnostics, see csr_batch_diagnostics */
    CSR_ARR_SUMGAIN1,   /* (n)      */
    CSR_ARR_EFFQ_LEVEL, /* (n)      */
    CSR_ARR_EFFQ_TREND, /* (n)      */
    CSR_ARR_MUNCTRACE,  /* (n)      */
    CSR_ARR_BACKGROUND,       /* (n) current background (subtracted from the data), see csr_batch_background_* */
    CSR_ARR_BACKGROUND_NEXT,  /* (n) proposal of the last csr_batch_background_update */
    CSR_ARR_COUNT
};
enum {
    CSR_EXPORT_FORWARD = 1u << 0,   /* D, xf, Pf, pnoise */
    CSR_EXPORT_SMOOTH  = 1u << 1,   /* xs, Ps, lag */
    CSR_EXPORT_RESID   = 1u << 2,   /* resid */
    CSR_EXPORT_MULT    = 1u << 3    /* lambda, kappa, qscale */
};
/* Convert device-internal (block-transposed) results into reference-layout device arrays. */
int csr_batch_export(csr_ctx *ctx, uint32_t what);
/* D2H one exported array of one chain. */
int csr_batch_download(csr_ctx *ctx, int32_t chain, int32_t array_id, void *host_dst);
/* Device pointer + element count of an exported array (all chains, chain c at csr_batch_chain_offset). */
int csr_batch_device_array(csr_ctx *ctx, int32_t array_id, void **dev_ptr, int64_t *n_elems);
int64_t csr_batch_chain_offset(csr_ctx *ctx, int32_t chain);   /* in bins, -1 on error */
int csr_synchronize(csr_ctx *ctx);

/* Instrumentation: kernel timing with HIP events on the library's stream. */
typedef struct csr_kernel_time {
    char name[48];
    int64_t launches;
    double total_ms;
} csr_kernel_time;
int csr_profile_enable(csr_ctx *ctx, int32_t on);          /* also clears accumulated times */
int csr_profile_read(csr_ctx *ctx, csr_kernel_time *out, int32_t capacity, int32_t *n_out);

/* ---- per-interval output diagnostics (core.py:7734-7878 `_perIntervalOutputDiagnosticTracks`,
 * a per-bin Python loop in the reference) ---------------------------------------------------------------------------
 * From the resident forward pass of every chain: sumGain0/1 (total Kalman gain on level / trend, re-derived from the
 * STORED float32 filtered covariance of the previous bin exactly as the reference does), effectiveQLevel/Trend and
 * muncTrace = sum_j max(v_j+pad,1e-12)/lambda.  flags: CSR_USE_LAMBDA / CSR_USE_KAPPA / CSR_USE_QSCALE select the
 * resident multipliers (absent = None in the reference call); without CSR_USE_KAPPA the forward pass's own pNoise is
 * the effective process noise (core.py:7826-7848).  Results: CSR_ARR_SUMGAIN0.. arrays (download / device_array).
 * The five remaining reference tracks (baseQ*, preKappaQ*, processQScale) are O(1)-per-bin functions of Q0 and qScale
 * and are formed by the host mirror. */
int csr_batch_diagnostics(csr_ctx *ctx, uint32_t flags);
/* Same on host buffers, shaped like the reference call (drop-in for core._perIntervalOutputDiagnosticTracks):
 * Pf (n,d,d), munc (m,n); lambda / kappa / qscale / pnoise may be NULL (pnoise: (>=n-1,d,d)); outputs (n) float32. */
int csr_output_diagnostics(const csr_model *mdl, int64_t m, int64_t n, const float *Pf, const float *munc,
                           const float *lambda, const float *kappa, const float *qscale, const float *pnoise,
                           float *sum_gain0, float *sum_gain1, float *effq_level, float *effq_trend,
                           float *munc_trace);

/* ---- natives of the background update between ECM phases ------------------------------------
 * csr_solve_background replaces cconsenrich.csolveZeroCenteredBackground (pyx:944-1096) for a BATCH of independent
 * chains: solves (diag(w) + lam_first D1'D1 + lam D2'D2) x = rhs per chain (pentadiagonal SPD, fp64), optionally with
 * the zero-sum Lagrange correction.  weight / rhs / out: the chains' vectors concatenated (chain c has n[c] entries).
 * The reference's sequential LDL' is replaced by an exact two-level partition (see csrc/csr_background.h);
 * block_len = 0 picks the default partition size.  bad_index[c] (chain-local) / bad_value[c]: first pivot that had to be
 * raised to the 1e-12 floor, or -1 -- the reference raises RuntimeError in that case; `out` is filled regardless. */
int csr_solve_background(int32_t n_chains, const int64_t *n, const double *weight, const double *rhs, double lam,
                         double lam_first, int32_t zero_center, int32_t block_len, double *out, int64_t *bad_index,
                         double *bad_value);
/* cconsenrich.cbackgroundWeightedStatsWithSupport (pyx:9700-9724): weight[i] = sum_j inv_var[j,i],
 * rhs[i] = sum_j inv_var[j,i]*resid[j,i] in fp64 over float32 (m,n) C-order matrices; *support = #{weight > 0}. */
int csr_background_weighted_stats(int64_t m, int64_t n, const float *resid, const float *inv_var, double *weight,
                                  double *rhs, int64_t *support);

/* Device-resident background update for a whole batch (core.py:5064-5137 + 8085-8378): weight / rhs tracks from the
 * resident ORIGINAL data, munc, smoothed level (and lambda), conditioning guard, the pentadiagonal solve and the
 * asymmetric-IRLS wrapper (<= max_passes re-solves with the negative-part penalty), all chains in lock-step.  Nothing
 * crosses PCIe but a few scalars per chain.  The proposal lands in CSR_ARR_BACKGROUND_NEXT; csr_batch_background_apply
 * makes it the current background, which csr_batch_stats / the residuals then subtract from the data in float32 exactly
 * like the reference's `dataAdjusted` (core.py:3253). */
typedef struct csr_bg_cfg {
    double lam_first, lam;              /* core.py:7478-7491 `_backgroundPenaltyWeightsFromSpan` */
    double negative_penalty_multiplier; /* fitParams.backgroundNegativePenaltyMultiplier; <= 0 / non-finite: plain solve */
    int32_t zero_center;                /* fitParams.ECM_zeroCenterBackground */
    int32_t use_nonnegative;            /* fitParams.useNonnegativeBackground */
    int32_t use_lambda;                 /* multiply 1/max(munc+pad,1e-8) by clip(lambda) (core.py:5065-5074) */
    int32_t use_initial;                /* bit 0 (CSR_BG_INIT_FROM_CURRENT): seed the first solve with the current background's
                                           negative mask (initialBackground, core.py:8306-8316); bit 1 (CSR_BG_ZERO_STATE):
                                           the smoothed level is taken as zero -- the background warm start from the weighted
                                           data, core.py:2809-2910 `_estimateBackgroundWarmStart` (no fit needs to be resident) */
    int32_t max_passes;                 /* reference: 5 */
    int32_t block_len;                  /* partition size of the solver, 0 = default */
} csr_bg_cfg;
enum { CSR_BG_INIT_FROM_CURRENT = 1, CSR_BG_ZERO_STATE = 2 };
enum { CSR_BG_OK = 0, CSR_BG_NO_SUPPORT = 1, CSR_BG_BAD_PIVOT = 2, CSR_BG_UNRELIABLE = 3, CSR_BG_NONFINITE = 4 };
typedef struct csr_bg_out {
    int64_t support;                    /* bins with positive weight */
    double weight_sum, weight_scale;    /* sum of weights; median of the positive weights (IRLS penalty scale) */
    double roundoff_index;              /* eps * (1 + (4 lam_first + 16 lam) / mean positive weight), core.py:8160-8187 */
    double shift_rms;                   /* sqrt(sum w (next - current)^2 / sum w), core.py:5199-5215 */
    double proposal_rms, reference_rms; /* same weighting of next / current (core.py:5216-5240: shift tolerance scale) */
    int64_t bad_index;                  /* first modified pivot (CSR_BG_BAD_PIVOT) */
    double bad_value;
    int32_t passes;                     /* IRLS re-solves performed */
    int32_t status;                     /* CSR_BG_*: the reference raises for 2, 3, 4 and returns zeros for 1 */
} csr_bg_out;
int csr_batch_background_update(csr_ctx *ctx, const csr_bg_cfg *cfg, csr_bg_out *out /* n_chains */);
/* current background := last proposal for the chains with take[c] != 0 (NULL: all).  Invalidates the statistics. */
int csr_batch_background_apply(csr_ctx *ctx, const unsigned char *take);
/* H2D a background track for one chain (NULL: zeros). */
int csr_batch_set_background(csr_ctx *ctx, int32_t chain, const float *background);

/* ---- terms of the penalised objective of the outer stop rule (core.py:4418-4538) ---------------------------
 * For every chain, from the resident variances, multipliers and CURRENT background: robust precision penalties
 * 0.5 nu sum(x - log x) (core.py:3161-3179; kappa skips the first bin), roughness penalties 0.5 lam sum d^2
 * (core.py:3182-3204), negative-part penalty 0.5 mult median(positive weights) sum min(bg,0)^2 (core.py:4431-4463; float64
 * weight track sum_j clip(lambda)/max(munc+pad,1e-8)), effective observation count (core.py:2981-2986).  The caller adds
 * the forward NLL of (data - background) (csr_batch_stats + csr_batch_forward_masked) and divides by the count. */
typedef struct csr_objective_cfg {
    double nu, lam_first, lam, negative_penalty_multiplier, pad;
    int32_t use_lambda_penalty, use_kappa_penalty, use_lambda_weights, use_nonnegative;
} csr_objective_cfg;
typedef struct csr_objective_terms {
    double robust_observation_penalty, robust_process_penalty, first_difference_penalty, second_difference_penalty,
        negative_penalty, weight_median;
    int64_t effective_observation_count;
} csr_objective_terms;
int csr_batch_objective_terms(csr_ctx *ctx, const csr_objective_cfg *cfg, csr_objective_terms *out);
/* The two per-phase diagnostics of `runConsenrich` that read the (m, n) matrices, as per-bin float64 tracks of ONE chain from
 * the resident data / variances, the smoothed level of the last ECM phase and the CURRENT background:
 *   rel[i]  = level_i - sum_j (data_ji - bg_i) w_ji / sum_j w_ji with w_ji = 1 / max(munc_ji + pad, 1e-12), summed row by row in
 *             float64 over the cells with finite data and finite positive munc_ji + pad, NaN where there is none: the track whose
 *             sign changes per kb `_relativeSignChangePerKB` counts (core.py:2647-2700; called at core.py:4980, 5485);
 *   fit[i]  = sum_j iv_ji (r_ji - g_i)^2, cnt[i] = #cells with finite r and finite positive iv (both NULL: skipped), with the
 *             matrices of the background update (float32 iv = 1 / max(munc + pad, 1e-8) (* clip(lambda) if use_lambda), float32
 *             r = data - level, model pad; core.py:5064-5076) and g = the PROPOSAL of the last csr_batch_background_update:
 *             0.5 sum(fit) is `background_weighted_residual_objective`, sum(cnt) the effective observation count of
 *             `_scoreBackgroundFitObjective` (core.py:4540-4606; called at core.py:5161).
 * rel / fit: n doubles, cnt: n int32 (host).  The caller's part is O(n). */
int csr_batch_phase_tracks(csr_ctx *ctx, int32_t chain, int32_t use_lambda, double pad, double *rel, double *fit, int32_t *cnt);
/* The final forward gain summary of `runConsenrich`'s run diagnostics (`_finalForwardReplicateGainContigSummary`,
 * core.py:7671-7731) for one chain, from the resident forward pass: per replicate j the gains
 * g_k = max(Pf00_k, 0) * clip(lambda_k, lam_lo, lam_hi) / max(munc_jk + pad, 1e-12) in float64 (lambda = 1 unless use_lambda) and, over
 * the FINITE ones, out[j*9 + ...] = {count, mean, standard deviation (np.std), then the six order statistics x[lo], x[hi] around
 * the positions (count - 1) q of q = 0.25, 0.5, 0.75 (NumPy's 'linear' method interpolates between exactly these)}.  The
 * reference sorts m rows of n float64 values on the host; here a byte-wise radix select on the device (csrc/csr_gain.h), exact.
 * out: m * 9 doubles.  (ABI 6) */
int csr_batch_gain_summary(csr_ctx *ctx, int32_t chain, int32_t use_lambda, double pad, double lam_lo, double lam_hi, double *out);
/* csr_batch_forward for the chains with chain_mask[c] != 0 only (NULL: all); the others keep their resident results. */
int csr_batch_forward_masked(csr_ctx *ctx, uint32_t flags, const unsigned char *chain_mask, double *sum_d,
                             double *sum_nll);

/* ---- track writer ------------------------------------------------------------------------------
 * bedGraph text of one track, byte for byte what the reference emits (consenrich.py:9797-9805: pandas to_csv with
 * sep="\t", header=False, index=False, float_format="%.4f", lineterminator="\n"; NaN -> empty field, +-inf -> inf/-inf).
 * Rows: chrom \t start \t end \t value \n.  Intervals: starts/ends (n int64 each) or, if NULL, start0 + k*step with
 * end = min(start + step, end_cap) (end_cap <= 0: no cap).  transform: CSR_BGW_NONE, CSR_BGW_ROUND4 (np.round(x, 4) in
 * float32 = core.getPrimaryState, core.py:6145-6166) or CSR_BGW_SQRT (uncertainty = sqrt(P00), consenrich.py:9476).
 * Returns the number of bytes of the text; writes it to `out` if out != NULL and out_capacity is large enough (call
 * once with out = NULL to size the buffer).  Negative on error. */
enum { CSR_BGW_NONE = 0, CSR_BGW_ROUND4 = 1, CSR_BGW_SQRT = 2 };
int64_t csr_format_bedgraph(const char *chrom, int64_t n, const int64_t *starts, const int64_t *ends, int64_t start0,
                            int64_t step, int64_t end_cap, const float *values, int32_t transform, char *out,
                            int64_t out_capacity);
/* Same for component `comp` of an exported array of one chain of a batch (values never leave the device as floats). */
int64_t csr_batch_format_bedgraph(csr_ctx *ctx, int32_t chain, int32_t array_id, int32_t comp, int32_t transform,
                                  const char *chrom, int64_t start0, int64_t step, int64_t end_cap, char *out,
                                  int64_t out_capacity);

/* bigWig (io.py:530 `convertBedGraphToBigWig`, io.py:633-760 `_convertBedGraphToBigWigPyBigWig`: the reference converts its
 * bedGraph files with pyBigWig).  The fixed-record body of a bigWig file (Kent et al. 2010, BigWig/BigBed file format) for one
 * track of one chain, formatted on the device from the exported array like csr_batch_format_bedgraph:
 *   csr_batch_bigwig_sections  the UNCOMPRESSED data sections, back to back: 24-byte header (chromId, chromStart, chromEnd,
 *                              itemStep 0, itemSpan 0, type 1 = bedGraph, reserved, itemCount) + 12-byte items (start, end,
 *                              float32 value), items_per_section items each (the last one fewer), + the total summary;
 *   csr_batch_bigwig_zoom      reduction records of bins_per_record consecutive intervals (32 bytes: chromId, start, end,
 *                              validCount, min, max, sum, sumSquares as float32) for one zoom level.
 * An item's value is what the reference's file holds: float32 of the decimal text "%.4f" of the (transformed) track value --
 * pyBigWig receives the values parsed back from the bedGraph rows.  Both return the byte count (call with out = NULL to size
 * the buffer), negative on error.  Chromosome tree, R-tree index, zlib of the blocks and the headers are O(sections) host
 * work (consenrich_amd/writers.py). */
typedef struct csr_bw_summary {
    int64_t bases_covered, non_finite;      /* non_finite: items whose value is NaN / inf (the reference refuses such rows) */
    double min_val, max_val, sum_data, sum_squares;
} csr_bw_summary;
int64_t csr_batch_bigwig_sections(csr_ctx *ctx, int32_t chain, int32_t array_id, int32_t comp, int32_t transform,
                                  uint32_t chrom_id, int64_t start0, int64_t step, int64_t end_cap, int32_t items_per_section,
                                  unsigned char *out, int64_t out_capacity, csr_bw_summary *total);
int64_t csr_batch_bigwig_zoom(csr_ctx *ctx, int32_t chain, int32_t array_id, int32_t comp, int32_t transform,
                              uint32_t chrom_id, int64_t start0, int64_t step, int64_t end_cap, int64_t bins_per_record,
                              unsigned char *out, int64_t out_capacity);

/* ---- natives of the delete-block uncertainty calibration (cuncertainty.pyx) -------------------
 * csr_observation_total_information = cobservationTotalInformation (pyx:97-157); csr_fold_mask_and_information =
 * cmakeFoldMaskAndInformation (pyx:160-305) after its argument validation.  munc is float32 (munc_is_f64 = 0) or float64,
 * (m,n) C-order; active (m,n) uint8; lambda NULL = useLambda False; reps (block_count, slots) int64; nominal may be NULL.
 * Same summation orders as the reference, IEEE division / sqrt: the tracks are bit-identical. */
int csr_observation_total_information(int64_t m, int64_t n, const void *munc, int32_t munc_is_f64, const uint8_t *active,
                                      const double *lambda, double pad, double rho, double *total);
int csr_fold_mask_and_information(int64_t m, int64_t n, int64_t block_len, int64_t fold, const int32_t *block_fold,
                                  const int64_t *reps_count, const int64_t *reps, int64_t slots, const void *munc,
                                  int32_t munc_is_f64, const uint8_t *active, const double *total, const double *lambda,
                                  double pad, double rho, uint8_t *mask, double *kept, double *heldout, double *h,
                                  double *nominal);
/* A fold as an extra chain of a batch (uncertainty.py:1370-1419 runs one fit per fold): chain `dst` (same length as
 * `src`) receives src's data and src's variances with the fold's deleted cells set to masked_variance (the reference:
 * 1e30, constants.py:387); kept / heldout / h (n doubles each, host) as above with every cell active, total computed on
 * the fly.  Nothing but the fold spec and the three tracks crosses PCIe. */
int csr_batch_make_fold(csr_ctx *ctx, int32_t src, int32_t dst, int64_t block_len, int64_t fold, const int32_t *block_fold,
                        const int64_t *reps_count, const int64_t *reps, int64_t slots, int32_t use_lambda, double pad,
                        double rho, float masked_variance, double *kept, double *heldout, double *h);

/* ---- the initial process-noise (Q0) seed ---------------------------------------------------------
 * csr_qseed_same_track = cEstimateSameTrackProcessNoiseTransitions (cconsenrich.pyx:1441-1797), csr_qseed_pooled =
 * cEstimatePooledProcessNoiseTransitions (pyx:1800-1902), csr_qseed_posterior = cQSeedPosteriorFromTransitions
 * (pyx:1905-2146), all after the Python-level argument validation.  Matrices are (m,n) C-order float64 + uint8 mask, as
 * the reference's natives take them; outputs need capacity min(n-1, max_transition_samples > 0 ? that : n-1) (same
 * track) or n-1 (pooled).  Data-dependent errors return nonzero with the reference's message in csr_last_error().
 * The sampled columns are gathered and reduced on the device; the bounded tail (two quantiles of <= precision_sample_cap
 * precisions, the stable ordering of <= 32 000 signal levels, the 64-point grid posterior over <= 2048 transitions) is
 * host arithmetic inside the library, in the reference's operation order (bit-identical transitions). */
typedef struct csr_qseed_sample_cfg {
    double precision_cap_quantile, precision_cap_multiplier;   /* core.py:277-278: 0.95, 20 */
    int64_t max_transition_samples, precision_sample_cap, signal_panel_size;   /* core.py:273-276: 32000, 32000, 2048 */
} csr_qseed_sample_cfg;
typedef struct csr_qseed_sample_diag {
    int64_t pair_count, sampled_pair_count, precision_sample_count, scan_count, candidate_count, selected_count;
    int32_t capped_mode, reserved;
    double precision_cap, precision_cap_fraction, transition_sample_fraction;
} csr_qseed_sample_diag;
typedef struct csr_qseed_post_cfg {
    double q_floor, q_cap /* may be +inf */, robust_t_nu, q_seed_prior_level;
    int64_t min_transitions;        /* core.py:272: 8 */
    double prior_log_sd, default_t_nu;   /* core.py:279-280: log 4, 8 */
    int64_t grid_size;              /* core.py:275: 64 */
} csr_qseed_post_cfg;
typedef struct csr_qseed_post {
    int64_t transition_count;
    int32_t ok, reserved;
    double effective_transition_count, median_sampling_variance, prior_level, posterior_mode, posterior_median,
        posterior_q05, posterior_q95, transition_q90;
} csr_qseed_post;
int csr_qseed_same_track(int64_t m, int64_t n, const double *data, const double *obs_var, const uint8_t *active,
                         const csr_qseed_sample_cfg *cfg, double *deltas, double *sampling_var, double *weights,
                         int64_t *out_count, csr_qseed_sample_diag *diag);
int csr_qseed_pooled(int64_t m, int64_t n, const double *data, const double *obs_var, const uint8_t *active,
                     double *deltas, double *sampling_var, double *weights, int64_t *out_count);
int csr_qseed_posterior(int64_t count, const double *deltas, const double *sampling_var, const double *weights,
                        const csr_qseed_post_cfg *cfg, csr_qseed_post *out);
/* The caller `_estimateInitialProcessNoiseFromData` (core.py:3621-3780) for every chain of a batch, on the resident
 * float32 data / variance matrices (nothing is converted or masked on the host; only the sampled columns are read):
 * same-track estimate -> pooled fallback -> median-observation-variance fallback -> clamps.  out: n_chains records.
 * source: 0 sameTrackEB, 1 pooledEB, 2 observationVarianceFloor, 3 minQ; reason: 0 ok, 1 fallback_observation_variance,
 * 2 fallback_min_q, 3 insufficient_transition_support (never final).  state_dim 2 = levelTrend, 1 = level. */
typedef struct csr_qseed_cfg {
    csr_qseed_sample_cfg sample;
    double pad, min_q, max_q /* < 0 or non-finite: no cap */, delta_f, robust_t_nu /* NaN: default */, q_seed_prior_level;
    int64_t min_transitions;
    double prior_log_sd, default_t_nu;
    int64_t grid_size;
    int32_t state_dim, reserved;
} csr_qseed_cfg;
typedef struct csr_qseed_out {
    double q_level, q_trend;                 /* diagonal of matrixQ before its float32 cast (qSeedLevelFinal / TrendFinal) */
    double level_pre_clamp, trend_pre_clamp;
    int32_t source, reason;
    csr_qseed_sample_diag sample;            /* of the same-track scan */
    csr_qseed_post post;                     /* of the estimate that was used (ok = 0: none was) */
} csr_qseed_out;
int csr_batch_qseed(csr_ctx *ctx, const csr_qseed_cfg *cfg, csr_qseed_out *out);

/* ---- the one collective of the path -- final track gather over RCCL / xGMI ----------------------------------
 * The reference fits chromosomes in a sequential loop of one process (consenrich.py:8809) and has no communication at all;
 * here contigs are sharded over the GPUs of a node (one process and one csr_ctx per GPU, no data-path collective) and the
 * per-bin output tracks are gathered ONCE at the end.  RCCL is bound at run time (dlopen librccl.so.1): only these entry
 * points need it.  No PyTorch: rank 0 creates the 128-byte unique id, the caller's launcher distributes it (bench.py: a file
 * on the node), every rank creates its communicator on its context's device. */
typedef struct csr_comm csr_comm;
int csr_comm_unique_id(char *id128);                       /* ncclGetUniqueId; 128 bytes */
csr_comm *csr_comm_create(csr_ctx *ctx, const char *id128, int32_t world, int32_t rank);     /* NULL on failure */
void csr_comm_destroy(csr_comm *comm);
int csr_comm_world(csr_comm *comm);
int csr_comm_rank(csr_comm *comm);
/* *value := max over ranks (ncclAllReduce on the library's stream + stream synchronisation: also a barrier that completes
 * only when every rank's queue is drained). */
int csr_comm_allreduce_max(csr_comm *comm, double *value);
/* *value := sum over ranks: an all-reduce of 1.0 tells the caller how many ranks the communicator really spans. */
int csr_comm_allreduce_sum(csr_comm *comm, double *value);
int csr_comm_barrier(csr_comm *comm);
/* (smoothed level xs[:,0], its variance Ps[:,0,0]) of every bin of the rank's chains, packed chain after chain on the device
 * straight from the exported arrays (CSR_EXPORT_SMOOTH; no host bounce), then ncclAllGather: every rank ends with
 * world x cap_bins (level, variance) pairs; rank r's chains start at pair r * cap_bins.  cap_bins = max over ranks of the
 * rank's total bins (same value on every rank).  host_out: world * cap_bins * 2 floats, or NULL.  Fails when the exported
 * arrays are not those of the resident fit (a pass ran since the last CSR_EXPORT_SMOOTH). */
int csr_batch_gather_tracks(csr_ctx *ctx, csr_comm *comm, int64_t cap_bins, float *host_out);

typedef struct csr_run_stats {
    int64_t blocks;             /* speculative blocks in the batch */
    int64_t fix_launches;       /* validation/fix-up kernel launches so far */
    int64_t reruns_p, reruns_x, reruns_b;  /* blocks re-run because the speculative carry-in was not bit-equal */
    int32_t block_len, warm_p, warm_x, warm_b;
    int32_t x_tol_ulps;
    int32_t pipeline_redos;     /* optimistic (deferred) validations that failed and re-ran their pipeline synchronously */
    int64_t local_repairs;      /* blocks a warm-started ECM sweep validated against their neighbour INSIDE the speculative
                                   kernel, found wanting and re-ran there (as of the last read-back) */
    int32_t ws_warm_f, ws_warm_b;   /* windows (bins) of the warm-started ECM sweeps, forward / smoother */
    int64_t sb_bailouts;        /* bit-exact state chain: single launches (k_sb_async) that gave up on a bounded wait; the pass
                                   form then ran instead (same results) */
    int64_t tail_groups;        /* bit-exact steps: groups of chains whose smoother / residuals were launched on their own (the
                                   first ones while the state chain of the other chains was still running) */
    int64_t nat_first_use_off_main; /* reference-layout arrays whose first use (allocation + zeroing) happened while a tail group's
                                   stream was current; harmless since ABI 4 (the zeroing is waited for on the host before the
                                   array is handed out) -- counted so that a test can show the path was exercised */
} csr_run_stats;
int csr_get_run_stats(csr_ctx *ctx, csr_run_stats *out);

#ifdef __cplusplus
}
#endif
#endif
