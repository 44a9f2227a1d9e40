"""SURVEY a12 behind the reference's OWN signature: ``runConsenrich(matrixData, matrixMunc, deltaF, minQ, maxQ, *, ...)``
(core.py:3861-3916: same parameter names, order and defaults; return tuple of core.py:6126-6142) on the DEVICE-RESIDENT fit
(`driver.run_consenrich_batch` on a one-chromosome batch) instead of the per-call drop-in path.  A maintainer binds it with

    import consenrich.core, consenrich_amd.core_api
    consenrich.core.runConsenrich = consenrich_amd.core_api.runConsenrich

and the reference's CLI (consenrich.py:9204-9249) and its calibration fold loop (uncertainty.py:1414) reach the resident fit.

Three layers, so that the same call can be replayed on the CPU twin in the tests:
  * `resolve_call`   -- pure host code: the reference's validation (its ValueError texts) and the mapping of its ~50 arguments
                        onto `ModelParams` / `FitConfig`; nothing is silently ignored: what this path cannot honour raises;
  * `runConsenrich`  -- uploads the two matrices once, runs the resident fit, downloads the final pass;
  * `assemble_result`-- pure host code: tuple variants and the run-diagnostics dict from the final pass's arrays.

Arguments that the reference's own body never reads (processNoiseWarmupECMIters, processNoiseWarmupOuterPasses: signature
only, core.py:3900-3901) and its logging knobs (logIndentLevel, logRunRole) are accepted and have no effect, like there.
`projectStateDuringFiltering` is accepted and has no effect, like there too: the reference only forwards it to
`cforwardPass` / `cforwardPassLevel` (core.py:4284, 4391), whose loops never read it (pyx:6393-6632) -- this
package's own `cforwardPass` mirror accepts and ignores it the same way.  `intervalSizeBP` feeds the one diagnostic it
feeds there (relative sign changes per kb, core.py:2647-2700, 4980).
The run-diagnostics mapping carries the reference's complete key set (core.py:5943-5999), and every record of
`post_process_noise_fit["fixed_background_ecm"]` the keys the reference's own contract test reads (test_core.py:4111-4163:
multiplier summaries and bound hits, sign-change rate, background-fit objective, the ECM's convergence state and -- with
`trackOptimizationPath` -- its per-iteration rows): `PassDiagnostics`, from per-bin tracks the device forms out of the resident
matrices (`csr_batch_phase_tracks`) and (n,) tracks downloaded per phase, only when `returnDiagnostics` asks for them."""
from __future__ import annotations

import atexit
import operator
import os
import threading
from dataclasses import dataclass, field
from typing import Any, Optional

import numpy as np

from . import _lib as L
from .batch import DeviceBatch, ModelParams
from .driver import ChainFit, FitConfig, run_consenrich_batch

# constants.py:134-135, 149-163, 268-275, 387
STATE_MODEL_LEVEL, STATE_MODEL_LEVEL_TREND = "level", "levelTrend"
PROCESS_NOISE_CALIBRATION_FIXED_DIAGONAL, PROCESS_NOISE_CALIBRATION_FIXED = "fixedDiagonal", "fixed"
PROCESS_DEFAULT_NOISE_CALIBRATION = PROCESS_NOISE_CALIBRATION_FIXED_DIAGONAL
PROCESS_DEFAULT_Q_SEED_PRIOR_LEVEL = 1.0e-5
PROCESS_DEFAULT_WARMUP_ECM_ITERS, PROCESS_DEFAULT_WARMUP_OUTER_PASSES = 50, 2
PROCESS_DEFAULT_PRECISION_MULTIPLIER_MIN, PROCESS_DEFAULT_PRECISION_MULTIPLIER_MAX = 5.0e-3, 5.0e3
FIT_DEFAULT_T_INNER_ITERS = 5
FIT_DEFAULT_USE_NONNEGATIVE_BACKGROUND = True
FIT_DEFAULT_BACKGROUND_NEGATIVE_PENALTY_MULTIPLIER = 1.0
MASKED_OBSERVATION_VARIANCE = np.float32(1.0e30)
FIXED_PROCESS_Q = 1.0e-4                                    # core.py:4038

_DEVICE = 0


def set_device(device: int) -> None:
    """GPU the next calls run on (one process per GPU: a rank sets its LOCAL_RANK once)."""
    global _DEVICE
    _DEVICE = int(device)


# One device context per GPU is kept between calls (streams, mailboxes, work buffers: 40-60 ms per call to create and destroy; the
# reference's CLI makes one call per chromosome).  `csr_batch_configure` releases everything of the previous batch before it lays
# out the next one, like the per-call callables of `cconsenrich` do on the library's default context.  A call that fails drops
# its context; concurrent calls from other threads get contexts of their own.  CONSENRICH_AMD_CORE_API_KEEP_CONTEXT=0: a fresh
# context per call.  `release_device()` frees the kept context's device memory.
_KEPT: dict = {}
_KEPT_LOCK = threading.Lock()


def _keep_contexts() -> bool:
    return os.environ.get("CONSENRICH_AMD_CORE_API_KEEP_CONTEXT", "1") != "0"


class _Context:
    """`with _Context(device) as batch:` -- the kept context of that GPU if it is free, else a temporary one."""

    def __init__(self, device: int):
        self.device, self.batch, self.kept = int(device), None, False

    def __enter__(self) -> DeviceBatch:
        if _keep_contexts() and _KEPT_LOCK.acquire(blocking=False):
            self.kept = True
            self.batch = _KEPT.pop(self.device, None)
        if self.batch is None:
            try:
                self.batch = DeviceBatch(self.device)
            except BaseException:
                if self.kept:
                    _KEPT_LOCK.release()
                raise
        return self.batch

    def __exit__(self, exc_type, exc, tb):
        try:
            if self.kept and exc_type is None:
                _KEPT[self.device] = self.batch
            else:
                self.batch.close()
        finally:
            if self.kept:
                _KEPT_LOCK.release()
        return False


def release_device(device: Optional[int] = None) -> None:
    """Destroy the context(s) kept between calls (all GPUs when `device` is None)."""
    with _KEPT_LOCK:
        for dev in [d for d in list(_KEPT) if device is None or d == int(device)]:
            _KEPT.pop(dev).close()


atexit.register(release_device)


# ---- the reference's argument checks (core.py:2202-2291, 2703-2780), restated with its messages -------------------------------
def _finite_positive(name, value):
    v = float(value)
    if not np.isfinite(v) or v <= 0.0:
        raise ValueError(f"`{name}` must be positive and finite")
    return v


def _finite_nonnegative(name, value):
    v = float(value)
    if not np.isfinite(v) or v < 0.0:
        raise ValueError(f"`{name}` must be nonnegative and finite")
    return v


def _bounds(prefix, lo, hi):
    lo_, hi_ = _finite_positive(f"{prefix}PrecisionMultiplierMin", lo), _finite_positive(f"{prefix}PrecisionMultiplierMax", hi)
    if hi_ < lo_:
        raise ValueError(f"`{prefix}PrecisionMultiplierMax` must be >= `{prefix}PrecisionMultiplierMin`")
    return lo_, hi_


def _process_bounds(lo, hi, nu, state_dim):
    hi_ = _finite_positive("processPrecisionMultiplierMax", hi)
    lo_raw = float(lo)
    if not np.isfinite(lo_raw):
        raise ValueError("`processPrecisionMultiplierMin` must be finite")
    if lo_raw < 0.0:                                        # "auto": the convexity-preserving bound (core.py:2231-2245)
        if nu is None:
            raise ValueError("`fitParams.ECM_robustTNu` must be positive and finite when "
                             "`processParams.precisionMultiplierMin` is negative.")
        nu_ = _finite_positive("ECM_robustTNu", nu)
        lo_ = (nu_ + float(state_dim)) / (2.0 * nu_) + 1.0e-4
        if hi_ < lo_:
            raise ValueError("`processPrecisionMultiplierMax` must be >= the auto `processPrecisionMultiplierMin` "
                             f"convexity-preserving lower bound {float(lo_):.6g}")
    else:
        lo_ = _finite_positive("processPrecisionMultiplierMin", lo_raw)
    if hi_ < lo_:
        raise ValueError("`processPrecisionMultiplierMax` must be >= `processPrecisionMultiplierMin`")
    return lo_, hi_


def _optional_vector(name, value, n):
    if value is None:
        return None
    arr = np.asarray(value, dtype=np.float32)
    if arr.shape != (int(n),):
        raise ValueError(f"`{name}` must have length {int(n)}")
    if not np.all(np.isfinite(arr)):
        raise ValueError(f"`{name}` must contain only finite values")
    return np.ascontiguousarray(arr, dtype=np.float32)


def _optional_q(value):
    if value is None:
        return None
    arr = np.asarray(value, dtype=np.float32)
    if arr.shape != (2, 2):
        raise ValueError("`initialProcessQ` must have shape (2, 2)")
    if not np.all(np.isfinite(arr)):
        raise ValueError("`initialProcessQ` must contain only finite values")
    if arr[0, 0] <= 0.0 or arr[1, 1] <= 0.0:
        raise ValueError("`initialProcessQ` diagonal entries must be positive")
    if not np.allclose(arr, arr.T, rtol=1.0e-5, atol=1.0e-8):
        raise ValueError("`initialProcessQ` must be symmetric")
    try:
        np.linalg.cholesky(arr.astype(np.float64) + 1.0e-8 * np.eye(2))
    except Exception as exc:
        raise ValueError("`initialProcessQ` must be positive definite") from exc
    return np.ascontiguousarray(arr, dtype=np.float32)


def construct_matrix_f(deltaF: float) -> np.ndarray:
    """core.constructMatrixF (core.py:2164-2176)."""
    F = np.eye(2, dtype=np.float32)
    F[0, 1] = np.float32(deltaF)
    return F


def construct_matrix_q(min_diag, q00, q01, q10, q11, tol=1.0e-8) -> np.ndarray:
    """core.constructMatrixQ (core.py:3781-3858): diagonal floor, |Q01| <= 0.99 sqrt(Q00 Q11), Cholesky check."""
    floor = _finite_positive("minDiagQ", min_diag)
    diag = lambda v: floor if (v is None or not np.isfinite(float(v))) else max(float(v), floor)     # noqa: E731
    Q = np.empty((2, 2), np.float32)
    Q[0, 0], Q[1, 1] = np.float32(diag(q00)), np.float32(diag(q11))
    Q[0, 1], Q[1, 0] = np.float32(q01), np.float32(q10)
    if not np.allclose(Q[0, 1], Q[1, 0], rtol=0.0, atol=1e-4):
        raise ValueError(f"Matrix is not symmetric: Q=\n{Q}")
    cap = np.float32(0.99) * np.sqrt(Q[0, 0] * Q[1, 1]).astype(np.float32)
    Q[0, 1] = np.clip(Q[0, 1], -cap, cap)
    Q[1, 0] = Q[0, 1]
    try:
        np.linalg.cholesky(Q.astype(np.float64) + tol * np.eye(2))
    except Exception as ex:
        raise ValueError(f"Process noise covariance Q is not positive definite:\n{Q}") from ex
    return Q


def clamp_process_noise_matrix(Q0, state_model, minQ, maxQ) -> np.ndarray:
    """core._clampProcessNoiseMatrix (core.py:3525-3546): diagonal into [minQ, cap], cap = inf for a negative maxQ."""
    floor = _finite_positive("minQ", minQ)
    cap = np.inf if float(maxQ) < 0.0 else max(float(maxQ), floor)
    q = np.asarray(Q0, np.float64)
    clamp = lambda v: min(max(float(v), floor), cap)                                                  # noqa: E731
    if state_model == STATE_MODEL_LEVEL:
        return np.asarray([[clamp(q[0, 0])]], np.float32)
    return construct_matrix_q(floor, clamp(q[0, 0]), float(q[0, 1]), float(q[1, 0]), clamp(q[1, 1]))


def background_penalties(blockLenIntervals, smoothness=1.0):
    """core._backgroundPenaltyWeightsFromSpan (core.py:7479-7491)."""
    span = max(2.0, float(blockLenIntervals))
    s = float(smoothness)
    return float(max(1.0, s * span * span / 4.0)), float(max(1.0, s * span ** 4 / 16.0))


def track_summary(values) -> dict:
    """core._metadataTrackSummary (core.py:2390-2409)."""
    a = np.asarray(values, np.float64).reshape(-1)
    ok = np.isfinite(a)
    if not bool(ok.all()):
        a = a[ok]
    if a.size == 0:
        return {k: None for k in ("min", "q05", "median", "mean", "q95", "max")}
    lo, hi = float(a.min()), float(a.max())
    if lo == hi:                    # a constant track (base Q, unit scales): every order statistic is that value, no selection passes
        return {"min": lo, "q05": lo, "median": lo, "mean": float(a.mean()), "q95": lo, "max": lo}
    # one selection pass for both quantiles (np.quantile partitions around all the positions it needs at once; the values are
    # those of two separate calls), one for the median (np.median: the mean of the two middle values, not the lerp of np.quantile)
    q05, q95 = np.quantile(a, [0.05, 0.95])
    return {"min": lo, "q05": float(q05), "median": float(np.median(a)), "mean": float(a.mean()), "q95": float(q95), "max": hi}


def track_summaries(tracks) -> list:
    """`track_summary` of several tracks; long ones side by side on a few threads (NumPy's selection releases the GIL)."""
    tracks = list(tracks)
    if len(tracks) < 2 or max(np.size(t) for t in tracks) < (1 << 16) or _workers() < 2:
        return [track_summary(t) for t in tracks]
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=min(len(tracks), _workers())) as pool:
        return list(pool.map(track_summary, tracks))


def _workers() -> int:
    try:
        return max(1, min(8, len(os.sched_getaffinity(0))))
    except AttributeError:          # not on Linux
        return max(1, min(8, os.cpu_count() or 1))


def _map_rows(fn, rows):
    """fn over the rows of a matrix, in order; large matrices through a small thread pool (NumPy releases the GIL in its loops)."""
    rows = list(rows)
    if len(rows) < 2 or rows[0].size < (1 << 16) or _workers() < 2:
        return [fn(r) for r in rows]
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=_workers()) as pool:
        return list(pool.map(fn, rows))


def _map_cols(fn, n, chunk=1 << 18):
    """fn(k0, k1) over column ranges that tile [0, n): every range runs the reference's own row-by-row accumulation, so sums keep
    their order; ranges are independent (thread pool)."""
    spans = [(k0, min(k0 + chunk, n)) for k0 in range(0, n, chunk)]
    if len(spans) < 2 or _workers() < 2:
        return [fn(a, b) for a, b in spans]
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=_workers()) as pool:
        return list(pool.map(lambda ab: fn(*ab), spans))


@dataclass
class RunPlan:
    """One validated `runConsenrich` call: what runs on the device (`model`, `cfg`, the matrices, warm starts) and what shapes
    the result (`ret`)."""
    data: np.ndarray
    munc: np.ndarray
    state_model: str
    model: ModelParams
    cfg: FitConfig
    block_len_intervals: int
    q0: Optional[np.ndarray]                    # float32 (d,d) when fixed / given; None = seeded from the data on the device
    q_policy: str
    initial_background: Optional[np.ndarray]
    initial_lambda: Optional[np.ndarray]
    initial_kappa: Optional[np.ndarray]
    requested_kappa: bool
    ret: dict = field(default_factory=dict)
    interval_size_bp: Optional[int] = None
    q_given: bool = False                       # initialProcessQ was passed (process-noise calibration "skipped", core.py:5689-5700)


def resolve_call(matrixData, matrixMunc, deltaF, minQ, maxQ, *, stateInit, stateCovarInit, boundState, stateLowerBound,
                 stateUpperBound, blockLenIntervals, intervalSizeBP=None, projectStateDuringFiltering=False, pad=1.0e-4,
                 ECM_fixedBackgroundIters=50, ECM_fixedBackgroundRtol=1.0e-4, t_innerIters=FIT_DEFAULT_T_INNER_ITERS,
                 ECM_robustTNu=8.0, ECM_useObsPrecisionReweighting=True, ECM_useProcessPrecisionReweighting=True,
                 ECM_useAPN=False, ECM_zeroCenterBackground=False, ECM_outerIters=3, ECM_minOuterIters=None,
                 ECM_backgroundShiftRtol=1.0e-3, ECM_outerNLLRtol=1.0e-4, ECM_backgroundSmoothness=1.0, fitBackground=True,
                 useNonnegativeBackground=FIT_DEFAULT_USE_NONNEGATIVE_BACKGROUND,
                 backgroundNegativePenaltyMultiplier=FIT_DEFAULT_BACKGROUND_NEGATIVE_PENALTY_MULTIPLIER, returnScales=True,
                 returnBackground=False, stateModel=STATE_MODEL_LEVEL_TREND,
                 processNoiseCalibration=PROCESS_DEFAULT_NOISE_CALIBRATION, qSeedPriorLevel=PROCESS_DEFAULT_Q_SEED_PRIOR_LEVEL,
                 processNoiseWarmupECMIters=PROCESS_DEFAULT_WARMUP_ECM_ITERS,
                 processNoiseWarmupOuterPasses=PROCESS_DEFAULT_WARMUP_OUTER_PASSES, observationPrecisionMultiplierMin=0.25,
                 observationPrecisionMultiplierMax=4.0, processPrecisionMultiplierMin=PROCESS_DEFAULT_PRECISION_MULTIPLIER_MIN,
                 processPrecisionMultiplierMax=PROCESS_DEFAULT_PRECISION_MULTIPLIER_MAX, observationMask=None,
                 initialBackground=None, initialObservationPrecision=None, initialProcessPrecision=None, initialProcessQ=None,
                 trackOptimizationPath=False, returnPrecisionDiagnostics=False, returnDiagnostics=False, logIndentLevel=0,
                 logRunRole=None) -> RunPlan:
    """core.py:3952-4115 (argument handling) and :5658-5690 (F, base Q0), without touching a GPU."""
    data = np.ascontiguousarray(matrixData, dtype=np.float32)                       # core.py:2737-2756
    munc = np.ascontiguousarray(matrixMunc, dtype=np.float32)
    if data.ndim == 1:
        data = data[None, :]
    elif data.ndim != 2:
        raise ValueError(f"matrixData must be 1D or 2D (got ndim={data.ndim})")
    if munc.ndim == 1:
        munc = munc[None, :]
    elif munc.ndim != 2:
        raise ValueError(f"matrixMunc must be 1D or 2D (got ndim={munc.ndim})")
    if data.shape != munc.shape:
        raise ValueError("matrixData and matrixMunc must have identical shapes")
    if observationMask is not None:                                                 # core.py:2759-2780
        mask = np.asarray(observationMask, dtype=bool)
        if mask.ndim == 1:
            mask = np.broadcast_to(mask[None, :], munc.shape)
        if mask.shape != munc.shape:
            raise ValueError("observationMask must match matrixData shape")
        munc = munc.copy(order="C")
        munc[~mask] = MASKED_OBSERVATION_VARIANCE
    pad = _finite_nonnegative("pad", pad)
    m, n = data.shape
    if n < 2:
        raise ValueError("need at least 2 intervals for smoothing")
    if intervalSizeBP is not None and int(intervalSizeBP) <= 0:
        raise ValueError("intervalSizeBP must be positive when provided")
    # projectStateDuringFiltering: forwarded to the forward passes by the reference (core.py:4284, 4391), read by neither loop
    requested_kappa = bool(ECM_useProcessPrecisionReweighting)
    use_apn = bool(ECM_useAPN)
    use_kappa = requested_kappa and not use_apn                                     # core.py:3974-3976
    if stateModel is None:
        state_model = STATE_MODEL_LEVEL_TREND
    else:
        state_model = str(stateModel).strip()
        if state_model not in (STATE_MODEL_LEVEL, STATE_MODEL_LEVEL_TREND):
            raise ValueError(f"stateModel must be one of ('level', 'levelTrend') (got {stateModel!r})")
    d = 1 if state_model == STATE_MODEL_LEVEL else 2
    lam_bounds = _bounds("observation", observationPrecisionMultiplierMin, observationPrecisionMultiplierMax)
    kap_bounds = _process_bounds(processPrecisionMultiplierMin, processPrecisionMultiplierMax, ECM_robustTNu, d)
    bg0 = _optional_vector("initialBackground", initialBackground, n)
    lam0 = _optional_vector("initialObservationPrecision", initialObservationPrecision, n)
    if lam0 is not None:
        lam0 = np.ascontiguousarray(np.clip(lam0, *lam_bounds), np.float32)
    kap0 = _optional_vector("initialProcessPrecision", initialProcessPrecision, n)
    if kap0 is not None:
        kap0 = np.ascontiguousarray(np.clip(kap0, *kap_bounds), np.float32)
    q_given = _optional_q(initialProcessQ)
    minQ = _finite_positive("minQ", minQ)
    maxQ = float(maxQ)
    if np.isnan(maxQ):
        raise ValueError("`maxQ` must not be NaN")
    max_q_apn = np.inf if maxQ < 0.0 else max(maxQ, minQ)
    mode = str(processNoiseCalibration).strip()
    if mode not in (PROCESS_NOISE_CALIBRATION_FIXED_DIAGONAL, PROCESS_NOISE_CALIBRATION_FIXED):
        raise ValueError(f"processNoiseCalibration must be one of ('fixedDiagonal', 'fixed') (got {processNoiseCalibration!r})")
    fixed_q = q_given is None and mode == PROCESS_NOISE_CALIBRATION_FIXED
    if fixed_q and minQ > FIXED_PROCESS_Q:
        raise ValueError("`minQ` must not exceed the fixed process Q")
    if fixed_q and np.isfinite(max_q_apn) and max_q_apn < FIXED_PROCESS_Q:
        raise ValueError("`maxQ` must be negative or at least the fixed process Q")
    if isinstance(t_innerIters, (bool, np.bool_)):
        raise ValueError("t_innerIters must be a positive integer")
    try:
        t_inner = operator.index(t_innerIters)
    except TypeError as ex:
        raise ValueError("t_innerIters must be a positive integer") from ex
    if t_inner <= 0:
        raise ValueError("t_innerIters must be a positive integer")
    outer = max(1, int(ECM_outerIters))
    min_outer = 3 if ECM_minOuterIters is None else max(1, int(ECM_minOuterIters))
    neg_mult = None if backgroundNegativePenaltyMultiplier is None else float(backgroundNegativePenaltyMultiplier)
    if neg_mult is not None and not np.isfinite(neg_mult):
        raise ValueError("`backgroundNegativePenaltyMultiplier` must be finite or None")
    if int(blockLenIntervals) <= 0:
        raise ValueError("blockLenIntervals must be positive")
    if not np.isfinite(float(stateCovarInit)) or not np.isfinite(float(stateInit)):
        raise ValueError("stateInit and stateCovarInit must be finite")

    # transition matrix and base process noise (core.py:5658-5690)
    if state_model == STATE_MODEL_LEVEL:
        delta_fit = 1.0
    else:
        delta_fit = float(deltaF)
        if not np.isfinite(delta_fit) or delta_fit <= 0.0:
            raise ValueError("deltaF must be a positive finite fixed step size")
    F = construct_matrix_f(delta_fit)
    if q_given is not None:
        q0, policy = clamp_process_noise_matrix(q_given, state_model, minQ, maxQ), PROCESS_NOISE_CALIBRATION_FIXED
    elif mode == PROCESS_NOISE_CALIBRATION_FIXED:
        q0 = clamp_process_noise_matrix(construct_matrix_q(minQ, FIXED_PROCESS_Q, 0.0, 0.0, FIXED_PROCESS_Q), state_model, minQ, maxQ)
        policy = mode
    else:
        q0, policy = None, mode                             # estimated from the data on the device (core.py:5666-5676)
    q_model = np.zeros((2, 2), np.float32)
    if q0 is not None:
        q_model[:d, :d] = q0[:d, :d]
    else:
        q_model[0, 0] = q_model[1, 1] = np.float32(FIXED_PROCESS_Q)       # placeholder until the seed replaces it per chain
    model = ModelParams(state_dim=d, F=tuple(map(tuple, F.astype(np.float64))), Q0=tuple(map(tuple, q_model.astype(np.float64))),
                        state_init=float(stateInit), state_covar_init=float(stateCovarInit), pad=float(pad),
                        lambda_bounds=lam_bounds, kappa_bounds=kap_bounds,
                        apn=(float(minQ), float(max_q_apn) if np.isfinite(max_q_apn) else 3.0e38, 5.0, 10.0, 2.0))   # pyx defaults of dStat*
    cfg = FitConfig(penalties=background_penalties(int(blockLenIntervals), float(ECM_backgroundSmoothness)),
                    ecm_iters=int(ECM_fixedBackgroundIters), ecm_rtol=float(ECM_fixedBackgroundRtol), inner_iters=int(t_inner),
                    nu=float(ECM_robustTNu), use_lambda=bool(ECM_useObsPrecisionReweighting), use_kappa=use_kappa,
                    use_apn=use_apn, fit_background=bool(fitBackground), zero_center=bool(ECM_zeroCenterBackground),
                    use_nonnegative=bool(useNonnegativeBackground), neg_multiplier=0.0 if neg_mult is None else neg_mult,
                    outer_passes=outer, min_outer=min_outer, shift_rtol=float(max(ECM_backgroundShiftRtol, 0.0)),
                    seed_q=q0 is None, min_q=float(minQ), max_q=float(maxQ), delta_f=float(delta_fit),
                    q_seed_prior_level=float(qSeedPriorLevel), outer_nll_rtol=float(max(ECM_outerNLLRtol, 0.0)), pad=float(pad))
    ret = {"scales": bool(returnScales), "background": bool(returnBackground),
           "precision": bool(returnPrecisionDiagnostics), "diagnostics": bool(returnDiagnostics),
           "bound_state": bool(boundState), "lower": float(stateLowerBound), "upper": float(stateUpperBound),
           "track_path": bool(trackOptimizationPath), "use_apn": use_apn, "max_q_apn": float(max_q_apn), "min_q": float(minQ)}
    return RunPlan(data=data, munc=munc, state_model=state_model, model=model, cfg=cfg,
                   block_len_intervals=int(blockLenIntervals), q0=q0, q_policy=policy, initial_background=bg0,
                   initial_lambda=lam0 if cfg.use_lambda else None, initial_kappa=kap0 if use_kappa else None,
                   requested_kappa=requested_kappa, ret=ret,
                   interval_size_bp=None if intervalSizeBP is None else int(intervalSizeBP), q_given=q_given is not None)


# ---- host restatements of the reference's diagnostics helpers (pure NumPy on downloaded / host arrays) ---------------------------
def metadata_float(value):
    """diagnostics.metadataFloat (diagnostics.py:19-23): None for a non-finite value."""
    v = float(value)
    return v if np.isfinite(v) else None


def process_noise_calibration_support(data, munc, pad) -> dict:
    """core._processNoiseCalibrationSupport (core.py:2989-3055): counts of usable cells / transitions.  (Evaluated range by range
    with one bin of overlap for the adjacency counts: the float64 copies of two (m, n) matrices are never whole in memory.)"""
    data, munc = np.asarray(data), np.asarray(munc)
    n = int(data.shape[1])
    half = 0.5 * float(MASKED_OBSERVATION_VARIANCE)

    def part(k0, k1):
        hi = min(k1 + 1, n)                                   # one bin of look-ahead: transitions k -> k + 1 with k in [k0, k1)
        d64, m64 = np.asarray(data[:, k0:hi], np.float64), np.asarray(munc[:, k0:hi], np.float64)
        obs_var = m64 + float(pad)
        unmasked = np.isfinite(m64) & (m64 < half)
        positive = np.isfinite(obs_var) & (obs_var > 0.0)
        active = np.isfinite(d64) & unmasked & positive
        act_iv = np.any(active, axis=0)
        own = k1 - k0
        adj = int(np.count_nonzero(act_iv[1:] & act_iv[:-1])) if hi - k0 >= 2 else 0
        same = int(np.count_nonzero(np.any(active[:, 1:] & active[:, :-1], axis=0))) if hi - k0 >= 2 else 0
        return (int(np.count_nonzero(np.isfinite(d64[:, :own]))), int(np.count_nonzero((unmasked & positive)[:, :own])),
                int(np.count_nonzero(active[:, :own])), int(np.count_nonzero(act_iv[:own])), adj, same)

    parts = _map_cols(part, n)
    finite_n, pos_n, act_n, iv_n, adj_n, same_n = (int(sum(p[i] for p in parts)) for i in range(6))
    reason = None
    if finite_n <= 0:
        reason = "no_finite_data"
    elif pos_n <= 0:
        reason = "no_positive_observation_variance"
    elif act_n <= 0:
        reason = "no_active_observations"
    elif adj_n <= 0:
        reason = "no_active_adjacent_transitions"
    return {"finiteDataCount": finite_n, "positiveObservationVarianceCount": pos_n, "activeObservationCount": act_n,
            "activeIntervalCount": iv_n, "intervalTransitionCount": int(max(n - 1, 0)),
            "activeAdjacentTransitionCount": adj_n, "sameTrackAdjacentTransitionCount": same_n,
            "processNoiseCalibrationCanRun": bool(reason is None), "processNoiseCalibrationSkipReason": reason}


def process_noise_q_boundary_diagnostics(Q0, state_model, minQ, maxQ) -> dict:
    """core._processNoiseQBoundaryDiagnostics (core.py:3058-3101)."""
    d = 1 if state_model == STATE_MODEL_LEVEL else 2
    floor = _finite_positive("minQ", minQ)
    mq = float(maxQ)
    cap = float("inf") if (mq < 0.0 or not np.isfinite(mq)) else float(max(mq, floor))
    q = np.asarray(Q0, np.float64)

    def clamp(v):
        v = float(v)
        if not np.isfinite(v):
            v = floor
        v = max(v, floor)
        return float(min(v, cap)) if np.isfinite(cap) else float(v)

    lvl = clamp(q[0, 0])
    trd = 0.0 if d == 1 else clamp(q[1, 1])
    hit_lf, hit_tf = bool(lvl <= 1.0001 * floor), bool(d == 2 and trd <= 1.0001 * floor)
    hit_lc = bool(np.isfinite(cap) and lvl >= 0.9999 * cap)
    hit_tc = bool(d == 2 and np.isfinite(cap) and trd >= 0.9999 * cap)
    hit_f, hit_c = bool(hit_lf or hit_tf), bool(hit_lc or hit_tc)
    status = "floor_and_cap" if (hit_f and hit_c) else "floor" if hit_f else "cap" if hit_c else "interior"
    return {"preKappaQLevel": lvl, "preKappaQTrend": trd, "qFloor": float(floor), "qCap": float(cap),
            "hitQLevelFloor": hit_lf, "hitQTrendFloor": hit_tf, "hitQLevelCap": hit_lc, "hitQTrendCap": hit_tc,
            "hitQFloor": hit_f, "hitQCap": hit_c, "qBoundaryStatus": status}


def static_process_noise_calibration_diagnostics(*, policy, status, reason, Q0, state_model, minQ, maxQ, support,
                                                 warm_start_process_noise) -> dict:
    """core._staticProcessNoiseCalibrationDiagnostics (core.py:3104-3158)."""
    d = 1 if state_model == STATE_MODEL_LEVEL else 2
    bnd = process_noise_q_boundary_diagnostics(Q0, state_model, minQ, maxQ)
    q_final = clamp_process_noise_matrix(Q0, state_model, minQ, maxQ)
    lvl, trd = float(bnd["preKappaQLevel"]), float(bnd["preKappaQTrend"])
    ratio = 0.0 if d == 1 else trd / max(lvl, float(bnd["qFloor"]))
    out = {"processNoisePolicy": policy, "processNoiseCalibrationStatus": status, "processNoiseCalibrationReason": reason,
           "stateModel": state_model, "preKappaQLevel": lvl, "preKappaQTrend": trd, "rawTrendLevelRatio": float(ratio),
           "effectiveTrendLevelRatio": float(ratio), "logQLevel": float(np.log(max(lvl, float(bnd["qFloor"])))),
           "logQTrend": 0.0 if d == 1 else float(np.log(max(trd, float(bnd["qFloor"])))),
           "usedInitialProcessQFallback": bool(status != "estimated" and float(warm_start_process_noise) <= 0.0),
           "matrixQ0Final": q_final.astype(float).tolist(), "warmStartProcessNoise": float(warm_start_process_noise),
           "globalScale": 1.0, "windowCount": 0, "validTransitionCount": 0, "qScaleClampFraction": 0.0}
    out.update(bnd)
    out.update(dict(support))
    return out


def summarize_precision_boundary_hits(*, observationPrecision, observationPrecisionMin, observationPrecisionMax,
                                      processPrecision, processPrecisionMin, processPrecisionMax) -> dict:
    """diagnostics.summarizePrecisionBoundaryHits (diagnostics.py:181-247): final multipliers pinned to their bounds."""
    def one(values, lower, upper, skip_first=False):
        if values is None:
            return {"enabled": False, "total": 0, "lower": 0, "upper": 0, "lower_fraction": None, "upper_fraction": None}
        arr = np.asarray(values, np.float64).reshape(-1)
        if skip_first and arr.size > 0:
            arr = arr[1:]
        fin = arr[np.isfinite(arr)]
        total = int(fin.size)
        lo = int(np.sum(np.isclose(fin, float(lower), rtol=0.0, atol=1.0e-6 * max(abs(float(lower)), 1.0))))
        hi = int(np.sum(np.isclose(fin, float(upper), rtol=0.0, atol=1.0e-6 * max(abs(float(upper)), 1.0))))
        return {"enabled": True, "total": total, "lower": lo, "upper": hi,
                "lower_fraction": metadata_float(lo / float(total)) if total > 0 else None,
                "upper_fraction": metadata_float(hi / float(total)) if total > 0 else None}

    return {"observation": one(observationPrecision, observationPrecisionMin, observationPrecisionMax),
            "process": one(processPrecision, processPrecisionMin, processPrecisionMax, skip_first=True),
            "bounds": {"observation": [float(observationPrecisionMin), float(observationPrecisionMax)],
                       "process": [float(processPrecisionMin), float(processPrecisionMax)]}}


def precision_bound_hits(values, lower, upper, skip_first=False):
    """core._precisionBoundHits (core.py:2591-2611): fractions of the finite multipliers at or beyond each bound."""
    if values is None:
        return None, None
    arr = np.asarray(values, np.float64).reshape(-1)
    if skip_first and arr.size > 1:
        arr = arr[1:]
    fin = arr[np.isfinite(arr)]
    if fin.size == 0:
        return None, None
    lo = int(np.count_nonzero(fin <= float(lower))) if np.isfinite(float(lower)) else 0
    hi = int(np.count_nonzero(fin >= float(upper))) if np.isfinite(float(upper)) else 0
    return metadata_float(lo / float(fin.size)), metadata_float(hi / float(fin.size))


def relative_sign_change_per_kb(state_level, data, munc, *, interval_size_bp, background=None, pad=0.0):
    """core._relativeSignChangePerKB + _signChangePerKB (core.py:2614-2700): sign changes per kb of (smoothed level - the
    inverse-variance weighted mean of the background-adjusted observations), ignoring values below 1 % of the mean magnitude."""
    if state_level is None or interval_size_bp is None or int(interval_size_bp) <= 0:
        return None
    x = np.asarray(state_level, np.float64).reshape(-1)
    d, v = np.asarray(data), np.asarray(munc)
    if d.ndim != 2 or v.shape != d.shape or d.shape[1] != x.size:
        return None
    bg = np.zeros(x.size) if background is None else np.asarray(background, np.float64).reshape(-1)
    if bg.size != x.size:
        return None
    mean = np.full(x.size, np.nan)
    x_ok = np.isfinite(x)

    def cols(k0, k1):       # the reference's row-by-row accumulation (core.py:2675-2692) on one range of bins
        tot, wsum = np.zeros(k1 - k0), np.zeros(k1 - k0)
        for j in range(d.shape[0]):
            row, den = np.asarray(d[j, k0:k1], np.float64), np.asarray(v[j, k0:k1], np.float64) + float(pad)
            ok = x_ok[k0:k1] & np.isfinite(row) & np.isfinite(den) & (den > 0.0)
            if not np.any(ok):
                continue
            w = 1.0 / np.maximum(den[ok], 1.0e-12)
            tot[ok] += (row[ok] - bg[k0:k1][ok]) * w
            wsum[ok] += w
        has = wsum > 0.0
        out = np.full(k1 - k0, np.nan)
        out[has] = tot[has] / wsum[has]
        mean[k0:k1] = out

    _map_cols(cols, x.size)
    return sign_change_per_kb(x - mean, interval_size_bp)


def sign_change_per_kb(values, interval_size_bp):
    """core._signChangePerKB (core.py:2614-2644): sign changes of a track per kb of its span; values below 1 % of the mean
    magnitude (and exact zeros) do not count as a side."""
    if values is None or interval_size_bp is None or int(interval_size_bp) <= 0:
        return None
    arr = np.asarray(values, np.float64).reshape(-1)
    if arr.size == 0:
        return None
    fin = arr[np.isfinite(arr)]
    if fin.size == 0:
        return None
    mean_abs = float(np.mean(np.abs(fin), dtype=np.float64))
    if not np.isfinite(mean_abs):
        return None
    if 0.01 * mean_abs > 0.0:
        fin = fin[np.abs(fin) >= 0.01 * mean_abs]
    sg = np.sign(fin)
    sg = sg[sg != 0.0]
    changes = int(np.count_nonzero(sg[1:] * sg[:-1] < 0.0)) if sg.size >= 2 else 0
    span_kb = float(arr.size) * float(int(interval_size_bp)) / 1000.0
    if not np.isfinite(span_kb) or span_kb <= 0.0:
        return None
    return metadata_float(float(changes) / span_kb)


def _sum_squares(v) -> float:
    """sum of squares in float64 with NumPy's own pairwise reduction -- NOT `np.dot`: a BLAS level-1 call wakes one spinning
    thread per host core, and inside a CPU quota (a container's share of a large host) that stalls the whole process, the
    device's launch path included, for the rest of the scheduler period (measured: ~70 ms per call next to a 12 ms ECM phase).
    The reference's own `np.dot` here has no fixed summation order either (it is whatever its BLAS does)."""
    v = np.asarray(v, np.float64)
    return float(np.sum(v * v, dtype=np.float64))


def multiplier_summary(values, lower, upper, skip_first=False):
    """core._observationLambdaSummary / _processKappaSummary (core.py:2338-2375): (mean, median) of the finite multipliers clipped
    to their bounds; the first kappa multiplies nothing and is left out (skip_first).  A float32 track is selected in float32
    (half the bytes to partition) and averaged in float64: the same two numbers as the reference's float64 expressions, since
    float32 -> float64 is exact and order-preserving."""
    if values is None:
        return None, None
    arr = np.asarray(values).reshape(-1)
    if arr.dtype != np.float32:
        arr = arr.astype(np.float64)
    if skip_first and arr.size > 1:
        arr = arr[1:]
    ok = np.isfinite(arr)
    fin = arr if bool(ok.all()) else arr[ok]
    if fin.size == 0:
        return None, None
    clipped = np.clip(fin, fin.dtype.type(lower), fin.dtype.type(upper)) if fin.dtype == np.float32 and \
        float(np.float32(lower)) == float(lower) and float(np.float32(upper)) == float(upper) else \
        np.clip(fin.astype(np.float64), float(lower), float(upper))
    k = clipped.size // 2
    if clipped.size % 2:
        median = float(np.partition(clipped, k)[k])
    else:
        part = np.partition(clipped, [k - 1, k])
        median = (float(part[k - 1]) + float(part[k])) / 2.0
    return metadata_float(float(np.mean(clipped.astype(np.float64, copy=False)))), metadata_float(median)


def background_fit_objective(data, munc, state_level, lam, background, *, pad, lambda_bounds, penalties, use_nonnegative,
                             negative_penalty_multiplier) -> dict:
    """`_scoreBackgroundFitObjective` (core.py:4540-4606) of a background PROPOSAL against the phase that produced it, with the
    matrices that phase's update forms (core.py:5064-5076): float32 inverse variances 1 / max(munc + pad, 1e-8) times the
    clipped observation precision, float32 residuals data - smoothed level; then in float64
    0.5 sum w (r - g)^2 + roughness penalties (`_backgroundObjectivePenalty`, core.py:3182-3204) + negative-part penalty
    0.5 (multiplier x median positive weight track) sum min(g, 0)^2, per finite cell of positive weight.
    Evaluated range by range (`_map_cols`); the per-range sums are added in order."""
    d, v = np.asarray(data, np.float32), np.asarray(munc, np.float32)
    n = d.shape[1]
    x = np.asarray(state_level, np.float32).reshape(-1)
    g = np.asarray(background, np.float64).reshape(-1)
    prec = None if lam is None else np.clip(np.asarray(lam, np.float32).reshape(1, n), np.float32(lambda_bounds[0]),
                                            np.float32(lambda_bounds[1])).astype(np.float32)
    track = np.zeros(n)

    def cols(k0, k1):
        inv = (np.float32(1.0) / np.maximum(v[:, k0:k1] + np.float32(pad), np.float32(1.0e-8))).astype(np.float32)
        if prec is not None:
            inv *= prec[:, k0:k1]
        res = d[:, k0:k1] - x[None, k0:k1]
        inv64, res64 = inv.astype(np.float64), res.astype(np.float64)
        fit = res64 - g[None, k0:k1]
        track[k0:k1] = np.sum(inv64, axis=0, dtype=np.float64)
        return (float(np.sum(inv64 * fit * fit, dtype=np.float64)),
                int(np.count_nonzero(np.isfinite(res64) & np.isfinite(inv64) & (inv64 > 0.0))))

    parts = _map_cols(cols, n)
    weighted = 0.5 * float(sum(p[0] for p in parts))
    count = float(max(1, sum(p[1] for p in parts)))
    lam_first, lam_second = penalties
    d1 = np.diff(g)
    first = 0.5 * float(lam_first) * _sum_squares(d1) if g.size >= 2 else 0.0
    d2 = np.diff(g, n=2)
    second = 0.5 * float(lam_second) * _sum_squares(d2) if g.size >= 3 else 0.0
    negative = 0.0
    mult = negative_penalty_multiplier
    if use_nonnegative and mult is not None and float(mult) > 0.0:
        pos = track[np.isfinite(track) & (track > 0.0)]
        scale = float(np.median(pos)) if pos.size else 1.0
        if not np.isfinite(scale) or scale <= 0.0:
            scale = 1.0
        negative = 0.5 * float(float(mult) * scale) * float(np.sum(np.minimum(g, 0.0) ** 2, dtype=np.float64))
    objective = float(weighted + (first + second) + negative)
    return {"background_weighted_residual_objective": weighted, "background_smoothness_penalty": float(first + second),
            "background_first_difference_penalty": float(first), "background_second_difference_penalty": float(second),
            "background_negative_penalty": float(negative), "background_objective": objective,
            "background_objective_per_cell": float(objective / count), "background_effective_observation_count": count}


class PassDiagnostics:
    """The per-phase summaries the reference computes from its host arrays after every fixed-background ECM phase
    (core.py:4946-4990 in the loop, :5456-5517 for the final phase) and after every background proposal (:5161-5197), for any
    chain of a batch.  The two that read the (m, n) matrices -- the sign-change rate and the weighted residual term of the
    background-fit objective -- come as per-bin float64 tracks from the device (`DeviceBatch.phase_tracks`: the reference's
    per-cell arithmetic on the resident matrices); everything else is O(n) on (n,) tracks downloaded per phase, on worker threads
    while the device runs the next phase.  They are diagnostics only (no stop rule reads them): `driver.fit_batch` runs without
    this object unless they were asked for (`returnDiagnostics`)."""

    def __init__(self, cfg: FitConfig, model: ModelParams, interval_size_bp, overlap: bool = True, workers: Optional[int] = None):
        self.cfg, self.model, self.interval_size_bp = cfg, model, interval_size_bp
        self.previous_per_cell = {}
        # overlap: the device calls (downloads, the tracks kernel) happen in the caller's thread at once; the NumPy part runs on a
        # few worker threads while the caller goes on to launch the next phase -- the methods then return a Future of the
        # summary instead of the summary.  The jobs of ONE chain run in submission order (the pass-to-pass test of the objective
        # needs that): each waits for its chain's previous job, which was queued before it and is therefore running or done.
        self._pool = None
        self._sets, self._slot, self._busy, self._prev = {}, {}, {}, {}
        if overlap:
            from concurrent.futures import ThreadPoolExecutor

            self._pool = ThreadPoolExecutor(max_workers=max(1, min(4, _workers()) if workers is None else int(workers)))

    @staticmethod
    def _after(prev, fn, *args):
        if prev is not None:
            prev.result()
        return fn(*args)

    def _run(self, c, key, fn, *args):
        if self._pool is None:
            return fn(*args)
        job = self._pool.submit(self._after, self._prev.get(c), fn, *args)
        self._prev[c] = self._busy[key] = job
        return job

    def _buffers(self, c, n):
        """Two sets of host tracks per chain, used in turn: a worker reads one set while the chain's next phase fills the other;
        fresh arrays per phase would spend more time in first-touch page faults than in the copies."""
        slot = self._slot.get(c, 0)
        self._slot[c] = slot ^ 1
        key = (c, slot)
        if self._busy.get(key) is not None:         # the job that read this set two phases ago
            self._busy.pop(key).result()
        if key not in self._sets or self._sets[key]["rel"].shape != (n,):
            self._sets[key] = {"lambda": np.empty(n, np.float32), "kappa": np.empty(n, np.float32), "g": np.empty(n, np.float32),
                               "rel": np.empty(n, np.float64), "fit": np.empty(n, np.float64), "cnt": np.empty(n, np.int32)}
        return key, self._sets[key]

    def _fetch(self, batch, c, with_fit):
        cfg = self.cfg
        key, buf = self._buffers(c, batch.chain_lens[c])
        batch.export(L.EXPORT_MULT)
        lam = batch.download(c, "lambda", out=buf["lambda"]) if cfg.use_lambda else None
        kap = batch.download(c, "kappa", out=buf["kappa"]) if cfg.use_kappa else None
        rel, fit, cnt = batch.phase_tracks(c, float(cfg.pad), with_fit=with_fit, use_lambda=cfg.use_lambda,
                                           out=(buf["rel"], buf["fit"], buf["cnt"]))
        return lam, kap, rel, fit, cnt, buf, key

    def _summaries(self, lam, kap, rel) -> dict:
        lam_b, kap_b = self.model.lambda_bounds, self.model.kappa_bounds
        lam_mean, lam_median = multiplier_summary(lam, *lam_b)
        kap_mean, kap_median = multiplier_summary(kap, *kap_b, skip_first=True)
        lam_lo, lam_hi = precision_bound_hits(lam, *lam_b)
        kap_lo, kap_hi = precision_bound_hits(kap, *kap_b, skip_first=True)
        return {"observation_lambda_mean": lam_mean, "observation_lambda_median": lam_median,
                "process_kappa_mean": kap_mean, "process_kappa_median": kap_median,
                "observation_lambda_lower_bound_hits": lam_lo, "observation_lambda_upper_bound_hits": lam_hi,
                "process_kappa_lower_bound_hits": kap_lo, "process_kappa_upper_bound_hits": kap_hi,
                "relative_sign_change_per_kb": sign_change_per_kb(rel, self.interval_size_bp)}

    def phase(self, batch, c):
        """after an ECM phase that no background update follows (the final phase; the single phase without a background fit)"""
        lam, kap, rel, _, _, _, key = self._fetch(batch, c, False)
        return self._run(c, key, self._summaries, lam, kap, rel)

    def loop_pass(self, batch, c, update_info):
        """after `background_update` (its per-chain record: `update_info`), BEFORE `background_apply`: the phase's summaries and
        the objective of the proposal (`_scoreBackgroundFitObjective`, core.py:4540-4606) with its pass-to-pass test"""
        lam, kap, rel, fit, cnt, buf, key = self._fetch(batch, c, True)
        g = batch.download(c, "background_next", out=buf["g"])
        return self._run(c, key, self._loop_summaries, c, lam, kap, rel, fit, cnt, g, float(update_info.get("weight_scale", 1.0)))

    def _loop_summaries(self, c, lam, kap, rel, fit, cnt, g, weight_scale) -> dict:
        cfg = self.cfg
        out = self._summaries(lam, kap, rel)
        g = np.asarray(g, np.float64)
        weighted = 0.5 * float(np.sum(fit, dtype=np.float64))
        count = float(max(1, int(np.sum(cnt, dtype=np.int64))))
        lam_first, lam_second = cfg.penalties
        d1, d2 = np.diff(g), np.diff(g, n=2)
        first = 0.5 * float(lam_first) * _sum_squares(d1) if g.size >= 2 else 0.0
        second = 0.5 * float(lam_second) * _sum_squares(d2) if g.size >= 3 else 0.0
        negative = 0.0
        mult = cfg.neg_multiplier
        if cfg.use_nonnegative and mult is not None and float(mult) > 0.0:
            # median of the positive weight track: the update's own (same float32 inverse variances, core.py:4562-4575 / 8287-8296)
            scale = weight_scale if (np.isfinite(weight_scale) and weight_scale > 0.0) else 1.0
            negative = 0.5 * float(float(mult) * scale) * float(np.sum(np.minimum(g, 0.0) ** 2, dtype=np.float64))
        objective = float(weighted + (first + second) + negative)
        cur, prev = objective / count, self.previous_per_cell.get(c, float("nan"))
        change = tol = float("nan")
        stable = False
        if np.isfinite(prev) and np.isfinite(cur):                                       # core.py:5172-5194
            change = abs(cur - prev)
            tol = float(cfg.outer_nll_rtol) * max(abs(cur), abs(prev), 1.0)
            stable = bool(change <= tol)
        self.previous_per_cell[c] = cur
        out.update({"background_objective": metadata_float(objective), "background_objective_per_cell": metadata_float(cur),
                    "background_objective_change_per_cell": metadata_float(change),
                    "background_objective_threshold_per_cell": metadata_float(tol),
                    "background_objective_stable": stable,
                    "background_weighted_residual_objective": metadata_float(weighted),
                    "background_fit_effective_observation_count": int(count)})
        return out

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None


def _metadata_value(value):
    """core.runConsenrich._processNoiseCalibrationMetadataValue + _diagnosticScalar (core.py:5921-5931, 2315-2327)."""
    if isinstance(value, np.ndarray):
        return value.astype(float).tolist()
    if isinstance(value, dict):
        return {str(k): _metadata_value(v) for k, v in value.items()}
    if isinstance(value, (list, tuple)):
        return [_metadata_value(v) for v in value]
    if isinstance(value, np.generic):
        value = value.item()
    if isinstance(value, bool):
        return bool(value)
    if isinstance(value, int):
        return int(value)
    if isinstance(value, float):
        return value if np.isfinite(value) else None
    return value


def input_only_summaries(plan: RunPlan) -> dict:
    """The parts of the run diagnostics that read nothing but the call's inputs: the support record of the process-noise
    calibration (core.py:3046-3100) and the summary of the observation-noise trace sum_j max(munc_j + pad, 1e-12) (replicates
    added in their order, bin range by bin range).  `run_plan` starts them on a thread beside the device fit."""
    pad_f = float(plan.cfg.pad)
    n = plan.data.shape[1]
    r_trace = np.empty(n)

    def trace_cols(k0, k1):
        acc = np.zeros(k1 - k0)
        for row in plan.munc:
            acc += np.maximum(np.asarray(row[k0:k1], np.float64) + pad_f, 1.0e-12)
        r_trace[k0:k1] = acc

    _map_cols(trace_cols, n)
    return {"observation_r_trace": track_summary(r_trace),
            "support": process_noise_calibration_support(plan.data, plan.munc, pad_f)}


def run_diagnostics(plan: RunPlan, fit: ChainFit, final: dict) -> dict:
    """The reference's `runDiagnostics` (core.py:5944-5999) from the arrays of the final pass: the keys its own tests read
    (test_core.py:4055-4110) and the per-pass record; host arithmetic on downloaded tracks, same formulas."""
    cfg, d = plan.cfg, plan.model.state_dim
    n = plan.data.shape[1]
    q0 = np.asarray(final["matrixQ0"], np.float64)
    lam = final.get("lambdaExp") if cfg.use_lambda else None
    kap = final.get("processPrecExp") if cfg.use_kappa else None
    # final forward gain of every replicate (core.py:7671-7731)
    p00 = np.maximum(np.asarray(final["stateCovarForward"], np.float64)[:, 0, 0], 0.0)
    prec = np.ones(n) if lam is None else np.clip(np.asarray(lam, np.float64), *plan.model.lambda_bounds)
    gain = {k: [] for k in ("mean", "median", "sd", "iqr", "count")}
    base_gain = p00 * prec
    pad_f = float(cfg.pad)

    def gain_row(row):      # one replicate: a sort-like pass over n values (the replicates are independent: a small thread pool)
        var = np.maximum(np.asarray(row, np.float64) + pad_f, 1.0e-12)
        g = base_gain / var
        ok = np.isfinite(g)
        if not ok.all():
            g = g[ok]
        if g.size == 0:
            return 0, float("nan"), float("nan"), float("nan"), float("nan")
        q25, q75 = np.quantile(g, [0.25, 0.75])
        return int(g.size), float(g.mean()), float(np.median(g)), float(np.std(g)), float(q75 - q25)

    if final.get("gainSummary") is not None:
        # the device evaluated it on the resident final pass (DeviceBatch.gain_summary: exact order statistics by radix select,
        # no (m, n) float64 rows on the host)
        gain = {k: list(v) for k, v in final["gainSummary"].items()}
    else:
        for cnt, mean, med, sd, iqr in _map_rows(gain_row, plan.munc):
            gain["count"].append(cnt); gain["mean"].append(mean); gain["median"].append(med); gain["sd"].append(sd); gain["iqr"].append(iqr)
    early = plan.ret.pop("_early", None)            # input-only summaries started beside the device fit (`run_plan`), if any
    early = early.result() if early is not None else input_only_summaries(plan)
    r_trace_summary, support = early["observation_r_trace"], early["support"]
    # effective process noise tracks (core.py:2420-2519): base / kappa, or the stored process noise without kappa
    base_l = np.full(n, q0[0, 0])
    base_t = np.full(n, q0[1, 1]) if d == 2 else np.zeros(n)
    eff_l, eff_t = base_l.copy(), base_t.copy()
    if kap is not None:
        k = np.maximum(np.clip(np.asarray(kap, np.float64), *plan.model.kappa_bounds), np.finfo(np.float64).tiny)
        eff_l, eff_t = base_l / k, base_t / k
    elif final.get("pNoiseForward") is not None and n > 1:
        pn = np.asarray(final["pNoiseForward"], np.float64)
        ok = np.all(np.isfinite(pn[: n - 1].reshape(n - 1, -1)), axis=1)
        eff_l[1:] = np.where(ok, pn[: n - 1, 0, 0], eff_l[1:])
        if d == 2:
            eff_t[1:] = np.where(ok, pn[: n - 1, 1, 1], eff_t[1:])
    lvl, trd, trace = track_summaries([eff_l, eff_t, eff_l + eff_t])
    use_apn = bool(plan.ret["use_apn"])
    policy = "adaptive_process_noise" if use_apn else ("student_t_kappa" if cfg.use_kappa else "base")
    disabled = bool(plan.requested_kappa and use_apn and not cfg.use_kappa)
    qd = {"policy": policy, "apn_enabled": use_apn, "process_precision_reweighting_requested": bool(plan.requested_kappa),
          "process_precision_reweighting_effective": bool(cfg.use_kappa),
          "process_precision_reweighting_disabled_by_apn": disabled,
          "baseQLevel": float(q0[0, 0]), "baseQTrend": float(q0[1, 1]) if d == 2 else 0.0,
          "preKappaQLevel": track_summary(base_l), "preKappaQTrend": track_summary(base_t),
          "effectiveQLevel": lvl, "effectiveQTrend": trd, "effectiveQTrace": trace, "processQScale": track_summary(np.ones(n))}
    for name, s in (("Level", lvl), ("Trend", trd), ("Trace", trace)):
        for stat in ("median", "min", "max"):
            qd[f"effectiveQ{name}{stat.capitalize()}"] = s[stat]
    nis = np.asarray(final["NIS"], np.float64)
    nis = nis[np.isfinite(nis)]
    lam_b, kap_b = plan.model.lambda_bounds, plan.model.kappa_bounds
    # the fit-level copies of the LAST phase's summaries (core.py:5473-5517 -> 5622-5627): taken from that phase's record when the
    # fit ran with `PassDiagnostics`, else evaluated here (the multipliers of the last ECM phase are the final ones)
    last = fit.loop_diagnostics[-1] if fit.loop_diagnostics else {}
    keys = ("observation_lambda_lower_bound_hits", "observation_lambda_upper_bound_hits", "process_kappa_lower_bound_hits",
            "process_kappa_upper_bound_hits", "relative_sign_change_per_kb")
    if all(k in last for k in keys):
        extras = {k: last[k] for k in keys}
    else:
        lam_lo, lam_hi = precision_bound_hits(lam, *lam_b)
        kap_lo, kap_hi = precision_bound_hits(kap, *kap_b, skip_first=True)
        extras = {"observation_lambda_lower_bound_hits": lam_lo, "observation_lambda_upper_bound_hits": lam_hi,
                  "process_kappa_lower_bound_hits": kap_lo, "process_kappa_upper_bound_hits": kap_hi,
                  "relative_sign_change_per_kb": relative_sign_change_per_kb(
                      fit.ecm_state_level, plan.data, plan.munc, interval_size_bp=plan.interval_size_bp,
                      background=final.get("background"), pad=float(cfg.pad))}
    post = fit.post_process_noise_fit(cfg, extras=extras)
    # process-noise calibration record (core.py:5686-5737, 5921-5942)
    q_full = np.zeros((2, 2), np.float64)
    q_full[:d, :d] = q0[:d, :d]
    common = dict(Q0=q_full if d == 2 else q_full[:1, :1], state_model=plan.state_model, minQ=float(plan.ret["min_q"]),
                  maxQ=float(cfg.max_q), support=support)
    if plan.q_given:
        calib = static_process_noise_calibration_diagnostics(policy=PROCESS_NOISE_CALIBRATION_FIXED, status="skipped",
                                                             reason="initial_process_q", warm_start_process_noise=1.0, **common)
    elif cfg.seed_q:
        calib = static_process_noise_calibration_diagnostics(policy=plan.q_policy, status="estimated",
                                                             reason="data_derived_q_estimate", warm_start_process_noise=1.0,
                                                             **common)
        calib.update(dict(fit.q_seed or {}))
        calib["validTransitionCount"] = int((fit.q_seed or {}).get("qSeedTransitionCount", 0))
    else:
        calib = static_process_noise_calibration_diagnostics(policy=plan.q_policy, status="skipped", reason=plan.q_policy,
                                                             warm_start_process_noise=0.0, **common)
    calib.update(support)
    calib["resolvedMinQ"] = float(plan.ret["min_q"])
    calib["resolvedMaxQ"] = float(plan.ret["max_q_apn"])
    calib["transitionCount"] = float(max(n - 1, 0))
    calib["processQScaleSummary"] = track_summary(np.ones(n, np.float32))
    calib = {k: _metadata_value(v) for k, v in calib.items() if not str(k).startswith("_")}
    return {"state_model": plan.state_model, "final_nll": metadata_float(fit.final_nll),
            "final_forward_nis": metadata_float(float(nis.mean()) if nis.size else float("nan")),
            "final_forward_gain_contig_summary": gain,
            "precision_reweighting_boundary_hits": summarize_precision_boundary_hits(
                observationPrecision=lam, observationPrecisionMin=lam_b[0], observationPrecisionMax=lam_b[1],
                processPrecision=kap if not use_apn else None, processPrecisionMin=kap_b[0], processPrecisionMax=kap_b[1]),
            "process_noise_calibration": calib,
            "post_process_noise_fit": post, "optimization_path_tracked": bool(plan.ret["track_path"]),
            "process_precision_reweighting_requested": bool(plan.requested_kappa),
            "process_precision_reweighting_effective": bool(cfg.use_kappa),
            "process_precision_reweighting_disabled_by_apn": disabled, "adaptive_process_noise_effective": use_apn,
            "process_q_policy": policy, "process_q_diagnostics": qd, "observation_r_trace": r_trace_summary}


def assemble_result(plan: RunPlan, fit: ChainFit, final: dict) -> tuple:
    """core.py:6001-6142: the return tuple variants.  `final` holds the final pass: stateSmoothed (n,2), stateCovarSmoothed
    (n,2,2) (level model already zero-padded), postFitResiduals (n,m), NIS (n,), intervalToBlockMap, background,
    stateCovarForward, pNoiseForward, lambdaExp / processPrecExp (or None), matrixQ0, outputTracks (when requested)."""
    xs = np.asarray(final["stateSmoothed"], np.float32)
    if plan.ret["bound_state"]:
        xs = xs.copy()
        np.clip(xs[:, 0], np.float32(plan.ret["lower"]), np.float32(plan.ret["upper"]), out=xs[:, 0])
    out = [xs, np.asarray(final["stateCovarSmoothed"], np.float32), np.asarray(final["postFitResiduals"], np.float32),
           np.asarray(final["NIS"], np.float32)]
    if plan.ret["scales"]:
        out.append(np.asarray(final["intervalToBlockMap"], np.int32))
    if plan.ret["background"]:
        out.append(np.asarray(final["background"], np.float32))
    cfg = plan.cfg
    use_apn = bool(plan.ret["use_apn"])
    diag = run_diagnostics(plan, fit, final) if (plan.ret["diagnostics"] or plan.ret["precision"]) else None
    if plan.ret["precision"]:
        out.append({
            "precision_track_diagnostics": True, "state_model": plan.state_model, "ECM_useAPN": use_apn,
            "process_precision_reweighting_requested": bool(plan.requested_kappa),
            "process_precision_reweighting_effective": bool(cfg.use_kappa),
            "process_precision_reweighting_disabled_by_apn": bool(plan.requested_kappa and use_apn and not cfg.use_kappa),
            "process_q_policy": diag["process_q_policy"], "process_q_diagnostics": diag["process_q_diagnostics"],
            "observationPrecisionMultiplierMin": float(plan.model.lambda_bounds[0]),
            "observationPrecisionMultiplierMax": float(plan.model.lambda_bounds[1]),
            "processPrecisionMultiplierMin": float(plan.model.kappa_bounds[0]),
            "processPrecisionMultiplierMax": float(plan.model.kappa_bounds[1]),
            "lambdaExp": None if final.get("lambdaExp") is None else np.asarray(final["lambdaExp"], np.float32),
            "processPrecExp": None if final.get("processPrecExp") is None else np.asarray(final["processPrecExp"], np.float32),
            "matrixQ0": np.asarray(final["matrixQ0"], np.float32),
            "outputTracks": {k: np.asarray(v, np.float32) for k, v in final["outputTracks"].items()}})
    if plan.ret["diagnostics"]:
        out.append(diag)
    return tuple(out)


def runConsenrich(
    matrixData: np.ndarray,
    matrixMunc: np.ndarray,
    deltaF: float,
    minQ: float,
    maxQ: float,
    *,
    stateInit: float,
    stateCovarInit: float,
    boundState: bool,
    stateLowerBound: float,
    stateUpperBound: float,
    blockLenIntervals: int,
    intervalSizeBP: Optional[int] = None,
    projectStateDuringFiltering: bool = False,
    pad: float = 1.0e-4,
    ECM_fixedBackgroundIters: int = 50,
    ECM_fixedBackgroundRtol: float = 1.0e-4,
    t_innerIters: int = FIT_DEFAULT_T_INNER_ITERS,
    ECM_robustTNu: float = 8.0,
    ECM_useObsPrecisionReweighting: bool = True,
    ECM_useProcessPrecisionReweighting: bool = True,
    ECM_useAPN: bool = False,
    ECM_zeroCenterBackground: bool = False,
    ECM_outerIters: int = 3,
    ECM_minOuterIters: Optional[int] = None,
    ECM_backgroundShiftRtol: float = 1.0e-3,
    ECM_outerNLLRtol: float = 1.0e-4,
    ECM_backgroundSmoothness: float = 1.0,
    fitBackground: bool = True,
    useNonnegativeBackground: bool = FIT_DEFAULT_USE_NONNEGATIVE_BACKGROUND,
    backgroundNegativePenaltyMultiplier: Optional[float] = (
        FIT_DEFAULT_BACKGROUND_NEGATIVE_PENALTY_MULTIPLIER
    ),
    returnScales: bool = True,
    returnBackground: bool = False,
    stateModel: Optional[str] = STATE_MODEL_LEVEL_TREND,
    processNoiseCalibration: str = PROCESS_DEFAULT_NOISE_CALIBRATION,
    qSeedPriorLevel: float = PROCESS_DEFAULT_Q_SEED_PRIOR_LEVEL,
    processNoiseWarmupECMIters: int = PROCESS_DEFAULT_WARMUP_ECM_ITERS,
    processNoiseWarmupOuterPasses: int = PROCESS_DEFAULT_WARMUP_OUTER_PASSES,
    observationPrecisionMultiplierMin: float = 0.25,
    observationPrecisionMultiplierMax: float = 4.0,
    processPrecisionMultiplierMin: float = PROCESS_DEFAULT_PRECISION_MULTIPLIER_MIN,
    processPrecisionMultiplierMax: float = PROCESS_DEFAULT_PRECISION_MULTIPLIER_MAX,
    observationMask: Optional[np.ndarray] = None,
    initialBackground: Optional[np.ndarray] = None,
    initialObservationPrecision: Optional[np.ndarray] = None,
    initialProcessPrecision: Optional[np.ndarray] = None,
    initialProcessQ: Optional[np.ndarray] = None,
    trackOptimizationPath: bool = False,
    returnPrecisionDiagnostics: bool = False,
    returnDiagnostics: bool = False,
    logIndentLevel: int = 0,
    logRunRole: Optional[str] = None,
):
    """`consenrich.core.runConsenrich` (core.py:3861) on the device-resident fit; see the module docstring."""
    kw = dict(locals())
    for k in ("matrixData", "matrixMunc", "deltaF", "minQ", "maxQ"):
        kw.pop(k)
    plan = resolve_call(matrixData, matrixMunc, deltaF, minQ, maxQ, **kw)
    fit, final = run_plan(plan, device=_DEVICE)
    return assemble_result(plan, fit, final)


def run_plan(plan: RunPlan, device: int = 0):
    """The device part: one chromosome as a one-chain batch through `run_consenrich_batch`; returns (ChainFit, final arrays)."""
    m, n = plan.data.shape
    cfg = plan.cfg
    early_pool = None
    if plan.ret["diagnostics"] or plan.ret["precision"]:
        from concurrent.futures import ThreadPoolExecutor

        early_pool = ThreadPoolExecutor(max_workers=1)
        plan.ret["_early"] = early_pool.submit(input_only_summaries, plan)
        early_pool.shutdown(wait=False)                 # (the submitted job still runs to its end)
    with _Context(device) as b:
        b.configure(plan.model, m, [n])
        b.upload(0, plan.data, plan.munc)
        if plan.initial_lambda is not None or plan.initial_kappa is not None:       # warm-started multipliers (core.py:4637-4648)
            b.upload_multipliers(0, plan.initial_lambda, plan.initial_kappa, None)
        passes = PassDiagnostics(cfg, plan.model, plan.interval_size_bp) if plan.ret["diagnostics"] else None
        try:
            fits, results = run_consenrich_batch(
                b, cfg, block_len_intervals=plan.block_len_intervals, model_q0=None if plan.q0 is None else _pad_q(plan.q0),
                initial_background=None if plan.initial_background is None else [plan.initial_background],
                return_background=True, return_precision_diagnostics=True, download=True,
                initial_lambda=plan.initial_lambda is not None, initial_kappa=plan.initial_kappa is not None,
                keep_ecm_state=False,
                pass_diagnostics=passes, track_path=bool(plan.ret["track_path"]))
        finally:
            if passes is not None:
                passes.close()
        fit, res = fits[0], results[0]
        final = {"stateSmoothed": res[0], "stateCovarSmoothed": res[1], "postFitResiduals": res[2], "NIS": res[3],
                 "intervalToBlockMap": res[4], "background": res[5], "outputTracks": res[6]["outputTracks"],
                 "lambdaExp": res[6]["lambdaExp"], "processPrecExp": res[6]["processPrecExp"],
                 "matrixQ0": res[6]["matrixQ0"], "stateCovarForward": b.download(0, "Pf"),
                 "pNoiseForward": b.download(0, "pnoise")}
        if plan.ret["diagnostics"] or plan.ret["precision"]:
            final["gainSummary"] = b.gain_summary(0, float(cfg.pad), use_lambda=bool(cfg.use_lambda),
                                                  lambda_bounds=plan.model.lambda_bounds)
    d = plan.model.state_dim
    final["matrixQ0"] = np.asarray(final["matrixQ0"], np.float32)[:d, :d] if d == 1 else np.asarray(final["matrixQ0"], np.float32)
    return fit, final


def _pad_q(q0):
    q = np.zeros((2, 2), np.float32)
    q0 = np.asarray(q0, np.float32)
    q[: q0.shape[0], : q0.shape[1]] = q0
    return q
