"""Initial process-noise (Q0) seed on the GPU.

Mirrors, with the reference's positional interfaces, validation messages, return shapes and diagnostics keys:
  ``cEstimateSameTrackProcessNoiseTransitions``  /root/reference/src/consenrich/cconsenrich.pyx:1441-1797
  ``cEstimatePooledProcessNoiseTransitions``     pyx:1800-1902
  ``cQSeedPosteriorFromTransitions``             pyx:1905-2146
and ``estimate_initial_process_noise`` = ``core._estimateInitialProcessNoiseFromData`` (core.py:3621-3780) on float32
matrices, which runs on the device-resident copy (``DeviceBatch.qseed`` does the same for every chain of a batch
without any upload).  The device gathers and reduces the sampled columns; the bounded tail (quantiles of the precision
sample, signal panel, 64-point grid posterior) is C++ host arithmetic inside libconsenrich_amd in the reference's
operation order.  Transitions are bit-identical to the reference.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib as L

U8P = C.POINTER(C.c_uint8)

# core.py:272-280, constants.py:149
QINIT_MIN_TRANSITIONS = 8
QINIT_MAX_TRANSITIONS = 32_000
QINIT_SIGNAL_PANEL_SIZE = 2048
QINIT_GRID_SIZE = 64
QINIT_PRECISION_SAMPLE_CAP = 32_000
QINIT_PRECISION_CAP_QUANTILE = 0.95
QINIT_PRECISION_CAP_MULTIPLIER = 20.0
QINIT_PRIOR_LOG_SD = math.log(4.0)
QINIT_DEFAULT_T_NU = 8.0
Q_SEED_PRIOR_LEVEL = 1.0e-5

SOURCES = ("sameTrackEB", "pooledEB", "observationVarianceFloor", "minQ")
REASONS = ("ok", "fallback_observation_variance", "fallback_min_q", "insufficient_transition_support")


def _call(rc):
    """data-dependent errors carry the reference's ValueError text"""
    if rc != 0:
        msg = L.last_error()
        if msg.startswith(("active ", "deltas ", "samplingVariances ", "transitionWeights ", "q seed ", "precisionSampleCap",
                           "`")):
            raise ValueError(msg)
        raise L.ConsenrichAMDError(msg or f"consenrich_amd call failed (rc={rc})")


def _sample_index(i, items, samples):
    return int(math.floor(((float(i) + 0.5) * float(items)) / float(samples)))


def _sample_diag(dg, precisionSampleCap, maxTransitionSamples, n):
    """pyx:1771-1797"""
    out = {"pairCount": int(dg.pair_count), "precisionCap": float(dg.precision_cap),
           "precisionCapFraction": float(dg.precision_cap_fraction),
           "candidateTransitionCount": int(dg.candidate_count), "selectedTransitionCount": int(dg.selected_count)}
    if dg.capped_mode:
        scan = int(dg.scan_count)
        out.update({"sampledPairCount": int(dg.sampled_pair_count), "precisionSamplePairCount": int(dg.precision_sample_count),
                    "sampledTransitionCount": scan, "transitionSampleFraction": float(dg.transition_sample_fraction),
                    "precisionSampleCap": int(precisionSampleCap), "maxTransitionSamples": int(maxTransitionSamples),
                    "sampledTransitionIndices": [_sample_index(i, n - 1, scan) for i in range(scan)] if scan <= 1024 else None})
    return out


def cEstimateSameTrackProcessNoiseTransitions(matrixData, obsVar, activeObservation, precisionCapQuantile,
                                              precisionCapMultiplier, maxTransitionSamples=0, precisionSampleCap=32000,
                                              signalPanelSize=0):
    if signalPanelSize < 0:                                                                     # pyx:1532-1546
        raise ValueError("signalPanelSize must be nonnegative")
    if (not math.isfinite(precisionCapQuantile)) or precisionCapQuantile < 0.0 or precisionCapQuantile > 1.0:
        raise ValueError("precisionCapQuantile must be in [0, 1]")
    if (not math.isfinite(precisionCapMultiplier)) or precisionCapMultiplier <= 0.0:
        raise ValueError("precisionCapMultiplier must be positive")
    data = np.ascontiguousarray(matrixData, np.float64)
    obs = np.ascontiguousarray(obsVar, np.float64)
    act = np.ascontiguousarray(activeObservation, np.uint8)
    if data.ndim != 2:
        raise ValueError("matrixData must be a 2D array")
    if obs.shape != data.shape:
        raise ValueError("obsVar shape must match matrixData")
    if act.shape != data.shape:
        raise ValueError("activeObservation shape must match matrixData")
    m, n = data.shape
    if n < 2 or m <= 0:
        e = np.empty(0)
        return e, e.copy(), e.copy(), {"pairCount": 0, "precisionCap": float("nan"), "precisionCapFraction": 0.0,
                                       "candidateTransitionCount": 0, "selectedTransitionCount": 0}
    L.require_gpu()
    cap = n - 1 if not (0 < maxTransitionSamples < n - 1) else int(maxTransitionSamples)
    d, s, w = np.empty(cap), np.empty(cap), np.empty(cap)
    cfg = L.QseedSampleCfg(float(precisionCapQuantile), float(precisionCapMultiplier), int(maxTransitionSamples),
                           int(precisionSampleCap), int(signalPanelSize))
    dg, cnt = L.QseedSampleDiag(), C.c_int64()
    _call(L.lib().csr_qseed_same_track(m, n, L.dp(data), L.dp(obs), act.ctypes.data_as(U8P), C.byref(cfg), L.dp(d), L.dp(s),
                                       L.dp(w), C.byref(cnt), C.byref(dg)))
    k = int(cnt.value)
    return d[:k], s[:k], w[:k], _sample_diag(dg, precisionSampleCap, maxTransitionSamples, n)


def cEstimatePooledProcessNoiseTransitions(matrixData, obsVar, activeObservation):
    data = np.ascontiguousarray(matrixData, np.float64)
    obs = np.ascontiguousarray(obsVar, np.float64)
    act = np.ascontiguousarray(activeObservation, np.uint8)
    if data.ndim != 2:
        raise ValueError("matrixData must be a 2D array")
    if obs.shape != data.shape:
        raise ValueError("obsVar shape must match matrixData")
    if act.shape != data.shape:
        raise ValueError("activeObservation shape must match matrixData")
    m, n = data.shape
    if n < 2 or m <= 0:
        e = np.empty(0)
        return e, e.copy(), e.copy()
    L.require_gpu()
    d, s, w = np.empty(n - 1), np.empty(n - 1), np.empty(n - 1)
    cnt = C.c_int64()
    _call(L.lib().csr_qseed_pooled(m, n, L.dp(data), L.dp(obs), act.ctypes.data_as(U8P), L.dp(d), L.dp(s), L.dp(w),
                                   C.byref(cnt)))
    k = int(cnt.value)
    return d[:k], s[:k], w[:k]


def _post_dict(p, source):
    """pyx:2013-2019, 2123-2146"""
    if not p.ok:
        return {"ok": False, "source": str(source), "reason": "insufficient_transition_support",
                "transitionCount": int(p.transition_count), "effectiveTransitionCount": float(p.effective_transition_count)}
    return {"ok": True, "source": str(source), "reason": "ok", "transitionCount": int(p.transition_count),
            "effectiveTransitionCount": float(p.effective_transition_count),
            "medianSamplingVariance": float(p.median_sampling_variance), "priorLevel": float(p.prior_level),
            "posteriorModeLevel": float(p.posterior_mode), "posteriorMedianLevel": float(p.posterior_median),
            "posteriorQ05Level": float(p.posterior_q05), "posteriorQ95Level": float(p.posterior_q95),
            "transitionQ90": float(p.transition_q90)}


def cQSeedPosteriorFromTransitions(deltas, samplingVariances, transitionWeights, qFloor, qCap, robustTNu, source,
                                   qSeedPriorLevel, minTransitions, priorLogSd, defaultTNu, gridSize):
    d = np.ascontiguousarray(deltas, np.float64).reshape(-1)
    s = np.ascontiguousarray(samplingVariances, np.float64).reshape(-1)
    w = np.ascontiguousarray(transitionWeights, np.float64).reshape(-1)
    if d.shape[0] != s.shape[0] or d.shape[0] != w.shape[0]:                                   # pyx:1977-1996
        raise ValueError("transition arrays must have the same length")
    if (not math.isfinite(qFloor)) or qFloor <= 0.0:
        raise ValueError("qFloor must be positive finite")
    if math.isfinite(qCap) and qCap <= 0.0:
        raise ValueError("qCap must be positive or infinite")
    if (not math.isfinite(qSeedPriorLevel)) or qSeedPriorLevel <= 0.0:
        raise ValueError("qSeedPriorLevel must be positive finite")
    if math.isfinite(qCap) and qSeedPriorLevel > qCap:
        raise ValueError("`qSeedPriorLevel` must not exceed `maxQ`")
    if minTransitions <= 0:
        raise ValueError("minTransitions must be positive")
    if (not math.isfinite(priorLogSd)) or priorLogSd <= 0.0:
        raise ValueError("priorLogSd must be positive finite")
    if (not math.isfinite(defaultTNu)) or defaultTNu <= 0.0:
        raise ValueError("defaultTNu must be positive finite")
    if gridSize <= 0:
        raise ValueError("gridSize must be positive")
    cfg = L.QseedPostCfg(float(qFloor), float(qCap), float(robustTNu), float(qSeedPriorLevel), int(minTransitions),
                         float(priorLogSd), float(defaultTNu), int(gridSize))
    p = L.QseedPost()
    _call(L.lib().csr_qseed_posterior(d.shape[0], L.dp(d), L.dp(s), L.dp(w), C.byref(cfg), C.byref(p)))
    return _post_dict(p, source)


# ---- the caller (core.py:3621-3780) -------------------------------------------------------------------------------
def seed_config(*, pad, stateModel, minQ, maxQ, deltaF, robustTNu, qSeedPriorLevel=Q_SEED_PRIOR_LEVEL) -> L.QseedCfg:
    cfg = L.QseedCfg()
    cfg.sample = L.QseedSampleCfg(QINIT_PRECISION_CAP_QUANTILE, QINIT_PRECISION_CAP_MULTIPLIER, QINIT_MAX_TRANSITIONS,
                                  QINIT_PRECISION_SAMPLE_CAP, QINIT_SIGNAL_PANEL_SIZE)
    cfg.pad, cfg.min_q, cfg.max_q, cfg.delta_f = float(pad), float(minQ), float(maxQ), float(deltaF)
    cfg.robust_t_nu = float("nan") if robustTNu is None else float(robustTNu)
    cfg.q_seed_prior_level = float(qSeedPriorLevel)
    cfg.min_transitions, cfg.prior_log_sd, cfg.default_t_nu = QINIT_MIN_TRANSITIONS, QINIT_PRIOR_LOG_SD, QINIT_DEFAULT_T_NU
    cfg.grid_size = QINIT_GRID_SIZE
    cfg.state_dim = 2 if stateModel == "levelTrend" else 1
    return cfg


def seed_result(o: L.QseedOut, min_q: float):
    """(matrixQ float32 (2,2), diagnostics dict) of core.py:3738-3780 from one csr_qseed_out record"""
    Q = np.zeros((2, 2), np.float32)
    Q[0, 0], Q[1, 1] = np.float32(max(o.q_level, min_q)), np.float32(max(o.q_trend, min_q))      # core.py:3820-3831
    p, nan = o.post, float("nan")
    ok = bool(p.ok)
    q_before = float(o.level_pre_clamp)
    changed = bool(abs(o.q_level / max(q_before, min_q) - 1.0) > 1.0e-6) if (math.isfinite(q_before) and q_before > 0.0) else False
    diag = {
        "qSeedSource": SOURCES[o.source], "qSeedReason": REASONS[o.reason],
        "qSeedTransitionCount": int(p.transition_count),
        "qSeedEffectiveTransitionCount": float(p.effective_transition_count),
        "qSeedPairCount": int(o.sample.pair_count),
        "qSeedCandidateTransitionCount": int(o.sample.candidate_count),
        "qSeedSelectedTransitionCount": int(o.sample.selected_count),
        "qSeedPrecisionCapFraction": float(o.sample.precision_cap_fraction),
        "qSeedPriorLevel": float(p.prior_level) if ok else nan,
        "qSeedPosteriorMedianLevel": float(p.posterior_median) if ok else nan,
        "qSeedPosteriorModeLevel": float(p.posterior_mode) if ok else nan,
        "qSeedPosteriorQ05Level": float(p.posterior_q05) if ok else nan,
        "qSeedPosteriorQ95Level": float(p.posterior_q95) if ok else nan,
        "qSeedTransitionQ90": float(p.transition_q90) if ok else nan,
        "qSeedGuardrailApplied": False,
        "qSeedLevelPreClamp": q_before, "qSeedTrendPreClamp": float(o.trend_pre_clamp),
        "qSeedLevelFinal": float(o.q_level), "qSeedTrendFinal": float(o.q_trend),
        "qSeedClampChanged": changed,
        "qSeedTrendLevelRatio": float(o.q_trend / max(o.q_level, min_q)),
        "qSeedMedianSamplingVariance": float(p.median_sampling_variance) if ok else nan,
    }
    return Q, diag


def estimate_initial_process_noise(*, matrixData, matrixMunc, pad, stateModel, minQ, maxQ, deltaF, robustTNu,
                                   qSeedPriorLevel=Q_SEED_PRIOR_LEVEL, device: int = 0):
    """``core._estimateInitialProcessNoiseFromData`` for one (m, n) float32 matrix pair: uploads them once and runs
    ``DeviceBatch.qseed``.  Returns (matrixQ, diagnostics) like the reference."""
    from .batch import DeviceBatch, ModelParams

    data = np.ascontiguousarray(matrixData, np.float32)
    munc = np.ascontiguousarray(matrixMunc, np.float32)
    if data.shape != munc.shape:
        raise ValueError("matrixData and matrixMunc must have matching shapes")
    if data.ndim != 2 or data.shape[0] < 1 or data.shape[1] < 1:
        raise ValueError("matrixData must be a non-empty 2D array")
    with DeviceBatch(device=device) as b:
        b.configure(ModelParams(pad=pad), data.shape[0], [data.shape[1]])
        b.upload(0, data, munc)
        return b.qseed(pad=pad, stateModel=stateModel, minQ=minQ, maxQ=maxQ, deltaF=deltaF, robustTNu=robustTNu,
                       qSeedPriorLevel=qSeedPriorLevel)[0]
