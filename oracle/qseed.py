"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the initial process-noise (Q0) seed.

* the three natives (cconsenrich.pyx:1441-1797, 1800-1902, 1905-2146) are restated in C (oracle/qseed_oracle.c); the
  wrappers below give them the reference's Python signatures, return shapes, diagnostics keys and error texts;
* `estimate_initial_process_noise` restates the pure-Python caller `_estimateInitialProcessNoiseFromData`
  (core.py:3621-3780, with `_activeProcessNoiseObservationMask` core.py:2989-3004, `_clampProcessNoise`
  core.py:3514-3522, `constructMatrixQ` core.py:3813-3858).  It takes the module providing the natives as its first
  argument, so tests/golden/make_golden.py runs the same composition on the compiled reference's natives to produce
  the golden vectors (`consenrich.core` itself cannot be imported here).
"""
from __future__ import annotations

import ctypes as C
import math
import sys

import numpy as np

from . import oracle as _orc

_DP = C.POINTER(C.c_double)
_U8 = C.POINTER(C.c_uint8)

# core.py:272-280
QINIT_MIN_TRANSITIONS = 8
QINIT_MAX_TRANSITIONS = 32_000
QINIT_SIGNAL_PANEL_SIZE = 2048
QINIT_GRID_SIZE = 64
QINIT_PRECISION_SAMPLE_CAP = 32_000
QINIT_PRECISION_CAP_QUANTILE = 0.95
QINIT_PRECISION_CAP_MULTIPLIER = 20.0
QINIT_PRIOR_LOG_SD = math.log(4.0)
QINIT_DEFAULT_T_NU = 8.0
MASKED_OBSERVATION_VARIANCE = float(np.float32(1.0e30))      # constants.py:387
Q_SEED_PRIOR_LEVEL = 1.0e-5                                  # constants.py:149

ERRORS = {
    1: "active matrixData values must be finite",
    2: "active obsVar values must be positive finite",
    3: "active transition values must be finite",
    4: "active transition precision must be positive finite",
    5: "precisionSampleCap must be positive",
    6: "active pooled observations must be finite with positive variance",
    7: "deltas must be finite",
    8: "samplingVariances must be nonnegative finite",
    9: "transitionWeights must be positive finite",
    10: "q seed posterior produced a nonfinite score",
    11: "q seed posterior normalization failed",
}


class _Diag(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("pairCount", "sampledPairCount", "precisionSampleCount", "scanCount",
                                         "candidateTransitionCount", "selectedTransitionCount")] + \
               [("cappedMode", C.c_int32), ("pad_", C.c_int32)] + \
               [(k, C.c_double) for k in ("precisionCap", "precisionCapFraction", "transitionSampleFraction")]


class _Post(C.Structure):
    _fields_ = [("transitionCount", C.c_int64), ("ok", C.c_int32), ("pad_", C.c_int32)] + \
               [(k, C.c_double) for k in ("effectiveTransitionCount", "medianSamplingVariance", "priorLevel",
                                          "posteriorModeLevel", "posteriorMedianLevel", "posteriorQ05Level",
                                          "posteriorQ95Level", "transitionQ90")]


def check_same_track_args(dataArr, obsArr, activeArr, precisionCapQuantile, precisionCapMultiplier, signalPanelSize):
    """argument validation of pyx:1532-1546 (shared with the product mirror's tests)"""
    if signalPanelSize < 0:
        raise ValueError("signalPanelSize must be nonnegative")
    if (not math.isfinite(precisionCapQuantile)) or precisionCapQuantile < 0.0 or precisionCapQuantile > 1.0:
        raise ValueError("precisionCapQuantile must be in [0, 1]")
    if (not math.isfinite(precisionCapMultiplier)) or precisionCapMultiplier <= 0.0:
        raise ValueError("precisionCapMultiplier must be positive")
    if dataArr.ndim != 2:
        raise ValueError("matrixData must be a 2D array")
    if obsArr.shape != dataArr.shape:
        raise ValueError("obsVar shape must match matrixData")
    if activeArr.shape != dataArr.shape:
        raise ValueError("activeObservation shape must match matrixData")


def sample_index(i, item_count, sample_count):
    """pyx:1431-1438"""
    return int(math.floor(((float(i) + 0.5) * float(item_count)) / float(sample_count)))


def same_track_diagnostics(d, precisionSampleCap, maxTransitionSamples, n):
    """the diagnostics dict of pyx:1771-1797 from the C struct's fields"""
    out = {"pairCount": int(d["pairCount"]), "precisionCap": float(d["precisionCap"]),
           "precisionCapFraction": float(d["precisionCapFraction"]),
           "candidateTransitionCount": int(d["candidateTransitionCount"]),
           "selectedTransitionCount": int(d["selectedTransitionCount"])}
    if d["cappedMode"]:
        scan = int(d["scanCount"])
        out.update({"sampledPairCount": int(d["sampledPairCount"]),
                    "precisionSamplePairCount": int(d["precisionSampleCount"]),
                    "sampledTransitionCount": scan,
                    "transitionSampleFraction": float(d["transitionSampleFraction"]),
                    "precisionSampleCap": int(precisionSampleCap), "maxTransitionSamples": int(maxTransitionSamples),
                    "sampledTransitionIndices": [sample_index(i, n - 1, scan) for i in range(scan)] if scan <= 1024 else None})
    return out


def cEstimateSameTrackProcessNoiseTransitions(matrixData, obsVar, activeObservation, precisionCapQuantile,
                                              precisionCapMultiplier, maxTransitionSamples=0, precisionSampleCap=32000,
                                              signalPanelSize=0):
    data = np.ascontiguousarray(matrixData, np.float64)
    obs = np.ascontiguousarray(obsVar, np.float64)
    act = np.ascontiguousarray(activeObservation, np.uint8)
    check_same_track_args(data, obs, act, precisionCapQuantile, precisionCapMultiplier, signalPanelSize)
    m, n = data.shape
    if n < 2 or m <= 0:
        e = np.empty(0)
        return e, e.copy(), e.copy(), {"pairCount": 0, "precisionCap": float("nan"), "precisionCapFraction": 0.0,
                                       "candidateTransitionCount": 0, "selectedTransitionCount": 0}
    cap = n - 1 if not (0 < maxTransitionSamples < n - 1) else int(maxTransitionSamples)
    d, s, w = np.empty(cap), np.empty(cap), np.empty(cap)
    dg = _Diag()
    f = _orc.lib().cor_qseed_same_track
    f.restype = C.c_int64
    f.argtypes = [C.c_int64, C.c_int64, _DP, _DP, _U8, C.c_double, C.c_double, C.c_int64, C.c_int64, C.c_int64,
                  _DP, _DP, _DP, C.POINTER(_Diag)]
    cnt = f(m, n, data.ctypes.data_as(_DP), obs.ctypes.data_as(_DP), act.ctypes.data_as(_U8), float(precisionCapQuantile),
            float(precisionCapMultiplier), int(maxTransitionSamples), int(precisionSampleCap), int(signalPanelSize),
            d.ctypes.data_as(_DP), s.ctypes.data_as(_DP), w.ctypes.data_as(_DP), C.byref(dg))
    if cnt < 0:
        raise ValueError(ERRORS[-cnt])
    diag = same_track_diagnostics({k: getattr(dg, k) for k, _ in _Diag._fields_}, precisionSampleCap, maxTransitionSamples, n)
    return d[:cnt], s[:cnt], w[:cnt], diag


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
    d, s, w = np.empty(n - 1), np.empty(n - 1), np.empty(n - 1)
    f = _orc.lib().cor_qseed_pooled
    f.restype = C.c_int64
    f.argtypes = [C.c_int64, C.c_int64, _DP, _DP, _U8, _DP, _DP, _DP]
    cnt = f(m, n, data.ctypes.data_as(_DP), obs.ctypes.data_as(_DP), act.ctypes.data_as(_U8), d.ctypes.data_as(_DP),
            s.ctypes.data_as(_DP), w.ctypes.data_as(_DP))
    if cnt < 0:
        raise ValueError(ERRORS[-cnt])
    return d[:cnt], s[:cnt], w[:cnt]


def check_posterior_args(n_d, n_s, n_w, qFloor, qCap, qSeedPriorLevel, minTransitions, priorLogSd, defaultTNu, gridSize):
    """argument validation of pyx:1977-1996 (shared with the product mirror's tests)"""
    if n_d != n_s or n_d != n_w:
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


def posterior_dict(p, source):
    """the result dict of pyx:2013-2019 / 2123-2146 from the C struct's fields"""
    if not p["ok"]:
        return {"ok": False, "source": str(source), "reason": "insufficient_transition_support",
                "transitionCount": int(p["transitionCount"]),
                "effectiveTransitionCount": float(p["effectiveTransitionCount"])}
    out = {"ok": True, "source": str(source), "reason": "ok", "transitionCount": int(p["transitionCount"])}
    for k in ("effectiveTransitionCount", "medianSamplingVariance", "priorLevel", "posteriorModeLevel",
              "posteriorMedianLevel", "posteriorQ05Level", "posteriorQ95Level", "transitionQ90"):
        out[k] = float(p[k])
    return out


def cQSeedPosteriorFromTransitions(deltas, samplingVariances, transitionWeights, qFloor, qCap, robustTNu, source,
                                   qSeedPriorLevel, minTransitions, priorLogSd, defaultTNu, gridSize):
    d = np.ascontiguousarray(deltas, np.float64).reshape(-1)
    s = np.ascontiguousarray(samplingVariances, np.float64).reshape(-1)
    w = np.ascontiguousarray(transitionWeights, np.float64).reshape(-1)
    check_posterior_args(d.shape[0], s.shape[0], w.shape[0], qFloor, qCap, qSeedPriorLevel, minTransitions, priorLogSd,
                         defaultTNu, gridSize)
    p = _Post()
    f = _orc.lib().cor_qseed_posterior
    f.restype = C.c_int
    f.argtypes = [C.c_int64, _DP, _DP, _DP] + [C.c_double] * 4 + [C.c_int64, C.c_double, C.c_double, C.c_int64,
                                                                   C.POINTER(_Post)]
    rc = f(d.shape[0], d.ctypes.data_as(_DP), s.ctypes.data_as(_DP), w.ctypes.data_as(_DP), float(qFloor), float(qCap),
           float(robustTNu), float(qSeedPriorLevel), int(minTransitions), float(priorLogSd), float(defaultTNu),
           int(gridSize), C.byref(p))
    if rc < 0:
        raise ValueError(ERRORS[-rc])
    return posterior_dict({k: getattr(p, k) for k, _ in _Post._fields_}, source)


NATIVES = sys.modules[__name__]      # this module provides the three natives under the reference's names


# ---- the pure-Python caller (core.py:3621-3780) ------------------------------------------------------------------
def _check_finite_positive(name, value):
    """core.py:2202-2206"""
    v = float(value)
    if (not math.isfinite(v)) or v <= 0.0:
        raise ValueError(f"`{name}` must be positive and finite")
    return v


def _clamp(value, q_floor, q_cap):
    """core.py:3514-3522"""
    v = float(value)
    if not math.isfinite(v):
        v = q_floor
    v = max(v, q_floor)
    if math.isfinite(q_cap):
        v = min(v, float(q_cap))
    return float(v)


def construct_matrix_q(min_diag, q00, q11):
    """core.py:3813-3858 for the diagonal call made by the seed (Q01 = Q10 = 0)"""
    Q = np.zeros((2, 2), np.float32)
    for i, v in ((0, q00), (1, q11)):
        v = float(v)
        Q[i, i] = np.float32(max(v, min_diag) if math.isfinite(v) else min_diag)
    return Q


def estimate_initial_process_noise(natives, *, matrixData, matrixMunc, pad, stateModel, minQ, maxQ, deltaF, robustTNu,
                                   qSeedPriorLevel=Q_SEED_PRIOR_LEVEL):
    """core.py:3621-3780; `stateModel` is "level" or "levelTrend".  Returns (matrixQ float32 (2,2), diagnostics)."""
    q_floor = _check_finite_positive("minQ", minQ)
    mx = float(maxQ)                                                           # core.py:3500-3511
    q_cap = float("inf") if (mx < 0.0 or not math.isfinite(mx)) else max(mx, q_floor)
    prior_floor = _check_finite_positive("minQ", qSeedPriorLevel)
    if math.isfinite(q_cap) and prior_floor > q_cap:
        raise ValueError("`qSeedPriorLevel` must not exceed `maxQ`")
    data = np.asarray(matrixData, np.float64)
    munc = np.asarray(matrixMunc, np.float64)
    if data.shape != munc.shape:
        raise ValueError("matrixData and matrixMunc must have matching shapes")
    obs_raw = munc + float(pad)
    with np.errstate(invalid="ignore"):                                       # core.py:2989-3004
        active = (np.isfinite(data) & np.isfinite(munc) & (munc < 0.5 * MASKED_OBSERVATION_VARIANCE)
                  & np.isfinite(obs_raw) & (obs_raw > 0.0))
    obs = np.maximum(obs_raw, 1.0e-12)
    nu = QINIT_DEFAULT_T_NU if robustTNu is None or not math.isfinite(float(robustTNu)) else float(robustTNu)
    d, s, w, same = natives.cEstimateSameTrackProcessNoiseTransitions(
        data, obs, active, float(QINIT_PRECISION_CAP_QUANTILE), float(QINIT_PRECISION_CAP_MULTIPLIER),
        int(QINIT_MAX_TRANSITIONS), int(QINIT_PRECISION_SAMPLE_CAP), int(QINIT_SIGNAL_PANEL_SIZE))
    post = (float(q_floor), float(q_cap), float(nu))
    tail = (float(prior_floor), int(QINIT_MIN_TRANSITIONS), float(QINIT_PRIOR_LOG_SD), float(QINIT_DEFAULT_T_NU),
            int(QINIT_GRID_SIZE))
    est = natives.cQSeedPosteriorFromTransitions(d, s, w, *post, "sameTrackEB", *tail)
    if not bool(est.get("ok", False)):
        pd_, ps_, pw_ = natives.cEstimatePooledProcessNoiseTransitions(data, obs, active)
        pooled = natives.cQSeedPosteriorFromTransitions(pd_, ps_, pw_, *post, "pooledEB", *tail)
        if bool(pooled.get("ok", False)):
            est = pooled
    source = str(est.get("source", "fallback"))
    reason = str(est.get("reason", "ok"))
    q_med = float(est.get("posteriorMedianLevel", float("nan")))
    q_before = q_med
    if not math.isfinite(q_before) or q_before <= 0.0:
        pool = obs[active]
        pool = pool[np.isfinite(pool) & (pool > 0.0)]
        fvar = float(np.median(pool)) if pool.size else float("nan")
        ok = math.isfinite(fvar) and fvar > 0.0
        q_before = 1.0e-4 * fvar if ok else q_floor
        source = "observationVarianceFloor" if math.isfinite(fvar) else "minQ"
        reason = "fallback_observation_variance" if math.isfinite(fvar) else "fallback_min_q"
    q_init = _clamp(q_before, q_floor, q_cap)
    if stateModel == "levelTrend":
        df = max(float(deltaF), 1.0e-12)
        q_trend_raw = q_init / (df * df)
        q_trend = _clamp(q_trend_raw, q_floor, q_cap)
    else:
        q_trend = q_init
        q_trend_raw = q_trend
    Q = construct_matrix_q(q_floor, q_init, q_trend)
    changed = bool(abs(q_init / max(q_before, q_floor) - 1.0) > 1.0e-6) if (math.isfinite(q_before) and q_before > 0.0) else False
    nan = float("nan")
    diag = {
        "qSeedSource": source, "qSeedReason": reason,
        "qSeedTransitionCount": int(est.get("transitionCount", 0)),
        "qSeedEffectiveTransitionCount": float(est.get("effectiveTransitionCount", 0.0)),
        "qSeedPairCount": int(same.get("pairCount", 0)),
        "qSeedCandidateTransitionCount": int(same.get("candidateTransitionCount", 0)),
        "qSeedSelectedTransitionCount": int(same.get("selectedTransitionCount", 0)),
        "qSeedPrecisionCapFraction": float(same.get("precisionCapFraction", 0.0)),
        "qSeedPriorLevel": float(est.get("priorLevel", nan)),
        "qSeedPosteriorMedianLevel": float(est.get("posteriorMedianLevel", nan)),
        "qSeedPosteriorModeLevel": float(est.get("posteriorModeLevel", nan)),
        "qSeedPosteriorQ05Level": float(est.get("posteriorQ05Level", nan)),
        "qSeedPosteriorQ95Level": float(est.get("posteriorQ95Level", nan)),
        "qSeedTransitionQ90": float(est.get("transitionQ90", nan)),
        "qSeedGuardrailApplied": False,
        "qSeedLevelPreClamp": float(q_before), "qSeedTrendPreClamp": float(q_trend_raw),
        "qSeedLevelFinal": float(q_init), "qSeedTrendFinal": float(q_trend),
        "qSeedClampChanged": changed,
        "qSeedTrendLevelRatio": float(q_trend / max(q_init, q_floor)),
        "qSeedMedianSamplingVariance": float(est.get("medianSamplingVariance", nan)),
    }
    return Q, diag
