"""TEST INFRASTRUCTURE ONLY -- CPU restatement (NumPy, float64) of the reference's per-interval output diagnostics,
SURVEY.md section 8(f) rank 2:

  * ``process_q_tracks``  follows /root/reference/src/consenrich/core.py:2420-2510 (`_processQTrackArrays`,
    returnFullQ=False);
  * ``output_diagnostic_tracks`` follows core.py:7734-7878 (`_perIntervalOutputDiagnosticTracks`), whose per-bin
    Python loop (core.py:7837-7865) is restated here in vectorised form: the loop carries no state (the "previous
    covariance" of bin k is the STORED float32 filtered covariance of bin k-1), so every bin is independent.

Pinning: `consenrich.core` cannot be imported in this image (third-party `itrigamma` / `structlog` are absent and
no stand-ins are written), so this restatement is pinned by the reference's own known-answer test
tests/test_core.py:2632-2698 (literal inputs and expected values, reproduced in tests/test_oracle_diagnostics.py) plus
an independent scalar re-derivation in that test file.  Never imported by ``consenrich_amd``.
"""
from __future__ import annotations

import numpy as np

_TINY = np.finfo(np.float64).tiny


def _clipped(mult, lo, hi, n, what):
    """optional multiplier track -> float64, clipped to [lo, hi] and floored at the smallest normal (core.py:7773-7783)"""
    if mult is None:
        return None
    v = np.asarray(mult, dtype=np.float64).reshape(-1)
    if v.shape != (n,):
        raise ValueError(f"{what} length must match interval count")
    return np.maximum(np.clip(v, float(lo), float(hi)), _TINY)


def process_q_tracks(matrixQ0, n, state_dim, processPrecExp=None, processQScale=None, pNoiseForward=None,
                     procPrecisionMultiplierMin=5e-3, procPrecisionMultiplierMax=5e3):
    """core.py:2420-2510.  Returns float64 tracks keyed like the reference's dict."""
    d = int(state_dim)
    q0 = np.asarray(matrixQ0, dtype=np.float64)[:d, :d]
    qs = np.ones(n)
    if processQScale is not None:
        qs = np.asarray(processQScale, dtype=np.float64).reshape(-1)
        if qs.shape != (n,):
            raise ValueError("processQScale length must match interval count")
        if not np.all(np.isfinite(qs)):
            raise ValueError("processQScale contains non-finite values")
        qs = np.maximum(qs, _TINY)
        if n:
            qs = qs.copy()
            qs[0] = 1.0                                          # core.py:2455-2457
    base_level = np.full(n, q0[0, 0])
    base_trend = np.full(n, q0[1, 1]) if d == 2 else np.zeros(n)
    pre_level, pre_trend = base_level * qs, base_trend * qs
    eff_level, eff_trend = pre_level.copy(), pre_trend.copy()
    if processPrecExp is not None:                               # core.py:2475-2492
        kap = np.asarray(processPrecExp, dtype=np.float64).reshape(-1)
        if kap.shape != (n,):
            raise ValueError("processPrecExp length must match interval count")
        if not np.all(np.isfinite(kap)):
            raise ValueError("processPrecExp contains non-finite values")
        kap = np.maximum(np.clip(kap, float(procPrecisionMultiplierMin), float(procPrecisionMultiplierMax)), _TINY)
        if n:
            eff_level, eff_trend = pre_level / kap, pre_trend / kap
    elif pNoiseForward is not None and n > 1:                    # core.py:2493-2510
        pn = np.asarray(pNoiseForward, dtype=np.float64)[: n - 1, :d, :d]
        ok = np.all(np.isfinite(pn.reshape(n - 1, -1)), axis=1)
        eff_level[1:] = np.where(ok, pn[:, 0, 0], eff_level[1:])
        if d == 2:
            eff_trend[1:] = np.where(ok, pn[:, 1, 1], eff_trend[1:])
    return {"baseQLevel": base_level, "baseQTrend": base_trend, "preKappaQLevel": pre_level,
            "preKappaQTrend": pre_trend, "effectiveQLevel": eff_level, "effectiveQTrend": eff_trend,
            "processQScale": qs}


def output_diagnostic_tracks(*, stateCovarForward, matrixMunc, matrixQ0, matrixF, stateCovarInit, state_dim,
                             lambdaExp=None, processPrecExp=None, processQScale=None, pNoiseForward=None, pad=1e-4,
                             obsPrecisionMultiplierMin=0.25, obsPrecisionMultiplierMax=4.0,
                             procPrecisionMultiplierMin=5e-3, procPrecisionMultiplierMax=5e3):
    """core.py:7734-7878.  Returns the reference's ten float32 tracks."""
    d = int(state_dim)
    covar = np.asarray(stateCovarForward, dtype=np.float64)
    munc = np.asarray(matrixMunc)
    n = int(covar.shape[0])
    if covar.ndim != 3 or covar.shape[1] < d or covar.shape[2] < d:
        raise ValueError("stateCovarForward shape does not match stateModel")
    if munc.ndim != 2 or int(munc.shape[1]) != n:
        raise ValueError("matrixMunc must have shape (trackCount, intervalCount)")
    F = np.asarray(matrixF, dtype=np.float64)
    q0 = np.asarray(matrixQ0, dtype=np.float64)[:d, :d]

    lam = _clipped(lambdaExp, obsPrecisionMultiplierMin, obsPrecisionMultiplierMax, n, "lambdaExp")
    if lam is None:
        lam = np.ones(n)
    # trace of the effective observation covariance and total observation precision per bin (core.py:7785-7799);
    # non-finite terms are skipped
    R = np.maximum(munc.astype(np.float64) + float(pad), 1.0e-12)
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        eff = R / lam[None, :]
        inv = lam[None, :] / R
    munc_trace = np.where(np.isfinite(eff), eff, 0.0).sum(axis=0)
    sum_inv_r = np.where(np.isfinite(inv), inv, 0.0).sum(axis=0)

    qt = process_q_tracks(matrixQ0, n, d, processPrecExp, processQScale, pNoiseForward,
                          procPrecisionMultiplierMin, procPrecisionMultiplierMax)
    qs = qt["processQScale"]
    kap = _clipped(processPrecExp, procPrecisionMultiplierMin, procPrecisionMultiplierMax, n, "processPrecExp")

    # effective process noise entering bin k (core.py:7838-7848)
    q_eff = q0[None, :, :] * qs[:, None, None]
    if kap is not None:
        q_eff = q_eff / kap[:, None, None]
    elif pNoiseForward is not None and n > 1:
        pn = np.asarray(pNoiseForward, dtype=np.float64)[: n - 1, :d, :d]
        ok = np.all(np.isfinite(pn.reshape(n - 1, -1)), axis=1)
        q_eff[1:] = np.where(ok[:, None, None], pn, q_eff[1:])
    # previous (stored) filtered covariance; the prior for the first bin (core.py:7818, 7866-7869)
    prev = np.empty((n, d, d))
    if n:
        prev[0] = np.eye(d) * float(stateCovarInit)
        prev[1:] = covar[:-1, :d, :d]
    if d == 2:
        pred = np.einsum("ij,kjl,ml->kim", F, prev, F) + q_eff   # F P F^T + Q  (core.py:7850)
        pred10 = pred[:, 1, 0]
    else:
        pred = prev + q_eff
        pred10 = np.zeros(n)
    pred00 = np.maximum(pred[:, 0, 0], 0.0)
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        denom = 1.0 + pred00 * sum_inv_r
        ok = np.isfinite(denom) & (denom > 0.0)
        scale = np.where(ok, sum_inv_r / np.where(ok, denom, 1.0), 0.0)
    out = {k: v.astype(np.float32) for k, v in qt.items()}
    out["muncTrace"] = munc_trace.astype(np.float32)
    out["sumGain0"] = np.where(ok, pred00 * scale, 0.0).astype(np.float32)
    out["sumGain1"] = np.where(ok, pred10 * scale, 0.0).astype(np.float32)
    return out
