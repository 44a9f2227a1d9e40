"""Host mirror of the reference's per-interval output diagnostics.

``perIntervalOutputDiagnosticTracks`` has the keyword interface, validation messages and return dict of
``consenrich.core._perIntervalOutputDiagnosticTracks`` (/root/reference/src/consenrich/core.py:7734-7878), whose body
is a per-bin Python loop with 2x2 NumPy products (core.py:7837-7865, ~10 s per chromosome).  Here the per-bin work
(trace of the effective observation covariance, total observation precision, predicted covariance from the stored
filtered covariance, total gains, effective process noise) runs in one HIP kernel through ``csr_output_diagnostics``;
the five tracks that are O(1) functions of Q0 and qScale are formed with NumPy as the reference does
(core.py:2420-2473).  Drop-in::

    import consenrich.core as core, consenrich_amd.diagnostics as amd
    core._perIntervalOutputDiagnosticTracks = amd.perIntervalOutputDiagnosticTracks

No CPU fallback: raises ConsenrichAMDError without the library or a GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

STATE_MODEL_LEVEL, STATE_MODEL_LEVEL_TREND = "level", "levelTrend"      # constants.py:134-135


def _state_dim(stateModel) -> int:
    mode = STATE_MODEL_LEVEL_TREND if stateModel is None else str(stateModel).strip()      # core.py:2186-2199
    if mode == STATE_MODEL_LEVEL_TREND:
        return 2
    if mode == STATE_MODEL_LEVEL:
        return 1
    raise ValueError("`stateModel` must be one of 'level', 'levelTrend'")


def _f32_vec(a, n, what, finite=False):
    if a is None:
        return None
    v = np.asarray(a, dtype=np.float64).reshape(-1)
    if v.shape != (n,):
        raise ValueError(f"{what} length must match interval count")
    if finite and not np.all(np.isfinite(v)):
        raise ValueError(f"{what} contains non-finite values")
    return np.ascontiguousarray(v, dtype=np.float32)


def perIntervalOutputDiagnosticTracks(*, stateCovarForward, matrixMunc, matrixQ0, matrixF, stateCovarInit, stateModel,
                                      lambdaExp, processPrecExp, processQScale, pNoiseForward, pad,
                                      obsPrecisionMultiplierMin, obsPrecisionMultiplierMax,
                                      procPrecisionMultiplierMin, procPrecisionMultiplierMax):
    d = _state_dim(stateModel)
    covar = np.asarray(stateCovarForward)
    munc = np.asarray(matrixMunc)
    q0 = np.asarray(matrixQ0, dtype=np.float64)
    f = np.asarray(matrixF, dtype=np.float64)
    if covar.ndim != 3 or covar.shape[1] < d or covar.shape[2] < d:
        raise ValueError("stateCovarForward shape does not match stateModel")
    n = int(covar.shape[0])
    if munc.ndim != 2 or int(munc.shape[1]) != n:
        raise ValueError("matrixMunc must have shape (trackCount, intervalCount)")
    if q0.ndim != 2 or q0.shape[0] < d or q0.shape[1] < d:
        raise ValueError("matrixQ0 shape does not match stateModel")
    if d == 2 and f.shape != (2, 2):
        raise ValueError("matrixF must have shape (2, 2) for level-trend tracks")
    lam = _f32_vec(lambdaExp, n, "lambdaExp")
    qs = _f32_vec(processQScale, n, "processQScale", finite=True)
    kap = _f32_vec(processPrecExp, n, "processPrecExp", finite=True)
    pn = None
    if pNoiseForward is not None and kap is None:
        pn = np.asarray(pNoiseForward)
        if pn.ndim != 3 or pn.shape[0] < max(n - 1, 0) or pn.shape[1] < d or pn.shape[2] < d:
            raise ValueError("pNoiseForward shape does not match stateModel")
        pn = np.ascontiguousarray(pn[: max(n - 1, 0), :d, :d], dtype=np.float32)

    # O(1)-per-bin tracks (core.py:2443-2473)
    tiny = np.finfo(np.float64).tiny
    qs64 = np.ones(n)
    if qs is not None:
        qs64 = np.maximum(np.asarray(processQScale, dtype=np.float64).reshape(-1), tiny)
        if n:
            qs64 = qs64.copy()
            qs64[0] = 1.0
    base_level = np.full(n, float(q0[0, 0]))
    base_trend = np.full(n, float(q0[1, 1])) if d == 2 else np.zeros(n)
    out = {
        "baseQLevel": base_level.astype(np.float32), "baseQTrend": base_trend.astype(np.float32),
        "preKappaQLevel": (base_level * qs64).astype(np.float32),
        "preKappaQTrend": (base_trend * qs64).astype(np.float32),
        "processQScale": qs64.astype(np.float32),
    }
    names = ("sumGain0", "sumGain1", "effectiveQLevel", "effectiveQTrend", "muncTrace")
    if n == 0:
        out.update({k: np.zeros(0, np.float32) for k in names})
        return out

    L.require_gpu()
    lib = L.lib()
    mdl = L.Model()
    mdl.state_dim = d
    mdl.F[:] = [float(f[0, 0]), float(f[0, 1]), float(f[1, 0]), float(f[1, 1])] if d == 2 else [1.0, 0.0, 0.0, 1.0]
    mdl.Q0[:] = [float(q0[0, 0]), float(q0[0, 1]), float(q0[1, 0]), float(q0[1, 1])] if d == 2 else \
        [float(q0[0, 0]), 0.0, 0.0, 0.0]
    mdl.state_init, mdl.state_covar_init, mdl.pad = 0.0, float(stateCovarInit), float(pad)
    mdl.w_min, mdl.w_max = float(obsPrecisionMultiplierMin), float(obsPrecisionMultiplierMax)
    mdl.k_min, mdl.k_max = float(procPrecisionMultiplierMin), float(procPrecisionMultiplierMax)
    pf = np.ascontiguousarray(covar[:, :d, :d], dtype=np.float32)
    mu = np.ascontiguousarray(munc, dtype=np.float32)
    res = {k: np.empty(n, np.float32) for k in names}
    if qs is not None:
        qs = qs.copy()
        qs[0] = 1.0
    L.check(lib.csr_output_diagnostics(C.byref(mdl), int(mu.shape[0]), n, L.fp(pf), L.fp(mu), L.fp(lam), L.fp(kap),
                                       L.fp(qs), L.fp(pn), *(L.fp(res[k]) for k in names)))
    out.update(res)
    return out
