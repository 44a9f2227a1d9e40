"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of the CPU oracle (oracle/consenrich_oracle.c).

Mirrors the keyword interface of the reference's ``consenrich.cconsenrich`` hot-path callables
(/root/reference/src/consenrich/cconsenrich.pyx:6393, 6635, 6853, 7052, 7153, 7660, 710, 818) so parity tests can
call oracle / reference / HIP product with the same kwargs.  Never imported by ``consenrich_amd``.
Parity is pinned against the compiled reference (oracle/_ref) and tests/golden (see consenrich_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("ORACLE_LIB") or os.path.join(_HERE, "libconsenrich_oracle.so")   # ORACLE_LIB: sanitizer build


class _Model(C.Structure):
    _fields_ = [
        ("state_dim", C.c_int32),
        ("F", C.c_double * 4),
        ("Q0", C.c_double * 4),
        ("state_init", C.c_double),
        ("state_covar_init", C.c_double),
        ("pad", C.c_double),
        ("w_min", C.c_double),
        ("w_max", C.c_double),
        ("k_min", C.c_double),
        ("k_max", C.c_double),
        ("apn_min_q", C.c_double),
        ("apn_max_q", C.c_double),
        ("apn_thresh", C.c_double),
        ("apn_scale", C.c_double),
        ("apn_pc", C.c_double),
    ]


_FP = C.POINTER(C.c_float)
_IP = C.POINTER(C.c_int32)
_DP = C.POINTER(C.c_double)


class _FwdIO(C.Structure):
    _fields_ = [
        ("m", C.c_int64),
        ("n", C.c_int64),
        ("data", _FP),
        ("munc", _FP),
        ("block_map", _IP),
        ("block_count", C.c_int64),
        ("lam", _FP),
        ("kappa", _FP),
        ("qscale", _FP),
        ("use_apn", C.c_int32),
        ("return_nll", C.c_int32),
        ("store_nll_in_d", C.c_int32),
        ("D", _FP),
        ("xf", _FP),
        ("Pf", _FP),
        ("pnoise", _FP),
    ]


class _FwdOut(C.Structure):
    _fields_ = [("sum_d", C.c_double), ("sum_nll", C.c_double), ("invalid_block_index", C.c_int64)]


class _EcmCfg(C.Structure):
    _fields_ = [
        ("max_iters", C.c_int64),
        ("inner_iters", C.c_int64),
        ("rtol", C.c_double),
        ("nu", C.c_double),
        ("use_lambda", C.c_int32),
        ("use_kappa", C.c_int32),
        ("use_apn", C.c_int32),
    ]


class _EcmOut(C.Structure):
    _fields_ = [
        ("iters_done", C.c_int64),
        ("final_nll", C.c_double),
        ("initial_nll", C.c_double),
        ("abs_rel_change", C.c_double),
        ("rel_improvement", C.c_double),
        ("stable_iters", C.c_int64),
        ("nll_increase_count", C.c_int64),
        ("converged", C.c_int32),
        ("skipped", C.c_int32),
        ("has_initial_nll", C.c_int32),
        ("invalid_block_index", C.c_int64),
    ]


_lib = None


def build() -> str:
    """Compile the C restatement (gcc, seconds). Building the checker is not using it."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.cor_forward.argtypes = [C.POINTER(_Model), C.POINTER(_FwdIO), C.POINTER(_FwdOut)]
        _lib.cor_forward.restype = None
        _lib.cor_backward.argtypes = [C.POINTER(_Model), C.c_int64, C.c_int64, _FP, _FP, _FP, _FP,
                                      _FP, _FP, _FP, C.c_int64, _FP]
        _lib.cor_backward.restype = None
        _lib.cor_ecm.argtypes = [C.POINTER(_Model), C.POINTER(_EcmCfg), C.c_int64, C.c_int64, _FP, _FP,
                                 _IP, C.c_int64, _FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP,
                                 _DP, C.POINTER(_EcmOut)]
        _lib.cor_ecm.restype = None
        _lib.cor_transition_sums.argtypes = [C.c_int32, C.c_int64, _DP, _DP, _DP, _DP, _DP, _DP,
                                             C.POINTER(C.c_int64)]
        _lib.cor_transition_sums.restype = None
    return _lib


def _f32(x):
    return float(np.float32(x))


def _fp(a):
    return None if a is None else a.ctypes.data_as(_FP)


def _c32(a, ndim=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if ndim is not None and a.ndim != ndim:
        raise ValueError("Buffer has wrong number of dimensions")
    return a


def _model(state_dim, matrixF, matrixQ0, stateInit, stateCovarInit, pad, oMin, oMax, pMin, pMax,
           apn=(1.0e-4, 1000.0, 5.0, 10.0, 2.0)):
    mdl = _Model()
    mdl.state_dim = state_dim
    if state_dim == 2:
        F = np.asarray(matrixF, dtype=np.float32)
        Q = np.asarray(matrixQ0, dtype=np.float32)
        mdl.F[:] = [float(F[0, 0]), float(F[0, 1]), float(F[1, 0]), float(F[1, 1])]
        mdl.Q0[:] = [float(Q[0, 0]), float(Q[0, 1]), float(Q[1, 0]), float(Q[1, 1])]
    else:
        Q = np.asarray(matrixQ0, dtype=np.float32)
        mdl.F[:] = [1.0, 0.0, 0.0, 1.0]
        mdl.Q0[:] = [float(Q[0, 0]), 0.0, 0.0, 0.0]
    mdl.state_init = _f32(stateInit)
    mdl.state_covar_init = _f32(stateCovarInit)
    mdl.pad = _f32(pad)
    mdl.w_min, mdl.w_max = _f32(oMin), _f32(oMax)
    mdl.k_min, mdl.k_max = _f32(pMin), _f32(pMax)
    mdl.apn_min_q, mdl.apn_max_q, mdl.apn_thresh, mdl.apn_scale, mdl.apn_pc = [_f32(v) for v in apn]
    return mdl


def _check_bounds(lo, hi, obs):
    if lo <= 0.0 or hi <= 0.0 or hi < lo:
        raise ValueError(("observation" if obs else "process")
                         + " precision multiplier bounds must satisfy 0 < min <= max")


def _coerce_qscale(qs, n):
    arr = np.ascontiguousarray(qs, dtype=np.float32).reshape(-1)
    if arr.shape[0] != n:
        raise ValueError("processQScale length must match intervalCount")
    if not np.all(np.isfinite(arr) & (arr > 0)):
        raise ValueError("processQScale must contain only positive finite values")
    if n > 0 and abs(float(arr[0]) - 1.0) > 1.0e-6:
        raise ValueError("processQScale[0] must be 1.0")
    return arr


def _forward(state_dim, matrixData, matrixPluginMuncInit, matrixF, matrixQ0, intervalToBlockMap, blockCount,
             stateInit, stateCovarInit, pad, stateForward, stateCovarForward, pNoiseForward, vectorD,
             returnNLL, storeNLLInD, lambdaExp, processPrecExp, useObs, useProc, useAPN, oMin, oMax, pMin,
             pMax, apn, processQScale):
    data = _c32(matrixData, 2)
    munc = _c32(matrixPluginMuncInit, 2)
    m, n = data.shape
    use_qs = processQScale is not None
    use_lam = bool(useObs) and (lambdaExp is not None)
    use_kap = bool(useProc) and (processPrecExp is not None) and ((not useAPN) or use_qs)
    qs = _coerce_qscale(processQScale, n) if use_qs else None
    if n <= 0 or m <= 0:
        D = np.empty(n, dtype=np.float32) if vectorD is None else vectorD
        return (np.float32(0.0), 0, D, 0.0) if returnNLL else (np.float32(0.0), 0, D)
    if blockCount <= 0:
        raise ValueError("blockCount must be positive")
    if munc.shape != data.shape:
        raise ValueError("matrixPluginMuncInit shape must match matrixData shape")
    _check_bounds(_f32(oMin), _f32(oMax), True)
    _check_bounds(_f32(pMin), _f32(pMax), False)
    bm = np.ascontiguousarray(intervalToBlockMap, dtype=np.int32)
    if bm.shape[0] < n:
        raise ValueError("intervalToBlockMap length must match intervalCount")
    lam = kap = None
    if use_lam:
        lam = _c32(lambdaExp, 1)
        if lam.shape[0] != n:
            raise ValueError("lambdaExp length must match intervalCount")
    if use_kap:
        kap = _c32(processPrecExp, 1)
        if kap.shape[0] != n:
            raise ValueError("processPrecExp length must match intervalCount")
    if vectorD is None:
        vectorD = np.empty(n, dtype=np.float32)
    mdl = _model(state_dim, matrixF, matrixQ0, stateInit, stateCovarInit, pad, oMin, oMax, pMin, pMax, apn)
    if state_dim == 1 and mdl.Q0[0] <= 0.0:
        raise ValueError("matrixQ0[0, 0] must be positive")
    qdiag = 0.5 * (mdl.Q0[0] + mdl.Q0[3]) if state_dim == 2 else mdl.Q0[0]
    if qdiag <= 1.0e-12:
        useAPN = False
    io = _FwdIO()
    io.m, io.n = m, n
    io.data, io.munc = _fp(data), _fp(munc)
    io.block_map = bm.ctypes.data_as(_IP)
    io.block_count = int(blockCount)
    io.lam, io.kappa, io.qscale = _fp(lam), _fp(kap), _fp(qs)
    io.use_apn, io.return_nll, io.store_nll_in_d = int(bool(useAPN)), int(bool(returnNLL)), int(bool(storeNLLInD))
    io.D = _fp(vectorD)
    if stateForward is not None:
        io.xf, io.Pf, io.pnoise = _fp(stateForward), _fp(stateCovarForward), _fp(pNoiseForward)
    out = _FwdOut()
    lib().cor_forward(C.byref(mdl), C.byref(io), C.byref(out))
    if out.invalid_block_index >= 0:
        raise ValueError("intervalToBlockMap has out-of-range block id")
    phi = float(np.float32(out.sum_d / float(n)))
    if returnNLL:
        return (phi, 0, vectorD, float(out.sum_nll))
    return (phi, 0, vectorD)


def cforwardPass(matrixData, matrixPluginMuncInit, matrixF, matrixQ0, intervalToBlockMap, blockCount, stateInit,
                 stateCovarInit, pad=1.0e-4, projectStateDuringFiltering=False, stateLowerBound=0.0,
                 stateUpperBound=0.0, chunkSize=1000000, stateForward=None, stateCovarForward=None,
                 pNoiseForward=None, vectorD=None, returnNLL=False, storeNLLInD=False, lambdaExp=None,
                 processPrecExp=None, ECM_useObsPrecisionReweighting=True,
                 ECM_useProcessPrecisionReweighting=True, ECM_useAPN=False, obsPrecisionMultiplierMin=0.25,
                 obsPrecisionMultiplierMax=4.0, procPrecisionMultiplierMin=0.25, procPrecisionMultiplierMax=4.0,
                 APN_minQ=1.0e-4, APN_maxQ=1000.0, APN_dStatThresh=5.0, APN_dStatScale=10.0, APN_dStatPC=2.0,
                 processQScale=None):
    return _forward(2, matrixData, matrixPluginMuncInit, matrixF, matrixQ0, intervalToBlockMap, blockCount,
                    stateInit, stateCovarInit, pad, stateForward, stateCovarForward, pNoiseForward, vectorD,
                    returnNLL, storeNLLInD, lambdaExp, processPrecExp, ECM_useObsPrecisionReweighting,
                    ECM_useProcessPrecisionReweighting, ECM_useAPN, obsPrecisionMultiplierMin,
                    obsPrecisionMultiplierMax, procPrecisionMultiplierMin, procPrecisionMultiplierMax,
                    (APN_minQ, APN_maxQ, APN_dStatThresh, APN_dStatScale, APN_dStatPC), processQScale)


def cforwardPassLevel(matrixData, matrixPluginMuncInit, matrixQ0, intervalToBlockMap, blockCount, stateInit,
                      stateCovarInit, pad=1.0e-4, chunkSize=1000000, stateForward=None, stateCovarForward=None,
                      pNoiseForward=None, vectorD=None, returnNLL=False, storeNLLInD=False, lambdaExp=None,
                      processPrecExp=None, ECM_useObsPrecisionReweighting=True,
                      ECM_useProcessPrecisionReweighting=True, ECM_useAPN=False, obsPrecisionMultiplierMin=0.25,
                      obsPrecisionMultiplierMax=4.0, procPrecisionMultiplierMin=0.25,
                      procPrecisionMultiplierMax=4.0, APN_minQ=1.0e-4, APN_maxQ=1000.0, APN_dStatThresh=5.0,
                      APN_dStatScale=10.0, APN_dStatPC=2.0, processQScale=None):
    return _forward(1, matrixData, matrixPluginMuncInit, None, matrixQ0, intervalToBlockMap, blockCount,
                    stateInit, stateCovarInit, pad, stateForward, stateCovarForward, pNoiseForward, vectorD,
                    returnNLL, storeNLLInD, lambdaExp, processPrecExp, ECM_useObsPrecisionReweighting,
                    ECM_useProcessPrecisionReweighting, ECM_useAPN, obsPrecisionMultiplierMin,
                    obsPrecisionMultiplierMax, procPrecisionMultiplierMin, procPrecisionMultiplierMax,
                    (APN_minQ, APN_maxQ, APN_dStatThresh, APN_dStatScale, APN_dStatPC), processQScale)


def _backward(d, matrixData, matrixF, stateForward, stateCovarForward, pNoiseForward, stateSmoothed,
              stateCovarSmoothed, lagCovSmoothed, postFitResiduals):
    data = _c32(matrixData, 2)
    m, n = data.shape
    xs = np.empty((n, d), np.float32) if stateSmoothed is None else stateSmoothed
    Ps = np.empty((n, d, d), np.float32) if stateCovarSmoothed is None else stateCovarSmoothed
    lag = np.empty((max(n - 1, 1), d, d), np.float32) if lagCovSmoothed is None else lagCovSmoothed
    res = np.empty((n, m), np.float32) if postFitResiduals is None else postFitResiduals
    if n <= 0:
        return (xs, Ps, lag, res)
    mdl = _model(d, matrixF, np.eye(2, dtype=np.float32), 0.0, 0.0, 0.0, 1.0, 1.0, 1.0, 1.0)
    lib().cor_backward(C.byref(mdl), m, n, _fp(data), _fp(_c32(stateForward)), _fp(_c32(stateCovarForward)),
                       _fp(_c32(pNoiseForward)), _fp(xs), _fp(Ps), _fp(lag), int(lag.shape[0]), _fp(res))
    return (xs, Ps, lag, res)


def cbackwardPass(matrixData, matrixF, stateForward, stateCovarForward, pNoiseForward, chunkSize=1000000,
                  stateSmoothed=None, stateCovarSmoothed=None, lagCovSmoothed=None, postFitResiduals=None):
    return _backward(2, matrixData, matrixF, stateForward, stateCovarForward, pNoiseForward, stateSmoothed,
                     stateCovarSmoothed, lagCovSmoothed, postFitResiduals)


def cbackwardPassLevel(matrixData, stateForward, stateCovarForward, pNoiseForward, chunkSize=1000000,
                       stateSmoothed=None, stateCovarSmoothed=None, lagCovSmoothed=None, postFitResiduals=None):
    return _backward(1, matrixData, None, stateForward, stateCovarForward, pNoiseForward, stateSmoothed,
                     stateCovarSmoothed, lagCovSmoothed, postFitResiduals)


def _ecm(d, matrixData, matrixPluginMuncInit, matrixF, matrixQ0, intervalToBlockMap, blockCount, stateInit,
         stateCovarInit, iters, rtol, pad, nu, oMin, oMax, pMin, pMax, useObs, useProc, useAPN, apn, tInner,
         returnIntermediates, returnDiagnostics, lambdaExpInit, processPrecExpInit, trackOptimizationPath,
         processQScale):
    data = _c32(matrixData, 2)
    munc = _c32(matrixPluginMuncInit, 2)
    m, n = data.shape
    use_qs = processQScale is not None
    lam = kap = None
    if useObs:
        if lambdaExpInit is None:
            lam = np.ones(n, np.float32)
        else:
            lam = np.array(lambdaExpInit, dtype=np.float32, copy=True, order="C")
            if lam.shape[0] != n:
                raise ValueError("lambdaExpInit length must match intervalCount")
            if not np.all(np.isfinite(lam)):
                raise ValueError("lambdaExpInit must contain only finite values")
            np.clip(lam, oMin, oMax, out=lam)
    use_kappa = bool(useProc) and ((not useAPN) or use_qs)
    if use_kappa:
        if processPrecExpInit is None:
            kap = np.ones(n, np.float32)
        else:
            kap = np.array(processPrecExpInit, dtype=np.float32, copy=True, order="C").reshape(-1)
            if kap.shape[0] != n:
                raise ValueError("processPrecExpInit length must match intervalCount")
            if not np.all(np.isfinite(kap)):
                raise ValueError("processPrecExpInit must contain only finite values")
            np.clip(kap, pMin, pMax, out=kap)
    qs = _coerce_qscale(processQScale, n) if use_qs else None
    mdl = _model(d, matrixF, matrixQ0, stateInit, stateCovarInit, pad, oMin, oMax, pMin, pMax, apn)
    if not (n <= 5 and (n <= 0 or m <= 0)):
        if blockCount <= 0:
            raise ValueError("blockCount must be positive")
        _check_bounds(mdl.w_min, mdl.w_max, True)
        _check_bounds(mdl.k_min, mdl.k_max, False)
        if np.asarray(intervalToBlockMap).shape[0] < n:
            raise ValueError("intervalToBlockMap length must match intervalCount")
        if munc.shape != data.shape:
            raise ValueError("matrixPluginMuncInit shape must match matrixData shape")
        if d == 2 and (mdl.Q0[0] * mdl.Q0[3] - mdl.Q0[1] * mdl.Q0[2]) == 0.0:
            raise ValueError("matrixQ0 is singular")
        if d == 1 and mdl.Q0[0] <= 0.0:
            raise ValueError("matrixQ0[0, 0] must be positive")
    qdiag = 0.5 * (mdl.Q0[0] + mdl.Q0[3]) if d == 2 else mdl.Q0[0]
    apn_eff = bool(useAPN) and not (qdiag <= 1.0e-12)
    cfg = _EcmCfg(int(iters), int(tInner), _f32(rtol), _f32(nu), int(bool(useObs)), int(use_kappa), int(apn_eff))
    bm = np.ascontiguousarray(intervalToBlockMap, dtype=np.int32)
    xf = np.empty((n, d), np.float32)
    Pf = np.empty((n, d, d), np.float32)
    pn = np.empty((n, d, d), np.float32)
    xs = np.empty((n, d), np.float32)
    Ps = np.empty((n, d, d), np.float32)
    lag = np.empty((max(n - 1, 1), d, d), np.float32)
    res = np.empty((n, m), np.float32)
    Dscr = np.empty(max(n, 1), np.float32)
    path = np.zeros(max(int(iters), 1), np.float64)
    out = _EcmOut()
    lib().cor_ecm(C.byref(mdl), C.byref(cfg), m, n, _fp(data), _fp(munc), bm.ctypes.data_as(_IP),
                  int(blockCount), _fp(qs), _fp(lam), _fp(kap), _fp(xf), _fp(Pf), _fp(pn), _fp(xs), _fp(Ps),
                  _fp(lag), _fp(res), _fp(Dscr), path.ctypes.data_as(_DP), C.byref(out))
    if out.invalid_block_index >= 0:
        raise ValueError("intervalToBlockMap has out-of-range block id")
    if out.skipped:
        diag = {
            "iters_done": 0, "max_iters": int(iters), "converged": False, "skipped": True,
            "skip_reason": "too_few_intervals" if n > 0 else "empty_input",
            "fallback": "filter_smoother_only", "stable_iters": 0, "patience_target": 2,
            "initial_nll": float(out.final_nll), "final_nll": float(out.final_nll),
            "final_abs_rel_change": None, "final_rel_improvement": None, "nll_increase_count": 0,
        }
    else:
        hi = bool(out.has_initial_nll)
        diag = {
            "iters_done": int(out.iters_done), "max_iters": int(iters), "converged": bool(out.converged),
            "skipped": False, "skip_reason": None, "fallback": None, "stable_iters": int(out.stable_iters),
            "patience_target": 2, "initial_nll": float(out.initial_nll) if hi else None,
            "final_nll": float(out.final_nll),
            "final_abs_rel_change": float(out.abs_rel_change) if hi else None,
            "final_rel_improvement": float(out.rel_improvement) if hi else None,
            "nll_increase_count": int(out.nll_increase_count),
        }
    if trackOptimizationPath:
        diag["optimization_path"] = [float(v) for v in path[: int(out.iters_done)]]
    head = (int(out.iters_done), float(out.final_nll))
    if returnIntermediates:
        body = head + (xs, Ps, lag, res, lam, kap)
        return body + (diag,) if returnDiagnostics else body
    return head + (diag,) if returnDiagnostics else head


def cfixedBackgroundECM(matrixData, matrixPluginMuncInit, matrixF, matrixQ0, intervalToBlockMap, blockCount,
                        stateInit, stateCovarInit, ECM_fixedBackgroundIters=50, ECM_fixedBackgroundRtol=1.0e-4,
                        pad=1.0e-4, ECM_robustTNu=8.0, obsPrecisionMultiplierMin=0.25,
                        obsPrecisionMultiplierMax=4.0, procPrecisionMultiplierMin=0.25,
                        procPrecisionMultiplierMax=4.0, ECM_useObsPrecisionReweighting=True,
                        ECM_useProcessPrecisionReweighting=True, ECM_useAPN=False, APN_minQ=1.0e-4,
                        APN_maxQ=1000.0, APN_dStatThresh=5.0, APN_dStatScale=10.0, APN_dStatPC=2.0,
                        t_innerIters=5, returnIntermediates=False, returnDiagnostics=False, lambdaExpInit=None,
                        processPrecExpInit=None, trackOptimizationPath=False, logIterations=True,
                        processQScale=None):
    return _ecm(2, matrixData, matrixPluginMuncInit, matrixF, matrixQ0, intervalToBlockMap, blockCount,
                stateInit, stateCovarInit, ECM_fixedBackgroundIters, ECM_fixedBackgroundRtol, pad, ECM_robustTNu,
                obsPrecisionMultiplierMin, obsPrecisionMultiplierMax, procPrecisionMultiplierMin,
                procPrecisionMultiplierMax, ECM_useObsPrecisionReweighting, ECM_useProcessPrecisionReweighting,
                ECM_useAPN, (APN_minQ, APN_maxQ, APN_dStatThresh, APN_dStatScale, APN_dStatPC), t_innerIters,
                returnIntermediates, returnDiagnostics, lambdaExpInit, processPrecExpInit,
                trackOptimizationPath, processQScale)


def cfixedBackgroundECMLevel(matrixData, matrixPluginMuncInit, matrixQ0, intervalToBlockMap, blockCount,
                             stateInit, stateCovarInit, ECM_fixedBackgroundIters=50,
                             ECM_fixedBackgroundRtol=1.0e-4, pad=1.0e-4, ECM_robustTNu=8.0,
                             obsPrecisionMultiplierMin=0.25, obsPrecisionMultiplierMax=4.0,
                             procPrecisionMultiplierMin=0.25, procPrecisionMultiplierMax=4.0,
                             ECM_useObsPrecisionReweighting=True, ECM_useProcessPrecisionReweighting=True,
                             ECM_useAPN=False, APN_minQ=1.0e-4, APN_maxQ=1000.0, APN_dStatThresh=5.0,
                             APN_dStatScale=10.0, APN_dStatPC=2.0, t_innerIters=5, returnIntermediates=False,
                             returnDiagnostics=False, lambdaExpInit=None, processPrecExpInit=None,
                             trackOptimizationPath=False, logIterations=True, processQScale=None):
    return _ecm(1, matrixData, matrixPluginMuncInit, None, matrixQ0, intervalToBlockMap, blockCount, stateInit,
                stateCovarInit, ECM_fixedBackgroundIters, ECM_fixedBackgroundRtol, pad, ECM_robustTNu,
                obsPrecisionMultiplierMin, obsPrecisionMultiplierMax, procPrecisionMultiplierMin,
                procPrecisionMultiplierMax, ECM_useObsPrecisionReweighting, ECM_useProcessPrecisionReweighting,
                ECM_useAPN, (APN_minQ, APN_maxQ, APN_dStatThresh, APN_dStatScale, APN_dStatPC), t_innerIters,
                returnIntermediates, returnDiagnostics, lambdaExpInit, processPrecExpInit,
                trackOptimizationPath, processQScale)


def _sums(d, xs, Ps, lag, F):
    xs = np.ascontiguousarray(xs, np.float64)
    Ps = np.ascontiguousarray(Ps, np.float64)
    lag = np.ascontiguousarray(lag, np.float64)
    n = xs.shape[0]
    if xs.ndim != 2 or xs.shape[1] != d:
        raise ValueError(f"stateSmoothed must have shape (n, {d})")
    if Ps.shape != (n, d, d):
        raise ValueError(f"stateCovarSmoothed must have shape (n, {d}, {d})")
    if lag.ndim != 3 or lag.shape[0] < max(n - 1, 0) or lag.shape[1:] != (d, d):
        raise ValueError(f"lagCovSmoothed must have shape (n - 1, {d}, {d})")
    Fa = np.ascontiguousarray(np.eye(2) if F is None else F, np.float64)
    if Fa.shape != (2, 2):
        raise ValueError("matrixF must have shape (2, 2)")
    sl, st, cnt = C.c_double(0), C.c_double(0), C.c_int64(0)
    lib().cor_transition_sums(d, n, xs.ctypes.data_as(_DP), Ps.ctypes.data_as(_DP), lag.ctypes.data_as(_DP),
                              Fa.ctypes.data_as(_DP), C.byref(sl), C.byref(st), C.byref(cnt))
    return (sl.value, st.value, int(cnt.value))


def cExpectedTransitionResidualSums(stateSmoothed, stateCovarSmoothed, lagCovSmoothed, matrixF):
    return _sums(2, stateSmoothed, stateCovarSmoothed, lagCovSmoothed, matrixF)


def cExpectedTransitionResidualSumsLevel(stateSmoothed, stateCovarSmoothed, lagCovSmoothed):
    return _sums(1, stateSmoothed, stateCovarSmoothed, lagCovSmoothed, None)


# ---------------------------------------------------------------------------------------------------------------------
# background update natives (pyx:9700-9724, pyx:944-1096)
# ---------------------------------------------------------------------------------------------------------------------
_DP = C.POINTER(C.c_double)


def cbackgroundWeightedStatsWithSupport(residualMatrix, invVarMatrix):
    res = np.ascontiguousarray(residualMatrix, dtype=np.float32)
    inv = np.ascontiguousarray(invVarMatrix, dtype=np.float32)
    if res.ndim != 2 or inv.shape != res.shape:
        raise ValueError("residualMatrix and invVarMatrix must have identical 2D shapes")
    m, n = res.shape
    w, r = np.empty(n), np.empty(n)
    f = lib().cor_background_stats
    f.restype = C.c_int64
    f.argtypes = [C.c_int64, C.c_int64, _FP, _FP, _DP, _DP]
    support = f(m, n, _fp(res), _fp(inv), w.ctypes.data_as(_DP), r.ctypes.data_as(_DP))
    return w, r, int(support)


def csolveZeroCenteredBackground(weightTrack, rhsTrack, lam, zeroCenter=True, lamFirst=0.0):
    w = np.ascontiguousarray(weightTrack, dtype=np.float64)
    r = np.ascontiguousarray(rhsTrack, dtype=np.float64)
    n = w.shape[0]
    if r.shape[0] != n:
        raise ValueError("weightTrack and rhsTrack must have the same length")
    if not np.isfinite(lamFirst) or lamFirst < 0.0:
        raise ValueError("lamFirst must be finite and nonnegative")
    if not np.isfinite(lam) or lam < 0.0:
        raise ValueError("lam must be finite and nonnegative")
    out = np.zeros(n)
    f = lib().cor_solve_background
    f.restype = C.c_int64
    f.argtypes = [C.c_int64, _DP, _DP, C.c_double, C.c_int, C.c_double, _DP, _DP]
    badv = C.c_double(0.0)
    bad = f(n, w.ctypes.data_as(_DP), r.ctypes.data_as(_DP), float(lam), int(bool(zeroCenter)), float(lamFirst),
            out.ctypes.data_as(_DP), C.byref(badv))
    if bad >= 0:
        raise RuntimeError("roughness-penalized LDL factorization required pivot "
                           f"modification at index {bad} (pivot={badv.value:.6g}, floor={1.0e-12:.6g}).")
    return out


# ---------------------------------------------------------------------------------------------------------------------
# delete-block calibration natives (cuncertainty.pyx:97-157, 160-305)
# ---------------------------------------------------------------------------------------------------------------------
_U8 = C.POINTER(C.c_uint8)
_I32 = C.POINTER(C.c_int32)
_I64 = C.POINTER(C.c_int64)


def _munc_arg(matrixMunc):
    a = np.asarray(matrixMunc)
    if a.dtype == np.float64:
        a = np.ascontiguousarray(a, np.float64)
        return a, 1
    a = np.ascontiguousarray(a, np.float32)
    return a, 0


def _check_rho_pad(pad, rho):
    if not np.isfinite(pad):
        raise ValueError("observation information pad must be finite")
    if not np.isfinite(rho) or rho < 0.0 or rho >= 1.0:
        raise ValueError("replicate dependence rho must be in [0, 1)")


def cobservationTotalInformation(matrixMunc, activeMask, lambdaExp, useLambda, pad, replicateDependenceRho=0.0):
    munc, f64 = _munc_arg(matrixMunc)
    act = np.ascontiguousarray(activeMask, np.uint8)
    lam = np.ascontiguousarray(lambdaExp, np.float64)
    m, n = munc.shape
    if act.shape != (m, n):
        raise ValueError("activeMask must match matrixMunc shape")
    if useLambda and lam.shape[0] != n:
        raise ValueError("fullObservationPrecision must match interval count")
    _check_rho_pad(pad, replicateDependenceRho)
    total = np.zeros(n)
    f = lib().cor_total_information
    f.restype = None
    f.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_int, _U8, _DP, C.c_double, C.c_double, _DP]
    f(m, n, munc.ctypes.data, f64, act.ctypes.data_as(_U8), lam.ctypes.data_as(_DP) if useLambda else None, float(pad),
      float(replicateDependenceRho), total.ctypes.data_as(_DP))
    return total


def check_fold_spec(m, n, blockLen, fold, blockFold, repsByBlockCount, repsByBlock):
    """argument validation of cuncertainty.pyx:175-231 (shared by the oracle and the product mirror's tests)"""
    if m < 1 or n < 1 or blockLen < 1:
        raise ValueError("invalid uncertainty calibration mask dimensions")
    if fold < 0:
        raise ValueError("fold must be nonnegative")
    bc = (n + blockLen - 1) // blockLen
    if blockFold.shape[0] != bc or repsByBlockCount.shape[0] != bc or repsByBlock.shape[0] != bc:
        raise ValueError("fold spec has inconsistent block count")
    slots = repsByBlock.shape[1]
    if slots < m:
        raise ValueError("fold spec replicate matrix must allow every sample")
    if np.any(blockFold < 0):
        raise ValueError("fold spec contains negative fold id")
    if np.any(repsByBlockCount < 1) or np.any(repsByBlockCount > m) or np.any(repsByBlockCount > slots):
        raise ValueError("fold spec deleted-replicate count is out of bounds")
    for b in range(bc):
        r = repsByBlock[b, : repsByBlockCount[b]]
        if np.any(r < 0) or np.any(r >= m):
            raise ValueError("fold spec replicate is out of bounds")
        if len(set(r.tolist())) != len(r):
            raise ValueError("fold spec contains a duplicate replicate")


def cmakeFoldMaskAndInformation(m, n, blockLen, fold, blockFold, repsByBlockCount, repsByBlock, matrixMunc, activeMask,
                                totalInfo, lambdaExp, useLambda, pad, replicateDependenceRho=0.0,
                                returnNominalHeldout=False):
    munc, f64 = _munc_arg(matrixMunc)
    act = np.ascontiguousarray(activeMask, np.uint8)
    bf = np.ascontiguousarray(blockFold, np.int32)
    rc = np.ascontiguousarray(repsByBlockCount, np.int64)
    rb = np.ascontiguousarray(repsByBlock, np.int64)
    tot = np.ascontiguousarray(totalInfo, np.float64)
    lam = np.ascontiguousarray(lambdaExp, np.float64)
    if m < 1 or n < 1 or blockLen < 1:
        raise ValueError("invalid uncertainty calibration mask dimensions")
    if fold < 0:
        raise ValueError("fold must be nonnegative")
    if munc.shape != (m, n):
        raise ValueError("matrixMunc shape does not match fold spec")
    if act.shape != (m, n):
        raise ValueError("activeMask must match matrixMunc shape")
    if tot.shape[0] != n:
        raise ValueError("total information must match interval count")
    if useLambda and lam.shape[0] != n:
        raise ValueError("fullObservationPrecision must match interval count")
    _check_rho_pad(pad, replicateDependenceRho)
    check_fold_spec(m, n, blockLen, fold, bf, rc, rb)
    mask = np.ones((m, n), np.uint8)
    kept, held, h = np.empty(n), np.zeros(n), np.empty(n)
    nominal = np.zeros(n) if returnNominalHeldout else None
    f = lib().cor_fold_mask_information
    f.restype = None
    f.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_int64, _I32, _I64, _I64, C.c_int64, C.c_void_p, C.c_int, _U8, _DP,
                  _DP, C.c_double, C.c_double, _U8, _DP, _DP, _DP, _DP]
    f(m, n, int(blockLen), int(fold), bf.ctypes.data_as(_I32), rc.ctypes.data_as(_I64), rb.ctypes.data_as(_I64),
      rb.shape[1], munc.ctypes.data, f64, act.ctypes.data_as(_U8), tot.ctypes.data_as(_DP),
      lam.ctypes.data_as(_DP) if useLambda else None, float(pad), float(replicateDependenceRho), mask.ctypes.data_as(_U8),
      kept.ctypes.data_as(_DP), held.ctypes.data_as(_DP), h.ctypes.data_as(_DP),
      nominal.ctypes.data_as(_DP) if nominal is not None else None)
    if returnNominalHeldout:
        return mask, kept, held, h, nominal
    return mask, kept, held, h
