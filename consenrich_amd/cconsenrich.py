"""Drop-in replacement for the hot-path callables of the reference's ``consenrich.cconsenrich`` extension.

Same names, keyword arguments, defaults, return tuples and error behaviour as the Cython originals
(/root/reference/src/consenrich/cconsenrich.pyx: cforwardPass :6393, cbackwardPass :6635, cforwardPassLevel :6853,
cbackwardPassLevel :7052, cfixedBackgroundECMLevel :7153, cfixedBackgroundECM :7660,
cExpectedTransitionResidualSums :710, cExpectedTransitionResidualSumsLevel :818); the arithmetic runs on an MI355X
through the C ABI in include/consenrich_amd.h (ctypes only).  ``core.py`` resolves these callables by module
attribute at call time (core.py:3286-3290), so ``consenrich.cconsenrich.<name> = consenrich_amd.cconsenrich.<name>``
 routes the reference's own ``runConsenrich`` through the GPU.

No CPU fallback: without the built library or without a GPU every call raises.
"""
from __future__ import annotations

import ctypes as C
import sys

import numpy as np

from . import _lib as L

def set_validation(x_tol_ulps: int) -> None:
    """Carry validation of the forward state chain for the calls in this module (see csr_set_validation):
    0 = bit-exact sequential semantics (the default, here and in the batch API), k > 0 = accept speculative carries within
    k float32 ulps (2 is the opt-in throughput setting; its contract is per pass, not through an ECM loop)."""
    L.check(L.lib().csr_set_validation(None, int(x_tol_ulps)))


__all__ = [
    "cforwardPass", "cbackwardPass", "cforwardPassLevel", "cbackwardPassLevel", "cfixedBackgroundECM",
    "cfixedBackgroundECMLevel", "cExpectedTransitionResidualSums", "cExpectedTransitionResidualSumsLevel",
]


# ------------------------------------------------------------------------------------------------------------------
# argument coercion with the typed-buffer semantics of the Cython signatures (mode="c" float32 / int32 ndarrays)
# ------------------------------------------------------------------------------------------------------------------
def _typed(arr, name, dtype, ndim, out=False):
    """The reference's typed-buffer argument checks (`np.ndarray[dtype, ndim, mode="c"]`, pyx:6394-6398).  out=True: the array
    is written in place (the library copies device results straight into it), so a read-only buffer is refused here like
    Cython's writable buffer acquisition does."""
    if not isinstance(arr, np.ndarray):
        raise TypeError(f"Argument '{name}' has incorrect type (expected numpy.ndarray, got {type(arr).__name__})")
    if arr.dtype != dtype:
        raise ValueError(f"Buffer dtype mismatch for '{name}': expected {np.dtype(dtype).name}, got {arr.dtype.name}")
    if arr.ndim != ndim:
        raise ValueError(f"Buffer has wrong number of dimensions (expected {ndim}, got {arr.ndim})")
    if not arr.flags.c_contiguous:
        raise ValueError("ndarray is not C-contiguous")
    if out and not arr.flags.writeable:
        raise ValueError("buffer source array is read-only")
    return arr


def _f32(x) -> float:
    return float(np.float32(x))


def _validate_bounds(lo, hi, obs):  # pyx:143-151
    if lo <= 0.0 or hi <= 0.0 or hi < lo:
        raise ValueError(("observation" if obs else "process")
                         + " precision multiplier bounds must satisfy 0 < min <= max")


def _coerce_qscale(qs, n):  # pyx:101-131
    arr = np.ascontiguousarray(qs, dtype=np.float32).reshape(-1)
    if arr.shape[0] != n:
        raise ValueError("processQScale length must match intervalCount")
    if not np.all(np.isfinite(arr) & (arr > 0.0)):
        raise ValueError("processQScale must contain only positive finite values")
    if n > 0 and abs(float(arr[0]) - 1.0) > 1.0e-6:
        raise ValueError("processQScale[0] must be 1.0")
    return arr


def _check_block_map(bm, n, blockCount):  # pyx:389-392 + 6624
    head = bm[:n]
    if head.size and (int(head.min()) < 0 or int(head.max()) >= blockCount):
        raise ValueError("intervalToBlockMap has out-of-range block id")


def _model(d, matrixF, matrixQ0, stateInit, stateCovarInit, pad, oMin, oMax, pMin, pMax,
           apn=(1.0e-4, 1000.0, 5.0, 10.0, 2.0)):
    mdl = L.Model()
    mdl.state_dim = d
    if d == 2:
        mdl.F[:] = [float(matrixF[0, 0]), float(matrixF[0, 1]), float(matrixF[1, 0]), float(matrixF[1, 1])]
        mdl.Q0[:] = [float(matrixQ0[0, 0]), float(matrixQ0[0, 1]), float(matrixQ0[1, 0]), float(matrixQ0[1, 1])]
    else:
        mdl.F[:] = [1.0, 0.0, 0.0, 1.0]
        mdl.Q0[:] = [float(matrixQ0[0, 0]), 0.0, 0.0, 0.0]
    mdl.state_init, mdl.state_covar_init, mdl.pad = _f32(stateInit), _f32(stateCovarInit), _f32(pad)
    mdl.w_min, mdl.w_max, mdl.k_min, mdl.k_max = _f32(oMin), _f32(oMax), _f32(pMin), _f32(pMax)
    mdl.apn_min_q, mdl.apn_max_q, mdl.apn_thresh, mdl.apn_scale, mdl.apn_pc = (_f32(v) for v in apn)
    return mdl


# ------------------------------------------------------------------------------------------------------------------
# forward pass
# ------------------------------------------------------------------------------------------------------------------
def _forward(d, matrixData, matrixPluginMuncInit, matrixF, matrixQ0, intervalToBlockMap, blockCount, stateInit,
             stateCovarInit, pad, stateForward, stateCovarForward, pNoiseForward, vectorD, returnNLL, storeNLLInD,
             lambdaExp, processPrecExp, useObs, useProc, useAPN, oMin, oMax, pMin, pMax, apn, processQScale):
    data = _typed(matrixData, "matrixData", np.float32, 2)
    munc = _typed(matrixPluginMuncInit, "matrixPluginMuncInit", np.float32, 2)
    if d == 2:
        matrixF = _typed(matrixF, "matrixF", np.float32, 2)
    matrixQ0 = _typed(matrixQ0, "matrixQ0", np.float32, 2)
    bm = _typed(intervalToBlockMap, "intervalToBlockMap", np.int32, 1)
    m, n = data.shape
    doStore = stateForward is not None
    useLambda = bool(useObs) and (lambdaExp is not None)
    useQS = processQScale is not None
    useProcPrec = bool(useProc) and (processPrecExp is not None) and ((not useAPN) or useQS)
    lam = _typed(lambdaExp, "lambdaExp", np.float32, 1) if useLambda else None
    kap = _typed(processPrecExp, "processPrecExp", np.float32, 1) if useProcPrec else None
    qs = _coerce_qscale(processQScale, n) if useQS else None

    if n <= 0 or m <= 0:  # pyx:6494-6501
        D = np.empty(n, dtype=np.float32) if vectorD is None else _typed(vectorD, "vectorD", np.float32, 1, out=True)
        return (np.float32(0.0), 0, D, 0.0) if returnNLL else (np.float32(0.0), 0, D)
    if blockCount <= 0:
        raise ValueError("blockCount must be positive")
    if munc.shape[0] != m or munc.shape[1] != n:
        raise ValueError("matrixPluginMuncInit shape must match matrixData shape")
    if d == 2:
        if matrixF.shape[0] < 2 or matrixF.shape[1] < 2:
            raise ValueError("matrixF must have at least shape (2, 2)")
        if matrixQ0.shape[0] < 2 or matrixQ0.shape[1] < 2:
            raise ValueError("matrixQ0 must have at least shape (2, 2)")
    else:
        if matrixQ0.shape[0] < 1 or matrixQ0.shape[1] < 1:
            raise ValueError("matrixQ0 must have at least shape (1, 1)")
        if float(matrixQ0[0, 0]) <= 0.0:
            raise ValueError("matrixQ0[0, 0] must be positive")
    _validate_bounds(_f32(oMin), _f32(oMax), True)
    _validate_bounds(_f32(pMin), _f32(pMax), False)
    if bm.shape[0] < n:
        raise ValueError("intervalToBlockMap length must match intervalCount")
    if useLambda and lam.shape[0] != n:
        raise ValueError("lambdaExp length must match intervalCount")
    if useProcPrec and kap.shape[0] != n:
        raise ValueError("processPrecExp length must match intervalCount")
    if vectorD is None:
        vectorD = np.empty(n, dtype=np.float32)
    else:
        vectorD = _typed(vectorD, "vectorD", np.float32, 1, out=True)
        if vectorD.shape[0] < n:
            raise ValueError("vectorD length must match intervalCount")
    if doStore:
        sf = _typed(stateForward, "stateForward", np.float32, 2, out=True)
        sc = _typed(stateCovarForward, "stateCovarForward", np.float32, 3, out=True)
        pn = _typed(pNoiseForward, "pNoiseForward", np.float32, 3, out=True)
        if sf.shape[0] < n or sf.shape[1] < d:
            raise ValueError(f"stateForward shape must match intervalCount by {d}")
        if sc.shape[0] < n or sc.shape[1] < d or sc.shape[2] < d:
            raise ValueError(f"stateCovarForward shape must match intervalCount by {d} by {d}")
        if n > 1 and (pn.shape[0] < n - 1 or pn.shape[1] < d or pn.shape[2] < d):
            raise ValueError(f"pNoiseForward shape must permit intervalCount minus one {d} by {d} entries")
    mdl = _model(d, matrixF, matrixQ0, stateInit, stateCovarInit, pad, oMin, oMax, pMin, pMax, apn)
    qdiag = 0.5 * (mdl.Q0[0] + mdl.Q0[3]) if d == 2 else mdl.Q0[0]
    if qdiag <= 1.0e-12:  # pyx:6575 / 6997
        useAPN = False
    _check_block_map(bm, n, blockCount)

    flags = 0
    flags |= L.USE_LAMBDA if useLambda else 0
    flags |= L.USE_KAPPA if useProcPrec else 0
    flags |= L.USE_QSCALE if useQS else 0
    flags |= L.USE_APN if useAPN else 0
    flags |= L.RETURN_NLL if returnNLL else 0
    flags |= L.NLL_IN_D if storeNLLInD else 0
    io = L.FwdIO()
    io.m, io.n = m, n
    io.data, io.munc = L.fp(data), L.fp(munc)
    io.lam, io.kappa, io.qscale = L.fp(lam), L.fp(kap), L.fp(qs)
    io.flags = flags
    # The Cython loops write through raw row-major pointers (pyx:498-508): the library does the same, straight into the
    # caller's (C-contiguous, checked above) buffers -- the first n*d, n*d*d and (n-1)*d*d floats; nothing is staged or
    # copied on the host (at chr1 x 32 the bounce buffers and their first-touch page faults cost more than the GPU pass)
    io.D = L.fp(vectorD)
    if doStore:
        pn_dst = pn if pn.size else np.empty((1, d, d), np.float32)      # n == 1: no entry is written
        io.xf, io.Pf, io.pnoise = L.fp(sf), L.fp(sc), L.fp(pn_dst)
    out = L.FwdOut()
    L.check(L.lib().csr_forward_pass(C.byref(mdl), C.byref(io), C.byref(out)))
    phiHat = float(np.float32(out.sum_d / float(n)))
    if returnNLL:
        return (phiHat, 0, vectorD, float(out.sum_nll))
    return (phiHat, 0, vectorD)


def cforwardPass(matrixData, matrixPluginMuncInit, matrixF, matrixQ0, intervalToBlockMap, blockCount, stateInit,
                 stateCovarInit, pad=1.0e-4, projectStateDuringFiltering=False, stateLowerBound=0.0,
                 stateUpperBound=0.0, chunkSize=1000000, stateForward=None, stateCovarForward=None,
                 pNoiseForward=None, vectorD=None, returnNLL=False, storeNLLInD=False, lambdaExp=None,
                 processPrecExp=None, ECM_useObsPrecisionReweighting=True,
                 ECM_useProcessPrecisionReweighting=True, ECM_useAPN=False, obsPrecisionMultiplierMin=0.25,
                 obsPrecisionMultiplierMax=4.0, procPrecisionMultiplierMin=0.25, procPrecisionMultiplierMax=4.0,
                 APN_minQ=1.0e-4, APN_maxQ=1000.0, APN_dStatThresh=5.0, APN_dStatScale=10.0, APN_dStatPC=2.0,
                 processQScale=None):
    """Forward Kalman filter, levelTrend model (pyx:6393-6632 -> pyx:291-529) on the GPU.

    ``projectStateDuringFiltering``, ``stateLower/UpperBound`` and ``chunkSize`` are accepted and, as in the
    reference, have no effect on the result.
    """
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
    """Forward filter, scalar level model (pyx:6853-7049 -> pyx:538-707) on the GPU."""
    return _forward(1, matrixData, matrixPluginMuncInit, None, matrixQ0, intervalToBlockMap, blockCount,
                    stateInit, stateCovarInit, pad, stateForward, stateCovarForward, pNoiseForward, vectorD,
                    returnNLL, storeNLLInD, lambdaExp, processPrecExp, ECM_useObsPrecisionReweighting,
                    ECM_useProcessPrecisionReweighting, ECM_useAPN, obsPrecisionMultiplierMin,
                    obsPrecisionMultiplierMax, procPrecisionMultiplierMin, procPrecisionMultiplierMax,
                    (APN_minQ, APN_maxQ, APN_dStatThresh, APN_dStatScale, APN_dStatPC), processQScale)


# ------------------------------------------------------------------------------------------------------------------
# backward pass
# ------------------------------------------------------------------------------------------------------------------
def _backward(d, matrixData, matrixF, stateForward, stateCovarForward, pNoiseForward, stateSmoothed,
              stateCovarSmoothed, lagCovSmoothed, postFitResiduals):
    data = _typed(matrixData, "matrixData", np.float32, 2)
    if d == 2:
        matrixF = _typed(matrixF, "matrixF", np.float32, 2)
    xf = _typed(stateForward, "stateForward", np.float32, 2)
    Pf = _typed(stateCovarForward, "stateCovarForward", np.float32, 3)
    pn = _typed(pNoiseForward, "pNoiseForward", np.float32, 3)
    m, n = data.shape
    xs = np.empty((n, d), np.float32) if stateSmoothed is None else _typed(stateSmoothed, "stateSmoothed", np.float32, 2, out=True)
    Ps = (np.empty((n, d, d), np.float32) if stateCovarSmoothed is None
          else _typed(stateCovarSmoothed, "stateCovarSmoothed", np.float32, 3, out=True))
    lag = (np.empty((max(n - 1, 1), d, d), np.float32) if lagCovSmoothed is None
           else _typed(lagCovSmoothed, "lagCovSmoothed", np.float32, 3, out=True))
    res = (np.empty((n, m), np.float32) if postFitResiduals is None
           else _typed(postFitResiduals, "postFitResiduals", np.float32, 2, out=True))
    if n <= 0:
        return (xs, Ps, lag, res)
    if m <= 0:
        raise ValueError("matrixData must have at least one track")
    mdl = _model(d, matrixF, np.eye(2, dtype=np.float32), 0.0, 0.0, 0.0, 1.0, 1.0, 1.0, 1.0)
    # inputs: the library reads n rows of xf / Pf and n - 1 rows of pNoise; a caller array whose rows are exactly (d) /
    # (d, d) wide is read in place, anything wider is packed first
    xf_c = xf if xf.shape[1] == d else np.ascontiguousarray(xf[:n, :d])
    Pf_c = Pf if Pf.shape[1:] == (d, d) else np.ascontiguousarray(Pf[:n, :d, :d])
    if pn.shape[1:] == (d, d) and pn.shape[0] >= max(n - 1, 1):
        pn_c = pn
    else:
        pn_c = np.zeros((max(n - 1, 1), d, d), np.float32)
        if n > 1:
            pn_c[: n - 1] = pn[: n - 1, :d, :d]
    # outputs: written in place when the caller's (or the freshly allocated) array has exactly the shape the pass fills;
    # larger preallocated arrays (`shape[0] >= n`, pyx:6545-6561) go through a bounce buffer and a sliced store
    rows = min(n - 1, lag.shape[0])  # `if k < lagCovSmoothedArr.shape[0]` pyx:6840
    xs_b = xs if xs.shape == (n, d) else np.empty((n, d), np.float32)
    Ps_b = Ps if Ps.shape == (n, d, d) else np.empty((n, d, d), np.float32)
    lag_b = lag if (lag.shape[1:] == (d, d) and lag.shape[0] >= max(n - 1, 1)) else np.zeros((max(n - 1, 1), d, d), np.float32)
    res_b = res if res.shape == (n, m) else np.empty((n, m), np.float32)
    L.check(L.lib().csr_backward_pass(C.byref(mdl), m, n, L.fp(data), L.fp(xf_c), L.fp(Pf_c), L.fp(pn_c),
                                      L.fp(xs_b), L.fp(Ps_b), L.fp(lag_b), int(lag_b.shape[0]), L.fp(res_b)))
    if xs_b is not xs:
        xs[:n, :d] = xs_b
    if Ps_b is not Ps:
        Ps[:n, :d, :d] = Ps_b
    if lag_b is not lag and rows > 0:
        lag[:rows, :d, :d] = lag_b[:rows]
    if res_b is not res:
        res[:n, :m] = res_b
    return (xs, Ps, lag, res)


def cbackwardPass(matrixData, matrixF, stateForward, stateCovarForward, pNoiseForward, chunkSize=1000000,
                  stateSmoothed=None, stateCovarSmoothed=None, lagCovSmoothed=None, postFitResiduals=None):
    """RTS smoother, levelTrend model (pyx:6635-6850) on the GPU."""
    return _backward(2, matrixData, matrixF, stateForward, stateCovarForward, pNoiseForward, stateSmoothed,
                     stateCovarSmoothed, lagCovSmoothed, postFitResiduals)


def cbackwardPassLevel(matrixData, stateForward, stateCovarForward, pNoiseForward, chunkSize=1000000,
                       stateSmoothed=None, stateCovarSmoothed=None, lagCovSmoothed=None, postFitResiduals=None):
    """RTS smoother, scalar level model (pyx:7052-7150) on the GPU."""
    return _backward(1, matrixData, None, stateForward, stateCovarForward, pNoiseForward, stateSmoothed,
                     stateCovarSmoothed, lagCovSmoothed, postFitResiduals)


# ------------------------------------------------------------------------------------------------------------------
# fixed-background ECM
# ------------------------------------------------------------------------------------------------------------------
def _replay_path(tag, path, rtol, log):
    """Rebuild the per-iteration convergence records (pyx:8337-8402) from the NLL path returned by the device loop."""
    records = []
    prev = 1.0e16
    have = False
    stable = 0
    for i, cur in enumerate(path):
        have_prev = have
        have = True
        if have_prev:
            delta, scale = abs(cur - prev), abs(prev)
        else:
            delta, scale = 0.0, abs(cur)
        scale = max(scale, abs(cur), 1.0)
        rel = (prev - cur) / scale if have_prev else 0.0
        absrel = delta / scale if have_prev else 0.0
        tol = rtol * scale
        prev = cur
        stable = stable + 1 if (have_prev and delta <= tol) else 0
        conv = stable >= 2
        if log:
            sys.stderr.write(f"\n\t[{tag}] iter={i + 1}\n")
            sys.stderr.write(f"\t[{tag}] NLL={cur:.6f}  REL={rel:+.6e}  ABSREL={absrel:.6e}  THRESH={tol:.6e}\n")
            sys.stderr.write(f"\t[{tag}] stable={stable}/2\n")
            if conv:
                sys.stderr.write(f"\t[{tag}] CONVERGED (ECM) iter={i + 1} \n")
        records.append({
            "iter": i + 1, "objective_name": "nll", "objective_value": float(cur),
            "change": float(delta) if have_prev else None,
            "relative_improvement": float(rel) if have_prev else None,
            "abs_relative_change": float(absrel) if have_prev else None,
            "threshold": float(tol) if have_prev else None, "stable_iters": int(stable), "patience_target": 2,
            "reset_iteration": bool(not have_prev), "converged": bool(conv),
        })
    return records


def _ecm(d, tag, matrixData, matrixPluginMuncInit, matrixF, matrixQ0, intervalToBlockMap, blockCount, stateInit,
         stateCovarInit, iters, rtol, pad, nu, oMin, oMax, pMin, pMax, useObs, useProc, useAPN, apn, tInner,
         returnIntermediates, returnDiagnostics, lambdaExpInit, processPrecExpInit, trackOptimizationPath,
         logIterations, processQScale):
    data = _typed(matrixData, "matrixData", np.float32, 2)
    munc = _typed(matrixPluginMuncInit, "matrixPluginMuncInit", np.float32, 2)
    if d == 2:
        matrixF = _typed(matrixF, "matrixF", np.float32, 2)
    matrixQ0 = _typed(matrixQ0, "matrixQ0", np.float32, 2)
    bm = _typed(intervalToBlockMap, "intervalToBlockMap", np.int32, 1)
    m, n = data.shape
    iters, tInner = int(iters), int(tInner)
    useQS = processQScale is not None
    lam = kap = None
    if useObs:  # pyx:7899-7910
        if lambdaExpInit is None:
            lam = np.ones(n, dtype=np.float32)
        else:
            lam = np.array(lambdaExpInit, dtype=np.float32, copy=True, order="C")
            if lam.shape[0] != n:
                raise ValueError("lambdaExpInit length must match intervalCount")
            if not np.all(np.isfinite(lam)):
                raise ValueError("lambdaExpInit must contain only finite values")
            np.clip(lam, _f32(oMin), _f32(oMax), out=lam)
    useKappa = bool(useProc) and ((not useAPN) or useQS)  # pyx:7912
    if useKappa:
        if processPrecExpInit is None:
            kap = np.ones(n, dtype=np.float32)
        else:
            kap = np.array(processPrecExpInit, dtype=np.float32, copy=True, order="C").reshape(-1)
            if kap.shape[0] != n:
                raise ValueError("processPrecExpInit length must match intervalCount")
            if not np.all(np.isfinite(kap)):
                raise ValueError("processPrecExpInit must contain only finite values")
            np.clip(kap, _f32(pMin), _f32(pMax), out=kap)
    qs = _coerce_qscale(processQScale, n) if useQS else None

    xs = np.empty((n, d), np.float32)
    Ps = np.empty((n, d, d), np.float32)
    lag = np.empty((max(n - 1, 1), d, d), np.float32)
    res = np.empty((n, m), np.float32)
    empty = n <= 0 or m <= 0
    mdl = _model(d, matrixF if d == 2 else None, matrixQ0, stateInit, stateCovarInit, pad, oMin, oMax, pMin, pMax, apn)
    if not empty:  # pyx:8002-8011 / 8131-8140 (same checks on both paths)
        if blockCount <= 0:
            raise ValueError("blockCount must be positive")
        if d == 2:
            _validate_bounds(mdl.w_min, mdl.w_max, True)
            _validate_bounds(mdl.k_min, mdl.k_max, False)
            if bm.shape[0] < n:
                raise ValueError("intervalToBlockMap length must match intervalCount")
            if munc.shape[0] != m or munc.shape[1] != n:
                raise ValueError("matrixPluginMuncInit shape must match matrixData shape")
            if (mdl.Q0[0] * mdl.Q0[3] - mdl.Q0[1] * mdl.Q0[2]) == 0.0:
                raise ValueError("matrixQ0 is singular")
        else:
            if munc.shape[0] != m or munc.shape[1] != n:
                raise ValueError("matrixPluginMuncInit shape must match matrixData shape")
            if mdl.Q0[0] <= 0.0:
                raise ValueError("matrixQ0[0, 0] must be positive")
            _validate_bounds(mdl.w_min, mdl.w_max, True)
            _validate_bounds(mdl.k_min, mdl.k_max, False)
            if bm.shape[0] < n:
                raise ValueError("intervalToBlockMap length must match intervalCount")
        _check_block_map(bm, n, blockCount)

    out = L.EcmOut()
    path = np.zeros(max(iters, 1), np.float64)
    if empty:
        out.skipped = 1
    else:
        qdiag = 0.5 * (mdl.Q0[0] + mdl.Q0[3]) if d == 2 else mdl.Q0[0]
        apn_eff = bool(useAPN) and not (qdiag <= 1.0e-12)
        cfg = L.EcmCfg(iters, tInner, _f32(rtol), _f32(nu), int(bool(useObs)), int(useKappa), int(apn_eff), 0)
        L.check(L.lib().csr_fixed_background_ecm(C.byref(mdl), C.byref(cfg), m, n, L.fp(data), L.fp(munc), L.fp(qs),
                                                 L.fp(lam), L.fp(kap), L.fp(xs), L.fp(Ps), L.fp(lag), L.fp(res),
                                                 L.dp(path), C.byref(out)))
    itersDone = int(out.iters_done)
    if out.skipped:
        finalNLL = float(out.final_nll)
        diag = {
            "iters_done": 0, "max_iters": iters, "converged": False, "skipped": True,
            "skip_reason": "too_few_intervals" if n > 0 else "empty_input", "fallback": "filter_smoother_only",
            "stable_iters": 0, "patience_target": 2, "initial_nll": finalNLL, "final_nll": finalNLL,
            "final_abs_rel_change": None, "final_rel_improvement": None, "nll_increase_count": 0,
        }
        records = []
    else:
        finalNLL = float(out.final_nll)
        hi = bool(out.has_initial_nll)
        diag = {
            "iters_done": itersDone, "max_iters": iters, "converged": bool(out.converged), "skipped": False,
            "skip_reason": None, "fallback": None, "stable_iters": int(out.stable_iters), "patience_target": 2,
            "initial_nll": float(out.initial_nll) if hi else None, "final_nll": finalNLL,
            "final_abs_rel_change": float(out.abs_rel_change) if hi else None,
            "final_rel_improvement": float(out.rel_improvement) if hi else None,
            "nll_increase_count": int(out.nll_increase_count),
        }
        records = _replay_path(tag, [float(v) for v in path[:itersDone]], _f32(rtol), bool(logIterations))
    if trackOptimizationPath:
        diag["optimization_path"] = records
    head = (itersDone, finalNLL)
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
    """Fixed-background ECM, levelTrend model (pyx:7660-8442): the whole filter/smoother/E-step loop stays on the
    GPU; only one NLL scalar per ECM iteration returns to the host for the stop rule (pyx:8337-8407)."""
    return _ecm(2, "cfixedBackgroundECM", matrixData, matrixPluginMuncInit, matrixF, matrixQ0, intervalToBlockMap,
                blockCount, stateInit, stateCovarInit, ECM_fixedBackgroundIters, ECM_fixedBackgroundRtol, pad,
                ECM_robustTNu, obsPrecisionMultiplierMin, obsPrecisionMultiplierMax, procPrecisionMultiplierMin,
                procPrecisionMultiplierMax, ECM_useObsPrecisionReweighting, ECM_useProcessPrecisionReweighting,
                ECM_useAPN, (APN_minQ, APN_maxQ, APN_dStatThresh, APN_dStatScale, APN_dStatPC), t_innerIters,
                returnIntermediates, returnDiagnostics, lambdaExpInit, processPrecExpInit, trackOptimizationPath,
                logIterations, processQScale)


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
    """Fixed-background ECM, scalar level model (pyx:7153-7657) on the GPU."""
    return _ecm(1, "cfixedBackgroundECMLevel", matrixData, matrixPluginMuncInit, None, matrixQ0,
                intervalToBlockMap, blockCount, stateInit, stateCovarInit, ECM_fixedBackgroundIters,
                ECM_fixedBackgroundRtol, pad, ECM_robustTNu, obsPrecisionMultiplierMin, obsPrecisionMultiplierMax,
                procPrecisionMultiplierMin, procPrecisionMultiplierMax, ECM_useObsPrecisionReweighting,
                ECM_useProcessPrecisionReweighting, ECM_useAPN,
                (APN_minQ, APN_maxQ, APN_dStatThresh, APN_dStatScale, APN_dStatPC), t_innerIters,
                returnIntermediates, returnDiagnostics, lambdaExpInit, processPrecExpInit, trackOptimizationPath,
                logIterations, processQScale)


# ------------------------------------------------------------------------------------------------------------------
# expected transition residual sums
# ------------------------------------------------------------------------------------------------------------------
def _tsums(d, stateSmoothed, stateCovarSmoothed, lagCovSmoothed, matrixF):
    xs = _as64(stateSmoothed, "stateSmoothed", 2)
    Ps = _as64(stateCovarSmoothed, "stateCovarSmoothed", 3)
    lag = _as64(lagCovSmoothed, "lagCovSmoothed", 3)
    n = xs.shape[0]
    need = max(n - 1, 0)
    if xs.shape[1] != d:
        raise ValueError(f"stateSmoothed must have shape (n, {d})")
    if Ps.shape[0] != n or Ps.shape[1] != d or Ps.shape[2] != d:
        raise ValueError(f"stateCovarSmoothed must have shape (n, {d}, {d})")
    if lag.shape[0] < need or lag.shape[1] != d or lag.shape[2] != d:
        raise ValueError(f"lagCovSmoothed must have shape (n - 1, {d}, {d})")
    F = None
    if d == 2:
        F = _as64(matrixF, "matrixF", 2)
        if F.shape[0] != 2 or F.shape[1] != 2:
            raise ValueError("matrixF must have shape (2, 2)")
    if n - 1 <= 0:
        return 0.0, 0.0, 0
    xs, Ps, lag = (np.ascontiguousarray(a) for a in (xs, Ps, lag[:need]))
    Fc = np.ascontiguousarray(F) if F is not None else None
    sl, st, cnt = C.c_double(0.0), C.c_double(0.0), C.c_int64(0)
    L.check(L.lib().csr_expected_transition_residual_sums(d, n, L.dp(xs), L.dp(Ps), L.dp(lag), L.dp(Fc),
                                                          C.byref(sl), C.byref(st), C.byref(cnt)))
    return float(sl.value), float(st.value), int(cnt.value)


def _as64(arr, name, ndim):
    if not isinstance(arr, np.ndarray):
        raise TypeError(f"Argument '{name}' has incorrect type (expected numpy.ndarray, got {type(arr).__name__})")
    if arr.dtype != np.float64:
        raise ValueError(f"Buffer dtype mismatch for '{name}': expected float64, got {arr.dtype.name}")
    if arr.ndim != ndim:
        raise ValueError(f"Buffer has wrong number of dimensions (expected {ndim}, got {arr.ndim})")
    return arr


def cExpectedTransitionResidualSums(stateSmoothed, stateCovarSmoothed, lagCovSmoothed, matrixF):
    """Sum_k max(E[w w^T]_00, 0), Sum_k max(E[w w^T]_11, 0), n-1 (pyx:710-815) on the GPU."""
    return _tsums(2, stateSmoothed, stateCovarSmoothed, lagCovSmoothed, matrixF)


def cExpectedTransitionResidualSumsLevel(stateSmoothed, stateCovarSmoothed, lagCovSmoothed):
    """Scalar level-model variant (pyx:818-863) on the GPU."""
    return _tsums(1, stateSmoothed, stateCovarSmoothed, lagCovSmoothed, None)


# ------------------------------------------------------------------------------------------------------------------
# natives of the background update (pyx:944-1096, pyx:9700-9724)
# ------------------------------------------------------------------------------------------------------------------
def _bad_pivot(index, value):
    return RuntimeError("roughness-penalized LDL factorization required pivot "
                        f"modification at index {int(index)} (pivot={float(value):.6g}, floor={1.0e-12:.6g}).")


def solveBackgroundBatch(weightTracks, rhsTracks, lam, zeroCenter=True, lamFirst=0.0, blockLen=0):
    """Many independent chains (chromosomes) in ONE device pass: lists of float64 vectors -> list of solutions.
    Same system and error behaviour per chain as `csolveZeroCenteredBackground`."""
    ws = [np.ascontiguousarray(w, dtype=np.float64).reshape(-1) for w in weightTracks]
    rs = [np.ascontiguousarray(r, dtype=np.float64).reshape(-1) for r in rhsTracks]
    if len(ws) != len(rs) or any(w.shape != r.shape for w, r in zip(ws, rs)):
        raise ValueError("weightTrack and rhsTrack must have the same length")
    if not np.isfinite(lamFirst) or lamFirst < 0.0:
        raise ValueError("lamFirst must be finite and nonnegative")
    if not np.isfinite(lam) or lam < 0.0:
        raise ValueError("lam must be finite and nonnegative")
    outs = [np.zeros(w.shape[0]) for w in ws]
    live = [i for i, w in enumerate(ws) if w.shape[0] > 0]
    if not live:
        return outs
    L.require_gpu()
    n = np.asarray([ws[i].shape[0] for i in live], np.int64)
    wcat, rcat = np.concatenate([ws[i] for i in live]), np.concatenate([rs[i] for i in live])
    out = np.zeros(wcat.shape[0])
    bad_i, bad_v = np.full(len(live), -1, np.int64), np.zeros(len(live))
    L.check(L.lib().csr_solve_background(len(live), n.ctypes.data_as(L.I64P), L.dp(wcat), L.dp(rcat), float(lam),
                                         float(lamFirst), int(bool(zeroCenter)), int(blockLen), L.dp(out),
                                         bad_i.ctypes.data_as(L.I64P), L.dp(bad_v)))
    for k in range(len(live)):
        if bad_i[k] >= 0:
            raise _bad_pivot(bad_i[k], bad_v[k])
    pos = 0
    for k, i in enumerate(live):
        outs[i] = out[pos: pos + n[k]].copy()
        pos += int(n[k])
    return outs


def csolveZeroCenteredBackground(weightTrack, rhsTrack, lam, zeroCenter=True, lamFirst=0.0):
    """pyx:944-1096.  float64 vectors in, float64 solution out; RuntimeError when a pivot had to be raised to 1e-12."""
    w = np.asarray(weightTrack)
    r = np.asarray(rhsTrack)
    for a, name in ((w, "weightTrack"), (r, "rhsTrack")):
        if a.dtype != np.float64 or a.ndim != 1:
            raise ValueError(f"Buffer dtype mismatch or wrong number of dimensions for {name} (expected 1-D float64)")
    if r.shape[0] != w.shape[0]:
        raise ValueError("weightTrack and rhsTrack must have the same length")
    return solveBackgroundBatch([w], [r], lam, zeroCenter, lamFirst)[0]


def cbackgroundWeightedStatsWithSupport(residualMatrix, invVarMatrix):
    """pyx:9700-9724: (weightTrack, rhsTrack, supportCount)."""
    res = np.ascontiguousarray(residualMatrix, dtype=np.float32)
    inv = np.ascontiguousarray(invVarMatrix, dtype=np.float32)
    if res.ndim != 2 or inv.ndim != 2 or inv.shape[0] != res.shape[0] or inv.shape[1] != res.shape[1]:
        raise ValueError("residualMatrix and invVarMatrix must have identical 2D shapes")
    m, n = res.shape
    w, r = np.empty(n), np.empty(n)
    if n == 0 or m == 0:
        w[:] = 0.0
        r[:] = 0.0
        return w, r, 0
    L.require_gpu()
    sup = C.c_int64(0)
    L.check(L.lib().csr_background_weighted_stats(m, n, L.fp(res), L.fp(inv), L.dp(w), L.dp(r), C.byref(sup)))
    return w, r, int(sup.value)


# natives of the initial process-noise seed (pyx:1441-2146), implemented in consenrich_amd/qseed.py
from .qseed import (  # noqa: E402,F401
    cEstimatePooledProcessNoiseTransitions,
    cEstimateSameTrackProcessNoiseTransitions,
    cQSeedPosteriorFromTransitions,
)
