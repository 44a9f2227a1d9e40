"""Host mirror of the delete-block calibration natives: the reference's
``consenrich.cuncertainty.cobservationTotalInformation`` (/root/reference/src/consenrich/cuncertainty.pyx:97-157) and
``cmakeFoldMaskAndInformation`` (:160-305) with their positional interfaces, validation messages and return tuples, on
the GPU through ``csr_observation_total_information`` / ``csr_fold_mask_and_information`` (bit-identical tracks).
Drop-in: ``setattr(consenrich.cuncertainty, name, getattr(consenrich_amd.cuncertainty, name))``.  For device-resident
fits use ``DeviceBatch.make_fold`` instead (the folds of a chromosome become extra chains of the same batch).
No CPU fallback."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

U8P = C.POINTER(C.c_uint8)
I32P = C.POINTER(C.c_int32)


def _munc(matrixMunc):
    a = np.asarray(matrixMunc)
    if a.dtype == np.float64:
        return np.ascontiguousarray(a, np.float64), 1
    return np.ascontiguousarray(a, np.float32), 0


def _check_pad_rho(pad, rho):
    if not np.isfinite(pad):
        raise ValueError("observation information pad must be finite")
    if not np.isfinite(rho) or rho < 0.0 or rho >= 1.0:
        raise ValueError("replicate dependence rho must be in [0, 1)")


def cobservationTotalInformation(matrixMunc, activeMask, lambdaExp, useLambda, pad, replicateDependenceRho=0.0):
    munc, f64 = _munc(matrixMunc)
    act = np.ascontiguousarray(activeMask, np.uint8)
    lam = np.ascontiguousarray(lambdaExp, np.float64)
    if munc.ndim != 2 or act.shape != munc.shape:
        raise ValueError("activeMask must match matrixMunc shape")
    m, n = munc.shape
    if useLambda and lam.shape[0] != n:
        raise ValueError("fullObservationPrecision must match interval count")
    _check_pad_rho(pad, replicateDependenceRho)
    total = np.zeros(n)
    if m == 0 or n == 0:
        return total
    L.require_gpu()
    L.check(L.lib().csr_observation_total_information(m, n, munc.ctypes.data, f64, act.ctypes.data_as(U8P),
                                                      L.dp(lam) if useLambda else None, float(pad),
                                                      float(replicateDependenceRho), L.dp(total)))
    return total


def _check_fold_spec(m, n, blockLen, fold, bf, rc, rb):
    """cuncertainty.pyx:175-231"""
    if m < 1 or n < 1 or blockLen < 1:
        raise ValueError("invalid uncertainty calibration mask dimensions")
    if fold < 0:
        raise ValueError("fold must be nonnegative")
    bc = (n + blockLen - 1) // blockLen
    if bf.shape[0] != bc or rc.shape[0] != bc or rb.shape[0] != bc:
        raise ValueError("fold spec has inconsistent block count")
    slots = rb.shape[1]
    if slots < m:
        raise ValueError("fold spec replicate matrix must allow every sample")
    if np.any(bf < 0):
        raise ValueError("fold spec contains negative fold id")
    if np.any(rc < 1) or np.any(rc > m) or np.any(rc > slots):
        raise ValueError("fold spec deleted-replicate count is out of bounds")
    live = np.arange(slots)[None, :] < rc[:, None]
    vals = np.where(live, rb, -1)
    if np.any(live & ((rb < 0) | (rb >= m))):
        raise ValueError("fold spec replicate is out of bounds")
    srt = np.sort(np.where(live, vals, np.arange(slots)[None, :] + m), axis=1)       # padding made distinct
    if np.any(srt[:, 1:] == srt[:, :-1]):
        raise ValueError("fold spec contains a duplicate replicate")


def cmakeFoldMaskAndInformation(m, n, blockLen, fold, blockFold, repsByBlockCount, repsByBlock, matrixMunc, activeMask,
                                totalInfo, lambdaExp, useLambda, pad, replicateDependenceRho=0.0,
                                returnNominalHeldout=False):
    m, n, blockLen, fold = int(m), int(n), int(blockLen), int(fold)
    munc, f64 = _munc(matrixMunc)
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
    _check_pad_rho(pad, replicateDependenceRho)
    _check_fold_spec(m, n, blockLen, fold, bf, rc, rb)
    mask = np.ones((m, n), np.uint8)
    kept, held, h = np.empty(n), np.zeros(n), np.empty(n)
    nominal = np.zeros(n) if returnNominalHeldout else None
    L.require_gpu()
    L.check(L.lib().csr_fold_mask_and_information(
        m, n, blockLen, fold, bf.ctypes.data_as(I32P), rc.ctypes.data_as(L.I64P), rb.ctypes.data_as(L.I64P), rb.shape[1],
        munc.ctypes.data, f64, act.ctypes.data_as(U8P), L.dp(tot), L.dp(lam) if useLambda else None, float(pad),
        float(replicateDependenceRho), mask.ctypes.data_as(U8P), L.dp(kept), L.dp(held), L.dp(h), L.dp(nominal)))
    if returnNominalHeldout:
        return mask, kept, held, h, nominal
    return mask, kept, held, h
