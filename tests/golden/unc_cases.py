"""Case table for the delete-block calibration natives (cuncertainty.pyx:97-157, 160-305).
Inputs are re-synthesised from the seed (the fold spec included: cuncertainty.pyx:60-94 draws it from NumPy's default_rng,
restated in `fold_spec`); unc_*.npz hold the REAL reference's outputs."""
from __future__ import annotations

import numpy as np


def fold_spec(m, n, block_len, folds, p_delete, seed):
    """cuncertainty.pyx:60-94 `cmakeFoldSpec` (same generator calls in the same order)"""
    bc = (n + block_len - 1) // block_len
    rng = np.random.default_rng(int(seed))
    order = rng.permutation(bc).astype(np.int32, copy=False)
    block_fold = np.empty(bc, np.int32)
    block_fold[order] = np.arange(bc, dtype=np.int32) % int(folds)
    count = np.empty(bc, np.intp)
    reps = np.full((bc, m), -1, np.intp)
    for b in range(bc):
        k = int(rng.binomial(m, p_delete))
        while k < 1 or (m > 1 and k >= m):
            k = int(rng.binomial(m, p_delete))
        count[b] = k
        reps[b, :k] = rng.choice(m, size=k, replace=False)
    return block_fold, count, reps


def cases():
    cs = []
    for name, m, n, bl, folds, rho, use_lam, f64 in (
            ("unc_m4_n20", 4, 20, 5, 2, 0.0, False, False), ("unc_m6_n1003_rho", 6, 1003, 25, 3, 0.2, True, False),
            ("unc_m32_n5000", 32, 5000, 50, 2, 0.0, True, False), ("unc_m3_n257_f64", 3, 257, 16, 2, 0.1, False, True),
            ("unc_m1_n64", 1, 64, 8, 2, 0.0, False, False)):
        cs.append(dict(name=name, m=m, n=n, block_len=bl, folds=folds, rho=rho, use_lam=use_lam, f64=f64, seed=len(name) + n))
    return cs


def inputs(case):
    m, n = case["m"], case["n"]
    rng = np.random.default_rng(case["seed"])
    munc = np.abs(rng.normal(0.3, 0.1, (m, n))) + 0.01
    munc = munc.astype(np.float64 if case["f64"] else np.float32)
    act = (rng.random((m, n)) > 0.05).astype(np.uint8)
    lam = np.exp(rng.normal(0, 0.3, n))
    spec = fold_spec(m, n, case["block_len"], case["folds"], 0.5, case["seed"] + 1)
    return munc, act, lam, spec


def run(mod, case):
    munc, act, lam, (bf, rc, rb) = inputs(case)
    m, n = case["m"], case["n"]
    out = {}
    tot = np.asarray(mod.cobservationTotalInformation(munc, act, lam, case["use_lam"], 1.0e-4, case["rho"]))
    out["total"] = tot
    for fold in range(case["folds"]):
        r = mod.cmakeFoldMaskAndInformation(m, n, case["block_len"], fold, bf, rc, rb, munc, act, tot, lam, case["use_lam"],
                                            1.0e-4, case["rho"], True)
        for key, val in zip(("mask", "kept", "heldout", "h", "nominal"), r):
            out[f"f{fold}_{key}"] = np.asarray(val)
    return out
