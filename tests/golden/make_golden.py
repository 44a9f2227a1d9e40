#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference hot path (oracle/_ref, built by `make -C oracle ref`).

Runs only in the build container (needs /root/reference + oracle/_ref).  The committed .npz files hold DATA only:
case parameters (seeds / shapes / flags, or small literal matrices) and the reference's outputs.  Inputs are
re-synthesised from the seed by tests/golden/cases.py, which both this script and the tests import.

Usage:  python tests/golden/make_golden.py          (rewrites every fixture, then cross-checks the C oracle bit-for-bit)
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import bg_cases  # noqa: E402
import unc_cases  # noqa: E402
import qseed_cases  # noqa: E402
import cases  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from oracle import qseed as orc_qseed  # noqa: E402
from oracle import ref_loader  # noqa: E402


def main() -> int:
    ref = ref_loader.load()
    if ref is None:
        print("reference build not available (make -C oracle ref)", file=sys.stderr)
        return 2
    n_written = 0
    for case in cases.all_cases():
        out_ref = cases.run_case(ref, case)
        out_orc = cases.run_case(orc, case)
        # the oracle restatement must reproduce the reference bit-for-bit on stored arrays
        for key, val in out_ref.items():
            other = out_orc[key]
            if isinstance(val, np.ndarray) and val.dtype.kind == "f":
                same = np.array_equal(val, other, equal_nan=True)
                if not same:
                    err = float(np.nanmax(np.abs(val.astype(np.float64) - other.astype(np.float64))))
                    tol = 1e-12 * max(1.0, float(np.nanmax(np.abs(val))))
                    if err > tol:
                        raise SystemExit(f"oracle != reference for case {case['name']} key {key}: max|d|={err}")
            else:
                if not np.array_equal(np.asarray(val), np.asarray(other)):
                    raise SystemExit(f"oracle != reference for case {case['name']} key {key}: {val} vs {other}")
        path = os.path.join(HERE, case["name"] + ".npz")
        np.savez_compressed(path, **cases.compact(case, out_ref))
        n_written += 1
        print(f"wrote {os.path.relpath(path, ROOT)}  ({os.path.getsize(path)} B)")
    # background-update natives (same rule: the oracle must equal the reference bit for bit)
    for case in bg_cases.solve_cases():
        out_ref = bg_cases.run_solve(ref, case)
        out_orc = bg_cases.run_solve(orc, case)
        for key, val in out_ref.items():
            if not np.array_equal(val, out_orc[key]):
                raise SystemExit(f"oracle != reference for case {case['name']} key {key}")
        path = os.path.join(HERE, case["name"] + ".npz")
        np.savez_compressed(path, **out_ref)
        n_written += 1
        print(f"wrote {os.path.relpath(path, ROOT)}  ({os.path.getsize(path)} B)")
    res, inv = bg_cases.stats_inputs()
    w_ref, r_ref, s_ref = ref.cbackgroundWeightedStatsWithSupport(res, inv)
    w_orc, r_orc, s_orc = orc.cbackgroundWeightedStatsWithSupport(res, inv)
    if not (np.array_equal(w_ref, w_orc) and np.array_equal(r_ref, r_orc) and s_ref == s_orc):
        raise SystemExit("oracle != reference for cbackgroundWeightedStatsWithSupport")
    np.savez_compressed(os.path.join(HERE, "bg_stats.npz"), weight=w_ref, rhs=r_ref, support=np.int64(s_ref))
    n_written += 1
    # delete-block calibration natives (cuncertainty.pyx)
    from consenrich import cuncertainty as ref_unc  # noqa: E402  (compiled by `make -C oracle ref`)

    for case in unc_cases.cases():
        munc, act, lam, spec = unc_cases.inputs(case)
        ref_spec = ref_unc.cmakeFoldSpec(case["m"], case["n"], case["block_len"], case["folds"], 0.5, case["seed"] + 1)
        if not all(np.array_equal(a, b) for a, b in zip(spec, ref_spec)):
            raise SystemExit(f"fold spec restatement != reference for {case['name']}")
        out_ref, out_orc = unc_cases.run(ref_unc, case), unc_cases.run(orc, case)
        for key, val in out_ref.items():
            if not np.array_equal(val, out_orc[key], equal_nan=True):
                raise SystemExit(f"oracle != reference for case {case['name']} key {key}")
        np.savez_compressed(os.path.join(HERE, case["name"] + ".npz"), **out_ref)
        n_written += 1
    # Q0 seed natives (pyx:1441-2146) and the caller's composition run on the reference's natives
    import functools

    for case in qseed_cases.native_cases():
        out_ref, out_orc = qseed_cases.run_native(ref, case), qseed_cases.run_native(orc_qseed, case)
        qseed_cases.same(out_ref, out_orc)
        np.savez_compressed(os.path.join(HERE, case["name"] + ".npz"), **out_ref)
        n_written += 1
    for case in qseed_cases.estimate_cases():
        out_ref = qseed_cases.run_estimate(functools.partial(orc_qseed.estimate_initial_process_noise, ref), case)
        out_orc = qseed_cases.run_estimate(functools.partial(orc_qseed.estimate_initial_process_noise, orc_qseed), case)
        qseed_cases.same(out_ref, out_orc)
        np.savez_compressed(os.path.join(HERE, case["name"] + ".npz"), **out_ref)
        n_written += 1
    print(f"{n_written} fixtures written; oracle == reference on all of them")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
