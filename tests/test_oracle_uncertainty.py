"""CPU: the oracle's delete-block calibration natives (oracle/consenrich_oracle.c
`cor_total_information`, `cor_fold_mask_information`) against golden vectors captured from the compiled reference
(cuncertainty.pyx:97-157, 160-305) -- bit for bit -- and live against it where it exists; argument validation."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import unc_cases  # noqa: E402

from oracle import oracle as orc  # noqa: E402
from oracle import ref_loader  # noqa: E402

GOLDEN = os.path.join(HERE, "golden")
CASES = {c["name"]: c for c in unc_cases.cases()}


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_golden(name):
    got = unc_cases.run(orc, CASES[name])
    gold = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert set(got) == set(gold.files)
    for k in gold.files:
        assert np.array_equal(got[k], gold[k], equal_nan=True), k


@pytest.mark.skipif(not ref_loader.available(), reason="reference build only exists in the build container")
def test_oracle_matches_live_reference_and_fold_spec():
    ref_loader.load()
    from consenrich import cuncertainty as ref

    case = dict(name="live", m=5, n=777, block_len=20, folds=3, rho=0.15, use_lam=True, f64=False, seed=99)
    a, b = unc_cases.run(ref, case), unc_cases.run(orc, case)
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    spec = unc_cases.fold_spec(5, 777, 20, 3, 0.5, 100)
    assert all(np.array_equal(x, y) for x, y in zip(spec, ref.cmakeFoldSpec(5, 777, 20, 3, 0.5, 100)))


def test_validation_messages():
    case = CASES["unc_m4_n20"]
    munc, act, lam, (bf, rc, rb) = unc_cases.inputs(case)
    tot = orc.cobservationTotalInformation(munc, act, lam, False, 1e-4, 0.0)
    args = lambda **kw: dict(dict(m=4, n=20, blockLen=5, fold=0, blockFold=bf, repsByBlockCount=rc, repsByBlock=rb,     # noqa: E731
                                  matrixMunc=munc, activeMask=act, totalInfo=tot, lambdaExp=lam, useLambda=False, pad=1e-4), **kw)
    with pytest.raises(ValueError, match="fold must be nonnegative"):
        orc.cmakeFoldMaskAndInformation(**args(fold=-1))
    with pytest.raises(ValueError, match="inconsistent block count"):
        orc.cmakeFoldMaskAndInformation(**args(blockFold=bf[:-1]))
    bad = rb.copy(); bad[0, 1] = bad[0, 0]
    with pytest.raises(ValueError, match="duplicate replicate"):
        orc.cmakeFoldMaskAndInformation(**args(repsByBlock=bad))
    with pytest.raises(ValueError, match="rho must be in"):
        orc.cmakeFoldMaskAndInformation(**args(), replicateDependenceRho=1.0)
    with pytest.raises(ValueError, match="activeMask must match"):
        orc.cobservationTotalInformation(munc, act[:, :-1], lam, False, 1e-4)
