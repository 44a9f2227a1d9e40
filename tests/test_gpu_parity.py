"""GPU parity tests (run with -m gpu on an MI355X): the HIP product path, called through the C ABI via the
reference-shaped Python mirror, against (a) the golden vectors captured from the real reference and (b) the CPU
oracle run live on the same inputs.  Tolerance: north_star's 1e-5 relative (+ 2e-6 absolute, the tolerance of the
reference's own known-answer tests, test_core.py:3341-3350) on float32 tracks; discrete outputs (ECM iteration
count, convergence flag) must match exactly.
"""
import glob
import os

import numpy as np
import pytest

import cases
from conftest import gpu_available

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1.0e-5, 2.0e-6
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def product():
    if not gpu_available():
        pytest.fail("GPU tests selected but no HIP device / library: the product has no CPU fallback")
    from consenrich_amd import cconsenrich

    return cconsenrich


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as orc

    orc.lib()
    return orc


CASES = {c["name"]: c for c in cases.all_cases()}


@pytest.mark.parametrize("xtol", [2, 0], ids=["ulp2", "exact"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_product_matches_golden(product, name, xtol):
    """Both carry-validation modes (default: 2-ulp acceptance; 0: bit-exact sequential semantics) must meet parity."""
    case = CASES[name]
    gold = np.load(os.path.join(GOLDEN, name + ".npz"))
    product.set_validation(xtol)
    try:
        got = cases.run_case(product, case)
    finally:
        product.set_validation(2)
    cases.compare(case, got, gold, RTOL, ATOL)


@pytest.mark.parametrize("name", ["fb_trend_n4096_m32", "fb_level_n4096_m32", "ecm_trend_n4096_m8_defaults",
                                  "fb_trend_n20000_m8", "fb_trend_n333_m5_mask"])
def test_product_matches_live_oracle(product, oracle, name):
    case = CASES[name]
    got = cases.run_case(product, case)
    ref = cases.run_case(oracle, case)
    for k, v in ref.items():
        if isinstance(v, np.ndarray) and v.dtype.kind == "f":
            np.testing.assert_allclose(got[k].astype(np.float64), v.astype(np.float64), rtol=RTOL, atol=ATOL,
                                       err_msg=f"{name}:{k}")
        else:
            assert np.asarray(got[k]).item() == pytest.approx(np.asarray(v).item(), rel=RTOL, abs=ATOL), f"{name}:{k}"


def _run_batch(block_len, warm, d, n_list, m, seed, flags_extra=0, xtol=0):
    from consenrich_amd import _lib as L
    from consenrich_amd.batch import DeviceBatch, ModelParams

    mp = ModelParams(state_dim=d, Q0=((1e-3, 0.0), (0.0, 1e-4)) if d == 2 else ((1e-3, 0.0), (0.0, 0.0)))
    out = {}
    with DeviceBatch(0, block_len=block_len, warm=warm, x_tol_ulps=xtol) as b:
        b.configure(mp, m, n_list)
        for c, n in enumerate(n_list):
            data, munc = cases.synth(n, m, seed + c, mask_frac=0.02, outlier_frac=0.01)
            lam, kap, qs = cases.multipliers(n, seed + c)
            b.upload(c, data, munc)
            b.upload_multipliers(c, lam, kap, qs)
        b.stats()
        sd, sn = b.forward(L.RETURN_NLL | L.USE_LAMBDA | L.USE_KAPPA | L.USE_QSCALE | flags_extra)
        b.backward()
        b.export(L.EXPORT_FORWARD | L.EXPORT_SMOOTH | L.EXPORT_RESID)
        out["sd"], out["sn"] = sd, sn
        for c in range(len(n_list)):
            for name in ("D", "xf", "Pf", "pnoise", "xs", "Ps", "lag", "resid"):
                out[(c, name)] = b.download(c, name)
        out["stats"] = b.run_stats()
    return out


@pytest.mark.parametrize("d", [2, 1])
def test_speculative_blocks_equal_sequential_recursion(product, d):
    """Results must not depend on block length / warm-up: tiny blocks with (deliberately insufficient) warm-up,
    repaired by the validation/fix-up pass, must reproduce the single-block sequential run.  levelTrend: bit for bit;
    level (double carries, tolerance-validated): to 1e-6 relative."""
    n_list = [5000, 37, 1, 12345, 64, 65]
    seq = _run_batch(32 * 512, (0, 0, 0), d, n_list, 4, 100)       # every chain is one block: pure sequential
    for blk, warm in ((32, (1, 1, 1)), (64, (2, 4, 2)), (256, (2, 8, 4)), (32, (0, 0, 0))):
        spec = _run_batch(blk, warm, d, n_list, 4, 100)
        for key, val in seq.items():
            if key == "stats":
                continue
            if d == 2 and not isinstance(key, str):
                assert np.array_equal(val, spec[key]), f"block={blk} warm={warm} {key}"
            else:
                np.testing.assert_allclose(spec[key], val, rtol=1e-6, atol=1e-7, err_msg=f"{blk} {warm} {key}")
        if warm == (0, 0, 0):
            assert spec["stats"]["reruns_p"] > 0 and spec["stats"]["reruns_b"] > 0   # the fix-up path really ran


def test_ulp_tolerant_validation_stays_within_parity_budget(product):
    """Default mode: speculative carries are accepted within 2 float32 ulps.  Against the exact sequential run the
    tracks must agree far inside the 1e-5 budget (a few ulps on the level; the trend inherits ulp(level)-sized noise,
    covered by the absolute tolerance), with large |x| (coarse ulps) to make the test bite."""
    n_list = [60000, 7000]
    seq = _run_batch(32 * 2048, (0, 0, 0), 2, n_list, 8, 300, xtol=0)
    tol = _run_batch(64, (1, 2, 1), 2, n_list, 8, 300, xtol=2)
    exact = _run_batch(64, (1, 2, 1), 2, n_list, 8, 300, xtol=0)
    assert tol["stats"]["reruns_x"] < exact["stats"]["reruns_x"]
    for key, val in seq.items():
        if key == "stats":
            continue
        if not isinstance(key, str):
            assert np.array_equal(val, exact[key]), key
        scale = 1.0 if isinstance(key, str) else float(np.abs(val).max())
        np.testing.assert_allclose(tol[key], val, rtol=2e-6, atol=2e-6 * max(scale, 1e-30) if isinstance(key, str)
                                   else 1e-6 * scale, err_msg=str(key))


def test_batch_chains_are_independent(product):
    """A chain's result must not depend on which other chains share the batch (contig sharding relies on it)."""
    a = _run_batch(64, (2, 4, 2), 2, [3000, 777], 3, 7)
    b = _run_batch(64, (2, 4, 2), 2, [3000], 3, 7)
    for name in ("D", "xf", "Pf", "xs", "Ps", "lag", "resid"):
        assert np.array_equal(a[(0, name)], b[(0, name)]), name


def test_reference_api_contract(product):
    """Contract details the reference tests pin (test_core.py:3427, 3078-3079, 2862-2895)."""
    n, m = 40, 2
    data, munc = cases.synth(n, m, 5)
    F = np.asarray(cases.F_TREND, np.float32)
    Q0 = np.diag([1e-3, 1e-4]).astype(np.float32)
    bm = np.zeros(n, np.int32)
    D = np.empty(n, np.float32)
    r = product.cforwardPass(matrixData=data, matrixPluginMuncInit=munc, matrixF=F, matrixQ0=Q0, intervalToBlockMap=bm,
                             blockCount=1, stateInit=0.0, stateCovarInit=1000.0, vectorD=D, returnNLL=True)
    assert r[1] == 0 and r[2] is D and np.isfinite(r[3])
    with pytest.raises(ValueError):     # lambdaExp must be interval-level (1-D)
        product.cforwardPass(matrixData=data, matrixPluginMuncInit=munc, matrixF=F, matrixQ0=Q0,
                             intervalToBlockMap=bm, blockCount=1, stateInit=0.0, stateCovarInit=1000.0,
                             lambdaExp=np.ones((m, n), np.float32), ECM_useObsPrecisionReweighting=True)
    with pytest.raises(ValueError, match="out-of-range block id"):
        product.cforwardPass(matrixData=data, matrixPluginMuncInit=munc, matrixF=F, matrixQ0=Q0,
                             intervalToBlockMap=bm + 3, blockCount=1, stateInit=0.0, stateCovarInit=1000.0)
    with pytest.raises(ValueError, match="blockCount must be positive"):
        product.cforwardPass(matrixData=data, matrixPluginMuncInit=munc, matrixF=F, matrixQ0=Q0,
                             intervalToBlockMap=bm, blockCount=0, stateInit=0.0, stateCovarInit=1000.0)
    out = product.cfixedBackgroundECM(matrixData=data, matrixPluginMuncInit=munc, matrixF=F, matrixQ0=Q0,
                                      intervalToBlockMap=bm, blockCount=1, stateInit=0.0, stateCovarInit=1.0,
                                      ECM_fixedBackgroundIters=1, ECM_useObsPrecisionReweighting=False,
                                      ECM_useProcessPrecisionReweighting=False, returnIntermediates=True,
                                      t_innerIters=1, logIterations=False)
    assert out[6] is None and out[7] is None and len(out) == 8
    e = product.cforwardPass(matrixData=np.empty((m, 0), np.float32), matrixPluginMuncInit=np.empty((m, 0), np.float32),
                             matrixF=F, matrixQ0=Q0, intervalToBlockMap=np.empty(0, np.int32), blockCount=1,
                             stateInit=0.0, stateCovarInit=1.0, returnNLL=True)
    assert e[0] == 0.0 and e[1] == 0 and e[3] == 0.0


def test_ecm_is_silent_unless_logging(product, capfd):
    n, m = 64, 2
    data, munc = cases.synth(n, m, 9)
    product.cfixedBackgroundECM(matrixData=data, matrixPluginMuncInit=munc, matrixF=np.asarray(cases.F_TREND, np.float32),
                                matrixQ0=np.diag([1e-3, 1e-4]).astype(np.float32),
                                intervalToBlockMap=np.zeros(n, np.int32), blockCount=1, stateInit=0.0,
                                stateCovarInit=1.0, ECM_fixedBackgroundIters=2, logIterations=False)
    cap = capfd.readouterr()
    assert cap.out == "" and cap.err == ""
