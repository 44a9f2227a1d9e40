"""CPU: the oracle (C restatement) against the golden vectors captured from the REAL reference, against independent
NumPy float64 restatements of the algorithm, and (build container only) live against the compiled reference."""
import os

import numpy as np
import pytest

import cases
from oracle import oracle as orc
from oracle import ref_loader

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CASES = {c["name"]: c for c in cases.all_cases()}


@pytest.fixture(scope="module", autouse=True)
def _build_oracle():
    orc.lib()


def test_every_case_has_a_fixture():
    import bg_cases
    import qseed_cases
    import unc_cases

    have = {f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz")}
    assert have == (set(CASES) | {c["name"] for c in bg_cases.solve_cases()} | {"bg_stats"}
                    | {c["name"] for c in unc_cases.cases()}
                    | {c["name"] for c in qseed_cases.native_cases() + qseed_cases.estimate_cases()})


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_reference_golden_vectors(name):
    """Bit-for-bit on stored float32 arrays; 1e-12 on fp64 scalars (the fixtures hold the reference's outputs)."""
    case = CASES[name]
    gold = np.load(os.path.join(GOLDEN, name + ".npz"))
    got = cases.run_case(orc, case)
    cases.compare(case, got, gold, rtol=1e-12, atol=0.0)
    for key in gold.files:                      # stored arrays: exact equality, sampled rows included
        if key.endswith("__rows"):
            base = key[: -len("__rows")]
            idx = cases.sample_index(got[base].shape[0])
            assert np.array_equal(got[base][idx], gold[key], equal_nan=True), key
        elif "__" not in key and isinstance(got[key], np.ndarray) and got[key].dtype == np.float32:
            assert np.array_equal(got[key], gold[key], equal_nan=True), key


@pytest.mark.skipif(not ref_loader.available(), reason="compiled reference only exists in the build container")
@pytest.mark.parametrize("name", ["fb_trend_n4096_m32", "fb_level_n333_m5_mask", "ecm_trend_n600_m6_converge",
                                  "ecm_level_n1000_m3_warm_qs", "fb_trend_n3000_m8_apn"])
def test_oracle_equals_live_reference(name):
    ref = ref_loader.load()
    a, b = cases.run_case(ref, CASES[name]), cases.run_case(orc, CASES[name])
    for k, v in a.items():
        if isinstance(v, np.ndarray):
            assert np.array_equal(v, b[k], equal_nan=True), k
        else:
            assert float(v) == pytest.approx(float(b[k]), rel=1e-12, abs=0.0), k


# ---- independent float64 restatements (the specifications the reference's own tests encode) -----------------------
def _level_kalman_f64(data, munc, q, x0, p0, pad):
    """Scalar level filter + RTS smoother in plain float64 (spec: test_core.py:522-592)."""
    z, v = np.asarray(data, np.float64), np.asarray(munc, np.float64)
    n = z.shape[1]
    w = 1.0 / np.maximum(v + pad, 1e-12)
    xf, pf = np.empty(n), np.empty(n)
    x, p = float(x0), float(p0)
    for k in range(n):
        p = p + q
        s0, s1 = w[:, k].sum(), (w[:, k] * (z[:, k] - x)).sum()
        x = x + p * s1 / (1.0 + p * s0)
        p = p / (1.0 + p * s0)
        xf[k], pf[k] = x, p
    xs, ps, lag = xf.copy(), pf.copy(), np.zeros(max(n - 1, 1))
    for k in range(n - 2, -1, -1):
        pp = max(pf[k] + q, 1e-12)
        j = pf[k] / pp
        xs[k] = xf[k] + j * (xs[k + 1] - xf[k])
        ps[k] = max(pf[k] + j * j * (ps[k + 1] - pp), 0.0)
        lag[k] = pf[k] + j * (ps[k + 1] - pp)
    return xf, pf, xs, ps, lag


def test_level_filter_smoother_matches_float64_specification():
    data = np.asarray(cases.LIT_A_DATA, np.float32)
    munc = np.asarray(cases.LIT_A_MUNC, np.float32)
    n = data.shape[1]
    q, x0, p0, pad = 0.06, -0.1, 0.8, 0.02
    xf, pf, pn = np.empty((n, 1), np.float32), np.empty((n, 1, 1), np.float32), np.empty((n, 1, 1), np.float32)
    r = orc.cforwardPassLevel(matrixData=data, matrixPluginMuncInit=munc,
                              matrixQ0=np.asarray([[q, 0.0], [0.0, 0.5]], np.float32),
                              intervalToBlockMap=np.zeros(n, np.int32), blockCount=1, stateInit=x0, stateCovarInit=p0,
                              pad=pad, stateForward=xf, stateCovarForward=pf, pNoiseForward=pn, returnNLL=True,
                              ECM_useObsPrecisionReweighting=False, ECM_useProcessPrecisionReweighting=False)
    assert np.isfinite(r[0]) and np.isfinite(r[3])
    xs, ps, lag, res = orc.cbackwardPassLevel(matrixData=data, stateForward=xf, stateCovarForward=pf, pNoiseForward=pn)
    exf, epf, exs, eps, elag = _level_kalman_f64(data, munc, np.float32(q), np.float32(x0), np.float32(p0), np.float32(pad))
    for got, exp in ((xf[:, 0], exf), (pf[:, 0, 0], epf), (xs[:, 0], exs), (ps[:, 0, 0], eps), (lag[: n - 1, 0, 0], elag[: n - 1])):
        np.testing.assert_allclose(got, exp, rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(res, data.T.astype(np.float64) - exs[:, None], rtol=2e-6, atol=2e-6)


def test_trend_with_identity_transition_equals_level_model():
    """levelTrend with F = I must reproduce the level model incl. lambda / kappa / qScale clamping and NLL-in-D
    (spec: test_core.py:3355-3461)."""
    lv = cases.run_case(orc, CASES["lit_levelB"])
    tr = cases.run_case(orc, CASES["lit_trendB_identityF"])
    assert float(tr["phi"]) == pytest.approx(float(lv["phi"]), rel=2e-6, abs=2e-6)
    assert float(tr["nll"]) == pytest.approx(float(lv["nll"]), rel=2e-6, abs=2e-6)
    n = lv["xf"].shape[0]
    for a, b in ((tr["D"], lv["D"]), (tr["xf"][:, :1], lv["xf"]), (tr["Pf"][:, :1, :1], lv["Pf"]),
                 (tr["pn"][:, :1, :1], lv["pn"]), (tr["xs"][:, :1], lv["xs"]), (tr["Ps"][:, :1, :1], lv["Ps"]),
                 (tr["lag"][: n - 1, :1, :1], lv["lag"][: n - 1]), (tr["resid"], lv["resid"])):
        np.testing.assert_allclose(a, b, rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("name", ["lit_ecm_trendC", "lit_ecm_levelC"])
def test_ecm_precision_updates_match_numpy_formulas(name):
    """lambda / kappa after one ECM inner iteration vs the closed forms (spec: test_core.py:2911-3079)."""
    case = CASES[name]
    out = cases.run_case(orc, case)
    inp = cases.inputs(case)
    d, nu, pad = case["d"], case["nu"], case["pad"]
    m, n = inp["data"].shape
    xs, Ps, lag = (np.asarray(out[k], np.float64) for k in ("xs", "Ps", "lag"))
    lam = []
    for k in range(n):
        p00 = max(Ps[k, 0, 0], 0.0)
        u2 = sum(((float(inp["data"][j, k]) - xs[k, 0]) ** 2 + p00) / (float(inp["munc"][j, k]) + float(np.float32(pad)))
                 for j in range(m))
        lam.append((nu + m) / (nu + u2))
    np.testing.assert_allclose(out["lam"], np.clip(lam, 0.1, 10.0), rtol=2e-6, atol=2e-6)
    F = np.asarray(case["F"], np.float32).astype(np.float64)[:d, :d] if d == 2 else np.ones((1, 1))
    Q = np.diag(np.asarray(case["Q0"], np.float32).astype(np.float64))[:d, :d]
    kap = [1.0]
    for k in range(n - 1):
        x, y = xs[k], xs[k + 1]
        exx, eyy = Ps[k] + np.outer(x, x), Ps[k + 1] + np.outer(y, y)
        exy = lag[k] + np.outer(x, y)
        ww = eyy - exy.T @ F.T - F @ exy + F @ exx @ F.T
        ww[np.diag_indices(d)] = np.maximum(np.diag(ww), 0.0)
        delta = float(np.sum(np.linalg.inv(Q) * ww.T))
        kap.append((nu + d) / (nu + max(delta, 0.0)))
    np.testing.assert_allclose(out["kap"], np.clip(kap, 0.1, 10.0), rtol=2e-6, atol=2e-6)


def test_transition_sums_match_matrix_formula():
    rng = np.random.default_rng(123)
    n = 19
    F = np.asarray([[1.0, 0.35], [0.02, 0.97]])
    xs = rng.normal(size=(n, 2))
    raw = rng.normal(scale=0.1, size=(n, 2, 2))
    Ps = raw + np.swapaxes(raw, 1, 2)
    lag = rng.normal(scale=0.05, size=(n - 1, 2, 2))
    eL = eT = 0.0
    for k in range(n - 1):
        exx0, exx1 = Ps[k] + np.outer(xs[k], xs[k]), Ps[k + 1] + np.outer(xs[k + 1], xs[k + 1])
        e01 = lag[k] + np.outer(xs[k], xs[k + 1])
        r = exx1 - e01.T @ F.T - F @ e01 + F @ exx0 @ F.T
        eL += max(r[0, 0], 0.0)
        eT += max(r[1, 1], 0.0)
    sL, sT, cnt = orc.cExpectedTransitionResidualSums(xs, Ps, lag, F)
    assert cnt == n - 1 and sL == pytest.approx(eL, rel=1e-12) and sT == pytest.approx(eT, rel=1e-12)
    s1 = orc.cExpectedTransitionResidualSumsLevel(xs[:, :1].copy(), Ps[:, :1, :1].copy(), lag[:, :1, :1].copy())
    e1 = sum(max(Ps[k + 1, 0, 0] + xs[k + 1, 0] ** 2 - 2 * (lag[k, 0, 0] + xs[k, 0] * xs[k + 1, 0]) + Ps[k, 0, 0] + xs[k, 0] ** 2, 0.0)
             for k in range(n - 1))
    assert s1[0] == pytest.approx(e1, rel=1e-12) and s1[1] == 0.0 and s1[2] == n - 1


def test_tiny_track_falls_back_to_filter_smoother_only():
    """n <= 5: iters 0, skipped diagnostics, finite outputs (spec: test_core.py:2718-2846)."""
    for name in ("ecm_trend_n5_tiny", "ecm_level_n5_tiny"):
        out = cases.run_case(orc, CASES[name])
        assert int(out["iters"]) == 0 and int(out["skipped"]) == 1 and np.isfinite(float(out["nll"]))
        for k in ("xs", "Ps", "lag", "resid", "lam", "kap"):
            assert np.all(np.isfinite(out[k])), k
