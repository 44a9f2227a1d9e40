"""CPU tests pinning the oracle's background-update natives (oracle/consenrich_oracle.c
`cor_solve_background`, `cor_background_stats`) to the REAL reference: committed golden vectors (bit for bit,
including the pivot-modification error text), the live reference build when present, the reference's own literal test
values (tests/test_core.py:95-110) and an independent dense specification."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import bg_cases  # noqa: E402

from oracle import oracle as orc  # noqa: E402
from oracle import ref_loader  # noqa: E402

GOLDEN = os.path.join(HERE, "golden")
CASES = {c["name"]: c for c in bg_cases.solve_cases()}


def dense_system(w, lam_first, lam):
    """diag(w) + lam_first D1'D1 + lam D2'D2 from explicit difference operators"""
    n = len(w)
    A = np.diag(np.asarray(w, np.float64))
    if n >= 2:
        D1 = np.zeros((n - 1, n))
        D1[np.arange(n - 1), np.arange(n - 1)] = -1.0
        D1[np.arange(n - 1), np.arange(1, n)] = 1.0
        A = A + lam_first * D1.T @ D1
    if n >= 3:
        D2 = np.zeros((n - 2, n))
        for i in range(n - 2):
            D2[i, i:i + 3] = [1.0, -2.0, 1.0]
        A = A + lam * D2.T @ D2
    return A


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_golden(name):
    got = bg_cases.run_solve(orc, CASES[name])
    gold = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert set(got) == set(gold.files)
    for k in gold.files:
        if k.endswith("_error"):
            assert str(got[k]) == str(gold[k])
        else:
            assert np.array_equal(got[k], gold[k]), k


def test_oracle_stats_match_golden():
    res, inv = bg_cases.stats_inputs()
    w, r, s = orc.cbackgroundWeightedStatsWithSupport(res, inv)
    gold = np.load(os.path.join(GOLDEN, "bg_stats.npz"))
    assert np.array_equal(w, gold["weight"]) and np.array_equal(r, gold["rhs"]) and s == int(gold["support"])
    # the reference test's own expectation (tests/test_core.py:2527-2541)
    np.testing.assert_allclose(w, np.sum(inv.astype(np.float64), axis=0), rtol=1e-7)
    np.testing.assert_allclose(r, np.sum(inv.astype(np.float64) * res.astype(np.float64), axis=0), rtol=1e-7)


@pytest.mark.skipif(not ref_loader.available(), reason="reference build only exists in the build container")
def test_oracle_matches_live_reference():
    ref = ref_loader.load()
    rng = np.random.default_rng(3)
    for n in (6, 131, 4099):
        w = np.abs(rng.normal(50, 20, n))
        r = rng.normal(0, 10, n)
        for lam_first, lam in ((0.0, 0.0), (3.0, 0.0), (0.0, 40.0), (1e3, 1e7)):
            for zc in (False, True):
                assert np.array_equal(ref.csolveZeroCenteredBackground(w, r, lam, zc, lamFirst=lam_first),
                                      orc.csolveZeroCenteredBackground(w, r, lam, zc, lamFirst=lam_first))


def test_reference_literals_and_dense_specification():
    with pytest.raises(RuntimeError, match="required pivot modification"):        # tests/test_core.py:94-101
        orc.csolveZeroCenteredBackground(np.zeros(3), np.zeros(3), 0.0, False, lamFirst=0.0)
    one = orc.csolveZeroCenteredBackground(np.asarray([2.0]), np.asarray([8.0]), 9.0, False, lamFirst=6.0)
    np.testing.assert_allclose(one, [4.0])                                          # tests/test_core.py:103-110
    w, r = np.asarray(bg_cases.LIT_W), np.asarray(bg_cases.LIT_R)
    lam_first, lam = bg_cases.penalties(3, 2.0)
    assert (lam_first, lam) == (pytest.approx(2.0 * 9 / 4), pytest.approx(2.0 * 81 / 16))
    A = dense_system(w, lam_first, lam)
    np.testing.assert_allclose(orc.csolveZeroCenteredBackground(w, r, lam, False, lamFirst=lam_first),
                               np.linalg.solve(A, r), rtol=1e-11)
    # zero-sum variant: KKT system [A 1; 1' 0]
    n = len(w)
    K = np.zeros((n + 1, n + 1))
    K[:n, :n] = A
    K[:n, n] = 1.0
    K[n, :n] = 1.0
    xz = np.linalg.solve(K, np.concatenate([r, [0.0]]))[:n]
    np.testing.assert_allclose(orc.csolveZeroCenteredBackground(w, r, lam, True, lamFirst=lam_first), xz, rtol=1e-10,
                               atol=1e-13)
    for bad in (dict(lam=-1.0), dict(lamFirst=np.inf)):
        with pytest.raises(ValueError):
            orc.csolveZeroCenteredBackground(w, r, bad.get("lam", 1.0), False, lamFirst=bad.get("lamFirst", 0.0))


# ---- the Python control flow around the natives (core.py:8085-8378), expectations of the reference's own tests ---------
def test_background_update_reference_expectations():
    from oracle import background as bgo

    # tests/test_core.py:58-92: one track, 64 bins, residual 1, invVar 8.53348
    res = np.ones((1, 64), np.float32)
    inv = np.full((1, 64), 8.53348, np.float32)
    w, r, _ = orc.cbackgroundWeightedStatsWithSupport(res, inv)
    healthy = bgo.solve_background(w, r, 401, 64.0, zero_center=False, use_nonnegative=False)
    np.testing.assert_allclose(healthy, 1.0, rtol=1.0e-3)
    warned, info = bgo.solve_background(w, r, 1771, 64.0, zero_center=False, use_nonnegative=False, return_info=True)
    assert np.isfinite(warned).all() and info["roundoff_index"] >= 1.0e-2          # the reference logs a warning here
    with pytest.raises(RuntimeError, match="exceeds float64 reliability"):
        bgo.solve_background(w, r, 6427, 128.0, zero_center=False, use_nonnegative=False)
    # tests/test_core.py:2542-2581: re-using the plain solution as initialBackground reproduces it; zero weights -> zeros
    n = 48
    x = np.linspace(-1.0, 1.0, n, dtype=np.float32)
    res = np.vstack([-0.15 + 0.9 * np.exp(-((x - 0.2) ** 2) / 0.05), -0.10 + 0.7 * np.exp(-((x + 0.25) ** 2) / 0.08),
                     0.05 * np.sin(np.linspace(0.0, 3.0 * np.pi, n, dtype=np.float32))]).astype(np.float32)
    inv = np.vstack([np.linspace(0.7, 1.8, n, dtype=np.float32), np.linspace(1.4, 0.6, n, dtype=np.float32),
                     np.full(n, 1.1, dtype=np.float32)])
    w, r, sup = orc.cbackgroundWeightedStatsWithSupport(res, inv)
    assert sup == n
    for nonneg in (False, True):
        plain = bgo.solve_background(w, r, 5, 0.8, zero_center=False, use_nonnegative=nonneg)
        reused = bgo.solve_background(w, r, 5, 0.8, zero_center=False, use_nonnegative=nonneg, initial=plain)
        np.testing.assert_allclose(reused, plain, atol=1.0e-5)
        assert plain.dtype == np.float32
    assert not bgo.solve_background(np.zeros(n), np.zeros(n), 5, 0.8).any()
    assert bgo.penalties(12, 0.5) == (pytest.approx(0.5 * 144 / 4.0), pytest.approx(0.5 * 12.0 ** 4 / 16.0))   # :2703-2714


def test_background_irls_penalises_negative_parts():
    from oracle import background as bgo

    rng = np.random.default_rng(8)
    n = 4000
    truth = 0.25 * np.sin(np.arange(n) / 300.0) - 0.05
    w = 100.0 * np.exp(rng.normal(0, 0.2, n))
    r = w * (truth + rng.normal(0, 0.05, n))
    plain = bgo.solve_background(w, r, 40, 2.0, use_nonnegative=False)
    soft, info = bgo.solve_background(w, r, 40, 2.0, use_nonnegative=True, multiplier=4.0, return_info=True)
    assert info["passes"] >= 1 and (plain < 0).any()
    assert np.minimum(soft, 0).sum() > np.minimum(plain, 0).sum()          # negative mass shrinks, softly


def test_penalized_objective_known_answers():
    """oracle/driver.py `penalized_objective` (core.py:4418-4538, 3161-3204, 2981-2986) on inputs whose terms are known
    in closed form"""
    from oracle import driver as odrv

    n, m = 11, 3
    munc = np.full((m, n), 0.25, np.float32)
    munc[1, 4] = np.float32(1.0e30)                       # masked cell: not an effective observation
    bg = (0.5 * np.arange(n) - 1.0).astype(np.float32)    # linear: second differences vanish; two negative bins (-1, -0.5)
    cfg = dict(nu=8.0, use_lambda=True, use_kappa=True, penalties=(3.0, 7.0), neg_multiplier=2.0, use_nonnegative=True,
               lambda_bounds=(0.25, 4.0), pad=1.0e-4)
    lam = np.ones(n, np.float32)
    kap = np.full(n, np.float32(np.e))
    o = odrv.penalized_objective(100.0, munc, bg, lam, kap, cfg)
    assert o["effective_observation_count"] == m * n - 1
    assert o["robust_observation_penalty"] == pytest.approx(0.5 * 8.0 * n * 1.0)                 # x - log x at x = 1
    assert o["robust_process_penalty"] == pytest.approx(0.5 * 8.0 * (n - 1) * (np.e - 1.0), rel=1e-6)   # first bin skipped
    assert o["background_first_difference_penalty"] == pytest.approx(0.5 * 3.0 * (n - 1) * 0.25)
    assert o["background_second_difference_penalty"] == pytest.approx(0.0, abs=1e-12)
    w_full = 3.0 / (0.25 + 1.0e-4)                        # median bin: all three tracks unmasked
    assert o["background_negative_penalty"] == pytest.approx(0.5 * 2.0 * w_full * (1.0 + 0.25), rel=1e-6)
    total = 100.0 + sum(o[k] for k in ("robust_observation_penalty", "robust_process_penalty",
                                       "background_first_difference_penalty", "background_second_difference_penalty",
                                       "background_negative_penalty"))
    assert o["penalized_objective"] == pytest.approx(total) and o["penalized_objective_per_cell"] == pytest.approx(total / 32)
    cfg["use_nonnegative"] = False
    cfg["use_lambda"] = False
    o2 = odrv.penalized_objective(100.0, munc, bg, lam, None, cfg)
    assert o2["background_negative_penalty"] == 0.0 and o2["robust_observation_penalty"] == 0.0 and o2["robust_process_penalty"] == 0.0
