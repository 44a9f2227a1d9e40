"""GPU parity tests of the initial process-noise (Q0) seed, through the C ABI:
the reference-shaped natives and the batch estimator against (a) the golden vectors captured from the compiled
reference and (b) the CPU oracle run live.  Transitions are integer-indexed gathers + fp64 arithmetic in the reference's
order: bit-for-bit.  Posterior summaries pass through libm log / exp / log1p / lgamma on the host: 1e-12."""
import functools
import math
import os

import numpy as np
import pytest

import qseed_cases as qc
from conftest import gpu_available

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def product():
    if not gpu_available():
        pytest.fail("GPU tests selected but no HIP device / library: the product has no CPU fallback")
    from consenrich_amd import cconsenrich

    return cconsenrich


def _gold(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("case", qc.native_cases(), ids=lambda c: c["name"])
def test_natives_match_golden(product, case):
    got = qc.run_native(product, case)
    qc.same(got, _gold(case["name"]), rtol=1e-12 if case["kind"] == "post" else 0.0)


@pytest.mark.parametrize("case", qc.estimate_cases(), ids=lambda c: c["name"])
def test_estimator_matches_golden(product, case):
    from consenrich_amd import qseed

    got = qc.run_estimate(qseed.estimate_initial_process_noise, case)
    gold = _gold(case["name"])
    assert np.array_equal(got["Q"], gold["Q"])
    qc.same(got, gold, rtol=1e-12)


def test_batch_of_uneven_chains_matches_live_oracle(product):
    """several chains in one batch (capped and uncapped scans, a one-bin chain, masked cells, NaN data) against the
    oracle's restatement of the caller, chain by chain"""
    from consenrich_amd.batch import DeviceBatch, ModelParams
    from oracle import qseed as oq

    rng = np.random.default_rng(77)
    m, lens = 12, [150_000, 1, 40_000, 7, 31_999, 2]
    mats = []
    for n in lens:
        x = np.cumsum(rng.normal(0, 0.07, n))
        data = (x[None, :] + rng.normal(0, 0.1, (m, n))).astype(np.float32)
        munc = (0.01 * np.exp(rng.normal(0, 0.6, (m, n)))).astype(np.float32)
        munc[rng.random((m, n)) < 0.15] = np.float32(1.0e30)
        data[rng.random((m, n)) < 0.02] = np.nan
        mats.append((data, munc))
    kw = dict(pad=1.0e-4, stateModel="levelTrend", minQ=1.0e-6, maxQ=1000.0, deltaF=0.5, robustTNu=8.0)
    with DeviceBatch() as b:
        b.configure(ModelParams(), m, lens)
        for i, (d, v) in enumerate(mats):
            b.upload(i, d, v)
        got = b.qseed(**kw)
        again = b.qseed(**kw)
    for i, (d, v) in enumerate(mats):
        Q, diag = oq.estimate_initial_process_noise(oq, matrixData=d, matrixMunc=v, **kw)
        assert np.array_equal(got[i][0], Q), (i, got[i][0], Q)
        assert set(got[i][1]) == set(diag)
        for k, val in diag.items():
            g = got[i][1][k]
            if isinstance(val, float):
                assert (math.isnan(val) and math.isnan(g)) or g == pytest.approx(val, rel=1e-12, abs=0.0), (i, k, g, val)
            else:
                assert g == val, (i, k, g, val)
        assert again[i][1] == got[i][1] or all(
            (a == b_) or (isinstance(a, float) and math.isnan(a) and math.isnan(b_))
            for a, b_ in zip(again[i][1].values(), got[i][1].values()))
    assert got[0][1]["qSeedSource"] == "sameTrackEB" and got[0][1]["qSeedSelectedTransitionCount"] == 2048
    assert got[1][1]["qSeedSource"] in ("observationVarianceFloor", "minQ")


def test_large_uncapped_native_matches_live_oracle(product):
    """maxTransitionSamples = 0 on a long matrix: every transition is scanned (prefix over 2*10^5 counts, 10^6 pairs)"""
    from oracle import qseed as oq

    rng = np.random.default_rng(5)
    m, n = 6, 200_001
    data = np.cumsum(rng.normal(0, 0.05, n))[None, :] + rng.normal(0, 0.2, (m, n))
    obs = 0.04 * np.exp(rng.normal(0, 0.5, (m, n)))
    act = rng.random((m, n)) > 0.3
    args = (0.9, 3.0, 0, 32000, 4096)
    a = product.cEstimateSameTrackProcessNoiseTransitions(data, obs, act, *args)
    b = oq.cEstimateSameTrackProcessNoiseTransitions(data, obs, act, *args)
    for x, y in zip(a[:3], b[:3]):
        assert np.array_equal(x, y)
    assert a[3] == b[3]
    pa = product.cEstimatePooledProcessNoiseTransitions(data, obs, act)
    pb = oq.cEstimatePooledProcessNoiseTransitions(data, obs, act)
    for x, y in zip(pa, pb):
        assert np.array_equal(x, y)


def test_error_texts(product):
    one = np.ones(8)
    tail = (1.0e-5, 8, math.log(4.0), 8.0, 64)
    with pytest.raises(ValueError, match="samplingVariances must be nonnegative finite"):
        product.cQSeedPosteriorFromTransitions(one, -one, one, 1.0e-5, 1.0, 8.0, "bad", *tail)
    with pytest.raises(ValueError, match="transition arrays must have the same length"):
        product.cQSeedPosteriorFromTransitions(one, one[:3], one, 1.0e-5, 1.0, 8.0, "bad", *tail)
    data, act = np.zeros((2, 9)), np.ones((2, 9), bool)
    with pytest.raises(ValueError, match="active obsVar values must be positive finite"):
        product.cEstimateSameTrackProcessNoiseTransitions(data, np.zeros((2, 9)), act, 0.95, 20.0)
    bad = data.copy()
    bad[1, 4] = np.nan
    with pytest.raises(ValueError, match="active matrixData values must be finite"):
        product.cEstimateSameTrackProcessNoiseTransitions(bad, np.ones((2, 9)), act, 0.95, 20.0)
    with pytest.raises(ValueError, match="active pooled observations"):
        product.cEstimatePooledProcessNoiseTransitions(bad, np.ones((2, 9)), act)
    with pytest.raises(ValueError, match="precisionSampleCap must be positive"):
        product.cEstimateSameTrackProcessNoiseTransitions(data, np.ones((2, 9)), act, 0.95, 20.0, 4, 0)
    with pytest.raises(ValueError, match=r"precisionCapQuantile must be in \[0, 1\]"):
        product.cEstimateSameTrackProcessNoiseTransitions(data, np.ones((2, 9)), act, -0.1, 20.0)
    e = product.cEstimateSameTrackProcessNoiseTransitions(np.zeros((2, 1)), np.ones((2, 1)), np.ones((2, 1), bool), 0.95, 20.0)
    assert e[0].size == 0 and e[3]["pairCount"] == 0 and math.isnan(e[3]["precisionCap"])
    from consenrich_amd import qseed

    with pytest.raises(ValueError, match="must not exceed"):
        qseed.estimate_initial_process_noise(matrixData=np.zeros((2, 9), np.float32), matrixMunc=np.ones((2, 9), np.float32),
                                             pad=1e-4, stateModel="level", minQ=1e-6, maxQ=1e-5, deltaF=1.0, robustTNu=8.0,
                                             qSeedPriorLevel=1e-3)


def test_reference_known_answers_for_the_caller(product):
    """tests/test_core.py:3462-3541 on the product"""
    from consenrich_amd import qseed

    rng = np.random.default_rng(2024)
    n, m, q_true, obs = 160, 4, 1.0e-2, 2.0e-3
    latent = np.cumsum(rng.normal(0.0, np.sqrt(q_true), size=n))
    data = np.vstack([latent + rng.normal(0.0, np.sqrt(obs), size=n) for _ in range(m)]).astype(np.float32)
    Q, d = qseed.estimate_initial_process_noise(matrixData=data, matrixMunc=np.full((m, n), obs, np.float32), pad=1.0e-4,
                                                stateModel="levelTrend", minQ=1.0e-5, maxQ=1.0, deltaF=1.0, robustTNu=8.0)
    assert d["qSeedSource"] == "sameTrackEB" and 0.3 * q_true <= d["qSeedLevelFinal"] <= 3.0 * q_true
    assert Q[0, 0] == pytest.approx(d["qSeedLevelFinal"]) and Q[1, 1] == pytest.approx(d["qSeedLevelFinal"])
    data = np.full((2, 20), np.nan, np.float32)
    data[0, ::2] = 0.0
    data[1, 1::2] = 1.0
    Q, d = qseed.estimate_initial_process_noise(matrixData=data, matrixMunc=np.full((2, 20), 0.1, np.float32), pad=1.0e-4,
                                                stateModel="level", minQ=1.0e-4, maxQ=1.0, deltaF=1.0, robustTNu=8.0)
    assert d["qSeedSource"] == "pooledEB" and d["qSeedTransitionCount"] == 19 and Q[0, 0] > 1.0e-4


@pytest.mark.parametrize("m", [43, 70])
def test_many_tracks_use_the_global_work_area(product, m):
    """m > 42: the per-thread columns no longer fit the workgroup's LDS, the kernel instantiation with a global work area
    runs instead; same bits"""
    from oracle import qseed as oq

    rng = np.random.default_rng(m)
    n = 700
    data = np.cumsum(rng.normal(0, 0.05, n))[None, :] + rng.normal(0, 0.2, (m, n))
    obs = 0.04 * np.exp(rng.normal(0, 0.5, (m, n)))
    act = rng.random((m, n)) > 0.2
    args = (0.95, 20.0, 300, 1000, 64)
    a = product.cEstimateSameTrackProcessNoiseTransitions(data, obs, act, *args)
    b = oq.cEstimateSameTrackProcessNoiseTransitions(data, obs, act, *args)
    for x, y in zip(a[:3], b[:3]):
        assert np.array_equal(x, y)
    assert a[3] == b[3]
