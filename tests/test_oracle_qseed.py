"""CPU tests pinning oracle/qseed.py + oracle/qseed_oracle.c (the initial process-noise seed).

(1) golden vectors captured from the compiled reference's natives (bit-for-bit);
(2) the reference's own known-answer assertions for the caller (tests/test_core.py:3462-3541);
(3) an independent NumPy/SciPy restatement of the posterior (Student-t log-pdf via scipy.stats, np.interp on the CDF;
    the specification tests/test_core.py:3705-3755 encodes) at the reference test's 1e-8;
(4) error texts of the natives' validation.
"""
import functools
import math
import os

import numpy as np
import pytest

import qseed_cases as qc
from oracle import qseed as oq
from oracle import ref_loader

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _gold(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("case", qc.native_cases(), ids=lambda c: c["name"])
def test_natives_reproduce_reference_golden_vectors(case):
    qc.same(qc.run_native(oq, case), _gold(case["name"]))


@pytest.mark.parametrize("case", qc.estimate_cases(), ids=lambda c: c["name"])
def test_estimator_reproduces_reference_golden_vectors(case):
    got = qc.run_estimate(functools.partial(oq.estimate_initial_process_noise, oq), case)
    qc.same(got, _gold(case["name"]))


@pytest.mark.skipif(not ref_loader.available(), reason="compiled reference only exists in the build container")
def test_natives_equal_live_reference():
    ref = ref_loader.load()
    for case in qc.native_cases():
        qc.same(qc.run_native(oq, case), qc.run_native(ref, case))


def test_reference_known_answers_for_the_caller():
    """assertions of tests/test_core.py:3484-3487, 3512-3515, 3536-3541 on the same inputs"""
    rng = np.random.default_rng(2024)
    n, m, q_true, obs = 160, 4, 1.0e-2, 2.0e-3
    latent = np.cumsum(rng.normal(0.0, np.sqrt(q_true), size=n))
    data = np.vstack([latent + rng.normal(0.0, np.sqrt(obs), size=n) for _ in range(m)]).astype(np.float32)
    Q, d = oq.estimate_initial_process_noise(oq, matrixData=data, matrixMunc=np.full((m, n), obs, np.float32), pad=1.0e-4,
                                             stateModel="levelTrend", minQ=1.0e-5, maxQ=1.0, deltaF=1.0, robustTNu=8.0)
    assert d["qSeedSource"] == "sameTrackEB" and 0.3 * q_true <= d["qSeedLevelFinal"] <= 3.0 * q_true
    assert Q[0, 0] == pytest.approx(d["qSeedLevelFinal"]) and Q[1, 1] == pytest.approx(d["qSeedLevelFinal"])

    data = np.zeros((5, 30), np.float32)
    data[0, 15:] = 100.0
    munc = np.full((5, 30), 0.1, np.float32)
    munc[0, :] = 1.0e-9
    Q, d = oq.estimate_initial_process_noise(oq, matrixData=data, matrixMunc=munc, pad=1.0e-4, stateModel="level",
                                             minQ=1.0e-5, maxQ=10.0, deltaF=1.0, robustTNu=8.0)
    assert d["qSeedSource"] == "sameTrackEB" and d["qSeedPrecisionCapFraction"] > 0.0 and d["qSeedLevelFinal"] < 1.0e-3

    data = np.full((2, 20), np.nan, np.float32)
    data[0, ::2] = 0.0
    data[1, 1::2] = 1.0
    Q, d = oq.estimate_initial_process_noise(oq, matrixData=data, matrixMunc=np.full((2, 20), 0.1, np.float32), pad=1.0e-4,
                                             stateModel="level", minQ=1.0e-4, maxQ=1.0, deltaF=1.0, robustTNu=8.0)
    assert d["qSeedSource"] == "pooledEB" and d["qSeedTransitionCount"] == 19
    assert np.isfinite(Q).all() and Q[0, 0] > 1.0e-4


def _weighted_quantile(v, w, q):
    o = np.argsort(v, kind="mergesort")
    v, w = np.asarray(v)[o], np.asarray(w)[o]
    cum = np.cumsum(w)
    t = q * cum[-1]
    i = int(np.searchsorted(cum, t, side="left"))
    if i == 0:
        return float(v[0])
    return float(v[i - 1] + (t - cum[i - 1]) / (cum[i] - cum[i - 1]) * (v[i] - v[i - 1]))


def test_posterior_matches_scipy_specification():
    from scipy import stats

    case = next(c for c in qc.native_cases() if c["name"] == "qseed_post_synthetic96")
    d, s2, w = qc.native_inputs(case)
    got = oq.cQSeedPosteriorFromTransitions(d, s2, w, 1.0e-5, 1.0e-2, 8.0, "x", 1.0e-5, 8, math.log(4.0), 8.0, 64)
    center = _weighted_quantile(d, w, 0.5)
    scale = 1.4826 * _weighted_quantile(np.abs(d - center), w, 0.5)
    med_s2 = _weighted_quantile(s2, w, 0.5)
    q_prior = max(scale * scale - med_s2, 1.0e-5)
    grid = np.exp(np.linspace(math.log(1.0e-5), math.log(1.0e-2), 64))
    wn = np.clip(w / _weighted_quantile(w, w, 0.5), 0.25, 4.0)
    lp = np.empty(64)
    for i, q in enumerate(grid):
        sc = np.sqrt(q + s2)
        lp[i] = np.sum(wn * (stats.t.logpdf(d / sc, df=8.0) - np.log(sc))) - 0.5 * ((math.log(q) - math.log(q_prior)) / math.log(4.0)) ** 2
    post = np.exp(lp - lp.max())
    post /= post.sum()
    cdf = np.cumsum(post)
    assert got["priorLevel"] == pytest.approx(q_prior, rel=1e-8)
    assert got["posteriorModeLevel"] == pytest.approx(grid[int(np.argmax(post))], rel=1e-8)
    for key, pr in (("posteriorMedianLevel", 0.5), ("posteriorQ05Level", 0.05), ("posteriorQ95Level", 0.95)):
        assert got[key] == pytest.approx(float(np.interp(pr, cdf, grid)), rel=1e-8, abs=1e-12)
    assert got["transitionQ90"] == pytest.approx(_weighted_quantile(np.maximum(d * d - s2, 0.0), w, 0.9), rel=1e-8)


def test_validation_errors():
    one = np.ones(8)
    tail = (1.0e-5, 8, math.log(4.0), 8.0, 64)
    with pytest.raises(ValueError, match="samplingVariances"):
        oq.cQSeedPosteriorFromTransitions(one, -one, one, 1.0e-5, 1.0, 8.0, "bad", *tail)
    with pytest.raises(ValueError, match="transitionWeights"):
        oq.cQSeedPosteriorFromTransitions(one, one, 0 * one, 1.0e-5, 1.0, 8.0, "bad", *tail)
    with pytest.raises(ValueError, match="same length"):
        oq.cQSeedPosteriorFromTransitions(one, one[:3], one, 1.0e-5, 1.0, 8.0, "bad", *tail)
    with pytest.raises(ValueError, match="must not exceed"):
        oq.cQSeedPosteriorFromTransitions(one, one, one, 1.0e-5, 1.0e-6, 8.0, "bad", *tail)
    data, act = np.zeros((2, 9)), np.ones((2, 9), bool)
    with pytest.raises(ValueError, match="obsVar values must be positive"):
        oq.cEstimateSameTrackProcessNoiseTransitions(data, np.zeros((2, 9)), act, 0.95, 20.0)
    with pytest.raises(ValueError, match=r"precisionCapQuantile must be in \[0, 1\]"):
        oq.cEstimateSameTrackProcessNoiseTransitions(data, np.ones((2, 9)), act, 1.5, 20.0)
    with pytest.raises(ValueError, match="activeObservation shape"):
        oq.cEstimateSameTrackProcessNoiseTransitions(data, np.ones((2, 9)), act[:, :4], 0.95, 20.0)
    bad = data.copy()
    bad[0, 3] = np.inf
    with pytest.raises(ValueError, match="active pooled observations"):
        oq.cEstimatePooledProcessNoiseTransitions(bad, np.ones((2, 9)), act)
    e = oq.cEstimateSameTrackProcessNoiseTransitions(np.zeros((2, 1)), np.ones((2, 1)), np.ones((2, 1), bool), 0.95, 20.0)
    assert e[0].size == 0 and e[3]["pairCount"] == 0 and math.isnan(e[3]["precisionCap"])
