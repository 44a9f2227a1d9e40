"""Case table for the initial process-noise (Q0) seed (cconsenrich.pyx:1441-2146, core.py:3621-3780).

Inputs are re-synthesised from the seed; qseed_*.npz hold the outputs of the REAL reference's natives (and, for the
`est_*` cases, of the reference caller's composition restated in oracle/qseed.py run ON the reference's natives)."""
from __future__ import annotations

import numpy as np

INF = float("inf")


def _latent(rng, n, q, m, obs):
    x = np.cumsum(rng.normal(0.0, np.sqrt(q), n))
    return np.vstack([x + rng.normal(0.0, np.sqrt(obs), n) for _ in range(m)])


# ---- native-level cases (float64 inputs, like tests/test_core.py:3700-3990) ---------------------------------------
def native_cases():
    return [
        dict(name="qseed_same_uncapped_m4_n160", kind="same", m=4, n=160, seed=11, mask=0.1, args=(0.95, 20.0, 0, 32000, 0)),
        dict(name="qseed_same_capped_m6_n5000", kind="same", m=6, n=5000, seed=12, mask=0.2, args=(0.9, 5.0, 700, 500, 128)),
        dict(name="qseed_same_panel_m3_n900", kind="same", m=3, n=900, seed=13, mask=0.0, args=(1.0, 1.0e12, 0, 32000, 100)),
        dict(name="qseed_same_artifact_m5_n30", kind="same", m=5, n=30, seed=14, mask=0.0, artifact=True,
             args=(0.95, 20.0, 32000, 32000, 2048)),
        dict(name="qseed_same_defaults_m3_n40001", kind="same", m=3, n=40001, seed=15, mask=0.05,
             args=(0.95, 20.0, 32000, 32000, 2048)),
        dict(name="qseed_same_m1_n64", kind="same", m=1, n=64, seed=16, mask=0.3, args=(0.95, 20.0, 20, 7, 5)),
        dict(name="qseed_same_m33_n300_ties", kind="same", m=33, n=300, seed=21, mask=0.1, ties=True,
             args=(0.95, 20.0, 0, 32000, 64)),
        dict(name="qseed_pooled_m4_n500", kind="pooled", m=4, n=500, seed=17, mask=0.4),
        dict(name="qseed_pooled_sparse_m2_n20", kind="pooled", m=2, n=20, seed=18, sparse=True),
        dict(name="qseed_post_synthetic96", kind="post", count=96, seed=0, args=(1.0e-5, 1.0e-2, 8.0, 1.0e-5, 8, np.log(4.0), 8.0, 64)),
        dict(name="qseed_post_grid256_nu3", kind="post", count=96, seed=0, args=(1.0e-5, 1.0e-2, 3.0, 1.0e-5, 8, np.log(4.0), 8.0, 256)),
        dict(name="qseed_post_open_cap", kind="post", count=300, seed=19, args=(1.0e-6, INF, float("nan"), 1.0e-5, 8, np.log(4.0), 8.0, 64)),
        dict(name="qseed_post_single_grid", kind="post", count=40, seed=20, args=(1.0e-3, 1.0e-3, 8.0, 1.0e-3, 8, np.log(4.0), 8.0, 64)),
        dict(name="qseed_post_insufficient", kind="post", count=5, seed=22, args=(1.0e-5, 1.0, 8.0, 1.0e-5, 8, np.log(4.0), 8.0, 64)),
    ]


def native_inputs(case):
    rng = np.random.default_rng(case["seed"])
    if case["kind"] == "post":
        c = case["count"]
        if case["seed"] == 0:                                       # the synthetic series of test_core.py:3929-3937
            x = (np.arange(c, dtype=np.float64) + 0.5) / c
            d = 0.05 * np.sin(2 * np.pi * x) + 0.0175 * np.sin(6 * np.pi * x) + 0.006 * np.cos(10 * np.pi * x)
            return d, np.full(c, 1.0e-4), 1.0 + 0.35 * np.cos(2 * np.pi * x)
        d = rng.normal(0, 0.05, c) * np.exp(rng.normal(0, 0.5, c))
        d[::7] = np.round(d[::7], 2)                                # ties in the weighted quantiles
        return d, 1.0e-4 * np.exp(rng.normal(0, 0.7, c)), np.exp(rng.normal(0, 1.2, c))
    m, n = case["m"], case["n"]
    if case.get("sparse"):                                          # test_core.py:3515-3519 (as float64 inputs)
        data = np.full((m, n), np.nan)
        data[0, ::2] = 0.0
        data[1, 1::2] = 1.0
        obs = np.full((m, n), 0.1001)
        return data, obs, np.isfinite(data)
    data = _latent(rng, n, 1.0e-2, m, 2.0e-3)
    obs = 2.0e-3 * np.exp(rng.normal(0, 0.5, (m, n)))
    if case.get("artifact"):
        data[:] = 0.0
        data[0, n // 2:] = 100.0
        obs[:] = 0.1001
        obs[0, :] = 1.0e-4
    if case.get("ties"):
        data = np.round(data, 1)
        obs = np.round(obs * 500) / 500 + 1.0e-3
    active = rng.random((m, n)) >= case.get("mask", 0.0)
    return data, obs, active


def run_native(mod, case):
    """-> dict of arrays / scalars (dict diagnostics flattened with a `d_` prefix)"""
    inp = native_inputs(case)
    out = {}
    if case["kind"] == "same":
        d, s, w, diag = mod.cEstimateSameTrackProcessNoiseTransitions(*inp, *case["args"])
        out.update(deltas=np.asarray(d), svar=np.asarray(s), weights=np.asarray(w))
        for k, v in diag.items():
            if k == "sampledTransitionIndices":
                v = np.asarray([-1] if v is None else v, np.int64)
            out["d_" + k] = np.asarray(v)
    elif case["kind"] == "pooled":
        d, s, w = mod.cEstimatePooledProcessNoiseTransitions(*inp)
        out.update(deltas=np.asarray(d), svar=np.asarray(s), weights=np.asarray(w))
    else:
        a = case["args"]
        r = mod.cQSeedPosteriorFromTransitions(*inp, a[0], a[1], a[2], "sameTrackEB", *a[3:])
        for k, v in r.items():
            out["d_" + k] = np.asarray(v)
    return out


# ---- caller-level cases (float32 inputs, like tests/test_core.py:3462-3541) ----------------------------------------
def estimate_cases():
    return [
        dict(name="qseed_est_trend_m4_n160", m=4, n=160, seed=31, model="levelTrend", minQ=1.0e-5, maxQ=1.0, deltaF=1.0, nu=8.0),
        dict(name="qseed_est_artifact_level", m=5, n=30, seed=32, model="level", minQ=1.0e-5, maxQ=10.0, deltaF=1.0, nu=8.0,
             artifact=True),
        dict(name="qseed_est_pooled_fallback", m=2, n=20, seed=33, model="level", minQ=1.0e-4, maxQ=1.0, deltaF=1.0, nu=8.0,
             sparse=True),
        dict(name="qseed_est_obsvar_fallback", m=3, n=6, seed=34, model="levelTrend", minQ=1.0e-6, maxQ=1000.0, deltaF=1.0,
             nu=8.0),
        dict(name="qseed_est_minq_fallback", m=2, n=50, seed=35, model="levelTrend", minQ=1.0e-6, maxQ=1000.0, deltaF=1.0,
             nu=8.0, all_masked=True),
        dict(name="qseed_est_capped_m8_n50000", m=8, n=50000, seed=36, model="levelTrend", minQ=1.0e-6, maxQ=1000.0,
             deltaF=0.5, nu=float("nan"), mask=0.1),
        dict(name="qseed_est_open_cap_m5_n3000", m=5, n=3000, seed=37, model="levelTrend", minQ=1.0e-6, maxQ=-1.0,
             deltaF=2.0, nu=6.0, mask=0.02, prior=1.0e-4),
    ]


def estimate_inputs(case):
    rng = np.random.default_rng(case["seed"])
    m, n = case["m"], case["n"]
    if case.get("sparse"):
        data = np.full((m, n), np.nan, np.float32)
        data[0, ::2] = 0.0
        data[1, 1::2] = 1.0
        return data, np.full((m, n), 0.1, np.float32)
    if case.get("artifact"):
        data = np.zeros((m, n), np.float32)
        data[0, n // 2:] = 100.0
        munc = np.full((m, n), 0.1, np.float32)
        munc[0, :] = 1.0e-9
        return data, munc
    data = _latent(rng, n, 1.0e-2, m, 2.0e-3).astype(np.float32)
    munc = (2.0e-3 * np.exp(rng.normal(0, 0.4, (m, n)))).astype(np.float32)
    if case.get("mask"):
        munc[rng.random((m, n)) < case["mask"]] = np.float32(1.0e30)
        data[rng.random((m, n)) < 0.01] = np.nan
    if case.get("all_masked"):
        munc[:] = np.float32(1.0e30)
    return data, munc


STRING_KEYS = ("qSeedSource", "qSeedReason")


def run_estimate(estimator, case):
    """estimator(matrixData=..., matrixMunc=..., pad=..., stateModel=..., minQ=..., maxQ=..., deltaF=..., robustTNu=...,
    qSeedPriorLevel=...) -> (Q, diagnostics); -> flat dict"""
    data, munc = estimate_inputs(case)
    Q, diag = estimator(matrixData=data, matrixMunc=munc, pad=1.0e-4, stateModel=case["model"], minQ=case["minQ"],
                        maxQ=case["maxQ"], deltaF=case["deltaF"], robustTNu=case["nu"],
                        qSeedPriorLevel=case.get("prior", 1.0e-5))
    out = {"Q": np.asarray(Q, np.float32)}
    for k, v in diag.items():
        out["d_" + k] = np.asarray(v)
    return out


def same(a, b, *, rtol=0.0):
    """compare two flat result dicts; rtol = 0 means bit-for-bit on floats"""
    assert set(a) == set(b), (sorted(a), sorted(b))
    for k in a:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        if x.dtype.kind in "US" or y.dtype.kind in "US":
            assert str(x) == str(y), k
        elif x.dtype.kind == "f" and rtol > 0:
            np.testing.assert_allclose(x, y, rtol=rtol, atol=0, equal_nan=True, err_msg=k)
        else:
            assert x.shape == y.shape and np.array_equal(x, y, equal_nan=(x.dtype.kind == "f")), (k, x, y)
