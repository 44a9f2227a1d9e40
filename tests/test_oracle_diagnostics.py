"""CPU tests pinning oracle/diagnostics.py (per-interval output diagnostics, core.py:7734-7878).

(1) the reference's own known-answer test (tests/test_core.py:2632-2698): literal inputs, expected values;
(2) an independent per-bin scalar derivation written from the documented model (predicted covariance of bin k from
    the stored filtered covariance of bin k-1, total gain = P_pred[:,0] * sumInvR / (1 + P_pred[0,0] sumInvR)).
"""
import numpy as np
import pytest

from oracle import diagnostics as dg


def _known_answer_inputs():
    covar = np.zeros((3, 2, 2), np.float32)
    covar[:, 0, 0] = [0.4, 0.5, 0.6]
    covar[:, 0, 1] = [0.03, 0.04, 0.05]
    covar[:, 1, 0] = covar[:, 0, 1]
    covar[:, 1, 1] = [0.2, 0.25, 0.3]
    return dict(
        stateCovarForward=covar,
        matrixMunc=np.asarray([[0.9, 1.9, 0.4], [1.1, 0.1, 0.6]], np.float32),
        matrixQ0=np.asarray([[0.2, 0.0], [0.0, 0.05]], np.float32),
        matrixF=np.asarray([[1.0, 0.1], [0.0, 1.0]], np.float32),
        stateCovarInit=1.0, state_dim=2,
        lambdaExp=np.asarray([1.0, 2.0, 0.5], np.float32),
        processPrecExp=np.asarray([1.0, 2.0, 4.0], np.float32),
        processQScale=np.asarray([9.0, 2.0, 3.0], np.float32),
        pNoiseForward=None, pad=0.1,
        obsPrecisionMultiplierMin=0.25, obsPrecisionMultiplierMax=4.0,
        procPrecisionMultiplierMin=0.25, procPrecisionMultiplierMax=4.0)


def test_reference_known_answers():
    """expected values of tests/test_core.py:2666-2698"""
    kw = _known_answer_inputs()
    t = dg.output_diagnostic_tracks(**kw)
    np.testing.assert_allclose(t["preKappaQLevel"], [0.2, 0.4, 0.6], rtol=1e-6)
    np.testing.assert_allclose(t["preKappaQTrend"], [0.05, 0.1, 0.15], rtol=1e-6)
    np.testing.assert_allclose(t["effectiveQLevel"], [0.2, 0.2, 0.15], rtol=1e-6)
    np.testing.assert_allclose(t["effectiveQTrend"], [0.05, 0.05, 0.0375], rtol=1e-6)
    np.testing.assert_allclose(t["processQScale"], [1.0, 2.0, 3.0])
    lam = kw["lambdaExp"]
    np.testing.assert_allclose(t["muncTrace"], np.sum((kw["matrixMunc"].astype(np.float64) + 0.1) / lam[None, :], axis=0),
                               rtol=1e-6)
    sum_inv_r0 = (1.0 / 1.0) + (1.0 / 1.2)
    pred00, pred10 = 1.21, 0.1
    denom = 1.0 + pred00 * sum_inv_r0
    assert t["sumGain0"][0] == pytest.approx(pred00 * sum_inv_r0 / denom, rel=1e-6)
    assert t["sumGain1"][0] == pytest.approx(pred10 * sum_inv_r0 / denom, rel=1e-6)
    for k in ("baseQLevel", "baseQTrend", "preKappaQLevel", "preKappaQTrend", "effectiveQLevel", "effectiveQTrend",
              "processQScale", "muncTrace", "sumGain0", "sumGain1"):
        assert t[k].dtype == np.float32 and t[k].shape == (3,)


def _scalar_spec(covar, munc, Q0, F, p_init, d, lam, kap, qs, pn, pad, wlo, whi, klo, khi):
    n = covar.shape[0]
    g0, g1, tr, el, et = (np.zeros(n) for _ in range(5))
    for k in range(n):
        w = 1.0 if lam is None else min(max(float(lam[k]), wlo), whi)
        s = 0.0
        for j in range(munc.shape[0]):
            r = max(float(munc[j, k]) + pad, 1e-12)
            if np.isfinite(r / w):
                tr[k] += r / w
            if np.isfinite(w / r):
                s += w / r
        q = 1.0 if (qs is None or k == 0) else float(qs[k])
        Q = np.asarray(Q0, np.float64)[:d, :d] * q
        if kap is not None:
            Q = Q / min(max(float(kap[k]), klo), khi)
        elif pn is not None and k > 0 and np.all(np.isfinite(pn[k - 1, :d, :d])):
            Q = np.asarray(pn[k - 1, :d, :d], np.float64)
        el[k] = Q[0, 0]
        et[k] = Q[1, 1] if d == 2 else 0.0
        prev = np.eye(d) * p_init if k == 0 else np.asarray(covar[k - 1, :d, :d], np.float64)
        Fm = np.asarray(F, np.float64) if d == 2 else np.eye(1)
        pred = Fm @ prev @ Fm.T + Q
        p00 = max(pred[0, 0], 0.0)
        den = 1.0 + p00 * s
        if np.isfinite(den) and den > 0:
            g0[k] = p00 * s / den
            g1[k] = (pred[1, 0] if d == 2 else 0.0) * s / den
    return dict(sumGain0=g0, sumGain1=g1, muncTrace=tr, effectiveQLevel=el, effectiveQTrend=et)


@pytest.mark.parametrize("d", [2, 1])
@pytest.mark.parametrize("mode", ["kappa", "pnoise", "plain"])
def test_matches_scalar_specification(d, mode):
    rng = np.random.default_rng(7 + d)
    n, m = 257, 5
    A = rng.normal(size=(n, d, d))
    covar = (A @ A.transpose(0, 2, 1) * 0.01 + np.eye(d) * 0.02).astype(np.float32)
    munc = (0.25 * np.exp(rng.normal(0, 0.4, (m, n)))).astype(np.float32)
    munc[1, 17] = 1e30                      # masked cell
    munc[:, 40] = 1e30                      # fully masked bin
    Q0 = np.diag([1e-3, 1e-4]).astype(np.float32)
    F = np.asarray([[1, 1], [0, 1]], np.float32)
    lam = np.exp(rng.normal(0, 1.0, n)).astype(np.float32)         # exceeds the clip range in places
    kap = np.exp(rng.normal(0, 3.0, n)).astype(np.float32) if mode == "kappa" else None
    qs = np.exp(rng.normal(0, 0.3, n)).astype(np.float32)
    pn = None
    if mode == "pnoise":
        pn = np.zeros((n - 1, d, d), np.float32)
        pn[:, 0, 0] = 1e-3 * np.exp(rng.normal(0, 0.5, n - 1))
        if d == 2:
            pn[:, 1, 1] = 1e-4
        pn[5, 0, 0] = np.nan                # non-finite entry falls back to Q0 * qScale (core.py:7841-7844)
    got = dg.output_diagnostic_tracks(stateCovarForward=covar, matrixMunc=munc, matrixQ0=Q0, matrixF=F,
                                      stateCovarInit=1000.0, state_dim=d, lambdaExp=lam, processPrecExp=kap,
                                      processQScale=qs, pNoiseForward=pn, pad=1e-4)
    ref = _scalar_spec(covar, munc, Q0, F, 1000.0, d, lam, kap, qs, pn, 1e-4, 0.25, 4.0, 5e-3, 5e3)
    for k, v in ref.items():
        np.testing.assert_allclose(got[k], v.astype(np.float32), rtol=2e-6, atol=0, err_msg=k)


def test_input_validation():
    kw = _known_answer_inputs()
    kw["processQScale"] = np.asarray([1.0, np.inf, 1.0], np.float32)
    with pytest.raises(ValueError, match="processQScale contains non-finite"):
        dg.output_diagnostic_tracks(**kw)
    kw = _known_answer_inputs()
    kw["lambdaExp"] = np.ones(4, np.float32)
    with pytest.raises(ValueError, match="lambdaExp length"):
        dg.output_diagnostic_tracks(**kw)
