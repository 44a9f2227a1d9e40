"""Hard data: few samples (m = 8, m = 5) with 2-3 % outlier cells, a 6-iteration kappa-ECM (the reference's default loop).

What the round-2 review asked to be gated: the smoothed track of the DEFAULT mode against the oracle on EVERY bin,
|xs - oracle| <= 1e-5 max|row| + 2e-6, with the same iteration count.

What the CPU part of this file establishes first (with the oracle, i.e. the reference's own arithmetic): on exactly this
data the reference's ECM map is ill-conditioned relative to that gate.  Moving 1 % of the input cells by ONE float32 ulp
(6e-8 relative) moves the reference's own smoothed state by several times the gate after six iterations (the kappa E-step,
pyx:8244-8298, divides differences of neighbouring smoothed levels by Q0 ~ 1e-3 / 1e-4 and feeds them back into the next
sweep's process noise).  An implementation that is not bit-identical to the sequential recursion carries ulp-level
differences in its states by construction and inherits that amplification; only the bit-exact mode (x_tol_ulps = 0) can
hold the gate there, which is why it is the default of every entry point, and why the 2-ulp throughput mode is an explicit
opt-in whose contract is stated per PASS (a single forward / backward pass is within a few ulps on this data too)."""
import numpy as np
import pytest

import cases

RTOL, ATOL = 1.0e-5, 2.0e-6
N_LIST = [40000, 9000, 700]


def _oracle():
    from oracle import oracle as orc

    orc.lib()
    return orc


def _ecm(orc, d, v, iters=6, kappa_init=None):
    n = d.shape[1]
    return orc.cfixedBackgroundECM(matrixData=d, matrixPluginMuncInit=v, matrixF=np.asarray(cases.F_TREND, np.float32),
                                   matrixQ0=np.diag([1e-3, 1e-4]).astype(np.float32),
                                   intervalToBlockMap=np.zeros(n, np.int32), blockCount=1, stateInit=0.0,
                                   stateCovarInit=1000.0, ECM_fixedBackgroundIters=iters, ECM_fixedBackgroundRtol=1e-7,
                                   ECM_useObsPrecisionReweighting=False, ECM_useProcessPrecisionReweighting=True,
                                   procPrecisionMultiplierMin=5e-3, procPrecisionMultiplierMax=5e3, t_innerIters=5,
                                   processPrecExpInit=kappa_init, returnIntermediates=True, returnDiagnostics=False,
                                   logIterations=False)


def _gate(ref_xs):
    scale = np.abs(ref_xs.astype(np.float64)).max(axis=1, keepdims=True)
    return RTOL * scale + ATOL


def _one_ulp_on_some_cells(d, seed, frac=0.01):
    rng = np.random.default_rng(seed)
    hit = rng.random(d.shape) < frac
    return np.where(hit, np.nextafter(d, np.float32(np.inf)), d).astype(np.float32)


def reference_self_sensitivity(orc, d, v, trials=3):
    """max over `trials` of (worst |xs(perturbed input) - xs| / gate) for the REFERENCE arithmetic itself."""
    r = _ecm(orc, d, v)
    gate = _gate(r[2])
    worst = 0.0
    for t in range(trials):
        r2 = _ecm(orc, _one_ulp_on_some_cells(d, t), v)
        worst = max(worst, float((np.abs(r2[2].astype(np.float64) - r[2]) / gate).max()))
    return worst, r


def test_reference_ecm_is_ill_conditioned_on_hard_data_and_well_conditioned_on_the_bench_recipe():
    """CPU, oracle only.  m = 8 with 3 % outlier cells: one float32 ulp on 1 % of the cells moves the reference's own result
    OUTSIDE the parity gate (measured 4.7 - 13.5 x); the bench recipe (m = 32, no outliers) moves by 0.02 x."""
    orc = _oracle()
    d, v = cases.synth(40000, 8, 5100, outlier_frac=0.03)
    hard, _ = reference_self_sensitivity(orc, d, v)
    assert hard > 3.0, hard
    d, v = cases.synth(40000, 32, 5100, outlier_frac=0.0)
    easy, _ = reference_self_sensitivity(orc, d, v)
    assert easy < 0.1, easy


def _batch_ecm(sets, m, **kw):
    from consenrich_amd import _lib as L
    from consenrich_amd.batch import DeviceBatch, ModelParams

    with DeviceBatch(0, **kw) as b:
        b.configure(ModelParams(state_dim=2), m, [d.shape[1] for d, _ in sets])
        for c, (d_, v_) in enumerate(sets):
            b.upload(c, d_, v_)
        b.stats()
        outs, paths = b.ecm(max_iters=6, inner_iters=5, rtol=1e-7, use_lambda=False, use_kappa=True)
        b.export(L.EXPORT_SMOOTH | L.EXPORT_MULT)
        got = [(int(o.iters_done), b.download(c, "xs"), b.download(c, "Ps"), b.download(c, "kappa")) for c, o in enumerate(outs)]
        return got, b.run_stats()


@pytest.mark.gpu
@pytest.mark.parametrize("m,outl", [(8, 0.03), (8, 0.02), (5, 0.02), (5, 0.03)])
def test_hard_data_ecm_in_the_default_mode_holds_the_gate_on_every_bin(m, outl):
    """DeviceBatch with its DEFAULTS (no validation argument): chains of 40 000 / 9 000 / 700 bins, six ECM iterations of five
    sweeps: same iteration count, every bin of the smoothed state inside the gate, kappa inside it too."""
    orc = _oracle()
    sets = [cases.synth(n, m, 5100 + i, outlier_frac=outl) for i, n in enumerate(N_LIST)]
    got, rs = _batch_ecm(sets, m)
    assert rs["x_tol_ulps"] == 0, rs          # the default IS the bit-exact mode
    for c, (d_, v_) in enumerate(sets):
        r = _ecm(orc, d_, v_)
        assert got[c][0] == r[0], (c, got[c][0], r[0])
        excess = np.abs(got[c][1].astype(np.float64) - r[2]) / _gate(r[2])
        k = np.unravel_index(np.argmax(excess), excess.shape)
        assert excess.max() <= 1.0, (m, outl, c, k, float(excess.max()))
        np.testing.assert_allclose(got[c][2], r[3], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(got[c][3], r[7], rtol=RTOL, atol=ATOL)


@pytest.mark.gpu
@pytest.mark.parametrize("m,outl", [(8, 0.03), (5, 0.02)])
def test_hard_data_single_pass_in_the_throughput_mode_holds_the_gate_on_every_bin(m, outl):
    """What bench.py's headline measures is ONE forward + backward pass.  With the hard data AND the hard multipliers (the
    oracle's kappa after six ECM iterations: 1 % of the bins at the lower bound) the 2-ulp mode is inside the gate on every
    bin of every track: a pass does not amplify (the state map's gain over a block is < 3)."""
    from consenrich_amd import _lib as L
    from consenrich_amd.batch import DeviceBatch, ModelParams

    orc = _oracle()
    sets = [cases.synth(n, m, 5100 + i, outlier_frac=outl) for i, n in enumerate(N_LIST)]
    refs = [_ecm(orc, d_, v_) for d_, v_ in sets]
    F = np.asarray(cases.F_TREND, np.float32)
    Q0 = np.diag([1e-3, 1e-4]).astype(np.float32)
    with DeviceBatch(0, x_tol_ulps=2) as b:
        b.configure(ModelParams(state_dim=2), m, N_LIST)
        for c, (d_, v_) in enumerate(sets):
            b.upload(c, d_, v_)
            b.upload_multipliers(c, None, refs[c][7], None)
        b.stats()
        b.forward_backward(L.RETURN_NLL | L.USE_KAPPA, want_sums=False)
        b.export(L.EXPORT_FORWARD | L.EXPORT_SMOOTH)
        worst = 0.0
        for c, (d_, v_) in enumerate(sets):
            n = N_LIST[c]
            xf, Pf, pn = np.zeros((n, 2), np.float32), np.zeros((n, 2, 2), np.float32), np.zeros((n, 2, 2), np.float32)
            orc.cforwardPass(matrixData=d_, matrixPluginMuncInit=v_, matrixF=F, matrixQ0=Q0,
                             intervalToBlockMap=np.zeros(n, np.int32), blockCount=1, stateInit=0.0, stateCovarInit=1000.0,
                             stateForward=xf, stateCovarForward=Pf, pNoiseForward=pn, processPrecExp=refs[c][7],
                             procPrecisionMultiplierMin=5e-3, procPrecisionMultiplierMax=5e3, returnNLL=True)
            bw = orc.cbackwardPass(matrixData=d_, matrixF=F, stateForward=xf, stateCovarForward=Pf, pNoiseForward=pn)
            for name, ref in (("xf", xf), ("xs", bw[0])):
                e = np.abs(b.download(c, name).astype(np.float64) - ref) / _gate(ref)
                worst = max(worst, float(e.max()))
                assert e.max() <= 1.0, (name, c, float(e.max()))
            for name, ref in (("Pf", Pf), ("Ps", bw[1]), ("lag", bw[2][: n - 1])):
                np.testing.assert_allclose(b.download(c, name), ref, rtol=RTOL, atol=ATOL, err_msg=f"{name} chain {c}")
        assert worst <= 0.2, worst        # measured: a few float32 ulps of the level, an order inside the gate


@pytest.mark.gpu
def test_throughput_mode_ecm_on_hard_data_moves_no_more_than_the_reference_moves_itself():
    """The 2-ulp mode (opt-in) on the hard data: after six ECM iterations its smoothed state is outside the gate (measured
    2.5 - 8 x) -- and so is the reference when 1 % of its input cells move by one float32 ulp (4.7 - 13.5 x).  Gated here: same
    iteration count, and an error within 3 x the reference's own sensitivity + 1 gate."""
    orc = _oracle()
    m, outl = 8, 0.03
    sets = [cases.synth(n, m, 5100 + i, outlier_frac=outl) for i, n in enumerate(N_LIST)]
    got, rs = _batch_ecm(sets, m, x_tol_ulps=2)
    assert rs["x_tol_ulps"] == 2
    for c, (d_, v_) in enumerate(sets):
        own, r = reference_self_sensitivity(orc, d_, v_)
        assert got[c][0] == r[0]
        err = float((np.abs(got[c][1].astype(np.float64) - r[2]) / _gate(r[2])).max())
        assert err <= 3.0 * own + 1.0, (c, err, own)


# ----------------------------------------------------------------------------------------------------------------------
# Round 4: the same hard recipe at CHROMOSOME size.  The default mode is exact to the library's own sequential kernel, not
# to the reference: the hoisted sufficient statistics (S1 = lambda S0u (zbar - x) instead of sum_j w_j (z_j - x),
# pyx:271-282) and the Newton-refined reciprocals differ from the reference's operations at ~2^-50 relative, which flips a
# float32 rounding of a carry about once per 1e7 stored values.  On chains of 700 ... 40 000 bins a 6-iteration ECM sees
# ~1 such flip; at chr21 / chr1 size flips are certain.  These tests MEASURE (a) the flip count of one forward pass and
# (b) whether the 6-iteration kappa-ECM still holds the gate, and write what they found to gpurun_out/.
# ----------------------------------------------------------------------------------------------------------------------
SCALE_CASES = [("chr21", 233550, 8, 0.03), ("chr21", 233550, 5, 0.02), ("chr1", 1244783, 8, 0.03), ("chr1", 1244783, 5, 0.02)]
# measured on MI355X (profiles/r04_parity_worst_hard_*.json); asserted with slack below
HARD_SCALE_BOUND = float(__import__("os").environ.get("CONSENRICH_AMD_HARD_SCALE_BOUND", "1.0"))


def _first_sweep_flips(orc, d, v, x_tol_ulps=None):
    """Bitwise comparison of ONE stored forward pass (no multipliers = the first sweep of the ECM loop, kappa = 1) with the
    oracle: number of float32 elements of xf / Pf / D that differ."""
    from consenrich_amd import _lib as L
    from consenrich_amd.batch import DeviceBatch, ModelParams

    n, m = d.shape[1], d.shape[0]
    F = np.asarray(cases.F_TREND, np.float32)
    Q0 = np.diag([1e-3, 1e-4]).astype(np.float32)
    xf, Pf, pn = np.zeros((n, 2), np.float32), np.zeros((n, 2, 2), np.float32), np.zeros((n, 2, 2), np.float32)
    D = np.zeros(n, np.float32)
    orc.cforwardPass(matrixData=d, matrixPluginMuncInit=v, matrixF=F, matrixQ0=Q0, intervalToBlockMap=np.zeros(n, np.int32),
                     blockCount=1, stateInit=0.0, stateCovarInit=1000.0, stateForward=xf, stateCovarForward=Pf,
                     pNoiseForward=pn, vectorD=D, returnNLL=True)
    kw = {} if x_tol_ulps is None else {"x_tol_ulps": x_tol_ulps}
    with DeviceBatch(0, **kw) as b:
        b.configure(ModelParams(state_dim=2), m, [n])
        b.upload(0, d, v)
        b.stats()
        b.forward(L.RETURN_NLL)
        b.export(L.EXPORT_FORWARD)
        gx, gP, gD = b.download(0, "xf"), b.download(0, "Pf"), b.download(0, "D")
    out = {}
    for name, g, r in (("xf", gx, xf), ("Pf", gP, Pf), ("D", gD, D)):
        diff = g.view(np.uint32) != r.view(np.uint32)
        out[name] = {"differ": int(diff.sum()), "of": int(diff.size),
                     "first_bin": int(np.argwhere(diff.reshape(n, -1).any(axis=1))[0, 0]) if diff.any() else -1,
                     "worst_rel": float((np.abs(g.astype(np.float64) - r) / np.maximum(np.abs(r.astype(np.float64)), 1e-30)).max())}
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("label,n,m,outl", SCALE_CASES, ids=[f"{c[0]}-m{c[2]}" for c in SCALE_CASES])
def test_hard_data_at_chromosome_size_default_mode(label, n, m, outl):
    """DeviceBatch defaults on ONE chromosome-sized chain of the hard recipe: flip count of the first sweep, then the
    6-iteration kappa-ECM against the oracle: same iteration count, worst |xs - oracle| / gate recorded and asserted."""
    import json
    import os

    orc = _oracle()
    d, v = cases.synth(n, m, 7100 + m, outlier_frac=outl)
    flips = _first_sweep_flips(orc, d, v)
    got, rs = _batch_ecm([(d, v)], m)
    assert rs["x_tol_ulps"] == 0, rs
    r = _ecm(orc, d, v)
    excess = np.abs(got[0][1].astype(np.float64) - r[2]) / _gate(r[2])
    k = np.unravel_index(np.argmax(excess), excess.shape)
    xs_flips = int((got[0][1].view(np.uint32) != r[2].view(np.uint32)).sum())
    kap_rel = float((np.abs(got[0][3].astype(np.float64) - r[7]) / (RTOL * np.abs(r[7].astype(np.float64)) + ATOL)).max())
    rec = {"case": f"{label} n={n} m={m} outliers={outl}", "mode": "default (x_tol_ulps=0)",
           "iters": [got[0][0], int(r[0])], "worst_xs_over_gate": float(excess.max()), "worst_at": [int(k[0]), int(k[1])],
           "bins_outside_gate": int((excess > 1.0).any(axis=1).sum()), "xs_float32_elements_differing_after_ecm": xs_flips,
           "kappa_worst_over_gate": kap_rel, "first_sweep_vs_oracle": flips,
           "flip_density_xf": flips["xf"]["differ"] / flips["xf"]["of"]}
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/parity_worst_hard_{label}_m{m}.json", "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps(rec))
    assert got[0][0] == r[0], rec
    assert excess.max() <= HARD_SCALE_BOUND, rec


@pytest.mark.gpu
def test_hard_data_at_chromosome_size_flip_density_of_the_two_modes():
    """What the default mode buys over the opt-in 2-ulp mode, in the one number that can be counted: float32 elements of the
    filtered state that differ from the reference after ONE forward pass over a chr1-sized hard chain."""
    import json
    import os

    orc = _oracle()
    d, v = cases.synth(1244783, 8, 7108, outlier_frac=0.03)
    exact = _first_sweep_flips(orc, d, v)
    ulp2 = _first_sweep_flips(orc, d, v, x_tol_ulps=2)
    rec = {"case": "chr1 n=1244783 m=8 outliers=0.03, one forward pass", "default": exact, "ulp2": ulp2}
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_flip_density_modes.json", "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps(rec))
    assert exact["xf"]["differ"] <= 64, rec                     # ~1e-7 per value expected: a handful per chromosome
    assert ulp2["xf"]["differ"] > 100 * max(exact["xf"]["differ"], 1), rec
