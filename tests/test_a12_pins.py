"""Row a12 (`runConsenrich` orchestration, core.py:3861-6142) pinned by the known answers the REFERENCE'S OWN TESTS hold for its
pure-Python glue.  `consenrich.core` cannot be imported in this image, so the reference's test bodies cannot run; what they
assert is data -- literal inputs and expected numbers / counts -- and is applied here to both restatements of the glue:
the CPU twin (oracle/driver.py, oracle/background.py) and the product (consenrich_amd/driver.py + the device-resident
background update).

  test_core.py:4533-4610  weighted-RMS shift gate: weights [100, 1], seed [100, 0], proposal [100.1, 10], rtol 0.05
                          => shift = sqrt(w . d^2 / 101), threshold = 0.05 * max(RMS(proposal), RMS(seed), 1), "stable" although
                          the MAX-norm shift (10) exceeds the max-norm threshold (5.005)
  test_core.py:4614-4693  three outer passes are run although every tolerance is met at once (ECM_minOuterIters default 3);
                          max(minOuter, outerIters) passes are planned; 4 / 4 / 2 / > 1 ECM calls for the four configurations;
                          the last ECM record carries final_fixed_background_ecm = True
  test_core.py:4471-4491  background warm start (`estimateProvisionalBackground`) of a constant 1.75 matrix recovers it to 1e-4
  test_core.py:4494-4530  warm_start summary: an initial background means no background pre-pass
"""
import math

import numpy as np
import pytest

SEED_G = np.asarray([100.0, 0.0], np.float32)
PROPOSAL_G = np.asarray([100.1, 10.0], np.float32)
LIT_DATA = np.asarray([[100.0, 0.0]], np.float32)
LIT_MUNC = np.asarray([[0.01, 1.0]], np.float32)


def _expected_gate():
    """The arithmetic of test_core.py:4583-4609, literally."""
    weights = np.asarray([100.0, 1.0], dtype=np.float64)
    delta = PROPOSAL_G.astype(np.float64) - SEED_G.astype(np.float64)
    shift = math.sqrt(float(np.dot(weights, delta * delta)) / 101.0)
    scale = max(math.sqrt(float(np.dot(weights, PROPOSAL_G * PROPOSAL_G)) / 101.0),
                math.sqrt(float(np.dot(weights, SEED_G * SEED_G)) / 101.0), 1.0)
    max_shift = float(np.max(np.abs(delta)))
    max_thr = 0.05 * max(float(np.max(np.abs(PROPOSAL_G))), float(np.max(np.abs(SEED_G))), 1.0)
    return weights, shift, 0.05 * scale, max_shift, max_thr


def test_shift_gate_known_answer_cpu_twin_and_product_host_function():
    from consenrich_amd import driver as prod
    from oracle import background as bgo
    from oracle import driver as twin

    weights, shift, thr, max_shift, max_thr = _expected_gate()
    # the weight track the reference hands to its solver (capturedWeights, test_core.py:4581-4582): pad = 0
    w, _r, _inv, _res = bgo.weight_rhs_tracks(LIT_DATA, LIT_MUNC, np.zeros(2, np.float32), np.float32(0.0))
    np.testing.assert_allclose(w, weights, rtol=0.0, atol=1.0e-5)          # float32 reciprocal of 0.01: 100 +- 4e-6
    g = twin.background_shift_gate(weights, PROPOSAL_G, SEED_G, 0.05)
    assert g["background_shift"] == pytest.approx(shift)
    assert g["background_shift_threshold"] == pytest.approx(thr)
    assert g["background_shift_stable"] is True
    assert max_shift > max_thr                                             # a max-norm gate would NOT be stable here
    p = prod.background_shift_gate(*prod.weighted_rms(weights, PROPOSAL_G, SEED_G), 0.05)
    assert p["background_shift"] == pytest.approx(shift)
    assert p["background_shift_threshold"] == pytest.approx(thr)
    assert p["background_shift_stable"] is True
    with pytest.raises(ValueError, match="shift RMS requires positive weights"):      # core.py:5204-5205
        twin.background_shift_gate([0.0, 0.0], PROPOSAL_G, SEED_G, 0.05)
    with pytest.raises(ValueError, match="shift RMS requires positive weights"):
        prod.weighted_rms([0.0, 0.0], PROPOSAL_G, SEED_G)


def test_planned_outer_passes_rule():
    from consenrich_amd import driver as prod
    from oracle import driver as twin

    for outer, min_outer, fit_bg, want in ((5, 3, True, 5), (2, 3, True, 3), (1, 1, True, 1), (0, 1, True, 1), (7, 3, False, 1)):
        assert twin.planned_outer_passes(dict(outer_passes=outer, min_outer=min_outer, fit_background=fit_bg)) == want
        assert prod.planned_outer_passes(prod.FitConfig(penalties=(1.0, 1.0), outer_passes=outer, min_outer=min_outer,
                                                        fit_background=fit_bg)) == want


def _three_pass_inputs():
    """Inputs of test_core.py:4614-4653."""
    rng = np.random.default_rng(23)
    n, m = 32, 3
    grid = np.linspace(0.0, 1.0, n, dtype=np.float32)
    data = np.vstack([grid + 0.02 * rng.normal(size=n) - 0.01, grid + 0.02 * rng.normal(size=n),
                      grid + 0.02 * rng.normal(size=n) + 0.01]).astype(np.float32)
    munc = np.full((m, n), 0.2, dtype=np.float32)
    return data, munc


def _three_pass_cfg(outer, min_outer=3, nll_rtol=1.0e9):
    from oracle import background as bgo

    # runConsenrich's signature defaults (core.py:3877-3888: observation AND process re-weighting on, nu 8, pad 1e-4,
    # multiplier bounds (0.25, 4) / (5e-3, 5e3), background smoothness 128, nonnegative background) with the test's overrides
    return dict(state_dim=2, F=[[1.0, 0.2], [0.0, 1.0]], Q0=np.diag([1.0e-3, 1.0e-5]).astype(np.float32), state_init=0.0,
                state_covar_init=1.0, pad=1.0e-4, ecm_iters=3, ecm_rtol=1.0e9, inner_iters=5, nu=8.0, use_lambda=True,
                use_kappa=True, lambda_bounds=(0.25, 4.0), kappa_bounds=(5.0e-3, 5.0e3), fit_background=True,
                zero_center=False, use_nonnegative=True, neg_multiplier=1.0, penalties=bgo.penalties(8, 128.0),
                outer_passes=outer, min_outer=min_outer, shift_rtol=1.0e9, outer_nll_rtol=nll_rtol, patience=2,
                block_len_intervals=8)


def test_three_outer_passes_despite_tolerance_cpu_twin():
    from oracle import driver as twin

    data, munc = _three_pass_inputs()
    h = twin.run_consenrich_chain(data, munc, _three_pass_cfg(5))
    assert h["ecm_calls"] == 4 and h["passes"] == 3                        # test_core.py:4663-4664
    assert h["loop"][-1]["final_fixed_background_ecm"] is True             # :4665-4670
    assert h["stop_reason"] == "background_objective_inner_stable" and h["converged"]
    assert [r["outer_stable_iters"] for r in h["loop"][:-1]] == [0, 1, 2]    # the first pass has no previous objective
    assert twin.run_consenrich_chain(data, munc, _three_pass_cfg(2))["ecm_calls"] == 4          # :4672-4674
    assert twin.run_consenrich_chain(data, munc, _three_pass_cfg(1, min_outer=1))["ecm_calls"] == 2     # :4676-4683
    assert twin.run_consenrich_chain(data, munc, _three_pass_cfg(4, min_outer=1, nll_rtol=0.0))["ecm_calls"] > 1   # :4685-4692


def test_background_warm_start_recovers_a_constant_cpu_twin():
    from oracle import background as bgo
    from oracle import driver as twin

    n = 40
    data = np.full((3, n), 1.75, np.float32)
    munc = np.full((3, n), 0.05, np.float32)
    cfg = dict(pad=1.0e-4, zero_center=False, use_nonnegative=False, neg_multiplier=1.0, penalties=bgo.penalties(8, 128.0))
    est, _passes = twin.background_warm_start(data, munc, cfg)
    assert est.shape == (n,) and np.all(np.isfinite(est))
    assert np.max(np.abs(est - 1.75)) < 1.0e-4                             # test_core.py:4487-4489


def test_warm_start_summary_cpu_twin():
    from oracle import driver as twin

    rng = np.random.default_rng(41)                                        # inputs of test_core.py:4494-4508
    n, m = 24, 2
    bg0 = np.linspace(0.20, 0.35, n, dtype=np.float32)
    latent = np.linspace(0.0, 0.1, n, dtype=np.float32)
    data = (bg0[None, :] + latent[None, :] + 0.005 * rng.normal(size=(m, n))).astype(np.float32)
    munc = np.full((m, n), 0.08, np.float32)
    cfg = _three_pass_cfg(1)
    cfg.update(F=[[1.0, 1.0], [0.0, 1.0]], ecm_iters=1, ecm_rtol=1.0e-4, block_len_intervals=6, shift_rtol=5.0e-3,
               outer_nll_rtol=5.0e-5)
    h = twin.run_consenrich_chain(data, munc, cfg, initial_background=bg0)
    assert h["warm_start"]["background"] is True and h["warm_start"]["background_prepass"] is False     # :4528-4530
    h2 = twin.run_consenrich_chain(data, munc, cfg)
    assert h2["warm_start"]["background"] is False and h2["warm_start"]["background_prepass"] is True


# ---------------------------------------------------------------------------------------------------------------------
# the product on the GPU
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_device_background_update_weights_and_shift_gate_known_answer():
    """The device-resident update (`csr_batch_background_update`) on the reference's literal: data [[100, 0]], variances
    [[0.01, 1]], pad 0, current background [100, 0].  Its weight track must be the reference's [100, 1]: the three weighted
    RMS values it returns are checked against the reference's formula evaluated with those LITERAL weights on the proposal the
    device produced, and the gate is formed from them."""
    from consenrich_amd import _lib as L
    from consenrich_amd import driver as prod
    from consenrich_amd.batch import DeviceBatch, ModelParams

    with DeviceBatch(0) as b:
        b.configure(ModelParams(state_dim=2, pad=0.0, state_covar_init=1.0), 1, [2])
        b.upload(0, LIT_DATA, LIT_MUNC)
        b.set_background(0, SEED_G)
        b.stats()
        b.ecm(max_iters=1, inner_iters=1, rtol=1e-4)                      # n <= 5: filter + smoother only (pyx:7998-8129)
        info = b.background_update(1.0, 1.0, zero_center=False, use_nonnegative=False, use_initial=True)[0]
        proposal = b.download(0, "background_next")
    weights = np.asarray([100.0, 1.0])
    shift, prop, ref = prod.weighted_rms(weights, proposal, SEED_G)
    assert info["shift_rms"] == pytest.approx(shift, rel=1e-6)
    assert info["proposal_rms"] == pytest.approx(prop, rel=1e-6)
    assert info["reference_rms"] == pytest.approx(math.sqrt(100.0 * 100.0 * 100.0 / 101.0), rel=1e-6)      # RMS of the seed, literal
    g = prod.background_shift_gate(info["shift_rms"], info["proposal_rms"], info["reference_rms"], 0.05)
    assert g["background_shift_threshold"] == pytest.approx(0.05 * max(prop, ref, 1.0), rel=1e-6)


@pytest.mark.gpu
def test_device_background_warm_start_recovers_a_constant():
    from oracle import background as bgo      # penalties formula only (test infrastructure)
    from consenrich_amd.batch import DeviceBatch, ModelParams

    n = 40
    data = np.full((3, n), 1.75, np.float32)
    munc = np.full((3, n), 0.05, np.float32)
    lam_first, lam = bgo.penalties(8, 128.0)
    with DeviceBatch(0) as b:
        b.configure(ModelParams(state_dim=2), 3, [n])
        b.upload(0, data, munc)
        b.background_update(lam_first, lam, zero_center=False, use_nonnegative=False, use_initial=False, zero_state=True)
        est = b.download(0, "background_next")
    assert np.all(np.isfinite(est)) and np.max(np.abs(est - 1.75)) < 1.0e-4


@pytest.mark.gpu
def test_run_consenrich_batch_runs_three_passes_and_keeps_the_reference_keys():
    """test_core.py:4614-4693 through the product: pass counts 3 / 3 / 1 (+ the final phase), the reference's stop reason,
    the keys its tests read; the CPU twin agrees pass for pass."""
    from consenrich_amd import driver as prod
    from consenrich_amd.batch import DeviceBatch, ModelParams
    from oracle import driver as twin

    data, munc = _three_pass_inputs()

    def run(outer, min_outer=3, nll_rtol=1.0e9):
        c = _three_pass_cfg(outer, min_outer, nll_rtol)
        cfg = prod.FitConfig(penalties=c["penalties"], ecm_iters=3, ecm_rtol=1.0e9, inner_iters=5, nu=8.0, use_lambda=True,
                             use_kappa=True, outer_passes=outer, min_outer=min_outer, shift_rtol=1.0e9, outer_nll_rtol=nll_rtol)
        with DeviceBatch(0) as b:
            b.configure(ModelParams(state_dim=2, F=((1.0, 0.2), (0.0, 1.0)), Q0=((1.0e-3, 0.0), (0.0, 1.0e-5)),
                                    state_covar_init=1.0), 3, [32])
            b.upload(0, data, munc)
            fits, _ = prod.run_consenrich_batch(b, cfg, block_len_intervals=8, model_q0=np.diag([1e-3, 1e-5]), download=False)
        return fits[0], twin.run_consenrich_chain(data, munc, c)

    f, h = run(5)
    meta = f.post_process_noise_fit(prod.FitConfig(penalties=(1.0, 1.0), outer_passes=5))
    assert f.passes == 3 and len(f.loop_diagnostics) == 4                  # 4 ECM phases (test_core.py:4663)
    assert meta["actual_outer_passes"] == 3                                # :4664
    assert meta["fixed_background_ecm"][-1]["final_fixed_background_ecm"] is True        # :4665-4670
    assert meta["outer_stop_reason"] == "background_objective_inner_stable" == h["stop_reason"]
    assert meta["warm_start"]["background"] is False and meta["warm_start"]["background_prepass"] is True
    first = meta["fixed_background_ecm"][0]
    for key in ("background_shift", "background_shift_threshold", "background_shift_stable", "outer_stable_iters",
                "outer_patience_target", "outer_objective_per_cell", "outer_inner_ecm_converged"):
        assert key in first, key                                           # the keys test_core.py:4604-4609 reads
    assert [r["outer_stable_iters"] for r in meta["fixed_background_ecm"][:-1]] == [r["outer_stable_iters"] for r in h["loop"][:-1]]
    for a, b_ in zip(meta["fixed_background_ecm"][:-1], h["loop"][:-1]):
        assert a["background_shift"] == pytest.approx(b_["background_shift"], rel=1e-4, abs=1e-7)
        assert a["outer_objective_per_cell"] == pytest.approx(b_["outer_objective_per_cell"], rel=1e-6)
    f2, h2 = run(2)
    assert f2.passes == 3 == h2["passes"] and len(f2.loop_diagnostics) == 4      # :4672-4674
    f1, h1 = run(1, min_outer=1)
    assert f1.passes == 1 == h1["passes"] and len(f1.loop_diagnostics) == 2      # :4676-4683
    f4, h4 = run(4, min_outer=1, nll_rtol=0.0)
    assert len(f4.loop_diagnostics) > 1 and f4.passes == h4["passes"]             # :4685-4692
