"""SURVEY a12 behind the reference's own signature (`consenrich_amd.core_api.runConsenrich`).

CPU part: the signature IS the reference's (names, order, defaults -- the literal list of core.py:3861-3916 is data here);
`resolve_call` raises the reference's ValueErrors and maps the arguments; the three literal invocations the reference's own
contract tests make (test_core.py:3992-4110 outer-pass smoke, :4206-4257 level model, :6051-6090 adaptive process noise)
replayed on the CPU twin give the reference's tuple shapes, dtypes, keys and finite numbers -- including what that test reads of
every ECM phase record (:4111-4203); seven more reference-held cases (:1270 Q bounds, :1297 t_innerIters reaches the ECM, :1882
silence, :2850 interval-level precision, :4289 / :4356 / :6094 through the seams the reference spies on).
GPU part: the same invocations through the product on the device, against the twin."""
import inspect

import numpy as np
import pytest

from conftest import gpu_available

# core.py:3861-3916: (name, default) in order; REQ = no default
REQ = inspect.Parameter.empty
REFERENCE_SIGNATURE = [
    ("matrixData", REQ), ("matrixMunc", REQ), ("deltaF", REQ), ("minQ", REQ), ("maxQ", REQ),
    ("stateInit", REQ), ("stateCovarInit", REQ), ("boundState", REQ), ("stateLowerBound", REQ), ("stateUpperBound", REQ),
    ("blockLenIntervals", REQ), ("intervalSizeBP", None), ("projectStateDuringFiltering", False), ("pad", 1.0e-4),
    ("ECM_fixedBackgroundIters", 50), ("ECM_fixedBackgroundRtol", 1.0e-4), ("t_innerIters", 5), ("ECM_robustTNu", 8.0),
    ("ECM_useObsPrecisionReweighting", True), ("ECM_useProcessPrecisionReweighting", True), ("ECM_useAPN", False),
    ("ECM_zeroCenterBackground", False), ("ECM_outerIters", 3), ("ECM_minOuterIters", None),
    ("ECM_backgroundShiftRtol", 1.0e-3), ("ECM_outerNLLRtol", 1.0e-4), ("ECM_backgroundSmoothness", 1.0),
    ("fitBackground", True), ("useNonnegativeBackground", True), ("backgroundNegativePenaltyMultiplier", 1.0),
    ("returnScales", True), ("returnBackground", False), ("stateModel", "levelTrend"),
    ("processNoiseCalibration", "fixedDiagonal"), ("qSeedPriorLevel", 1.0e-5), ("processNoiseWarmupECMIters", 50),
    ("processNoiseWarmupOuterPasses", 2), ("observationPrecisionMultiplierMin", 0.25),
    ("observationPrecisionMultiplierMax", 4.0), ("processPrecisionMultiplierMin", 5.0e-3),
    ("processPrecisionMultiplierMax", 5.0e3), ("observationMask", None), ("initialBackground", None),
    ("initialObservationPrecision", None), ("initialProcessPrecision", None), ("initialProcessQ", None),
    ("trackOptimizationPath", False), ("returnPrecisionDiagnostics", False), ("returnDiagnostics", False),
    ("logIndentLevel", 0), ("logRunRole", None),
]
TRACK_KEYS = ("baseQLevel", "baseQTrend", "effectiveQLevel", "effectiveQTrend", "muncTrace", "preKappaQLevel",
              "preKappaQTrend", "processQScale", "sumGain0", "sumGain1")


def _case_outer_pass_smoke():
    """test_core.py:3992-4053: literal inputs and keyword arguments"""
    rng = np.random.default_rng(0)
    n, m = 64, 3
    grid = np.linspace(0.0, 2.0 * np.pi, n, dtype=np.float32)
    sig = np.sin(grid).astype(np.float32)
    bg = np.linspace(-0.25, 0.25, n, dtype=np.float32)
    data = np.vstack([sig + bg + 0.05 * rng.normal(size=n) - 0.04, sig + bg + 0.05 * rng.normal(size=n),
                      sig + bg + 0.05 * rng.normal(size=n) + 0.03]).astype(np.float32)
    munc = np.full((m, n), 0.2, dtype=np.float32)
    kw = dict(deltaF=0.1, minQ=1.0e-6, maxQ=1.0, stateInit=0.0, stateCovarInit=1.0, boundState=False, stateLowerBound=0.0,
              stateUpperBound=0.0, blockLenIntervals=8, intervalSizeBP=1000, ECM_fixedBackgroundIters=3, ECM_outerIters=2,
              processNoiseWarmupECMIters=1, trackOptimizationPath=True, returnPrecisionDiagnostics=True,
              returnDiagnostics=True)
    return data, munc, kw


def _case_level_smoke():
    """test_core.py:4206-4238"""
    rng = np.random.default_rng(100)
    n, m = 42, 3
    grid = np.linspace(0.0, 2.0 * np.pi, n, dtype=np.float32)
    sig = (0.4 * np.sin(grid) + 0.15 * np.cos(2.0 * grid)).astype(np.float32)
    data = np.vstack([sig + 0.04 * rng.normal(size=n) - 0.02, sig + 0.04 * rng.normal(size=n),
                      sig + 0.04 * rng.normal(size=n) + 0.03]).astype(np.float32)
    munc = np.full((m, n), 0.10, dtype=np.float32)
    kw = dict(stateModel="level", deltaF=-10.0, minQ=1.0e-4, maxQ=1.0, stateInit=0.0, stateCovarInit=1.0, boundState=False,
              stateLowerBound=0.0, stateUpperBound=0.0, blockLenIntervals=7, ECM_fixedBackgroundIters=1, ECM_outerIters=1,
              ECM_minOuterIters=1, ECM_useProcessPrecisionReweighting=True, ECM_useAPN=False, processNoiseWarmupECMIters=1,
              returnDiagnostics=True)
    return data, munc, kw


def _case_apn_smoke():
    """test_core.py:6051-6083"""
    rng = np.random.default_rng(123)
    n, m = 48, 3
    grid = np.linspace(0.0, 2.0 * np.pi, n, dtype=np.float32)
    sig = np.sin(grid).astype(np.float32)
    data = np.vstack([sig + 0.08 * rng.normal(size=n) - 0.03, sig + 0.08 * rng.normal(size=n),
                      sig + 0.08 * rng.normal(size=n) + 0.02]).astype(np.float32)
    munc = np.full((m, n), 0.15, dtype=np.float32)
    kw = dict(deltaF=0.1, minQ=1.0e-6, maxQ=0.5, stateInit=0.0, stateCovarInit=1.0, boundState=False, stateLowerBound=0.0,
              stateUpperBound=0.0, blockLenIntervals=8, ECM_fixedBackgroundIters=2, ECM_outerIters=1,
              ECM_useProcessPrecisionReweighting=True, ECM_useAPN=True, processNoiseCalibration="fixedDiagonal")
    return data, munc, kw


def _case_background_smoke():
    """test_core.py:4164-4181: the same matrices, one pass, the background returned"""
    data, munc, _ = _case_outer_pass_smoke()
    kw = dict(deltaF=0.1, minQ=1.0e-6, maxQ=1.0, stateInit=0.0, stateCovarInit=1.0, boundState=False, stateLowerBound=0.0,
              stateUpperBound=0.0, blockLenIntervals=8, ECM_fixedBackgroundIters=1, ECM_outerIters=1,
              processNoiseWarmupECMIters=1, returnBackground=True)
    return data, munc, kw


CASES = {"outer_pass_smoke": _case_outer_pass_smoke, "level_smoke": _case_level_smoke, "apn_smoke": _case_apn_smoke,
         "background_smoke": _case_background_smoke}


def _twin_call(data, munc, kw):
    from consenrich_amd import core_api
    import twin_core

    k = dict(kw)
    plan = core_api.resolve_call(data, munc, k.pop("deltaF"), k.pop("minQ"), k.pop("maxQ"), **k)
    fit, final = twin_core.twin_run(plan)
    return plan, core_api.assemble_result(plan, fit, final)


def _check_contract(name, out, data):
    """what the reference's own tests assert about the tuple (test_core.py:4055-4110, 4240-4257, 6085-6090)"""
    m, n = data.shape
    xs, Ps, resid, nis = out[:4]
    assert xs.shape == (n, 2) and Ps.shape == (n, 2, 2) and resid.shape == (n, m) and nis.shape == (n,)
    assert all(np.asarray(a).dtype == np.float32 for a in (xs, Ps, resid, nis))
    assert np.all(np.isfinite(xs)) and np.all(np.isfinite(Ps)) and np.all(np.isfinite(nis))
    if name == "outer_pass_smoke":
        assert len(out) == 7
        prec, diag = out[-2], out[-1]
        assert prec["precision_track_diagnostics"] is True
        tracks = prec["outputTracks"]
        assert tuple(sorted(tracks)) == TRACK_KEYS
        assert all(np.asarray(t).shape == (n,) for t in tracks.values())
        assert diag["final_forward_nis"] == pytest.approx(float(np.mean(nis)), rel=1e-6)
        qd = diag["process_q_diagnostics"]
        assert qd["effectiveQTraceMin"] <= qd["effectiveQTraceMedian"] <= qd["effectiveQTraceMax"]
        want = m * (0.2 + 0.0001)
        for k in ("min", "median", "max"):
            assert diag["observation_r_trace"][k] == pytest.approx(want)
        lam = np.asarray(prec["lambdaExp"], np.float64)
        kap = np.asarray(prec["processPrecExp"], np.float64)
        assert lam.shape == (n,) and kap.shape == (n,)
        np.testing.assert_allclose(tracks["muncTrace"], want / lam, rtol=2.0e-6, atol=2.0e-6)
        gs = diag["final_forward_gain_contig_summary"]
        assert all(len(gs[k]) == m for k in ("mean", "median", "sd", "iqr", "count")) and all(v >= 0.0 for v in gs["sd"])
        post = diag["post_process_noise_fit"]
        assert post["planned_outer_passes"] == 3 and post["requested_outer_passes"] == 2       # max(min 3, requested 2)
        assert "background_prior" not in diag
        # test_core.py:4111-4163: the per-phase records
        records = post["fixed_background_ecm"]
        assert records
        loop = [r for r in records if not r.get("final_fixed_background_ecm")]
        assert loop
        for key in ("observation_lambda_mean", "observation_lambda_median", "background_objective_per_cell",
                    "background_objective_change_per_cell", "background_objective_threshold_per_cell",
                    "observation_lambda_lower_bound_hits", "observation_lambda_upper_bound_hits",
                    "process_kappa_lower_bound_hits", "process_kappa_upper_bound_hits"):
            assert key in loop[-1], key
        for key in ("observation_lambda_lower_bound_hits", "observation_lambda_upper_bound_hits",
                    "process_kappa_lower_bound_hits", "process_kappa_upper_bound_hits"):
            assert 0.0 <= loop[-1][key] <= 1.0
        assert loop[-1]["relative_sign_change_per_kb"] >= 0.0
        last = records[-1]
        assert last["final_fixed_background_ecm"] is True
        assert "final_abs_rel_change" in last and "stable_iters" in last and "patience_target" in last
        assert last["relative_sign_change_per_kb"] >= 0.0
        assert "background_objective_per_cell" not in last and "outer_objective_per_cell" not in last
        rows = last["optimization_path"]
        assert rows and [r["iter"] for r in rows] == sorted(r["iter"] for r in rows)
        assert rows[0]["reset_iteration"] is True and rows[0]["change"] is None and rows[0]["threshold"] is None
        # test_core.py:4186-4203
        assert all(np.isfinite(float(r["objective_value"])) for r in rows) and all(r["objective_name"] == "nll" for r in rows)
        q_info = diag["process_noise_calibration"]
        assert q_info["processNoisePolicy"] == "fixedDiagonal" and q_info["processNoiseCalibrationStatus"] == "estimated"
        assert "blockMode" not in q_info and "process_q_calibration" not in diag
        want_level, want_trend = q_info["qSeedLevelFinal"], q_info["qSeedTrendFinal"]
        assert q_info["processNoiseCalibrationReason"] == "data_derived_q_estimate"
        assert q_info["preKappaQLevel"] == pytest.approx(want_level, rel=5.0e-6)
        assert q_info["preKappaQTrend"] == pytest.approx(want_trend, rel=5.0e-6)
        np.testing.assert_allclose(q_info["matrixQ0Final"], np.diag([want_level, want_trend]), rtol=5.0e-6)
        assert "processQScaleSummary" in q_info and "processQScale" not in q_info
        # the fit-level copies are the last phase's / the last in-loop record's (core.py:5611-5627)
        assert post["relative_sign_change_per_kb"] == last["relative_sign_change_per_kb"]
        assert post["background_objective_per_cell"] == loop[-1]["background_objective_per_cell"]
        assert post["background_objective_change_per_cell"] == loop[-1]["background_objective_change_per_cell"]
        assert loop[0]["background_objective_change_per_cell"] is None and loop[0]["background_objective_stable"] is False
    elif name == "level_smoke":
        assert len(out) == 6                                     # 4 + block map + run diagnostics
        np.testing.assert_array_equal(xs[:, 1], np.zeros(n, np.float32))
        np.testing.assert_array_equal(Ps[:, 0, 1], np.zeros(n, np.float32))
        np.testing.assert_array_equal(Ps[:, 1, 1], np.zeros(n, np.float32))
        assert out[-1]["state_model"] == "level"
    elif name == "background_smoke":                             # test_core.py:4182-4185
        assert len(out) == 6
        background = np.asarray(out[-1])
        assert background.shape == (n,) and np.isfinite(background).all()
    else:
        assert len(out) == 5                                     # returnScales only


def test_signature_is_the_references():
    from consenrich_amd import core_api

    for fn in (core_api.runConsenrich, core_api.resolve_call):
        ps = list(inspect.signature(fn).parameters.values())
        assert [(p.name, p.default) for p in ps] == REFERENCE_SIGNATURE
        assert all(p.kind == p.POSITIONAL_OR_KEYWORD for p in ps[:5]) and all(p.kind == p.KEYWORD_ONLY for p in ps[5:])


def test_argument_mapping_and_the_references_errors():
    from consenrich_amd import core_api

    data, munc, kw = _case_outer_pass_smoke()
    k = dict(kw)
    plan = core_api.resolve_call(data, munc, k.pop("deltaF"), k.pop("minQ"), k.pop("maxQ"), **k)
    assert plan.cfg.penalties == (16.0, 256.0)                     # L = 8: max(1, L^2/4), max(1, L^4/16) (core.py:7479-7491)
    assert plan.cfg.seed_q and plan.q0 is None                     # fixedDiagonal: Q0 from the data
    assert plan.cfg.use_lambda and plan.cfg.use_kappa and not plan.cfg.use_apn
    assert plan.cfg.outer_passes == 2 and plan.cfg.min_outer == 3 and plan.cfg.ecm_iters == 3
    assert np.asarray(plan.model.F, np.float32)[0, 1] == np.float32(0.1)
    base = dict(deltaF=1.0, minQ=1e-6, maxQ=1.0, stateInit=0.0, stateCovarInit=1.0, boundState=False, stateLowerBound=0.0,
                stateUpperBound=0.0, blockLenIntervals=8)

    def call(**over):
        a = {**base, **over}
        return core_api.resolve_call(a.pop("matrixData", data), a.pop("matrixMunc", munc), a.pop("deltaF"), a.pop("minQ"),
                                     a.pop("maxQ"), **a)

    fixed = call(processNoiseCalibration="fixed")
    np.testing.assert_array_equal(fixed.q0, np.diag([1e-4, 1e-4]).astype(np.float32))
    auto = call(processPrecisionMultiplierMin=-1.0)               # auto = (nu + d) / (2 nu) + 1e-4 (core.py:2231-2245)
    assert auto.model.kappa_bounds[0] == pytest.approx((8.0 + 2.0) / 16.0 + 1.0e-4)
    masked = call(observationMask=np.arange(data.shape[1]) % 5 != 0)
    assert np.all(masked.munc[:, ::5] == np.float32(1.0e30)) and np.all(masked.munc[:, 1:5] == np.float32(0.2))
    for over, text in ((dict(matrixData=data[:, :1], matrixMunc=munc[:, :1]), "need at least 2 intervals"),
                       (dict(matrixMunc=munc[:2]), "identical shapes"),
                       (dict(minQ=0.0), "`minQ` must be positive and finite"),
                       (dict(maxQ=float("nan")), "`maxQ` must not be NaN"),
                       (dict(pad=-1.0), "`pad` must be nonnegative and finite"),
                       (dict(deltaF=-1.0), "deltaF must be a positive finite fixed step size"),
                       (dict(t_innerIters=0), "t_innerIters must be a positive integer"),
                       (dict(t_innerIters=True), "t_innerIters must be a positive integer"),
                       (dict(intervalSizeBP=0), "intervalSizeBP must be positive when provided"),
                       (dict(observationPrecisionMultiplierMin=5.0), "`observationPrecisionMultiplierMax` must be >="),
                       (dict(processNoiseCalibration="fixed", minQ=1e-3), "`minQ` must not exceed the fixed process Q"),
                       (dict(initialBackground=np.zeros(3)), "`initialBackground` must have length 64"),
                       (dict(initialProcessQ=np.eye(3)), "`initialProcessQ` must have shape (2, 2)"),
                       (dict(observationMask=np.ones((2, 2), bool)), "observationMask must match matrixData shape"),
                       (dict(backgroundNegativePenaltyMultiplier=float("inf")), "must be finite or None")):
        with pytest.raises(ValueError, match=text.replace("(", r"\(").replace(")", r"\)")):
            call(**over)


    # accepted without effect, like the reference (core.py:4284, 4391 forward it to loops that never read it)
    assert call(projectStateDuringFiltering=True).cfg == call().cfg


# ---- three more contract cases the reference's tests hold (test_core.py:4289-4353, 4356-4467, 6094-6150), mirrored literally.
# The reference spies on `cconsenrich.cfixedBackgroundECM` and fakes `core._estimateInitialProcessNoiseFromData` with
# monkeypatch.setattr; the same seams exist here: the twin calls `oracle.oracle.cfixedBackgroundECM` and
# `oracle.qseed.estimate_initial_process_noise`, the product `DeviceBatch.ecm` and `DeviceBatch.qseed`.
SEED_Q = np.diag([2.0e-5, 3.0e-5]).astype(np.float32)                  # test_core.py:4370


def _case_initial_process_q():
    """test_core.py:4290-4337"""
    rng = np.random.default_rng(17)
    n, m = 30, 3
    grid = np.linspace(0.0, 1.0, n, dtype=np.float32)
    data = np.vstack([grid + 0.02 * rng.normal(size=n) + offset for offset in (-0.01, 0.0, 0.01)]).astype(np.float32)
    munc = np.full((m, n), 0.10, dtype=np.float32)
    kw = dict(deltaF=1.0, minQ=1.0e-4, maxQ=0.5, stateInit=0.0, stateCovarInit=1.0, boundState=False, stateLowerBound=0.0,
              stateUpperBound=0.0, blockLenIntervals=8, ECM_fixedBackgroundIters=2, ECM_useProcessPrecisionReweighting=True,
              ECM_useAPN=False, fitBackground=False, initialProcessQ=np.diag([1.0e-3, 1.0e-4]).astype(np.float32),
              returnDiagnostics=True)
    return data, munc, kw


def _case_fixed_diagonal_uses_data_q():
    """test_core.py:4357-4417"""
    rng = np.random.default_rng(3)
    n, m = 36, 3
    grid = np.linspace(0.0, 2.0 * np.pi, n, dtype=np.float32)
    data = np.vstack([np.sin(grid) + 0.2 * rng.normal(size=n) + offset for offset in (-0.1, 0.0, 0.1)]).astype(np.float32)
    munc = np.full((m, n), 0.05, dtype=np.float32)
    kw = dict(deltaF=0.2, minQ=1.0e-6, maxQ=1.0, processNoiseCalibration="fixedDiagonal", stateInit=0.0, stateCovarInit=1.0,
              boundState=False, stateLowerBound=0.0, stateUpperBound=0.0, blockLenIntervals=6, ECM_fixedBackgroundIters=1,
              ECM_outerIters=1, ECM_minOuterIters=1, fitBackground=False, ECM_useProcessPrecisionReweighting=True,
              initialProcessPrecision=np.linspace(0.5, 1.8, n, dtype=np.float32), qSeedPriorLevel=7.0e-6,
              returnPrecisionDiagnostics=True, returnDiagnostics=True)
    return data, munc, kw


def _case_always_runs_ecm_with_apn():
    """test_core.py:6097-6138"""
    rng = np.random.default_rng(321)
    n, m = 40, 3
    sig = np.cos(np.linspace(0.0, 2.0 * np.pi, n, dtype=np.float32))
    data = np.vstack([sig + 0.05 * rng.normal(size=n) - 0.02, sig + 0.05 * rng.normal(size=n),
                      sig + 0.05 * rng.normal(size=n) + 0.01]).astype(np.float32)
    munc = np.full((m, n), 0.2, dtype=np.float32)
    kw = dict(deltaF=0.1, minQ=1.0e-6, maxQ=0.5, stateInit=0.0, stateCovarInit=1.0, boundState=False, stateLowerBound=0.0,
              stateUpperBound=0.0, blockLenIntervals=8, ECM_useAPN=True, ECM_useProcessPrecisionReweighting=True,
              processNoiseCalibration="fixedDiagonal")
    return data, munc, kw


def _fake_seed(n):
    return SEED_Q.copy(), {"qSeedSource": "test", "qSeedTransitionCount": n - 1, "qSeedLevelFinal": float(SEED_Q[0, 0]),
                           "qSeedTrendFinal": float(SEED_Q[1, 1])}


def _check_initial_process_q(out, ecm_modes, kw):
    """test_core.py:4339-4353"""
    diagnostics = out[-1]
    assert ecm_modes == [(True, False)]                    # ONE phase, (process re-weighting, APN)
    q_info = diagnostics["process_noise_calibration"]
    assert q_info["processNoiseCalibrationStatus"] == "skipped"
    assert q_info["processNoiseCalibrationReason"] == "initial_process_q"
    assert q_info["warmStartProcessNoise"] == 1.0
    np.testing.assert_allclose(q_info["matrixQ0Final"], kw["initialProcessQ"], rtol=0.0, atol=0.0)
    assert diagnostics["post_process_noise_fit"]["warm_start"]["background"] is False


def _check_fixed_diagonal_uses_data_q(out, seed_calls, ecm_calls, data, munc):
    """test_core.py:4419-4467"""
    n = data.shape[1]
    prec, run = out[-2], out[-1]
    q_info = run["process_noise_calibration"]
    assert len(seed_calls) == 1
    call = seed_calls[0]
    if "matrixData" in call:                               # (the twin's seed estimator takes the matrices; the device's reads the resident ones)
        np.testing.assert_array_equal(call["matrixData"], data)
        np.testing.assert_array_equal(call["matrixMunc"], munc)
    assert call["pad"] == pytest.approx(1.0e-4)
    assert call["stateModel"] == "levelTrend"
    assert call["minQ"] == pytest.approx(1.0e-6) and call["maxQ"] == pytest.approx(1.0)
    assert call["deltaF"] == pytest.approx(0.2) and call["robustTNu"] == pytest.approx(8.0)
    assert call["qSeedPriorLevel"] == pytest.approx(7.0e-6)
    assert len(ecm_calls) == 1
    np.testing.assert_allclose(ecm_calls[0]["matrixQ0"], SEED_Q, rtol=0.0, atol=1.0e-10)
    assert ecm_calls[0]["useAPN"] is False and ecm_calls[0]["useProcPrec"] is True
    assert "process_noise_warmup_fit" not in run
    assert q_info["processNoisePolicy"] == "fixedDiagonal"
    assert q_info["processNoiseCalibrationStatus"] == "estimated"
    assert q_info["processNoiseCalibrationReason"] == "data_derived_q_estimate"
    assert q_info["qSeedSource"] == "test"
    assert q_info["validTransitionCount"] == n - 1
    np.testing.assert_allclose(q_info["matrixQ0Final"], SEED_Q, rtol=0.0, atol=1.0e-10)
    tracks = prec["outputTracks"]
    np.testing.assert_allclose(tracks["processQScale"], np.ones(n))
    np.testing.assert_allclose(tracks["baseQLevel"], SEED_Q[0, 0])
    np.testing.assert_allclose(tracks["baseQTrend"], SEED_Q[1, 1])
    np.testing.assert_allclose(tracks["preKappaQLevel"], SEED_Q[0, 0])
    np.testing.assert_allclose(tracks["preKappaQTrend"], SEED_Q[1, 1])
    assert np.any(np.abs(tracks["effectiveQLevel"] - tracks["preKappaQLevel"]) > 1.0e-9)


def _check_always_runs_ecm_with_apn(out, calls, data):
    """test_core.py:6140-6150"""
    m, n = data.shape
    assert calls
    assert calls[-1] == (True, False)                      # (APN, process re-weighting): kappa is off under APN (core.py:3974)
    xs, Ps, resid, nis, *_ = out
    assert xs.shape == (n, 2) and Ps.shape == (n, 2, 2) and resid.shape == (n, m) and nis.shape == (n,)
    assert np.all(np.isfinite(xs))


def _spy_twin(monkeypatch, n):
    """the reference's two seams on the twin: every cfixedBackgroundECM call recorded, the seed estimator faked"""
    from oracle import oracle as orc
    from oracle import qseed as oq

    ecm_calls, seed_calls = [], []
    original = orc.cfixedBackgroundECM

    def spy_ecm(*args, **kwargs):
        ecm_calls.append({"matrixQ0": np.asarray(kwargs["matrixQ0"], np.float64).copy(), "useAPN": bool(kwargs.get("ECM_useAPN", False)),
                          "useProcPrec": bool(kwargs["ECM_useProcessPrecisionReweighting"])})
        return original(*args, **kwargs)

    def fake_seed(_natives, **kwargs):
        seed_calls.append(kwargs)
        return _fake_seed(n)

    monkeypatch.setattr(orc, "cfixedBackgroundECM", spy_ecm)
    return ecm_calls, seed_calls, lambda: monkeypatch.setattr(oq, "estimate_initial_process_noise", fake_seed)


def test_initial_process_q_skips_the_seed_on_the_cpu_twin(monkeypatch):
    from oracle import qseed as oq

    data, munc, kw = _case_initial_process_q()
    ecm_calls, _seed_calls, _ = _spy_twin(monkeypatch, data.shape[1])

    def fail_seed(*_a, **_k):
        raise AssertionError("explicit process Q must bypass data estimation")

    monkeypatch.setattr(oq, "estimate_initial_process_noise", fail_seed)
    _, out = _twin_call(data, munc, kw)
    _check_initial_process_q(out, [(c["useProcPrec"], c["useAPN"]) for c in ecm_calls], kw)


def test_fixed_diagonal_uses_the_data_q_on_the_cpu_twin(monkeypatch):
    data, munc, kw = _case_fixed_diagonal_uses_data_q()
    ecm_calls, seed_calls, install_fake_seed = _spy_twin(monkeypatch, data.shape[1])
    install_fake_seed()
    _, out = _twin_call(data, munc, kw)
    _check_fixed_diagonal_uses_data_q(out, seed_calls, ecm_calls, data, munc)


def test_apn_always_runs_the_ecm_on_the_cpu_twin(monkeypatch):
    data, munc, kw = _case_always_runs_ecm_with_apn()
    ecm_calls, _s, _ = _spy_twin(monkeypatch, data.shape[1])
    _, out = _twin_call(data, munc, kw)
    _check_always_runs_ecm_with_apn(out, [(c["useAPN"], c["useProcPrec"]) for c in ecm_calls], data)


def _spy_device(monkeypatch, n, fake_seed=False, forbid_seed=False):
    """the same seams on the product: DeviceBatch.ecm recorded (with the base process noise the batch holds), DeviceBatch.qseed faked"""
    from consenrich_amd.batch import DeviceBatch

    ecm_calls, seed_calls, chain_q = [], [], []
    original_ecm, original_set_q = DeviceBatch.ecm, DeviceBatch.set_chain_q

    def spy_set_q(self, qs):
        chain_q[:] = [np.asarray(q, np.float64).copy() for q in qs]
        return original_set_q(self, qs)

    def spy_ecm(self, *args, **kwargs):
        q = chain_q[0] if chain_q else np.asarray(self.model.Q0, np.float64)
        ecm_calls.append({"matrixQ0": np.asarray(q, np.float64)[:2, :2].copy(), "useAPN": bool(kwargs.get("use_apn", False)),
                          "useProcPrec": bool(kwargs.get("use_kappa", True))})
        return original_ecm(self, *args, **kwargs)

    def fake(self, **kwargs):
        if forbid_seed:
            raise AssertionError("explicit process Q must bypass data estimation")
        seed_calls.append(kwargs)
        return [_fake_seed(n)]

    monkeypatch.setattr(DeviceBatch, "ecm", spy_ecm)
    monkeypatch.setattr(DeviceBatch, "set_chain_q", spy_set_q)
    if fake_seed or forbid_seed:
        monkeypatch.setattr(DeviceBatch, "qseed", fake)
    return ecm_calls, seed_calls


def _device_call(data, munc, kw):
    if not gpu_available():
        pytest.fail("GPU tests selected but no HIP device / library: the product has no CPU fallback")
    from consenrich_amd import core_api

    k = dict(kw)
    return core_api.runConsenrich(data, munc, k.pop("deltaF"), k.pop("minQ"), k.pop("maxQ"), **k)


@pytest.mark.gpu
def test_initial_process_q_skips_the_seed_on_the_device(monkeypatch):
    data, munc, kw = _case_initial_process_q()
    ecm_calls, _ = _spy_device(monkeypatch, data.shape[1], forbid_seed=True)
    out = _device_call(data, munc, kw)
    _check_initial_process_q(out, [(c["useProcPrec"], c["useAPN"]) for c in ecm_calls], kw)
    np.testing.assert_allclose(ecm_calls[0]["matrixQ0"], kw["initialProcessQ"], rtol=0.0, atol=0.0)


@pytest.mark.gpu
def test_fixed_diagonal_uses_the_data_q_on_the_device(monkeypatch):
    data, munc, kw = _case_fixed_diagonal_uses_data_q()
    ecm_calls, seed_calls = _spy_device(monkeypatch, data.shape[1], fake_seed=True)
    out = _device_call(data, munc, kw)
    _check_fixed_diagonal_uses_data_q(out, seed_calls, ecm_calls, data, munc)


@pytest.mark.gpu
def test_apn_always_runs_the_ecm_on_the_device(monkeypatch):
    data, munc, kw = _case_always_runs_ecm_with_apn()
    ecm_calls, _ = _spy_device(monkeypatch, data.shape[1])
    out = _device_call(data, munc, kw)
    _check_always_runs_ecm_with_apn(out, [(c["useAPN"], c["useProcPrec"]) for c in ecm_calls], data)
    _, ref = _twin_call(data, munc, kw)
    for a, b in zip(out[:3], ref[:3]):
        np.testing.assert_allclose(a.astype(np.float64), b, rtol=1e-4, atol=2e-5)


# the reference's run-diagnostics key set (core.py:5943-5999; post_process_noise_fit: core.py:3373-3417; process_noise_calibration:
# core.py:3133-3158 + 3046-3055 + 3088-3100 + 5731-5737; precision_reweighting_boundary_hits: diagnostics.py:233-247)
RUN_DIAGNOSTICS_KEYS = {
    "state_model", "final_nll", "final_forward_nis", "final_forward_gain_contig_summary",
    "precision_reweighting_boundary_hits", "process_noise_calibration", "post_process_noise_fit", "optimization_path_tracked",
    "process_precision_reweighting_requested", "process_precision_reweighting_effective",
    "process_precision_reweighting_disabled_by_apn", "adaptive_process_noise_effective", "process_q_policy",
    "process_q_diagnostics", "observation_r_trace"}
POST_FIT_KEYS = {
    "requested_outer_passes", "min_outer_passes", "planned_outer_passes", "actual_outer_passes", "outer_converged",
    "outer_stop_reason", "background_shift", "background_shift_threshold", "background_objective",
    "background_objective_per_cell", "background_objective_change_per_cell", "background_objective_threshold_per_cell",
    "background_objective_stable", "outer_nll", "outer_nll_change", "outer_nll_threshold", "outer_nll_stable", "outer_objective",
    "outer_objective_per_cell", "outer_objective_change_per_cell", "outer_objective_threshold_per_cell",
    "outer_objective_stable", "outer_effective_observation_count", "observation_lambda_lower_bound_hits",
    "observation_lambda_upper_bound_hits", "process_kappa_lower_bound_hits", "process_kappa_upper_bound_hits",
    "relative_sign_change_per_kb", "outer_stable_iters", "outer_patience_target", "inner_ecm_converged", "warm_start",
    "all_ecm_converged", "max_nll_increase_count", "fixed_background_ecm"}
CALIBRATION_KEYS = {
    "processNoisePolicy", "processNoiseCalibrationStatus", "processNoiseCalibrationReason", "stateModel", "preKappaQLevel",
    "preKappaQTrend", "rawTrendLevelRatio", "effectiveTrendLevelRatio", "logQLevel", "logQTrend", "usedInitialProcessQFallback",
    "matrixQ0Final", "warmStartProcessNoise", "globalScale", "windowCount", "validTransitionCount", "qScaleClampFraction",
    "qFloor", "qCap", "hitQLevelFloor", "hitQTrendFloor", "hitQLevelCap", "hitQTrendCap", "hitQFloor", "hitQCap",
    "qBoundaryStatus", "finiteDataCount", "positiveObservationVarianceCount", "activeObservationCount", "activeIntervalCount",
    "intervalTransitionCount", "activeAdjacentTransitionCount", "sameTrackAdjacentTransitionCount",
    "processNoiseCalibrationCanRun", "processNoiseCalibrationSkipReason", "resolvedMinQ", "resolvedMaxQ", "transitionCount",
    "processQScaleSummary"}
WARM_START_KEYS = {"background", "background_prepass", "background_prepass_source", "observation_precision", "process_precision"}


def test_run_diagnostics_carry_the_references_key_set():
    data, munc, kw = _case_outer_pass_smoke()
    _, out = _twin_call(data, munc, kw)
    diag = out[-1]
    n = data.shape[1]
    assert set(diag) == RUN_DIAGNOSTICS_KEYS
    assert set(diag["post_process_noise_fit"]) == POST_FIT_KEYS
    assert set(diag["post_process_noise_fit"]["warm_start"]) == WARM_START_KEYS
    assert CALIBRATION_KEYS <= set(diag["process_noise_calibration"])            # (+ the seed estimator's own qSeed* keys)
    hits = diag["precision_reweighting_boundary_hits"]
    assert set(hits) == {"observation", "process", "bounds"} and hits["bounds"] == {"observation": [0.25, 4.0], "process": [5.0e-3, 5.0e3]}
    assert hits["observation"]["enabled"] and hits["observation"]["total"] == n
    assert hits["process"]["enabled"] and hits["process"]["total"] == n - 1       # the first kappa multiplies nothing (skipFirst)
    post = diag["post_process_noise_fit"]
    assert post["warm_start"] == {"background": False, "background_prepass": True,
                                  "background_prepass_source": "asymmetric_irls_weighted_data",
                                  "observation_precision": False, "process_precision": False}
    rate = post["relative_sign_change_per_kb"]                                    # intervalSizeBP = 1000: changes per interval
    assert rate is not None and 0.0 <= rate <= 1.0
    cal = diag["process_noise_calibration"]
    assert cal["processNoiseCalibrationStatus"] == "estimated" and cal["intervalTransitionCount"] == n - 1
    assert cal["activeObservationCount"] == data.size and cal["processNoiseCalibrationCanRun"] is True
    import json
    json.dumps(diag["process_noise_calibration"])                                 # metadata: plain Python values only (core.py:5921-5942)


def test_host_restatements_of_the_diagnostics_helpers():
    """the reference's small pure-NumPy helpers restated in core_api, on hand-made inputs with answers worked out by hand"""
    from consenrich_amd import core_api as ca

    hits = ca.summarize_precision_boundary_hits(observationPrecision=[0.25, 1.0, 4.0, 4.0, np.nan], observationPrecisionMin=0.25,
                                                observationPrecisionMax=4.0, processPrecision=[5e-3, 5e-3, 1.0, 5e3],
                                                processPrecisionMin=5e-3, processPrecisionMax=5e3)
    assert hits["observation"] == {"enabled": True, "total": 4, "lower": 1, "upper": 2, "lower_fraction": 0.25, "upper_fraction": 0.5}
    assert hits["process"] == {"enabled": True, "total": 3, "lower": 1, "upper": 1, "lower_fraction": 1 / 3, "upper_fraction": 1 / 3}
    off = ca.summarize_precision_boundary_hits(observationPrecision=None, observationPrecisionMin=0.25, observationPrecisionMax=4.0,
                                               processPrecision=None, processPrecisionMin=5e-3, processPrecisionMax=5e3)
    assert off["observation"]["enabled"] is False and off["process"]["lower_fraction"] is None
    assert ca.precision_bound_hits([0.1, 0.25, 1.0, 5.0], 0.25, 4.0) == (0.5, 0.25)
    assert ca.precision_bound_hits([9.0, 0.25, 1.0, 5.0], 0.25, 4.0, skip_first=True) == (1 / 3, 1 / 3)
    assert ca.precision_bound_hits(None, 0.25, 4.0) == (None, None)
    # sign changes: state - weighted mean = [+1, -1, +1, +1, -0.001 (below 1 % of the mean magnitude: dropped), -1] -> 3 changes
    data = np.zeros((2, 6), np.float32)
    munc = np.ones((2, 6), np.float32)
    state = np.asarray([1.0, -1.0, 1.0, 1.0, -0.001, -1.0])
    assert ca.relative_sign_change_per_kb(state, data, munc, interval_size_bp=500) == pytest.approx(3 / 3.0)
    assert ca.relative_sign_change_per_kb(state, data, munc, interval_size_bp=None) is None
    bnd = ca.process_noise_q_boundary_diagnostics(np.diag([1e-6, 0.5]), "levelTrend", 1e-6, 0.5)
    assert bnd["qBoundaryStatus"] == "floor_and_cap" and bnd["hitQLevelFloor"] and bnd["hitQTrendCap"] and not bnd["hitQLevelCap"]
    assert ca.process_noise_q_boundary_diagnostics(np.diag([1e-3, 1e-4]), "levelTrend", 1e-6, -1.0)["qCap"] == float("inf")
    sup = ca.process_noise_calibration_support(np.asarray([[1.0, np.nan, 2.0, 3.0]]), np.asarray([[0.1, 0.1, 1e30, 0.1]]), 1e-4)
    assert sup["finiteDataCount"] == 3 and sup["activeObservationCount"] == 2 and sup["activeAdjacentTransitionCount"] == 0
    assert sup["processNoiseCalibrationSkipReason"] == "no_active_adjacent_transitions"


def test_sign_change_helpers_give_the_references_known_answers():
    """test_core.py:4007-4041: the literal inputs and answers the reference's contract test holds for `_signChangePerKB` and
    `_relativeSignChangePerKB`; :165-174 for the penalty split -- on the product's helpers AND on the twin's restatement."""
    from consenrich_amd import core_api as ca
    from oracle import passdiag as pdg

    for sign, rel in ((ca.sign_change_per_kb, lambda s_, d_, v_, bp, **k: ca.relative_sign_change_per_kb(s_, d_, v_, interval_size_bp=bp, **k)),
                      (pdg.sign_change_per_kb, pdg.relative_sign_change_per_kb)):
        assert sign(np.asarray([1.0, -0.005, 1.0], np.float32), 1000) == pytest.approx(0.0)
        assert sign(np.asarray([1.0, -0.02, 1.0], np.float32), 1000) == pytest.approx(2.0 / 3.0)
        state = np.asarray([2.0, 2.0, 2.0], np.float32)
        data = np.asarray([[1.0, 3.0, 1.0], [3.0, 1.0, 3.0]], np.float32)
        munc = np.asarray([[0.1, 0.1, 0.1], [10.0, 10.0, 10.0]], np.float32)
        assert rel(state, data, munc, 1000, pad=0.0) == pytest.approx(2.0 / 3.0)
        bg = np.asarray([0.1, -0.2, 0.3], np.float32)
        assert rel(state, data + bg[None, :], munc, 1000, background=bg, pad=0.0) == pytest.approx(2.0 / 3.0)
    target = np.linspace(-0.4, 0.7, 9)
    lam_first, lam_second = ca.background_penalties(3, 2.0)
    total, first, second = pdg.objective_penalty(target, lam_first, lam_second)
    assert total - first - second == pytest.approx(0.0, abs=1.0e-12)
    assert first == pytest.approx(0.5 * lam_first * 8 * (1.1 / 8) ** 2) and second == pytest.approx(0.0, abs=1e-20)
    assert (lam_first, lam_second) == (max(1.0, 2.0 * 9 / 4.0), max(1.0, 2.0 * 81 / 16.0))       # core.py:7480-7493


def test_host_diagnostics_in_ranges_equal_the_whole_matrix_formulas():
    """The per-call entry evaluates its host-side summaries range by range / row by row on a thread pool; on a matrix large enough
    to take those paths (several 2^18-bin ranges) the results must be what the reference's whole-matrix expressions give."""
    from consenrich_amd import core_api as ca

    rng = np.random.default_rng(11)
    m, n = 3, (1 << 18) * 2 + 12345
    data = rng.normal(size=(m, n)).astype(np.float32)
    munc = (0.2 * np.exp(rng.normal(0, 0.3, size=(m, n)))).astype(np.float32)
    data[0, 5:9] = np.nan
    munc[1, (1 << 18) - 2:(1 << 18) + 3] = np.float32(1.0e30)            # masked cells across a range boundary
    munc[:, 1 << 18] = np.float32(1.0e30)                                # ... and a wholly inactive bin ON the boundary
    munc[2, 100] = np.float32(-1.0)                                      # a non-positive variance
    sup = ca.process_noise_calibration_support(data, munc, 1.0e-4)
    d64, m64 = data.astype(np.float64), munc.astype(np.float64)
    ov = m64 + 1.0e-4
    unmasked = np.isfinite(m64) & (m64 < 0.5e30)
    active = np.isfinite(d64) & unmasked & np.isfinite(ov) & (ov > 0.0)
    iv = active.any(axis=0)
    assert sup["finiteDataCount"] == int(np.isfinite(d64).sum()) and sup["activeObservationCount"] == int(active.sum())
    assert sup["positiveObservationVarianceCount"] == int((unmasked & np.isfinite(ov) & (ov > 0.0)).sum())
    assert sup["activeIntervalCount"] == int(iv.sum()) and sup["activeAdjacentTransitionCount"] == int((iv[1:] & iv[:-1]).sum())
    assert sup["sameTrackAdjacentTransitionCount"] == int((active[:, 1:] & active[:, :-1]).any(axis=0).sum())
    # sign changes: the reference's expressions on the whole matrix (core.py:2647-2700)
    state = rng.normal(size=n)
    bg = (0.1 * np.sin(np.arange(n) / 1000.0)).astype(np.float32)
    tot, ws = np.zeros(n), np.zeros(n)
    for j in range(m):
        den = m64[j] + 1.0e-4
        ok = np.isfinite(d64[j]) & np.isfinite(den) & (den > 0.0)
        w = 1.0 / np.maximum(den[ok], 1.0e-12)
        tot[ok] += (d64[j][ok] - bg.astype(np.float64)[ok]) * w
        ws[ok] += w
    arr = state - np.where(ws > 0, tot / np.where(ws > 0, ws, 1.0), np.nan)
    fin = arr[np.isfinite(arr)]
    fin = fin[np.abs(fin) >= 0.01 * np.mean(np.abs(fin))]
    sg = np.sign(fin)
    want = float(np.count_nonzero(sg[1:] * sg[:-1] < 0.0)) / (n * 25 / 1000.0)
    assert ca.relative_sign_change_per_kb(state, data, munc, interval_size_bp=25, background=bg, pad=1.0e-4) == want
    # rows through the pool come back in order
    assert ca._map_rows(lambda r: float(r[0]), munc) == [float(munc[j, 0]) for j in range(m)]
    # background-fit objective (core.py:4540-4606): range by range == the whole-matrix restatement of the twin
    from oracle import passdiag as pdg
    lam = np.exp(rng.normal(0, 0.8, size=n)).astype(np.float32)                    # some outside [0.25, 4]
    level = rng.normal(size=n).astype(np.float32)
    g = (bg - np.float32(0.05)).astype(np.float32)                                  # partly negative
    kwo = dict(pad=1.0e-4, lambda_bounds=(0.25, 4.0), penalties=(16.0, 256.0), use_nonnegative=True, negative_penalty_multiplier=1.5)
    assert np.isnan(ca.background_fit_objective(data, munc, level, lam, g, **kwo)["background_objective"])    # NaN cells: like np.sum there
    clean = np.nan_to_num(data)
    got = ca.background_fit_objective(clean, munc, level, lam, g, **kwo)
    inv, res = pdg.update_matrices(clean, munc, level, lam, 1.0e-4, (0.25, 4.0))
    want_o = pdg.background_fit_objective(res, inv, g, 16.0, 256.0, True, 1.5)
    assert set(got) == set(want_o)
    for key in want_o:
        assert got[key] == pytest.approx(want_o[key], rel=1e-12), key
    assert got["background_effective_observation_count"] == want_o["background_effective_observation_count"]
    assert got["background_negative_penalty"] > 0.0 and got["background_second_difference_penalty"] > 0.0
    for args in ((lam, 0.25, 4.0, False), (lam, 0.25, 4.0, True)):
        assert ca.multiplier_summary(*args) == (pdg.kappa_summary(*args[:3]) if args[3] else pdg.lambda_summary(*args[:3]))


def test_an_initial_lambda_weights_the_background_warm_start():
    """core.py:4663-4676: the background prepass is weighted by the INITIAL observation precision (clipped to its bounds) when
    one is given and no initial background is -- a different start than the unweighted prepass."""
    from oracle import driver as odrv

    data, munc, kw = _case_outer_pass_smoke()
    n = data.shape[1]
    munc = (munc * np.linspace(0.5, 2.0, 3, dtype=np.float32)[:, None]).astype(np.float32)
    lam0 = np.where(np.arange(n) % 2 == 0, 0.1, 9.0).astype(np.float32)         # outside the bounds on both sides
    k = dict(kw, initialObservationPrecision=lam0)
    plan, out = _twin_call(data, munc, k)
    assert plan.initial_lambda.min() == np.float32(0.25) and plan.initial_lambda.max() == np.float32(4.0)
    ws = out[-1]["post_process_noise_fit"]["warm_start"]
    assert ws["observation_precision"] is True and ws["background_prepass"] is True and ws["background"] is False
    cfg = __import__("twin_core").twin_cfg(plan)
    with_lam, _ = odrv.background_warm_start(data, munc, cfg, plan.initial_lambda)
    without, _ = odrv.background_warm_start(data, munc, cfg)
    assert float(np.abs(with_lam - without).max()) > 1e-3
    # the formula itself (core.py:2842-2857): weights = lambda / max(munc + pad, 1e-8), residual = data
    w = (lam0.clip(0.25, 4.0)[None, :] / np.maximum(munc + np.float32(1e-4), np.float32(1e-8))).astype(np.float32)
    from oracle import background as bgo
    want, _ = bgo.solve_background(w.sum(axis=0, dtype=np.float64), np.einsum("ij,ij->j", w, data, dtype=np.float64), 0,
                                   zero_center=False, use_nonnegative=True, multiplier=1.0, initial=None,
                                   penalties_override=cfg["penalties"], return_info=True)
    np.testing.assert_allclose(with_lam, want, rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
def test_an_initial_lambda_weights_the_background_warm_start_on_the_device():
    """the device prepass hard-coded use_lambda=False.  Device == twin with an initial lambda and no initial
    background (the prepass runs), and the warm-start summary says what was uploaded."""
    data, munc, kw = _case_outer_pass_smoke()
    n = data.shape[1]
    munc = (munc * np.linspace(0.5, 2.0, 3, dtype=np.float32)[:, None]).astype(np.float32)
    lam0 = np.where(np.arange(n) % 2 == 0, 0.1, 9.0).astype(np.float32)
    k = dict(kw, initialObservationPrecision=lam0, initialProcessPrecision=np.full(n, 1.3, np.float32), returnBackground=True)
    out = _device_call(data, munc, k)
    _, ref = _twin_call(data, munc, k)
    ws = out[-1]["post_process_noise_fit"]["warm_start"]
    assert ws == ref[-1]["post_process_noise_fit"]["warm_start"]
    assert ws["observation_precision"] is True and ws["process_precision"] is True and ws["background_prepass"] is True
    pg, pr = out[-1]["post_process_noise_fit"], ref[-1]["post_process_noise_fit"]
    assert [r["iters_done"] for r in pg["fixed_background_ecm"]] == [r["iters_done"] for r in pr["fixed_background_ecm"]]
    np.testing.assert_allclose(out[5].astype(np.float64), ref[5], rtol=1e-4, atol=2e-5)        # the fitted background
    np.testing.assert_allclose(out[0].astype(np.float64), ref[0], rtol=1e-4, atol=2e-5)
    assert pg["relative_sign_change_per_kb"] == pytest.approx(pr["relative_sign_change_per_kb"], abs=2.0 / n)
    assert out[-1]["precision_reweighting_boundary_hits"] == ref[-1]["precision_reweighting_boundary_hits"]


@pytest.mark.parametrize("name", sorted(CASES))
def test_reference_contract_cases_on_the_cpu_twin(name):
    data, munc, kw = CASES[name]()
    _, out = _twin_call(data, munc, kw)
    _check_contract(name, out, data)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_reference_contract_cases_on_the_device_match_the_twin(name):
    """The literal invocations through `core_api.runConsenrich` (device-resident fit) == the CPU twin: same tuple length,
    the discrete history (passes, ECM iterations, stop reason) equal, arrays within the parity tolerance."""
    if not gpu_available():
        pytest.fail("GPU tests selected but no HIP device / library: the product has no CPU fallback")
    from consenrich_amd import core_api

    data, munc, kw = CASES[name]()
    k = dict(kw)
    out = core_api.runConsenrich(data, munc, k.pop("deltaF"), k.pop("minQ"), k.pop("maxQ"), **k)
    _check_contract(name, out, data)
    _, ref = _twin_call(data, munc, kw)
    assert len(out) == len(ref)
    for i, (a, b) in enumerate(zip(out, ref)):
        if isinstance(a, np.ndarray):
            assert a.dtype == b.dtype and a.shape == b.shape, i
            if a.dtype.kind == "f":
                lvl = 1.0 if i != 0 else np.maximum(np.abs(b[:, :1]).astype(np.float64), 1.0)
                tol = 1e-4 if i != 3 else 5e-2                   # NIS amplifies one ulp of the level (close_mostly elsewhere)
                assert float((np.abs(a.astype(np.float64) - b) / (lvl * (np.abs(b) if i != 0 else 1.0) + 1e-3)).max()) <= tol, (name, i)
            else:
                np.testing.assert_array_equal(a, b)
    if kw.get("returnDiagnostics"):
        dg, dr = out[-1], ref[-1]
        pg, pr = dg["post_process_noise_fit"], dr["post_process_noise_fit"]
        for key in ("planned_outer_passes", "actual_outer_passes", "outer_stop_reason", "outer_converged"):
            assert pg[key] == pr[key], (name, key, pg[key], pr[key])
        assert [r["iters_done"] for r in pg["fixed_background_ecm"]] == [r["iters_done"] for r in pr["fixed_background_ecm"]]
        _compare_phase_records(pg["fixed_background_ecm"], pr["fixed_background_ecm"], data.shape[1], name)
        for key in ("background_objective", "background_objective_per_cell", "background_objective_change_per_cell",
                    "background_objective_threshold_per_cell", "relative_sign_change_per_kb"):
            assert (pg[key] is None) == (pr[key] is None), (name, key)
            if pr[key] is not None:
                assert pg[key] == pytest.approx(pr[key], rel=1e-3, abs=2.0 / data.shape[1]), (name, key)
        assert pg["background_objective_stable"] == pr["background_objective_stable"]
        assert dg["final_nll"] == pytest.approx(dr["final_nll"], rel=1e-6)
        assert dg["process_q_diagnostics"]["policy"] == dr["process_q_diagnostics"]["policy"]
        np.testing.assert_allclose(dg["process_q_diagnostics"]["baseQLevel"], dr["process_q_diagnostics"]["baseQLevel"], rtol=1e-6)


# ---- four more reference-held cases (test_core.py:1270-1295, 1297-1351, 1882-1921, 2850-2911), mirrored literally ----------------
@pytest.mark.parametrize("q_kwargs, message", [({"minQ": 2.0e-4, "maxQ": 1.0}, "minQ"), ({"minQ": 1.0e-6, "maxQ": 5.0e-5}, "maxQ")])
def test_fixed_process_q_respects_q_bounds(q_kwargs, message):
    """test_core.py:1270-1295: the fixed process Q (1e-4) must lie within [minQ, maxQ] -- raised before anything touches a device"""
    from consenrich_amd import core_api

    kwargs = dict(deltaF=0.1, stateInit=0.0, stateCovarInit=1.0, boundState=False, stateLowerBound=0.0, stateUpperBound=0.0,
                  blockLenIntervals=2, ECM_fixedBackgroundIters=1, processNoiseCalibration="fixed", **q_kwargs)
    with pytest.raises(ValueError, match=message):
        core_api.runConsenrich(np.zeros((2, 4), np.float32), np.ones((2, 4), np.float32), **kwargs)


def _case_t_inner_iters():
    """test_core.py:1319-1336"""
    return dict(deltaF=0.1, minQ=1.0e-4, maxQ=1.0, stateInit=0.0, stateCovarInit=1.0, boundState=False, stateLowerBound=0.0,
                stateUpperBound=0.0, blockLenIntervals=2, ECM_fixedBackgroundIters=1, ECM_outerIters=1, ECM_minOuterIters=1,
                ECM_useObsPrecisionReweighting=False, ECM_useProcessPrecisionReweighting=False, fitBackground=False,
                processNoiseCalibration="fixed")


def test_t_inner_iters_reaches_the_ecm_on_the_cpu_twin(monkeypatch):
    """test_core.py:1297-1351: a non-integer t_innerIters raises, an integer one is what the ECM native receives"""
    from consenrich_amd import core_api
    from oracle import oracle as orc

    seen, original = [], orc.cfixedBackgroundECM

    def spy(*args, **kwargs):
        seen.append(kwargs["t_innerIters"])
        return original(*args, **kwargs)

    monkeypatch.setattr(orc, "cfixedBackgroundECM", spy)
    data, munc, kw = np.zeros((2, 5), np.float32), np.ones((2, 5), np.float32), _case_t_inner_iters()
    with pytest.raises(ValueError, match="t_innerIters"):
        core_api.runConsenrich(data, munc, **kw, t_innerIters=1.5)
    _twin_call(data, munc, dict(kw, t_innerIters=4))
    assert seen == [4]


@pytest.mark.gpu
def test_t_inner_iters_reaches_the_ecm_on_the_device(monkeypatch):
    from consenrich_amd.batch import DeviceBatch

    seen, original = [], DeviceBatch.ecm

    def spy(self, *args, **kwargs):
        seen.append(kwargs["inner_iters"])
        return original(self, *args, **kwargs)

    monkeypatch.setattr(DeviceBatch, "ecm", spy)
    out = _device_call(np.zeros((2, 5), np.float32), np.ones((2, 5), np.float32), dict(_case_t_inner_iters(), t_innerIters=4))
    assert seen == [4] and out[0].shape == (5, 2)


def _case_silent():
    """test_core.py:1882-1916"""
    rng = np.random.default_rng(11)
    n = 18
    grid = np.linspace(0.0, 1.0, n, dtype=np.float32)
    data = np.vstack([grid + 0.01 * rng.normal(size=n), grid + 0.02 * rng.normal(size=n) + 0.05]).astype(np.float32)
    munc = np.full_like(data, 0.1, dtype=np.float32)
    kw = dict(deltaF=0.1, minQ=1.0e-4, maxQ=1.0, stateInit=0.0, stateCovarInit=1.0, boundState=False, stateLowerBound=0.0,
              stateUpperBound=0.0, blockLenIntervals=6, pad=1.0e-4, ECM_fixedBackgroundIters=1, ECM_fixedBackgroundRtol=0.0,
              ECM_outerIters=1, ECM_minOuterIters=1, ECM_backgroundShiftRtol=0.0, ECM_outerNLLRtol=0.0, fitBackground=False,
              processNoiseCalibration="fixed", returnScales=True)
    return data, munc, kw


def test_a_default_call_is_silent_on_the_cpu_twin(capfd):
    """test_core.py:1882-1921: nothing on stdout / stderr from a default call"""
    data, munc, kw = _case_silent()
    _twin_call(data, munc, kw)
    captured = capfd.readouterr()
    assert captured.out == "" and captured.err == ""


@pytest.mark.gpu
def test_a_default_call_is_silent_on_the_device(capfd):
    data, munc, kw = _case_silent()
    _device_call(data, munc, kw)                     # (the library is loaded by now: an earlier test of this module ran on the device)
    capfd.readouterr()
    out = _device_call(data, munc, kw)
    captured = capfd.readouterr()
    assert captured.out == "" and captured.err == "" and np.all(np.isfinite(out[0]))


def _case_interval_level_precision():
    """test_core.py:2850-2858"""
    from consenrich_amd import core_api

    n, m = 10, 2
    return dict(matrixData=np.zeros((m, n), np.float32), matrixPluginMuncInit=np.full((m, n), 0.2, np.float32),
                matrixF=core_api.construct_matrix_f(0.1).astype(np.float32), matrixQ0=np.diag([1.0e-4, 1.0e-4]).astype(np.float32),         # constructMatrixQ(minDiagQ=1e-4): diag (core.py:3781)
                intervalToBlockMap=np.zeros(n, np.int32), blockCount=1, stateInit=0.0, stateCovarInit=1.0), n, m


def _check_interval_level_precision(mod):
    """test_core.py:2860-2891 on a module with the reference's callables: the observation precision is one value per INTERVAL"""
    base, n, m = _case_interval_level_precision()
    out = mod.cfixedBackgroundECM(**base, ECM_fixedBackgroundIters=1, ECM_fixedBackgroundRtol=0.0, ECM_useObsPrecisionReweighting=True,
                                  ECM_useProcessPrecisionReweighting=False, returnIntermediates=True, t_innerIters=1)
    assert np.asarray(out[6]).shape == (n,)
    with pytest.raises(ValueError):
        mod.cforwardPass(**base, lambdaExp=np.ones((m, n), np.float32), ECM_useObsPrecisionReweighting=True)


def test_observation_precision_is_interval_level_only_on_the_oracle_and_in_the_call():
    """test_core.py:2850-2911"""
    from consenrich_amd import core_api
    from oracle import oracle as orc

    _check_interval_level_precision(orc)
    base, n, m = _case_interval_level_precision()
    with pytest.raises(ValueError):                                     # test_core.py:2893-2910
        core_api.runConsenrich(base["matrixData"], base["matrixPluginMuncInit"], deltaF=0.1, minQ=1.0e-4, maxQ=0.5, stateInit=0.0,
                               stateCovarInit=1.0, boundState=False, stateLowerBound=0.0, stateUpperBound=0.0, blockLenIntervals=4,
                               ECM_fixedBackgroundIters=1, initialObservationPrecision=np.ones((1, n), np.float32))


@pytest.mark.gpu
def test_observation_precision_is_interval_level_only_on_the_device():
    if not gpu_available():
        pytest.fail("GPU tests selected but no HIP device / library: the product has no CPU fallback")
    from consenrich_amd import cconsenrich as amd

    _check_interval_level_precision(amd)


@pytest.mark.gpu
def test_a_kept_context_gives_what_fresh_contexts_give(monkeypatch):
    """`core_api` keeps one device context per GPU between calls (the reference's CLI calls once per chromosome).  A sequence
    of calls of different shapes, state models and options through the kept context returns, array for array and BIT FOR BIT,
    what the same calls return on a fresh context each -- nothing of a call survives `csr_batch_configure`."""
    if not gpu_available():
        pytest.fail("GPU tests selected but no HIP device / library: the product has no CPU fallback")
    from consenrich_amd import core_api

    variants, data, munc = _variant_cases()
    calls = [(data, munc, _case_outer_pass_smoke()[2]), _case_level_smoke(), (data, munc, variants["fixed_q_and_mask"]),
             _case_apn_smoke(), (data, munc, variants["given_q_and_warm_starts"]), (data, munc, _case_outer_pass_smoke()[2])]
    big = np.random.default_rng(8)
    n_big = 5000
    calls.insert(2, ((np.sin(np.arange(n_big) / 50.0)[None, :] + 0.2 * big.normal(size=(6, n_big))).astype(np.float32),
                     np.full((6, n_big), 0.1, np.float32), dict(_case_background_smoke()[2], blockLenIntervals=50)))

    def run_all():
        outs = []
        for d_, v_, kw in calls:
            k = dict(kw)
            outs.append(core_api.runConsenrich(d_, v_, k.pop("deltaF"), k.pop("minQ"), k.pop("maxQ"), **k))
        return outs

    core_api.release_device()
    monkeypatch.setenv("CONSENRICH_AMD_CORE_API_KEEP_CONTEXT", "0")
    fresh = run_all()
    assert not core_api._KEPT
    monkeypatch.setenv("CONSENRICH_AMD_CORE_API_KEEP_CONTEXT", "1")
    kept = run_all()
    assert list(core_api._KEPT) == [0]
    for i, (a, b) in enumerate(zip(kept, fresh)):
        assert len(a) == len(b)
        for j, (x, y) in enumerate(zip(a, b)):
            if isinstance(x, np.ndarray):
                np.testing.assert_array_equal(x, y, err_msg=f"call {i} item {j}")
            elif isinstance(x, dict) and "outputTracks" in x:
                for key in TRACK_KEYS:
                    np.testing.assert_array_equal(x["outputTracks"][key], y["outputTracks"][key], err_msg=f"call {i} {key}")
            elif isinstance(x, dict):
                assert x["post_process_noise_fit"] == y["post_process_noise_fit"] and x["final_nll"] == y["final_nll"], f"call {i}"
    # a failing call drops its context; the next call starts a new one
    with pytest.raises(Exception):
        with core_api._Context(0) as b:
            raise RuntimeError("boom")
    assert not core_api._KEPT
    core_api.release_device()


def _compare_phase_records(got, want, n, name):
    """every key the twin's record of an ECM phase carries is in the device's, with the same discrete values and close numbers
    (fractions of n bins: within two bins; differences of NLLs: absolute)"""
    assert len(got) == len(want)
    for rg, rr in zip(got, want):
        assert set(rr) <= set(rg), (name, sorted(set(rr) - set(rg)))
        for key, w in rr.items():
            g = rg[key]
            if key == "optimization_path":
                assert [r["iter"] for r in g] == [r["iter"] for r in w]
                assert [set(r) for r in g] == [set(r) for r in w]
                assert [(r["reset_iteration"], r["converged"], r["stable_iters"]) for r in g] == \
                       [(r["reset_iteration"], r["converged"], r["stable_iters"]) for r in w]
                np.testing.assert_allclose([r["objective_value"] for r in g], [r["objective_value"] for r in w], rtol=1e-6)
            elif w is None or isinstance(w, (bool, str, int)):
                assert g == w, (name, key, g, w)
            else:
                assert g is not None, (name, key)
                assert g == pytest.approx(w, rel=1e-3, abs=max(2.0 / n, 1e-5)), (name, key, g, w)


def _case_records_without_background_fit():
    data, munc, kw = _case_outer_pass_smoke()
    return data, munc, dict(kw, fitBackground=False, returnPrecisionDiagnostics=False)


def _check_records_without_background_fit(out):
    """core.py:4994-5040: ONE phase, its record carries the phase summaries and a zero shift, no background objective"""
    post = out[-1]["post_process_noise_fit"]
    records = post["fixed_background_ecm"]
    assert len(records) == 1 and post["outer_stop_reason"] == "fit_background_false" and post["outer_converged"] is True
    r = records[0]
    assert r["background_shift"] == 0.0 and r["background_shift_threshold"] == 0.0 and r["background_shift_stable"] is True
    assert "background_objective_per_cell" not in r and "final_fixed_background_ecm" not in r
    assert r["observation_lambda_mean"] is not None and r["process_kappa_median"] is not None
    assert r["relative_sign_change_per_kb"] >= 0.0 and r["outer_stable_iters"] == 0 and r["outer_patience_target"] == 2
    assert r["optimization_path"] and r["optimization_path"][0]["reset_iteration"] is True
    assert post["background_objective"] is None and post["background_objective_stable"] is False
    assert post["relative_sign_change_per_kb"] == r["relative_sign_change_per_kb"]


def test_phase_record_without_a_background_fit_on_the_cpu_twin():
    data, munc, kw = _case_records_without_background_fit()
    _, out = _twin_call(data, munc, kw)
    _check_records_without_background_fit(out)


@pytest.mark.gpu
def test_phase_record_without_a_background_fit_on_the_device_matches_the_twin():
    data, munc, kw = _case_records_without_background_fit()
    out = _device_call(data, munc, kw)
    _check_records_without_background_fit(out)
    _, ref = _twin_call(data, munc, kw)
    _compare_phase_records(out[-1]["post_process_noise_fit"]["fixed_background_ecm"],
                           ref[-1]["post_process_noise_fit"]["fixed_background_ecm"], data.shape[1], "no background fit")


@pytest.mark.gpu
@pytest.mark.parametrize("state_dim", [2, 1])
def test_phase_tracks_equal_the_whole_matrix_restatement(state_dim):
    """`csr_batch_phase_tracks` (the per-bin tracks behind the sign-change rate and the background-fit objective) against the
    twin's whole-matrix NumPy restatement of core.py:2656-2696 / 4546-4552 on the SAME fit (smoothed level, multipliers,
    current background and proposal downloaded from the device): the weighted-mean track bit for bit (same float64 row-by-row
    accumulation), the per-bin objective sums to float64 rounding, the cell counts exactly.  Two chains (the second one's bins
    start at an offset), masked cells, a nonzero current background; the multipliers are the ECM's own (within their bounds)."""
    if not gpu_available():
        pytest.fail("GPU tests selected but no HIP device / library: the product has no CPU fallback")
    from consenrich_amd import _lib as L
    from consenrich_amd.batch import DeviceBatch, ModelParams
    from oracle import passdiag as pdg

    rng = np.random.default_rng(3)
    m, lens = 5, [3001, 777]
    mp = ModelParams(state_dim=state_dim)
    with DeviceBatch(0) as b:
        b.configure(mp, m, lens)
        host = []
        for c, n in enumerate(lens):
            sig = np.sin(np.arange(n) / 40.0)
            data = (sig[None, :] + 0.3 * rng.normal(size=(m, n))).astype(np.float32)
            munc = (0.2 * np.exp(rng.normal(0, 0.4, size=(m, n)))).astype(np.float32)
            munc[1, 10:20] = np.float32(1.0e30)
            munc[:, 50] = np.float32(1.0e30)
            b.upload(c, data, munc)
            b.set_background(c, (0.05 * np.cos(np.arange(n) / 90.0)).astype(np.float32))
            host.append((data, munc))
        b.stats()
        b.ecm(max_iters=2, inner_iters=2, rtol=1e-6, nu=8.0, use_lambda=True, use_kappa=True)
        info = b.background_update(16.0, 256.0, use_lambda=True, use_initial=True)
        b.export(L.EXPORT_SMOOTH | L.EXPORT_MULT)
        pad32 = float(np.float32(mp.pad))
        for c, n in enumerate(lens):
            data, munc = host[c]
            rel, fit, cnt = b.phase_tracks(c, 1.0e-4, with_fit=True, use_lambda=True)
            only_rel, none_fit, none_cnt = b.phase_tracks(c, 1.0e-4)
            assert none_fit is None and none_cnt is None
            np.testing.assert_array_equal(only_rel, rel)
            level, lam = b.download(c, "xs")[:, 0], b.download(c, "lambda")
            cur, nxt = b.download(c, "background"), b.download(c, "background_next")
            assert np.abs(cur).max() > 0.01 and float(np.abs(nxt - cur).max()) > 0.0
            np.testing.assert_array_equal(rel, pdg.relative_level_track(level, data, munc, background=cur, pad=1.0e-4))
            inv, res = pdg.update_matrices(data, munc, level, lam, pad32, mp.lambda_bounds)
            inv64, fr = inv.astype(np.float64), res.astype(np.float64) - nxt.astype(np.float64)[None, :]
            np.testing.assert_allclose(fit, np.sum(inv64 * fr * fr, axis=0), rtol=1e-14)
            np.testing.assert_array_equal(cnt, np.count_nonzero(np.isfinite(res) & np.isfinite(inv) & (inv > 0.0), axis=0))
            want = pdg.background_fit_objective(res, inv, nxt, 16.0, 256.0, True, 1.0)
            assert 0.5 * float(fit.sum()) == pytest.approx(want["background_weighted_residual_objective"], rel=1e-12)
            assert int(cnt.sum()) == int(want["background_effective_observation_count"]) == m * n
            track = inv64.sum(axis=0)                           # the update's weight track: its median is the penalty scale
            assert info[c]["weight_scale"] == pytest.approx(float(np.median(track[track > 0.0])), rel=1e-12)
        with pytest.raises(L.ConsenrichAMDError, match="pad must be finite and nonnegative"):
            b.phase_tracks(0, -1.0)


def _variant_cases():
    data, munc, kw = _case_outer_pass_smoke()
    n = data.shape[1]
    base = {k: v for k, v in kw.items() if k not in ("returnPrecisionDiagnostics", "returnDiagnostics", "trackOptimizationPath")}
    rng = np.random.default_rng(5)
    return {
        "background_and_bounds": dict(base, returnBackground=True, boundState=True, stateLowerBound=-0.5, stateUpperBound=0.6),
        "no_background_fit": dict(base, fitBackground=False, returnBackground=True, returnScales=False),
        "fixed_q_and_mask": dict(base, processNoiseCalibration="fixed", minQ=1.0e-6, observationMask=rng.random((3, n)) > 0.2,
                                 ECM_useObsPrecisionReweighting=False),
        "given_q_and_warm_starts": dict(base, initialProcessQ=np.asarray([[2e-3, 1e-4], [1e-4, 5e-4]], np.float32),
                                        initialBackground=np.linspace(-0.2, 0.2, n).astype(np.float32),
                                        initialProcessPrecision=np.full(n, 1.5, np.float32),
                                        initialObservationPrecision=np.full(n, 0.8, np.float32), returnPrecisionDiagnostics=True),
    }, data, munc


@pytest.mark.parametrize("name", ["background_and_bounds", "no_background_fit", "fixed_q_and_mask", "given_q_and_warm_starts"])
def test_argument_variants_on_the_cpu_twin(name):
    variants, data, munc = _variant_cases()
    kw = variants[name]
    _, out = _twin_call(data, munc, kw)
    n = data.shape[1]
    want = 4 + int(kw.get("returnScales", True)) + int(kw.get("returnBackground", False)) + int(kw.get("returnPrecisionDiagnostics", False))
    assert len(out) == want
    assert out[0].shape == (n, 2) and all(np.all(np.isfinite(np.asarray(a, np.float64))) for a in out[:4])
    if kw.get("boundState"):
        assert out[0][:, 0].min() >= np.float32(-0.5) and out[0][:, 0].max() <= np.float32(0.6)
    if name == "no_background_fit":
        np.testing.assert_array_equal(out[4], np.zeros(n, np.float32))          # background stays zero (core.py:5038-5040)
    if name == "given_q_and_warm_starts":
        np.testing.assert_allclose(out[-1]["matrixQ0"], [[2e-3, 1e-4], [1e-4, 5e-4]], rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["background_and_bounds", "no_background_fit", "fixed_q_and_mask", "given_q_and_warm_starts"])
def test_argument_variants_on_the_device_match_the_twin(name):
    if not gpu_available():
        pytest.fail("GPU tests selected but no HIP device / library: the product has no CPU fallback")
    from consenrich_amd import core_api

    variants, data, munc = _variant_cases()
    kw = dict(variants[name])
    out = core_api.runConsenrich(data, munc, kw.pop("deltaF"), kw.pop("minQ"), kw.pop("maxQ"), **kw)
    _, ref = _twin_call(data, munc, variants[name])
    assert len(out) == len(ref)
    for i, (a, b) in enumerate(zip(out, ref)):
        if isinstance(a, np.ndarray):
            assert a.dtype == b.dtype and a.shape == b.shape, (name, i)
            if a.dtype.kind == "f" and i != 3:
                np.testing.assert_allclose(a.astype(np.float64), b, rtol=1e-4, atol=2e-5, err_msg=f"{name} item {i}")
            elif a.dtype.kind != "f":
                np.testing.assert_array_equal(a, b)
    if isinstance(out[-1], dict):
        for k in TRACK_KEYS:
            np.testing.assert_allclose(out[-1]["outputTracks"][k], ref[-1]["outputTracks"][k], rtol=5e-2, atol=1e-6, err_msg=k)


@pytest.mark.gpu
def test_the_references_fold_loop_call_by_call_equals_the_batch_of_folds():
    """uncertainty.py:1370-1419 calls `runConsenrich(matrixData, matrixMunc, observationMask=mask, **fitKwargs)` once per fold.
    Through `core_api.runConsenrich` (one device-resident fit per call) and through `driver.run_folds_batch` (all folds as
    chains of one batch, masks made on the device) the tuples must be the same arrays bit for bit."""
    if not gpu_available():
        pytest.fail("GPU tests selected but no HIP device / library: the product has no CPU fallback")
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import cases
    import unc_cases
    from consenrich_amd import core_api
    from consenrich_amd.batch import DeviceBatch
    from consenrich_amd.driver import fold_chain_lengths, run_folds_batch
    from oracle import oracle as orc

    m, n, folds, fbl = 4, 2300, 2, 50
    data, munc = cases.synth(n, m, 8800)
    data = (data + (0.3 * np.sin(np.arange(n) / 300.0)).astype(np.float32)[None, :]).astype(np.float32)
    bf, rc, rb = unc_cases.fold_spec(m, n, fbl, folds, 0.4, 12)
    kw = dict(deltaF=1.0, minQ=1.0e-6, maxQ=1000.0, stateInit=0.0, stateCovarInit=1000.0, boundState=False, stateLowerBound=0.0,
              stateUpperBound=0.0, blockLenIntervals=40, ECM_fixedBackgroundIters=4, ECM_outerIters=2, ECM_minOuterIters=1,
              ECM_useObsPrecisionReweighting=False, returnScales=True, returnBackground=True)
    k = dict(kw)
    plan = core_api.resolve_call(data, munc, k.pop("deltaF"), k.pop("minQ"), k.pop("maxQ"), **k)
    spec = dict(data=data, munc=munc, folds=folds, fold_block_len=fbl, block_fold=bf, reps_count=rc, reps=rb, pad=float(np.float32(plan.model.pad)))
    with DeviceBatch(0) as b:
        b.configure(plan.model, m, fold_chain_lengths([spec]))
        fits, results, info = run_folds_batch(b, plan.cfg, [spec], block_len_intervals=plan.block_len_intervals)
    act = np.ones((m, n), np.uint8)
    tot = orc.cobservationTotalInformation(munc, act, np.ones(n), False, float(np.float32(plan.model.pad)), 0.0)
    for f in range(folds):
        mask = orc.cmakeFoldMaskAndInformation(m, n, fbl, f, bf, rc, rb, munc, act, tot, np.ones(n), False, float(np.float32(plan.model.pad)), 0.0)[0]
        k = dict(kw)
        out = core_api.runConsenrich(data, munc, k.pop("deltaF"), k.pop("minQ"), k.pop("maxQ"), observationMask=mask != 0, **k)
        assert len(out) == 6
        for a, b_ in zip(out, results[f]):
            assert np.array_equal(a, b_), f
