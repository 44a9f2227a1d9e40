"""TEST INFRASTRUCTURE ONLY -- CPU counterpart of consenrich_amd/driver.py: the outer alternation of
`runConsenrich` (fixed-background ECM phase <-> background update) composed from the oracle's natives.

Per chromosome and outer pass (core.py:4860-5390):
  1. dataAdjusted = float32(data - background)                                              core.py:3253-3256
  2. cfixedBackgroundECM(dataAdjusted, ..., lambdaExpInit / processPrecExpInit = previous)   core.py:3257-3290 (warm start)
  3. weight / rhs tracks from the ORIGINAL data and the smoothed level                       core.py:5064-5083
  4. solveZeroCenteredBackground(..., initialBackground = current background)               core.py:5124-5136
  5. weighted RMS shift, proposal / reference RMS, tolerance = rtol * max(RMS..., 1)         core.py:5199-5243
  6. background := proposal                                                                  core.py:5243-5247
  7. penalised objective of the adopted background with the current multipliers              core.py:4418-4538, 4750-4830
     = forward NLL + robust precision penalties (core.py:3161-3179) + roughness penalties (core.py:3182-3204)
       + negative-part penalty, per effective observation (core.py:2981-2986); stable when its per-cell change is within
       outer_nll_rtol * max(|current|, |previous|, 1)
  8. stop when shift-stable AND objective-stable AND inner ECM converged for `patience` consecutive passes after
     `min_outer` passes                                                                      core.py:5252-5376

Steps 7-8 restate pure-Python code of `consenrich.core`, which cannot be imported here: that part is
"parity unpinned" -- checked against the formulas' own NumPy restatement in the tests, not against reference outputs.
Never imported by ``consenrich_amd``.
"""
from __future__ import annotations

import numpy as np

from . import background as bgo
from . import oracle as orc
from . import passdiag as pdg


MASKED_HALF = 0.5 * float(np.float32(1.0e30))      # constants.py:387, core.py:2983-2985


def penalized_objective(forward_nll, munc, background, lam, kap, cfg):
    """core.py:4484-4538 given the forward NLL of (data - background): dict with the reference's keys."""
    nu = float(cfg["nu"])
    tiny = float(np.finfo(np.float64).tiny)
    obs_pen = proc_pen = 0.0
    if cfg["use_lambda"] and lam is not None:                                  # core.py:3171-3173
        v = np.maximum(np.asarray(lam, np.float64), tiny)
        obs_pen = float(0.5 * nu * np.sum(v - np.log(v)))
    if cfg["use_kappa"] and kap is not None:                                   # core.py:3174-3178
        v = np.maximum(np.asarray(kap, np.float64), tiny)
        if v.size > 1:
            v = v[1:]
        proc_pen = float(0.5 * nu * np.sum(v - np.log(v)))
    bg = np.asarray(background, np.float64).reshape(-1)
    lam_first, lam_second = cfg["penalties"]
    first = 0.5 * float(lam_first) * float(np.dot(np.diff(bg), np.diff(bg))) if bg.size >= 2 else 0.0
    second = 0.5 * float(lam_second) * float(np.dot(np.diff(bg, n=2), np.diff(bg, n=2))) if bg.size >= 3 else 0.0
    neg = 0.0
    mult = cfg["neg_multiplier"]
    if cfg["use_nonnegative"] and mult is not None and mult > 0.0:             # core.py:4078-4082, 4431-4463
        w = np.zeros(bg.size)
        prec = None if lam is None else np.clip(np.asarray(lam, np.float64).reshape(-1), *cfg["lambda_bounds"])
        for row in np.asarray(munc):
            inv = 1.0 / np.maximum(np.asarray(row, np.float64) + float(cfg["pad"]), 1.0e-8)
            if prec is not None:
                inv *= prec
            w += inv
        pos = w[np.isfinite(w) & (w > 0.0)]
        scale = float(np.median(pos)) if pos.size else 1.0
        if not np.isfinite(scale) or scale <= 0.0:
            scale = 1.0
        neg = 0.5 * float(mult * scale) * float(np.sum(np.minimum(bg, 0.0) ** 2, dtype=np.float64))
    m64 = np.asarray(munc, np.float64)
    count = int(max(1, np.count_nonzero(np.isfinite(m64) & (m64 < MASKED_HALF))))
    obj = float(forward_nll + obs_pen + proc_pen + (first + second) + neg)
    return {"forward_nll": float(forward_nll), "robust_observation_penalty": obs_pen, "robust_process_penalty": proc_pen,
            "background_first_difference_penalty": first, "background_second_difference_penalty": second,
            "background_negative_penalty": neg, "penalized_objective": obj, "penalized_objective_per_cell": obj / count,
            "effective_observation_count": count}


def background_shift_gate(weights, proposal, reference, rtol):
    """core.py:5199-5243: weighted RMS shift of the background proposal against the current background, its scale
    max(proposal RMS, reference RMS, 1) and the stability test -- with the reference's diagnostics keys (core.py:5262-5270).
    Pinned by the reference's own known answer (tests/test_core.py:4533-4610)."""
    w = np.asarray(weights, np.float64)
    sw = float(np.sum(w, dtype=np.float64))
    if sw <= 0.0:
        raise ValueError("shift RMS requires positive weights")
    g1, g0 = np.asarray(proposal, np.float64), np.asarray(reference, np.float64)
    delta = g1 - g0
    shift = float(np.sqrt(float(np.dot(w, delta * delta)) / sw))
    prop = float(np.sqrt(float(np.dot(w, g1 * g1)) / sw))
    ref = float(np.sqrt(float(np.dot(w, g0 * g0)) / sw))
    tol = float(float(rtol) * float(max(prop, ref, 1.0)))
    return {"background_shift": shift, "background_shift_threshold": tol, "background_shift_stable": bool(shift <= tol),
            "proposal_rms": prop, "reference_rms": ref}


def planned_outer_passes(cfg):
    """core.py:4704-4713: max(ECM_minOuterIters, max(1, ECM_outerIters)) when the background is fitted, else 1."""
    if not cfg["fit_background"]:
        return 1
    return max(int(cfg["min_outer"]), max(1, int(cfg["outer_passes"])))


def _apn_kwargs(cfg, n=None):
    """adaptive process noise (core.py:3273-3279: APN_minQ = minQ, APN_maxQ = max(maxQ, minQ) or inf); off unless cfg asks"""
    if not cfg.get("use_apn"):
        return {}
    # (the reference passes processQScale = ones with it, core.py:3282 / 4296: that disables the adaptation itself, pyx:510)
    return {"ECM_useAPN": True, "APN_minQ": float(cfg["apn"][0]), "APN_maxQ": float(cfg["apn"][1]), "processQScale": np.ones(int(n), np.float32)}


def ecm_record(diag, iters, nll, cfg, outer_pass):
    """core.py:3336-3352 `_normalizeFixedBackgroundECMDiagnostics` of the ECM's own mapping (pyx:8404-8440); the oracle's native
    returns the NLL path as plain floats: with `track_path` they become the reference's per-iteration rows (pyx:8337-8402)."""
    rec = {}
    for key, value in dict(diag).items():
        if isinstance(value, np.generic):
            value = value.item()
        if isinstance(value, float) and not np.isfinite(value):
            value = None
        rec[str(key)] = value
    if "optimization_path" in rec:
        rec["optimization_path"] = pdg.path_rows(rec["optimization_path"], cfg["ecm_rtol"])
    rec.setdefault("iters_done", int(iters))
    rec.setdefault("max_iters", int(cfg["ecm_iters"]))
    rec.setdefault("final_nll", float(nll))
    rec.setdefault("diagnostics_source", "cfixedBackgroundECM")
    rec["outer_pass"] = int(outer_pass)
    return rec


def fit_chain(data, munc, cfg, initial_background=None, initial_lambda=None, initial_kappa=None):
    data = np.ascontiguousarray(data, np.float32)
    munc = np.ascontiguousarray(munc, np.float32)
    m, n = data.shape
    d = cfg["state_dim"]
    bg = np.zeros(n, np.float32) if initial_background is None else np.ascontiguousarray(initial_background, np.float32).copy()
    lam, kap = initial_lambda, initial_kappa                 # warm-started multipliers (core.py:4637-4648)
    hist = {"ecm_iters": [], "nll": [], "shift": [], "irls_passes": [], "objective": [], "converged": False, "loop": [],
            "stop_reason": "max_outer_passes"}
    prev_obj = float("nan")
    fwd = orc.cforwardPass if d == 2 else orc.cforwardPassLevel
    lam_first, lam2 = cfg["penalties"]
    stable = 0
    bm = (np.arange(n) // cfg["block_len_intervals"]).astype(np.int32)
    ecm = orc.cfixedBackgroundECM if d == 2 else orc.cfixedBackgroundECMLevel
    kw = dict(matrixQ0=np.asarray(cfg["Q0"], np.float32), intervalToBlockMap=bm, blockCount=int(bm.max()) + 1,
              stateInit=cfg["state_init"], stateCovarInit=cfg["state_covar_init"], pad=cfg["pad"],
              ECM_fixedBackgroundIters=cfg["ecm_iters"], ECM_fixedBackgroundRtol=cfg["ecm_rtol"],
              t_innerIters=cfg["inner_iters"], ECM_robustTNu=cfg["nu"], returnIntermediates=True, returnDiagnostics=True,
              ECM_useObsPrecisionReweighting=cfg["use_lambda"], ECM_useProcessPrecisionReweighting=cfg["use_kappa"],
              obsPrecisionMultiplierMin=cfg["lambda_bounds"][0], obsPrecisionMultiplierMax=cfg["lambda_bounds"][1],
              procPrecisionMultiplierMin=cfg["kappa_bounds"][0], procPrecisionMultiplierMax=cfg["kappa_bounds"][1],
              logIterations=False, trackOptimizationPath=bool(cfg.get("track_path")), **_apn_kwargs(cfg, n))
    if d == 2:
        kw["matrixF"] = np.asarray(cfg["F"], np.float32)
    out = None
    prev_bg_obj = float("nan")
    neg_active = bool(cfg["use_nonnegative"] and cfg["neg_multiplier"] is not None and cfg["neg_multiplier"] > 0.0)   # core.py:4078
    inner_ok = obj_stable = False
    for p in range(planned_outer_passes(cfg)):
        adj = np.ascontiguousarray(data - bg[None, :], dtype=np.float32)
        out = ecm(matrixData=adj, matrixPluginMuncInit=munc, lambdaExpInit=lam, processPrecExpInit=kap, **kw)
        iters, nll, xs, Ps, lag, res, lam, kap, diag = out
        hist["ecm_iters"].append(int(iters))
        hist["nll"].append(float(nll))
        rec = ecm_record(diag, iters, nll, cfg, p + 1)
        rec.update(pdg.phase_summaries(data, munc, xs[:, 0], lam, kap, bg, cfg))               # core.py:4946-4984
        if not cfg["fit_background"]:
            hist["converged"] = True                       # core.py:5038-5040
            hist["stop_reason"] = "fit_background_false"
            rec.update({"background_shift": 0.0, "background_shift_threshold": 0.0, "background_shift_stable": True,
                        "outer_inner_ecm_converged": bool(diag["converged"]), "outer_stable_iters": 0,
                        "outer_patience_target": int(cfg["patience"])})
            hist["loop"].append(rec)
            break
        w, r, _, _ = bgo.weight_rhs_tracks(data, munc, xs[:, 0], np.float32(cfg["pad"]),
                                           lam if cfg["use_lambda"] else None, cfg["lambda_bounds"])
        nxt, info = bgo.solve_background(w, r, 0, zero_center=cfg["zero_center"], use_nonnegative=cfg["use_nonnegative"],
                                         multiplier=cfg["neg_multiplier"], initial=bg,
                                         penalties_override=(lam_first, lam2), return_info=True)
        # objective of the proposal against the phase that produced it (core.py:5161-5197)
        inv_m, res_m = pdg.update_matrices(data, munc, xs[:, 0], lam if cfg["use_lambda"] else None, cfg["pad"], cfg["lambda_bounds"])
        bgo_ = pdg.background_fit_objective(res_m, inv_m, nxt, lam_first, lam2, neg_active, cfg["neg_multiplier"])
        cur_bg = bgo_["background_objective_per_cell"]
        bg_change = bg_tol = float("nan")
        bg_stable = False
        if np.isfinite(prev_bg_obj) and np.isfinite(cur_bg):
            bg_change = abs(cur_bg - prev_bg_obj)
            bg_tol = cfg["outer_nll_rtol"] * max(abs(cur_bg), abs(prev_bg_obj), 1.0)
            bg_stable = bool(bg_change <= bg_tol)
        prev_bg_obj = cur_bg
        gate = background_shift_gate(w, nxt, bg, cfg["shift_rtol"])
        shift = gate["background_shift"]
        hist["shift"].append(shift)
        hist["irls_passes"].append(int(info["passes"]))
        bg = nxt
        # penalised objective of the adopted background (core.py:4465-4483: forward NLL of data - background with the
        # phase's multipliers)
        adj = np.ascontiguousarray(data - bg[None, :], dtype=np.float32)
        fkw = dict(matrixData=adj, matrixPluginMuncInit=munc, matrixQ0=kw["matrixQ0"], intervalToBlockMap=bm,
                   blockCount=kw["blockCount"], stateInit=cfg["state_init"], stateCovarInit=cfg["state_covar_init"],
                   pad=cfg["pad"], returnNLL=True, lambdaExp=lam, processPrecExp=kap if cfg["use_kappa"] else None,
                   ECM_useObsPrecisionReweighting=cfg["use_lambda"], ECM_useProcessPrecisionReweighting=cfg["use_kappa"],
                   obsPrecisionMultiplierMin=cfg["lambda_bounds"][0], obsPrecisionMultiplierMax=cfg["lambda_bounds"][1],
                   procPrecisionMultiplierMin=cfg["kappa_bounds"][0], procPrecisionMultiplierMax=cfg["kappa_bounds"][1],
                   **_apn_kwargs(cfg, n))
        if d == 2:
            fkw["matrixF"] = kw["matrixF"]
        obj = penalized_objective(float(fwd(**fkw)[3]), munc, bg, lam, kap, cfg)
        cur = obj["penalized_objective_per_cell"]
        obj_stable = bool(np.isfinite(prev_obj) and np.isfinite(cur)
                          and abs(cur - prev_obj) <= cfg["outer_nll_rtol"] * max(abs(cur), abs(prev_obj), 1.0))
        prev_obj = cur
        hist["objective"].append(obj)
        inner_ok = bool(diag["converged"])
        if gate["background_shift_stable"] and obj_stable and inner_ok:
            stable += 1
        else:
            stable = 0
        none = lambda v: float(v) if np.isfinite(v) else None                       # metadataFloat
        hist["loop"].append({**rec,
                             "background_objective": none(bgo_["background_objective"]),             # core.py:5273-5300
                             "background_objective_per_cell": none(cur_bg),
                             "background_objective_change_per_cell": none(bg_change),
                             "background_objective_threshold_per_cell": none(bg_tol),
                             "background_objective_stable": bg_stable,
                             "background_weighted_residual_objective": none(bgo_["background_weighted_residual_objective"]),
                             "background_fit_effective_observation_count": int(bgo_["background_effective_observation_count"]),
                             "converged": inner_ok, "background_shift": gate["background_shift"],
                             "background_shift_threshold": gate["background_shift_threshold"],
                             "background_shift_stable": gate["background_shift_stable"],
                             "outer_objective_per_cell": cur, "outer_objective_stable": obj_stable,
                             "outer_inner_ecm_converged": inner_ok, "outer_stable_iters": int(stable),
                             "outer_patience_target": int(cfg["patience"])})
        if p + 1 >= cfg["min_outer"] and stable >= cfg["patience"]:
            hist["converged"] = True
            hist["stop_reason"] = "background_objective_inner_stable"          # core.py:5371-5375
            break
    if cfg["fit_background"] and not hist["converged"]:                          # core.py:5377-5383
        if not inner_ok:
            hist["stop_reason"] = "max_outer_passes_inner_ecm_unconverged"
        elif not obj_stable:
            hist["stop_reason"] = "max_outer_passes_objective"
        elif stable < cfg["patience"]:
            hist["stop_reason"] = "max_outer_passes_patience"
    iters, nll, xs, Ps, lag, res, lam, kap, diag = out
    hist.update(passes=len(hist["ecm_iters"]), background=bg, xs=xs, Ps=Ps, resid=res, lam=lam, kap=kap)
    return hist


def warm_start_source(cfg):
    """`source` of `_estimateBackgroundWarmStart`'s diagnostics (core.py:2866-2884), reported as
    warm_start["background_prepass_source"] (core.py:4687-4692)."""
    if cfg["use_nonnegative"]:
        return "asymmetric_irls_zero_centered_weighted_data" if cfg["zero_center"] else "asymmetric_irls_weighted_data"
    return "zero_centered_banded_weighted_data" if cfg["zero_center"] else "banded_weighted_data"


def background_warm_start(data, munc, cfg, lam=None):
    """core.py:2809-2910 `_estimateBackgroundWarmStart` (called at core.py:4663 when the background is fitted and no initial
    background is given): the background solve of the weighted DATA (residual = data, float32 inverse variances times the
    INITIAL observation precision clipped to its bounds when one is supplied -- core.py:4669 passes lambdaExpLocal, :2847-2856 --
    no initial background)."""
    w, r, _, _ = bgo.weight_rhs_tracks(data, munc, np.zeros(np.asarray(data).shape[1], np.float32), np.float32(cfg["pad"]),
                                       lam, cfg.get("lambda_bounds", (0.25, 4.0)))
    out, info = bgo.solve_background(w, r, 0, zero_center=cfg["zero_center"], use_nonnegative=cfg["use_nonnegative"],
                                     multiplier=cfg["neg_multiplier"], initial=None, penalties_override=cfg["penalties"],
                                     return_info=True)
    return np.ascontiguousarray(out, np.float32), int(info["passes"])


def run_consenrich_chain(data, munc, cfg, initial_background=None, initial_lambda=None, initial_kappa=None):
    """The whole `runConsenrich` composition for one chromosome (module docstring of consenrich_amd/driver.py, steps 1-5):
    background warm start -> alternation loop -> FINAL fixed-background ECM phase (core.py:5385-5440) -> FINAL store-all
    forward / backward on data - background with the final multipliers (core.py:5560-5600, 4207-4336) -> the pieces of the
    return tuple (core.py:6126-6142).  Natives: the oracle's; glue: restated ("parity unpinned", see the header)."""
    data = np.ascontiguousarray(data, np.float32)
    munc = np.ascontiguousarray(munc, np.float32)
    m, n = data.shape
    d = cfg["state_dim"]
    warm_passes = None
    bg0 = initial_background
    lam0 = initial_lambda if cfg["use_lambda"] else None                                    # core.py:4639-4643
    kap0 = initial_kappa if (cfg["use_kappa"] and not cfg.get("use_apn")) else None         # core.py:4644-4650
    if bg0 is None and cfg["fit_background"]:
        bg0, warm_passes = background_warm_start(data, munc, cfg, lam0)
    hist = fit_chain(data, munc, cfg, initial_background=bg0, initial_lambda=lam0, initial_kappa=kap0)
    bg, lam, kap = hist["background"], hist["lam"], hist["kap"]
    ecm_xs_level = np.asarray(hist["xs"], np.float32)[:, 0].copy()          # smoothed level of the last ECM phase (core.py:4980)
    bm = (np.arange(n, dtype=np.int32) // cfg["block_len_intervals"]).astype(np.int32)
    Q0 = np.asarray(cfg["Q0"], np.float32)
    common = dict(matrixQ0=Q0, intervalToBlockMap=bm, blockCount=int(bm.max()) + 1, stateInit=cfg["state_init"],
                  stateCovarInit=cfg["state_covar_init"], pad=cfg["pad"],
                  ECM_useObsPrecisionReweighting=cfg["use_lambda"], ECM_useProcessPrecisionReweighting=cfg["use_kappa"],
                  obsPrecisionMultiplierMin=cfg["lambda_bounds"][0], obsPrecisionMultiplierMax=cfg["lambda_bounds"][1],
                  procPrecisionMultiplierMin=cfg["kappa_bounds"][0], procPrecisionMultiplierMax=cfg["kappa_bounds"][1],
                  **_apn_kwargs(cfg, n))
    if d == 2:
        common["matrixF"] = np.asarray(cfg["F"], np.float32)
    adj = np.ascontiguousarray(data - bg[None, :], dtype=np.float32)
    final = {}
    if cfg["fit_background"]:
        ecm = orc.cfixedBackgroundECM if d == 2 else orc.cfixedBackgroundECMLevel
        out = ecm(matrixData=adj, matrixPluginMuncInit=munc, lambdaExpInit=lam, processPrecExpInit=kap,
                  ECM_fixedBackgroundIters=cfg["ecm_iters"], ECM_fixedBackgroundRtol=cfg["ecm_rtol"],
                  t_innerIters=cfg["inner_iters"], ECM_robustTNu=cfg["nu"], returnIntermediates=True,
                  returnDiagnostics=True, logIterations=False, trackOptimizationPath=bool(cfg.get("track_path")), **common)
        iters, nll, _xs, _Ps, _lag, _res, lam, kap, diag = out
        ecm_xs_level = np.asarray(_xs, np.float32)[:, 0].copy()             # ... of the final phase (core.py:5485)
        final = dict(final_ecm_iters=int(iters), final_ecm_nll=float(nll), final_ecm_converged=bool(diag["converged"]))
        final_rec = ecm_record(diag, iters, nll, cfg, hist["passes"] + 1)                      # core.py:5441-5455
        final_rec["final_fixed_background_ecm"] = True
        final_rec.update(pdg.phase_summaries(data, munc, ecm_xs_level, lam, kap, bg, cfg))      # core.py:5456-5517
    xf, Pf, pn = np.empty((n, d), np.float32), np.empty((n, d, d), np.float32), np.empty((n, d, d), np.float32)
    D = np.empty(n, np.float32)
    fwd = orc.cforwardPass if d == 2 else orc.cforwardPassLevel
    phi, _, D, nll = fwd(matrixData=adj, matrixPluginMuncInit=munc, stateForward=xf, stateCovarForward=Pf, pNoiseForward=pn,
                         vectorD=D, returnNLL=True, storeNLLInD=False, lambdaExp=lam if cfg["use_lambda"] else None,
                         processPrecExp=kap if cfg["use_kappa"] else None, **{"processQScale": np.ones(n, np.float32), **common})
    bkw = dict(matrixData=adj, stateForward=xf, stateCovarForward=Pf, pNoiseForward=pn)
    if d == 2:
        bkw["matrixF"] = common["matrixF"]
    xs, Ps, lag, res = (orc.cbackwardPass if d == 2 else orc.cbackwardPassLevel)(**bkw)
    if d == 1:          # _padLevelStateArray / _padLevelCovarArray (core.py:4178-4192)
        xs2, Ps2 = np.zeros((n, 2), np.float32), np.zeros((n, 2, 2), np.float32)
        xs2[:, 0], Ps2[:, 0, 0] = xs[:, 0], Ps[:, 0, 0]
        xs, Ps = xs2, Ps2
    prepass = bool(initial_background is None and cfg["fit_background"])
    hist["warm_start"] = {"background": initial_background is not None,                       # core.py:4689-4695
                          "background_prepass": prepass,
                          "background_prepass_source": warm_start_source(cfg) if prepass else "",
                          "observation_precision": lam0 is not None, "process_precision": kap0 is not None}
    hist["ecm_calls"] = len(hist["ecm_iters"]) + (1 if cfg["fit_background"] else 0)
    if cfg["fit_background"]:
        hist["loop"].append(final_rec)
    hist.update(final, warm_start_passes=warm_passes, final_nll=float(nll), final_forward_nis=float(phi),
                out_xs=np.asarray(xs, np.float32), out_Ps=np.asarray(Ps, np.float32), out_resid=np.asarray(res, np.float32),
                out_NIS=np.asarray(D, np.float32), out_block_map=bm, out_background=bg, out_lam=lam, out_kap=kap,
                out_Pf=Pf, out_pn=pn, ecm_xs_level=ecm_xs_level)
    return hist
