"""Device-resident counterpart of the reference's estimator orchestration (`runConsenrich`,
core.py:3861-6142) for a whole batch of chromosomes on one GPU.

`run_consenrich_batch` composes, in the reference's order:

  0. [seed_q] every chromosome's base process noise from its own data (`DeviceBatch.qseed` + `set_chain_q`; the
     reference's fixedDiagonal calibration, core.py:5667-5686);
  1. background warm start from the weighted data (`_estimateBackgroundWarmStart`, core.py:4663-4690, 2809-2910) when the
     background is fitted and no initial background is given;
  2. the outer alternation `fit_batch` (core.py:4860-5376), per pass and for the chains still iterating (chromosomes are
     independent fits and stop independently):
       a. `stats()` -- the current background is subtracted from the data in float32 inside the statistics kernel
          (= the reference's `dataAdjusted`, core.py:3253-3256);
       b. `ecm(chain_mask=...)` -- the multipliers of the previous pass are resident, i.e. the warm start the reference
          passes as lambdaExpInit / processPrecExpInit (core.py:3257-3290);
       c. `background_update()` -- weight / rhs tracks from the ORIGINAL data and the smoothed level, conditioning guard,
          pentadiagonal solve, asymmetric IRLS seeded with the current background (core.py:5064-5136);
       d. shift test: weighted RMS shift <= rtol * max(proposal RMS, reference RMS, 1) (core.py:5199-5243);
       e. `background_apply(take=...)`: the proposal is adopted (core.py:5243-5247);
       f. penalised objective of the adopted background with the phase's multipliers (`_scorePenalizedObjective`,
          core.py:4418-4538): `stats()` + `forward_masked(RETURN_NLL | multipliers)` give the forward NLL of
          data - background, `objective_terms()` the robust-precision, roughness and negative-part penalties and the
          effective observation count; objective-stable when the per-cell change is within
          outer_nll_rtol * max(|cur|, |prev|, 1) (`_recordOuterObjective`, core.py:4750-4830);
       g. a chain stops when it was shift-stable AND objective-stable with a converged inner ECM for `patience`
          consecutive passes after `min_outer` passes (core.py:5252-5376);
  3. the FINAL fixed-background ECM phase with the converged background and warm-started multipliers, all chains
     (core.py:5385-5440);
  4. the FINAL store-all forward / backward pass on data - background with the final multipliers
     (`_runForwardBackward`, core.py:5560-5600, 4207-4336: returnNLL, NIS in D): THE tracks the reference returns and its
     CLI writes come from this pass;
  5. the return tuple of core.py:6126-6142 per chromosome: (stateSmoothed (n,2), stateCovarSmoothed (n,2,2),
     postFitResiduals (n,m), NIS (n,), intervalToBlockMap[, background][, precision diagnostics]); the level model's
     arrays are zero-padded to the levelTrend shapes (core.py:4178-4192, 6004-6006).

Steps 2d-2g restate pure-Python code of `consenrich.core`, which cannot be imported in the build image.  They are pinned by
the known answers the reference's own tests hold for them (tests/test_a12_pins.py): the weighted-RMS shift gate
(test_core.py:4533-4610), the planned pass count / minimum of three passes and the final fixed-background phase
(:4614-4693), the background warm start on a constant matrix (:4471-4491), the warm-start summary (:4494-4530); the
penalised objective (2f) has no reference-held literal and is checked against the CPU twin (oracle/driver.py) and a NumPy
restatement.  Everything the passes COMPUTE with natives is the reference's arithmetic.  The per-pass record is kept under
the reference's diagnostics keys (`ChainFit.loop_diagnostics`, `ChainFit.post_process_noise_fit()`; core.py:5262-5340,
3355-3417).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from . import _lib as L
from .batch import DeviceBatch


@dataclass
class FitConfig:
    penalties: Tuple[float, float]              # (lamFirst, lamSecond), core.py:7478-7491
    ecm_iters: int = 50
    ecm_rtol: float = 1.0e-6
    inner_iters: int = 5
    nu: float = 8.0
    use_lambda: bool = False
    use_kappa: bool = True
    use_apn: bool = False                        # ECM_useAPN as `runConsenrich` passes it: ALWAYS together with processQScale = ones
                                                 # (core.py:3282, 4296), which switches the adaptation itself off (pyx:510: `useAPN and
                                                 # not useProcessQScale`); what remains is kappa off (core.py:3975) and qScale = 1
    fit_background: bool = True
    zero_center: bool = False
    use_nonnegative: bool = True
    neg_multiplier: float = 1.0
    outer_passes: int = 32
    min_outer: int = 3
    shift_rtol: float = 5.0e-3
    seed_q: bool = False                         # per-chromosome Q0 from the data (core.py:5667, fixedDiagonal calibration)
    min_q: float = 1.0e-6                        # constants.py:147-149
    max_q: float = 1000.0
    delta_f: float = 1.0
    q_seed_prior_level: float = 1.0e-5
    outer_nll_rtol: float = 5.0e-5               # ECM_outerNLLRtol, constants.py:281
    pad: float = 1.0e-4                          # the Python-float pad of the objective's weight track (core.py:4506)
    patience: int = 2
    background_warm_start: bool = False          # step 1 (run_consenrich_batch sets it when no initial background is given)


def background_shift_gate(shift_rms: float, proposal_rms: float, reference_rms: float, rtol: float) -> dict:
    """core.py:5228-5243, 5262-5270 from the three weighted RMS values (the device computes them from the resident weight track,
    `csr_batch_background_update`): scale max(proposal RMS, reference RMS, 1), threshold rtol * scale, stable iff the shift is
    within it -- under the reference's diagnostics keys."""
    tol = float(float(rtol) * float(max(float(proposal_rms), float(reference_rms), 1.0)))
    return {"background_shift": float(shift_rms), "background_shift_threshold": tol,
            "background_shift_stable": bool(float(shift_rms) <= tol)}


def weighted_rms(weights, proposal, reference) -> Tuple[float, float, float]:
    """(shift, proposal, reference) weighted RMS of core.py:5199-5227 for host arrays (what the device returns per chain)."""
    w = np.asarray(weights, np.float64)
    sw = float(np.sum(w, dtype=np.float64))
    if sw <= 0.0:
        raise ValueError("shift RMS requires positive weights")
    g1, g0 = np.asarray(proposal, np.float64), np.asarray(reference, np.float64)
    d = g1 - g0
    return (float(np.sqrt(float(np.dot(w, d * d)) / sw)), float(np.sqrt(float(np.dot(w, g1 * g1)) / sw)),
            float(np.sqrt(float(np.dot(w, g0 * g0)) / sw)))


def planned_outer_passes(cfg: "FitConfig") -> int:
    """core.py:4704-4713: max(ECM_minOuterIters, max(1, ECM_outerIters)) passes are planned when the background is fitted."""
    if not cfg.fit_background:
        return 1
    return max(int(cfg.min_outer), max(1, int(cfg.outer_passes)))


@dataclass
class ChainFit:
    passes: int = 0
    converged: bool = False
    outer_stop_reason: str = "max_outer_passes"               # core.py:4747, 5040, 5371-5383
    loop_diagnostics: List[dict] = field(default_factory=list)     # one dict per ECM phase, the reference's keys (core.py:5262-5340)
    warm_start: dict = field(default_factory=dict)            # core.py:4689-4695
    planned_passes: int = 0
    ecm_iters: List[int] = field(default_factory=list)
    nll: List[float] = field(default_factory=list)
    shift: List[float] = field(default_factory=list)
    irls_passes: List[int] = field(default_factory=list)
    objective: List[dict] = field(default_factory=list)       # per pass: the reference's objective diagnostics
    q0: object = None                                         # seeded base process noise (float32 (2,2)) if seed_q
    q_seed: dict = field(default_factory=dict)                # its diagnostics (core.py:3751-3779)
    warm_start_passes: Optional[int] = None                   # IRLS passes of the background warm start
    final_ecm_iters: Optional[int] = None                     # final fixed-background ECM phase (core.py:5403)
    final_ecm_nll: Optional[float] = None
    final_ecm_converged: Optional[bool] = None
    final_nll: Optional[float] = None                         # sumNLL of the final forward pass (core.py:5583)
    final_forward_nis: Optional[float] = None                 # mean NIS of the final forward pass (core.py:5824)
    ecm_state_level: object = None                            # smoothed level of the LAST ECM phase (host float32 (n,)), kept on request:
                                                              # what the reference's sign-change diagnostic reads (core.py:4980, 5485)

    def post_process_noise_fit(self, cfg: "FitConfig", extras: Optional[dict] = None) -> dict:
        """The reference's `diagnostics["post_process_noise_fit"]` summary of this chain (`_fitDiagnosticsMetadata`,
        core.py:3355-3417) with its complete key set.  `extras`: values only the caller can compute from host arrays (the
        multiplier bound-hit fractions, the sign-change rate).  The `background_objective*` keys -- a diagnostic of the
        background solve alone that enters no stop rule (core.py:5160-5197; the rule reads the shift, the OUTER objective and the
        inner convergence, :5252-5376) -- come from the last in-loop record when the fit ran with `pass_diagnostics`
        (`core_api.runConsenrich` with returnDiagnostics does), else None."""
        def finite(v):
            return None if (v is None or not math.isfinite(float(v))) else float(v)

        def scalar(v):              # core._diagnosticScalar (core.py:2315-2327): plain Python values, non-finite floats -> None
            if isinstance(v, np.generic):
                v = v.item()
            if isinstance(v, float):
                return v if math.isfinite(v) else None
            return v

        loop = [r for r in self.loop_diagnostics if not r.get("final_fixed_background_ecm")]
        last = loop[-1] if loop else {}
        conv = [bool(r["converged"]) for r in self.loop_diagnostics if r.get("converged") is not None]
        incr = [int(r["nll_increase_count"]) for r in self.loop_diagnostics if r.get("nll_increase_count") is not None]
        # outer_nll*: the forward NLL of the adopted background and its pass-to-pass change (core.py:4781-4797)
        fwd = [r.get("outer_forward_nll") for r in loop if r.get("outer_forward_nll") is not None]
        nll_change = nll_tol = None
        nll_stable = False
        if len(fwd) >= 2 and math.isfinite(fwd[-1]) and math.isfinite(fwd[-2]):
            nll_change = abs(fwd[-1] - fwd[-2])
            nll_tol = cfg.outer_nll_rtol * max(abs(fwd[-1]), abs(fwd[-2]), 1.0)
            nll_stable = bool(nll_change <= nll_tol)
        out = {
            "requested_outer_passes": max(1, int(cfg.outer_passes)) if cfg.fit_background else 1,
            "min_outer_passes": int(cfg.min_outer) if cfg.fit_background else 1,
            "planned_outer_passes": int(self.planned_passes),
            "actual_outer_passes": int(self.passes),
            "outer_converged": bool(self.converged),
            "outer_stop_reason": str(self.outer_stop_reason),
            "background_shift": finite(last.get("background_shift", 0.0)),
            "background_shift_threshold": finite(last.get("background_shift_threshold")),
            "background_objective": finite(last.get("background_objective")),
            "background_objective_per_cell": finite(last.get("background_objective_per_cell")),
            "background_objective_change_per_cell": finite(last.get("background_objective_change_per_cell")),
            "background_objective_threshold_per_cell": finite(last.get("background_objective_threshold_per_cell")),
            "background_objective_stable": bool(last.get("background_objective_stable", False)),
            "outer_nll": finite(fwd[-1]) if fwd else None, "outer_nll_change": finite(nll_change),
            "outer_nll_threshold": finite(nll_tol), "outer_nll_stable": nll_stable,
            "outer_objective": finite(last.get("outer_objective")),
            "outer_objective_per_cell": finite(last.get("outer_objective_per_cell")),
            "outer_objective_change_per_cell": finite(last.get("outer_objective_change_per_cell")),
            "outer_objective_threshold_per_cell": finite(last.get("outer_objective_threshold_per_cell")),
            "outer_objective_stable": bool(last.get("outer_objective_stable", False)),
            "outer_effective_observation_count": int(last.get("outer_effective_observation_count", 0) or 0),
            "observation_lambda_lower_bound_hits": None, "observation_lambda_upper_bound_hits": None,
            "process_kappa_lower_bound_hits": None, "process_kappa_upper_bound_hits": None,
            "relative_sign_change_per_kb": None,
            "outer_stable_iters": int(last.get("outer_stable_iters", 0)),
            "outer_patience_target": int(cfg.patience),
            "inner_ecm_converged": bool(self.loop_diagnostics[-1].get("converged")) if self.loop_diagnostics else False,
            "warm_start": dict(self.warm_start),
            "all_ecm_converged": (bool(conv) and all(conv)) if conv else None,
            "max_nll_increase_count": max(incr) if incr else None,
            "fixed_background_ecm": [{k: ([dict(row) for row in v] if k == "optimization_path" else scalar(v))
                                      for k, v in r.items()} for r in self.loop_diagnostics],
        }
        if extras:
            out.update(extras)
        return out


def warm_start_source(cfg: "FitConfig") -> str:
    """`source` of `_estimateBackgroundWarmStart` (core.py:2866-2884), reported as warm_start["background_prepass_source"]."""
    if cfg.use_nonnegative:
        return "asymmetric_irls_zero_centered_weighted_data" if cfg.zero_center else "asymmetric_irls_weighted_data"
    return "zero_centered_banded_weighted_data" if cfg.zero_center else "banded_weighted_data"


def ecm_phase_record(out, path, cfg: "FitConfig", outer_pass: int, track_path: bool = False) -> dict:
    """One ECM phase as the reference records it: the diagnostics mapping `cfixedBackgroundECM(..., returnDiagnostics=True)`
    returns (pyx:8404-8440: iteration counts, convergence state, first / last NLL; with `trackOptimizationPath` the per-iteration
    rows of pyx:8337-8402) after `_normalizeFixedBackgroundECMDiagnostics` (core.py:3336-3352: source and outer pass added),
    from the batch ECM's per-chain record and NLL path.  A phase on <= 5 bins is the filter + smoother fallback (pyx:7998-8129):
    zero iterations, `skipped`, not converged -- like there."""
    from .cconsenrich import _replay_path

    skipped = bool(out.skipped)
    inner_ok = bool(out.converged) and not skipped
    hi = bool(out.has_initial_nll) and not skipped
    rec = {"iters_done": 0 if skipped else int(out.iters_done), "max_iters": int(cfg.ecm_iters), "converged": inner_ok,
           "skipped": skipped, "skip_reason": "too_few_intervals" if skipped else None,
           "fallback": "filter_smoother_only" if skipped else None,
           "stable_iters": 0 if skipped else int(out.stable_iters), "patience_target": 2,
           "initial_nll": float(out.final_nll) if skipped else (float(out.initial_nll) if hi else None),
           "final_nll": float(out.final_nll),
           "final_abs_rel_change": float(out.abs_rel_change) if hi else None,
           "final_rel_improvement": float(out.rel_improvement) if hi else None,
           "nll_increase_count": 0 if skipped else int(out.nll_increase_count),
           "diagnostics_source": "cfixedBackgroundECM", "outer_pass": int(outer_pass)}
    if track_path:
        rows = [] if skipped else [float(v) for v in np.asarray(path)[: int(out.iters_done)]]
        rec["optimization_path"] = _replay_path("", rows, float(np.float32(cfg.ecm_rtol)), False)
    return rec


def fit_batch(batch: DeviceBatch, cfg: FitConfig, keep_background: bool = False, initial_lambda: bool = False,
              initial_kappa: bool = False, pass_diagnostics=None, track_path: bool = False) -> List[ChainFit]:
    """Steps 0-2 (the alternation loop, core.py:4860-5376) on a configured batch with data uploaded.  The fit that is
    resident afterwards is the one of each chain's last in-loop ECM phase against the background of THAT phase -- the
    reference never returns it: `run_consenrich_batch` continues with the final phases.  keep_background: start from the
    background that is resident (an initial background uploaded with set_background) instead of zeros.
    initial_lambda / initial_kappa: the caller uploaded warm-start multipliers (`upload_multipliers`; the reference's
    initialObservationPrecision / initialProcessPrecision): the first ECM phase starts from them either way (they are the
    resident multipliers); an initial lambda also weights the background warm start (core.py:4663-4676 passes
    observationPrecision = lambdaExpLocal), and both are recorded in the warm-start summary (core.py:4689-4695).
    pass_diagnostics: optional object with `loop_pass(batch, chain, update_record)` / `phase(batch, chain)` returning the per-phase summaries
    the reference computes on the host (multiplier means / medians / bound hits, sign-change rate, background-fit
    objective; core.py:4946-4990, 5161-5197) as a dict or a Future of one -- `core_api.PassDiagnostics`; they are merged into the
    phase records (before this function returns) and enter no decision.  track_path: keep the per-iteration rows of every ECM phase (`trackOptimizationPath`)."""
    nc = len(batch.chain_lens)
    fits = [ChainFit() for _ in range(nc)]
    active = [True] * nc
    stable = [0] * nc
    pending = []                            # (summary still being computed, the record it belongs to): `_merge_summaries`
    if not keep_background:
        for c in range(nc):
            batch.set_background(c, None)
    if cfg.seed_q:
        # matrixQ0 of every chromosome from its own data (core.py:5667-5686), on the resident matrices
        seeds = batch.qseed(pad=cfg.pad, stateModel="levelTrend" if batch.d == 2 else "level", minQ=cfg.min_q,
                            maxQ=cfg.max_q, deltaF=cfg.delta_f, robustTNu=cfg.nu, qSeedPriorLevel=cfg.q_seed_prior_level)
        batch.set_chain_q([q[: batch.d, : batch.d] for q, _ in seeds])
        for c in range(nc):
            fits[c].q0, fits[c].q_seed = seeds[c]
    if cfg.fit_background and cfg.background_warm_start and not keep_background:
        # core.py:4663-4690: asymmetric-IRLS (or plain) solve of the weighted data themselves, no initial background
        info = batch.background_update(cfg.penalties[0], cfg.penalties[1], zero_center=cfg.zero_center,
                                       use_nonnegative=cfg.use_nonnegative,
                                       negative_penalty_multiplier=cfg.neg_multiplier,
                                       use_lambda=bool(cfg.use_lambda and initial_lambda),
                                       use_initial=False, zero_state=True)
        batch.background_apply(None)
        for c in range(nc):
            fits[c].warm_start_passes = int(info[c]["passes"])
    prev_obj = [float("nan")] * nc
    fwd_flags = (L.RETURN_NLL | (L.USE_LAMBDA if cfg.use_lambda else 0) | (L.USE_KAPPA if cfg.use_kappa else 0)
                 | ((L.USE_APN | L.USE_QSCALE) if cfg.use_apn else 0))
    have_stats = False
    planned = planned_outer_passes(cfg)
    last_inner = [False] * nc
    last_obj_stable = [False] * nc
    prepass = bool(cfg.fit_background and cfg.background_warm_start and not keep_background)
    for c in range(nc):
        fits[c].planned_passes = planned
        fits[c].warm_start = {"background": bool(keep_background),                       # core.py:4689-4695
                              "background_prepass": prepass,
                              "background_prepass_source": warm_start_source(cfg) if prepass else "",
                              "observation_precision": bool(cfg.use_lambda and initial_lambda),
                              "process_precision": bool(cfg.use_kappa and not cfg.use_apn and initial_kappa)}
    for p in range(planned):
        if not have_stats:
            batch.stats()
        outs, paths = batch.ecm(max_iters=cfg.ecm_iters, inner_iters=cfg.inner_iters, rtol=cfg.ecm_rtol, nu=cfg.nu,
                                use_lambda=cfg.use_lambda, use_kappa=cfg.use_kappa, use_apn=cfg.use_apn, use_qscale=cfg.use_apn,
                                chain_mask=active)
        for c in range(nc):
            if active[c]:
                fits[c].ecm_iters.append(int(outs[c].iters_done))
                fits[c].nll.append(float(outs[c].final_nll))
                fits[c].passes = p + 1
        if not cfg.fit_background:
            for c in range(nc):                         # core.py:4994-5062: one phase, no background model
                fits[c].converged = True
                fits[c].outer_stop_reason = "fit_background_false"
                rec = ecm_phase_record(outs[c], paths[c], cfg, 1, track_path)
                rec.update({"background_shift": 0.0, "background_shift_threshold": 0.0, "background_shift_stable": True})
                if pass_diagnostics is not None:                    # core.py:5009-5033
                    pending.append((pass_diagnostics.phase(batch, c), rec))
                rec.update({"outer_inner_ecm_converged": bool(rec["converged"]), "outer_stable_iters": 0,
                            "outer_patience_target": int(cfg.patience)})
                fits[c].loop_diagnostics.append(rec)
            break
        info = batch.background_update(cfg.penalties[0], cfg.penalties[1], zero_center=cfg.zero_center,
                                       use_nonnegative=cfg.use_nonnegative,
                                       negative_penalty_multiplier=cfg.neg_multiplier, use_lambda=cfg.use_lambda,
                                       use_initial=True)
        take = list(active)
        # host-side summaries of the phase and of the proposal (they read the CURRENT background and the proposal: before the apply)
        extra = [pass_diagnostics.loop_pass(batch, c, info[c]) if (pass_diagnostics is not None and active[c]) else {}
                 for c in range(nc)]
        batch.background_apply(take)            # the proposal of a pass is always adopted (core.py:5243)
        # penalised objective of the adopted background (core.py:5248-5251)
        batch.stats()
        have_stats = True
        _, fnll = batch.forward_masked(fwd_flags, take)
        terms = batch.objective_terms(cfg.nu, cfg.penalties[0], cfg.penalties[1], cfg.neg_multiplier, pad=cfg.pad,
                                      use_lambda_penalty=cfg.use_lambda, use_kappa_penalty=cfg.use_kappa,
                                      use_lambda_weights=cfg.use_lambda, use_nonnegative=cfg.use_nonnegative)
        for c in range(nc):
            if not active[c]:
                continue
            o = info[c]
            gate = background_shift_gate(o["shift_rms"], o["proposal_rms"], o["reference_rms"], cfg.shift_rtol)
            fits[c].shift.append(float(o["shift_rms"]))
            fits[c].irls_passes.append(int(o["passes"]))
            t = terms[c]
            obj = (float(fnll[c]) + t["robust_observation_penalty"] + t["robust_process_penalty"]
                   + (t["first_difference_penalty"] + t["second_difference_penalty"]) + t["negative_penalty"])
            cur = obj / t["effective_observation_count"]
            obj_stable = bool(math.isfinite(prev_obj[c]) and math.isfinite(cur)
                              and abs(cur - prev_obj[c]) <= cfg.outer_nll_rtol * max(abs(cur), abs(prev_obj[c]), 1.0))
            prev_obj[c] = cur
            fits[c].objective.append(dict(t, forward_nll=float(fnll[c]), penalized_objective=obj,
                                          penalized_objective_per_cell=cur, stable=obj_stable))
            rec = ecm_phase_record(outs[c], paths[c], cfg, p + 1, track_path)
            inner_ok = bool(rec["converged"])
            ok = gate["background_shift_stable"] and obj_stable and inner_ok
            stable[c] = stable[c] + 1 if ok else 0
            last_inner[c], last_obj_stable[c] = inner_ok, obj_stable
            tol_obj = cfg.outer_nll_rtol * max(abs(cur), abs(fits[c].objective[-2]["penalized_objective_per_cell"]), 1.0) \
                if len(fits[c].objective) >= 2 else float("nan")
            if pass_diagnostics is not None:
                pending.append((extra[c], rec))
            fits[c].loop_diagnostics.append(rec)
            rec.update({
                **gate,
                "outer_ecm_fit_nll": float(outs[c].final_nll), "outer_forward_nll": float(fnll[c]),
                "outer_objective": obj, "outer_objective_per_cell": cur,
                "outer_objective_change_per_cell": abs(cur - fits[c].objective[-2]["penalized_objective_per_cell"])
                if len(fits[c].objective) >= 2 else float("nan"),
                "outer_objective_threshold_per_cell": tol_obj, "outer_objective_stable": obj_stable,
                "outer_effective_observation_count": int(t["effective_observation_count"]),
                "outer_robust_observation_penalty": t["robust_observation_penalty"],
                "outer_robust_process_penalty": t["robust_process_penalty"],
                "outer_background_smoothness_penalty": t["first_difference_penalty"] + t["second_difference_penalty"],
                "outer_background_first_difference_penalty": t["first_difference_penalty"],
                "outer_background_second_difference_penalty": t["second_difference_penalty"],
                "outer_background_negative_penalty": t["negative_penalty"],
                "outer_inner_ecm_converged": inner_ok, "outer_stable_iters": int(stable[c]),
                "outer_patience_target": int(cfg.patience)})
            if p + 1 >= cfg.min_outer and stable[c] >= cfg.patience:
                fits[c].converged = True
                fits[c].outer_stop_reason = "background_objective_inner_stable"        # core.py:5371-5375
                active[c] = False
        if not any(active):
            break
    if cfg.fit_background:
        for c in range(nc):                                # core.py:5377-5383
            if fits[c].converged:
                continue
            if not last_inner[c]:
                fits[c].outer_stop_reason = "max_outer_passes_inner_ecm_unconverged"
            elif not last_obj_stable[c]:
                fits[c].outer_stop_reason = "max_outer_passes_objective"
            elif stable[c] < cfg.patience:
                fits[c].outer_stop_reason = "max_outer_passes_patience"
    _merge_summaries(pending)
    return fits


def _merge_summaries(pending) -> None:
    """`pass_diagnostics` may hand back its summaries as Futures (the host part runs beside the device's next phase): merge each
    into the record it belongs to; the keys are disjoint from the record's own."""
    for summary, record in pending:
        record.update(summary.result() if hasattr(summary, "result") else summary)
    pending.clear()


def precision_diagnostics(batch: DeviceBatch, cfg: FitConfig, chain: int, q0, stateModel: str) -> dict:
    """The reference's `returnPrecisionDiagnostics` dict (core.py:6040-6121) of one chain from the resident final pass:
    ten per-interval float32 tracks (core.py:7734-7878; `DeviceBatch.diagnostics` must have run) + multipliers + Q0."""
    n = batch.chain_lens[chain]
    d = batch.d
    q0 = np.asarray(q0, np.float32)
    base_level = np.full(n, float(q0[0, 0]))
    base_trend = np.full(n, float(q0[1, 1])) if d == 2 else np.zeros(n)
    ones = np.ones(n)                                       # processQScaleFinal = ones (core.py:5688)
    tracks = {
        "baseQLevel": base_level.astype(np.float32), "baseQTrend": base_trend.astype(np.float32),
        "preKappaQLevel": (base_level * ones).astype(np.float32),
        "preKappaQTrend": (base_trend * ones).astype(np.float32),
        "processQScale": ones.astype(np.float32),
    }
    for name in ("sumGain0", "sumGain1", "effectiveQLevel", "effectiveQTrend", "muncTrace"):
        tracks[name] = batch.download(chain, name)
    return {
        "precision_track_diagnostics": True,
        "state_model": stateModel,
        "ECM_useAPN": bool(cfg.use_apn),
        "process_precision_reweighting_requested": bool(cfg.use_kappa),
        "process_precision_reweighting_effective": bool(cfg.use_kappa),
        "process_precision_reweighting_disabled_by_apn": False,
        "lambdaExp": batch.download(chain, "lambda") if cfg.use_lambda else None,
        "processPrecExp": batch.download(chain, "kappa") if cfg.use_kappa else None,
        "matrixQ0": q0,
        "outputTracks": tracks,
    }


def run_consenrich_batch(batch: DeviceBatch, cfg: FitConfig, *, block_len_intervals: int, model_q0=None,
                         initial_background=None, return_background: bool = True,
                         return_precision_diagnostics: bool = True, download: bool = True,
                         initial_lambda: bool = False, initial_kappa: bool = False, keep_ecm_state: bool = False,
                         pass_diagnostics=None, track_path: bool = False):
    """Steps 0-5 of the module docstring.  Returns (fits, results): `fits` the per-chain history, `results` one tuple per
    chain in the reference's order (core.py:6126-6142 with returnScales=True): (stateSmoothed (n,2) float32,
    stateCovarSmoothed (n,2,2), postFitResiduals (n,m), NIS (n,), intervalToBlockMap (n,) int32[, background (n,)]
    [, precision diagnostics dict]).  download=False leaves everything on the device (results = None): the final pass's
    arrays are resident and exported (`DeviceBatch.download`, `bedgraph_bytes`, `device_array`).

    model_q0: the batch-wide matrixQ0 (2,2) used when cfg.seed_q is False (for the diagnostics dict; default: the
    reference's 1e-4 fixed diagonal is NOT assumed -- pass what the batch was configured with).
    initial_background: optional list of per-chain float32 tracks (the reference's `initialBackground`); None = background
    warm start from the weighted data when the background is fitted (core.py:4663), zeros otherwise.
    initial_lambda / initial_kappa: warm-start multipliers were uploaded (`fit_batch`).
    keep_ecm_state: keep the smoothed level of every chain's LAST ECM phase on the host (`ChainFit.ecm_state_level`; the
    reference's sign-change diagnostic reads that state, core.py:4980 / 5485 -- not the final pass's).
    pass_diagnostics / track_path: `fit_batch`; the final phase's record gets the same summaries (core.py:5456-5517)."""
    nc = len(batch.chain_lens)
    d = batch.d
    cfg = FitConfig(**{**cfg.__dict__})
    if initial_background is not None:
        for c in range(nc):
            batch.set_background(c, initial_background[c])
        cfg.background_warm_start = False
    else:
        cfg.background_warm_start = bool(cfg.fit_background)
    fits = fit_batch(batch, cfg, keep_background=initial_background is not None, initial_lambda=initial_lambda,
                     initial_kappa=initial_kappa, pass_diagnostics=pass_diagnostics, track_path=track_path)

    mult_flags = (L.USE_LAMBDA if cfg.use_lambda else 0) | (L.USE_KAPPA if cfg.use_kappa else 0)
    apn_flag = (L.USE_APN | L.USE_QSCALE) if cfg.use_apn else 0      # (the resident qScale track is all ones: core.py:5688)
    if cfg.fit_background:
        # final fixed-background ECM phase (core.py:5385-5440): every chain, converged background, warm-started multipliers
        batch.stats()
        outs, paths = batch.ecm(max_iters=cfg.ecm_iters, inner_iters=cfg.inner_iters, rtol=cfg.ecm_rtol, nu=cfg.nu,
                                use_lambda=cfg.use_lambda, use_kappa=cfg.use_kappa, use_apn=cfg.use_apn, use_qscale=cfg.use_apn)
        pending = []
        for c in range(nc):
            rec = ecm_phase_record(outs[c], paths[c], cfg, fits[c].passes + 1, track_path)         # core.py:5441-5455
            rec["final_fixed_background_ecm"] = True
            if pass_diagnostics is not None:
                pending.append((pass_diagnostics.phase(batch, c), rec))
            fits[c].final_ecm_iters = int(rec["iters_done"])
            fits[c].final_ecm_nll = float(outs[c].final_nll)
            fits[c].final_ecm_converged = bool(rec["converged"])
            fits[c].loop_diagnostics.append(rec)
    if keep_ecm_state:
        # every chain's last ECM phase: the final one, or (no background fit) the loop's single phase -- a chain that left
        # the alternation early still took part in the final phase when there is one
        batch.export(L.EXPORT_SMOOTH)
        for c in range(nc):
            fits[c].ecm_state_level = np.ascontiguousarray(batch.download(c, "xs")[:, 0])
    # final store-all forward / backward on data - background with the final multipliers (core.py:5560-5600).  The
    # statistics of the final background are resident (the ECM phase above or, without a background fit, the loop's)
    sum_d, sum_nll = batch.forward_backward(L.RETURN_NLL | mult_flags | apn_flag)
    batch.export(L.EXPORT_FORWARD | L.EXPORT_SMOOTH | L.EXPORT_RESID | L.EXPORT_MULT)
    if return_precision_diagnostics:
        batch.diagnostics(mult_flags | (L.USE_QSCALE if cfg.use_apn else 0))
    if cfg.fit_background:
        _merge_summaries(pending)              # (computed beside the final pass)
    for c in range(nc):
        fits[c].final_nll = float(sum_nll[c])
        fits[c].final_forward_nis = float(sum_d[c]) / float(batch.chain_lens[c])      # phiHat = sumD / n (pyx:6627)
    if not download:
        return fits, None
    state_model = "levelTrend" if d == 2 else "level"
    results = []
    for c in range(nc):
        n = batch.chain_lens[c]
        xs, ps = batch.download(c, "xs"), batch.download(c, "Ps")
        if d == 1:                              # _padLevelStateArray / _padLevelCovarArray (core.py:4178-4192)
            xs2, ps2 = np.zeros((n, 2), np.float32), np.zeros((n, 2, 2), np.float32)
            xs2[:, 0], ps2[:, 0, 0] = xs[:, 0], ps[:, 0, 0]
            xs, ps = xs2, ps2
        block_map = (np.arange(n, dtype=np.int64) // max(int(block_len_intervals), 1)).astype(np.int32)   # core.py:4105-4114
        res = [xs, ps, batch.download(c, "resid"), batch.download(c, "D"), block_map]
        if return_background:
            res.append(batch.download(c, "background"))
        if return_precision_diagnostics:
            q0 = fits[c].q0 if fits[c].q0 is not None else model_q0
            if q0 is None:
                raise ValueError("model_q0 is required for the precision diagnostics when cfg.seed_q is False")
            res.append(precision_diagnostics(batch, cfg, c, q0, state_model))
        results.append(tuple(res))
    return fits, results


def fold_chain_lengths(specs) -> List[int]:
    """Chain lengths of the batch `run_folds_batch` expects: chromosome c's folds are consecutive chains of its length."""
    out: List[int] = []
    for sp in specs:
        out += [int(np.asarray(sp["data"]).shape[1])] * int(sp["folds"])
    return out


def run_folds_batch(batch: DeviceBatch, cfg: FitConfig, specs, *, block_len_intervals: int, model_q0=None,
                    return_precision_diagnostics: bool = False):
    """The delete-block calibration fold loop (uncertainty.py:1370-1419: per fold `cmakeFoldMaskAndInformation`, then a FULL
    `runConsenrich(matrixData, matrixMunc, observationMask=mask, **fitKwargs)`) for every fold of every chromosome as ONE
    device-resident batch: each (chromosome, fold) is a chain; its masked variance matrix (1e30 in the deleted (replicate,
    block) cells, core.py:2759-2780) is made on the device from one upload per chromosome; `run_consenrich_batch` then runs
    every chain as a full fit -- its own Q0 seed on its masked matrices (cfg.seed_q), background warm start, alternation,
    final ECM phase, final pass.  Chains are independent fits, so a fold's result equals the single-chain fit of the
    host-masked matrices (tests: bit for bit in the default mode).

    specs: one dict per chromosome -- data, munc (float32 (m, n)), folds, fold_block_len, block_fold (int32 per block),
    reps_count (int64 per block), reps (int64 (blocks, slots)) as `cmakeFoldSpec` returns them, pad, rho (default 0).
    The batch must be configured with `fold_chain_lengths(specs)`.
    Returns (fits, results, info): fits / results per chain as `run_consenrich_batch` (background always returned),
    info[c][f] = (keptInformation, heldoutInformation, h) float64 tracks of fold f of chromosome c."""
    want = fold_chain_lengths(specs)
    if list(batch.chain_lens) != want:
        raise ValueError("the batch must be configured with fold_chain_lengths(specs)")
    info = []
    first = 0
    for sp in specs:
        nf = int(sp["folds"])
        if nf < 1:
            raise ValueError("folds must be positive")
        batch.upload(first, np.ascontiguousarray(sp["data"], np.float32), np.ascontiguousarray(sp["munc"], np.float32))
        tracks = [None] * nf
        # folds 1 .. nf-1 are masked copies of the uploaded matrices, fold 0 is then made in place
        for f in list(range(1, nf)) + [0]:
            tracks[f] = batch.make_fold(first, first + f, int(sp["fold_block_len"]), f, sp["block_fold"], sp["reps_count"],
                                        sp["reps"], pad=float(sp["pad"]), rho=float(sp.get("rho", 0.0)))
        info.append(tracks)
        first += nf
    fits, results = run_consenrich_batch(batch, cfg, block_len_intervals=block_len_intervals, model_q0=model_q0,
                                         return_background=True, return_precision_diagnostics=return_precision_diagnostics)
    return fits, results, info
