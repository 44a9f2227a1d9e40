"""Device-resident counterpart of the reference's outer alternation (`runConsenrich`, core.py:4860-5390):
fixed-background ECM phase <-> background update, for a whole batch of chromosomes on one GPU.

With `seed_q`, every chromosome first gets its own base process noise from its data (`DeviceBatch.qseed` +
`set_chain_q`; the reference's fixedDiagonal calibration, core.py:5667-5686).  Then, per outer pass, for the chains still
iterating (chromosomes are independent fits and stop independently):
  1. `DeviceBatch.stats()` -- the current background is subtracted from the data in float32 inside the statistics
     kernel (= the reference's `dataAdjusted`, core.py:3253-3256);
  2. `DeviceBatch.ecm(chain_mask=...)` -- the multipliers of the previous pass are resident, i.e. the warm start the
     reference passes as lambdaExpInit / processPrecExpInit (core.py:3257-3290);
  3. `DeviceBatch.background_update()` -- weight / rhs tracks from the ORIGINAL data and the smoothed level, conditioning
     guard, pentadiagonal solve, asymmetric IRLS seeded with the current background (core.py:5064-5136);
  4. shift test: weighted RMS shift <= rtol * max(proposal RMS, reference RMS, 1) (core.py:5199-5243);
  5. `DeviceBatch.background_apply(take=...)`: the proposal is adopted (core.py:5243-5247);
  6. penalised objective of the adopted background with the phase's multipliers (`_scorePenalizedObjective`,
     core.py:4418-4538): `stats()` + `forward_masked(RETURN_NLL | multipliers)` give the forward NLL of data - background
     (the statistics are the ones the next ECM phase starts from, so only the forward pass is extra, as in the
     reference), `objective_terms()` the robust-precision, roughness and negative-part penalties and the effective
     observation count; objective-stable when the per-cell change is within outer_nll_rtol * max(|cur|, |prev|, 1)
     (`_recordOuterObjective`, core.py:4750-4830);
  7. a chain stops when it was shift-stable AND objective-stable with a converged inner ECM for `patience` consecutive
     passes after `min_outer` passes (core.py:5252-5376).

Steps 6-7 restate pure-Python code of `consenrich.core`, which cannot be imported in the build image: they are checked
against the CPU twin (oracle/driver.py) and a NumPy restatement of the formulas, not against reference outputs ("parity
unpinned" for this part).  Everything the passes COMPUTE with natives is the reference's arithmetic.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Tuple

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


@dataclass
class ChainFit:
    passes: int = 0
    converged: bool = False
    ecm_iters: List[int] = field(default_factory=list)
    nll: List[float] = field(default_factory=list)
    shift: List[float] = field(default_factory=list)
    irls_passes: List[int] = field(default_factory=list)
    objective: List[dict] = field(default_factory=list)       # per pass: the reference's objective diagnostics
    q0: object = None                                         # seeded base process noise (float32 (2,2)) if seed_q
    q_seed: dict = field(default_factory=dict)                # its diagnostics (core.py:3751-3779)


def fit_batch(batch: DeviceBatch, cfg: FitConfig) -> List[ChainFit]:
    """Runs the alternation on a configured batch with data uploaded.  Afterwards download(): "xs", "Ps", "lag", "resid",
    "lambda", "kappa" (fit of the last ECM phase of each chain) and "background"."""
    nc = len(batch.chain_lens)
    fits = [ChainFit() for _ in range(nc)]
    active = [True] * nc
    stable = [0] * nc
    for c in range(nc):
        batch.set_background(c, None)
    if cfg.seed_q:
        # matrixQ0 of every chromosome from its own data (core.py:5667-5686), on the resident matrices
        seeds = batch.qseed(pad=cfg.pad, stateModel="levelTrend" if batch.d == 2 else "level", minQ=cfg.min_q,
                            maxQ=cfg.max_q, deltaF=cfg.delta_f, robustTNu=cfg.nu, qSeedPriorLevel=cfg.q_seed_prior_level)
        batch.set_chain_q([q[: batch.d, : batch.d] for q, _ in seeds])
        for c in range(nc):
            fits[c].q0, fits[c].q_seed = seeds[c]
    prev_obj = [float("nan")] * nc
    fwd_flags = L.RETURN_NLL | (L.USE_LAMBDA if cfg.use_lambda else 0) | (L.USE_KAPPA if cfg.use_kappa else 0)
    have_stats = False
    for p in range(cfg.outer_passes):
        if not have_stats:
            batch.stats()
        outs, _ = batch.ecm(max_iters=cfg.ecm_iters, inner_iters=cfg.inner_iters, rtol=cfg.ecm_rtol, nu=cfg.nu,
                            use_lambda=cfg.use_lambda, use_kappa=cfg.use_kappa, chain_mask=active)
        for c in range(nc):
            if active[c]:
                fits[c].ecm_iters.append(int(outs[c].iters_done))
                fits[c].nll.append(float(outs[c].final_nll))
                fits[c].passes = p + 1
        if not cfg.fit_background:
            break
        info = batch.background_update(cfg.penalties[0], cfg.penalties[1], zero_center=cfg.zero_center,
                                       use_nonnegative=cfg.use_nonnegative,
                                       negative_penalty_multiplier=cfg.neg_multiplier, use_lambda=cfg.use_lambda,
                                       use_initial=True)
        # natural-layout copies of this pass's fit (smoothed moments, residuals, multipliers) are taken before the
        # proposal is applied: applying a background invalidates the resident fit, the exported arrays stay downloadable
        batch.export(L.EXPORT_SMOOTH | L.EXPORT_RESID | L.EXPORT_MULT)
        take = list(active)
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
            scale = max(o["proposal_rms"], o["reference_rms"], 1.0)
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
            inner_ok = bool(outs[c].converged) or bool(outs[c].skipped == 1)
            ok = o["shift_rms"] <= cfg.shift_rtol * scale and obj_stable and inner_ok
            stable[c] = stable[c] + 1 if ok else 0
            if p + 1 >= cfg.min_outer and stable[c] >= cfg.patience:
                fits[c].converged = True
                active[c] = False
        if not any(active):
            break
    if not cfg.fit_background:
        batch.export(L.EXPORT_SMOOTH | L.EXPORT_RESID | L.EXPORT_MULT)
    # like the reference's loop, the returned fit is the one of the last ECM phase and "background" the last adopted
    # proposal; call stats() before any further pass so that the statistics see the final background
    return fits
