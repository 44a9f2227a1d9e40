"""Device-resident counterpart of the reference's outer alternation (`runConsenrich`, core.py:4860-5390):
fixed-background ECM phase <-> background update, for a whole batch of chromosomes on one GPU.

Per outer pass, for the chains still iterating (chromosomes are independent fits and stop independently):
  1. `DeviceBatch.stats()` -- the current background is subtracted from the data in float32 inside the statistics
     kernel (= the reference's `dataAdjusted`, core.py:3253-3256);
  2. `DeviceBatch.ecm(chain_mask=...)` -- the multipliers of the previous pass are resident, i.e. the warm start the
     reference passes as lambdaExpInit / processPrecExpInit (core.py:3257-3290);
  3. `DeviceBatch.background_update()` -- weight / rhs tracks from the ORIGINAL data and the smoothed level, conditioning
     guard, pentadiagonal solve, asymmetric IRLS seeded with the current background (core.py:5064-5136);
  4. shift test: weighted RMS shift <= rtol * max(proposal RMS, reference RMS, 1) (core.py:5199-5243);
  5. `DeviceBatch.background_apply(take=...)`; a chain stops when it was shift-stable with a converged inner ECM for
     `patience` consecutive passes after `min_outer` passes (core.py:5244-5376).

The reference's stop rule additionally requires its penalised objective to be stable (`_recordOuterObjective`,
core.py:4750-4830: one more forward-NLL pass per outer pass).  That term is not reproduced -- `consenrich.core` cannot be
imported in the build image to pin it -- so this driver (and its CPU twin oracle/driver.py, which the tests compare it
with) uses the two criteria above; everything the passes COMPUTE is the reference's arithmetic.
"""
from __future__ import annotations

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
    patience: int = 2


@dataclass
class ChainFit:
    passes: int = 0
    converged: bool = False
    ecm_iters: List[int] = field(default_factory=list)
    nll: List[float] = field(default_factory=list)
    shift: List[float] = field(default_factory=list)
    irls_passes: List[int] = field(default_factory=list)


def fit_batch(batch: DeviceBatch, cfg: FitConfig) -> List[ChainFit]:
    """Runs the alternation on a configured batch with data uploaded.  Afterwards download(): "xs", "Ps", "lag", "resid",
    "lambda", "kappa" (fit of the last ECM phase of each chain) and "background"."""
    nc = len(batch.chain_lens)
    fits = [ChainFit() for _ in range(nc)]
    active = [True] * nc
    stable = [0] * nc
    for c in range(nc):
        batch.set_background(c, None)
    for p in range(cfg.outer_passes):
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
        for c in range(nc):
            if not active[c]:
                continue
            o = info[c]
            scale = max(o["proposal_rms"], o["reference_rms"], 1.0)
            fits[c].shift.append(float(o["shift_rms"]))
            fits[c].irls_passes.append(int(o["passes"]))
            inner_ok = bool(outs[c].converged) or bool(outs[c].skipped == 1)
            stable[c] = stable[c] + 1 if (o["shift_rms"] <= cfg.shift_rtol * scale and inner_ok) else 0
            if p + 1 >= cfg.min_outer and stable[c] >= cfg.patience:
                fits[c].converged = True
                active[c] = False
        batch.background_apply(take)            # the proposal of a pass is always adopted (core.py:5243)
        if not any(active):
            break
    if not cfg.fit_background:
        batch.export(L.EXPORT_SMOOTH | L.EXPORT_RESID | L.EXPORT_MULT)
    # like the reference's loop, the returned fit is the one of the last ECM phase and "background" the last adopted
    # proposal; call stats() before any further pass so that the statistics see the final background
    return fits
