"""TEST INFRASTRUCTURE ONLY -- CPU counterpart of consenrich_amd/driver.py: the outer alternation of
`runConsenrich` (fixed-background ECM phase <-> background update) composed from the oracle's natives.

Per chromosome and outer pass (core.py:4860-5390):
  1. dataAdjusted = float32(data - background)                                              core.py:3253-3256
  2. cfixedBackgroundECM(dataAdjusted, ..., lambdaExpInit / processPrecExpInit = previous)   core.py:3257-3290 (warm start)
  3. weight / rhs tracks from the ORIGINAL data and the smoothed level                       core.py:5064-5083
  4. solveZeroCenteredBackground(..., initialBackground = current background)               core.py:5124-5136
  5. weighted RMS shift, proposal / reference RMS, tolerance = rtol * max(RMS..., 1)         core.py:5199-5243
  6. background := proposal; stop when shift-stable and inner ECM converged for `patience`
     consecutive passes after `min_outer` passes                                            core.py:5244-5376

NOT reproduced (documented in DESIGN.md): the penalised-objective stability term of the reference's stop rule
(`_recordOuterObjective`, core.py:4750-4830, an extra forward-NLL pass per outer pass) -- `consenrich.core` cannot be
imported here to pin it, so both drivers use the two criteria above.  Never imported by ``consenrich_amd``.
"""
from __future__ import annotations

import numpy as np

from . import background as bgo
from . import oracle as orc


def fit_chain(data, munc, cfg):
    data = np.ascontiguousarray(data, np.float32)
    munc = np.ascontiguousarray(munc, np.float32)
    m, n = data.shape
    d = cfg["state_dim"]
    bg = np.zeros(n, np.float32)
    lam = kap = None
    hist = {"ecm_iters": [], "nll": [], "shift": [], "irls_passes": [], "converged": False}
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
              logIterations=False)
    if d == 2:
        kw["matrixF"] = np.asarray(cfg["F"], np.float32)
    out = None
    for p in range(cfg["outer_passes"]):
        adj = np.ascontiguousarray(data - bg[None, :], dtype=np.float32)
        out = ecm(matrixData=adj, matrixPluginMuncInit=munc, lambdaExpInit=lam, processPrecExpInit=kap, **kw)
        iters, nll, xs, Ps, lag, res, lam, kap, diag = out
        hist["ecm_iters"].append(int(iters))
        hist["nll"].append(float(nll))
        if not cfg["fit_background"]:
            break
        w, r, _, _ = bgo.weight_rhs_tracks(data, munc, xs[:, 0], np.float32(cfg["pad"]),
                                           lam if cfg["use_lambda"] else None, cfg["lambda_bounds"])
        nxt, info = bgo.solve_background(w, r, 0, zero_center=cfg["zero_center"], use_nonnegative=cfg["use_nonnegative"],
                                         multiplier=cfg["neg_multiplier"], initial=bg,
                                         penalties_override=(lam_first, lam2), return_info=True)
        sw = float(w.sum())
        g1, g0 = nxt.astype(np.float64), bg.astype(np.float64)
        shift = float(np.sqrt(np.dot(w, (g1 - g0) ** 2) / sw))
        scale = max(float(np.sqrt(np.dot(w, g1 * g1) / sw)), float(np.sqrt(np.dot(w, g0 * g0) / sw)), 1.0)
        hist["shift"].append(shift)
        hist["irls_passes"].append(int(info["passes"]))
        bg = nxt
        if shift <= cfg["shift_rtol"] * scale and bool(diag["converged"]):
            stable += 1
        else:
            stable = 0
        if p + 1 >= cfg["min_outer"] and stable >= cfg["patience"]:
            hist["converged"] = True
            break
    iters, nll, xs, Ps, lag, res, lam, kap, diag = out
    hist.update(passes=len(hist["ecm_iters"]), background=bg, xs=xs, Ps=Ps, resid=res, lam=lam, kap=kap)
    return hist
