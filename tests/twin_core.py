"""TEST INFRASTRUCTURE: a validated `core_api.RunPlan` replayed on the CPU twin (oracle/driver.py, the oracle's natives) --
returns the same (ChainFit, final-pass arrays) pair the device path hands to `core_api.assemble_result`, so one call of the
reference's signature can be compared tuple for tuple between the GPU product and the CPU restatement."""
import numpy as np


def twin_cfg(plan):
    cfg, mp = plan.cfg, plan.model
    d = mp.state_dim
    return dict(state_dim=d, F=mp.F, Q0=None, state_init=mp.state_init, state_covar_init=mp.state_covar_init, pad=mp.pad,
                lambda_bounds=mp.lambda_bounds, kappa_bounds=mp.kappa_bounds, block_len_intervals=plan.block_len_intervals,
                penalties=cfg.penalties, ecm_iters=cfg.ecm_iters, ecm_rtol=cfg.ecm_rtol, inner_iters=cfg.inner_iters, nu=cfg.nu,
                use_lambda=cfg.use_lambda, use_kappa=cfg.use_kappa, use_apn=cfg.use_apn, apn=mp.apn,
                fit_background=cfg.fit_background, zero_center=cfg.zero_center, use_nonnegative=cfg.use_nonnegative,
                neg_multiplier=cfg.neg_multiplier, outer_passes=cfg.outer_passes, min_outer=cfg.min_outer,
                shift_rtol=cfg.shift_rtol, patience=cfg.patience, outer_nll_rtol=cfg.outer_nll_rtol,
                interval_size_bp=plan.interval_size_bp, track_path=bool(plan.ret["track_path"]))


def twin_run(plan):
    from consenrich_amd import core_api
    from consenrich_amd.driver import ChainFit
    from oracle import diagnostics as odiag
    from oracle import driver as odrv
    from oracle import qseed as oq

    cfg, mp = plan.cfg, plan.model
    d = mp.state_dim
    q0, q_seed = plan.q0, {}
    if q0 is None:                  # core.py:5666-5676 + the clamp of :5679-5684
        Q, q_seed = oq.estimate_initial_process_noise(oq, matrixData=plan.data, matrixMunc=plan.munc, pad=cfg.pad,
                                                      stateModel=plan.state_model, minQ=cfg.min_q, maxQ=cfg.max_q,
                                                      deltaF=cfg.delta_f, robustTNu=cfg.nu,
                                                      qSeedPriorLevel=cfg.q_seed_prior_level)
        q0 = core_api.clamp_process_noise_matrix(Q, plan.state_model, cfg.min_q, cfg.max_q)
    ocfg = twin_cfg(plan)
    q_pad = np.zeros((2, 2), np.float32)
    q_pad[:d, :d] = np.asarray(q0, np.float32)[:d, :d]
    ocfg["Q0"] = q_pad if d == 2 else q_pad[:1, :1]
    ref = odrv.run_consenrich_chain(plan.data, plan.munc, ocfg, initial_background=plan.initial_background,
                                    initial_lambda=plan.initial_lambda, initial_kappa=plan.initial_kappa)
    lam = ref["out_lam"] if cfg.use_lambda else None
    kap = ref["out_kap"] if cfg.use_kappa else None
    n = plan.data.shape[1]
    tracks = odiag.output_diagnostic_tracks(
        stateCovarForward=ref["out_Pf"], matrixMunc=plan.munc, matrixQ0=q_pad, matrixF=np.asarray(mp.F, np.float32),
        stateCovarInit=mp.state_covar_init, state_dim=d, lambdaExp=lam, processPrecExp=kap,
        processQScale=np.ones(n, np.float32), pNoiseForward=ref["out_pn"], pad=mp.pad,
        obsPrecisionMultiplierMin=mp.lambda_bounds[0], obsPrecisionMultiplierMax=mp.lambda_bounds[1],
        procPrecisionMultiplierMin=mp.kappa_bounds[0], procPrecisionMultiplierMax=mp.kappa_bounds[1])
    fit = ChainFit(passes=ref["passes"], converged=bool(ref["converged"]), outer_stop_reason=ref["stop_reason"],
                   loop_diagnostics=list(ref["loop"]), planned_passes=odrv.planned_outer_passes(ocfg),
                   warm_start=dict(ref["warm_start"]), ecm_state_level=ref["ecm_xs_level"],
                   ecm_iters=list(ref["ecm_iters"]), nll=list(ref["nll"]), q_seed=dict(q_seed),
                   final_ecm_iters=ref.get("final_ecm_iters"), final_nll=ref["final_nll"],
                   final_forward_nis=ref["final_forward_nis"])
    final = {"stateSmoothed": ref["out_xs"], "stateCovarSmoothed": ref["out_Ps"], "postFitResiduals": ref["out_resid"],
             "NIS": ref["out_NIS"], "intervalToBlockMap": ref["out_block_map"], "background": ref["out_background"],
             "outputTracks": tracks, "lambdaExp": lam, "processPrecExp": kap,
             "matrixQ0": q_pad if d == 2 else q_pad[:1, :1], "stateCovarForward": ref["out_Pf"], "pNoiseForward": ref["out_pn"]}
    return fit, final
