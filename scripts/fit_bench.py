"""Whole-genome device-resident fit (SURVEY a12 counterpart, consenrich_amd/driver.py): hg38 autosomes @200bp x 32 synthetic
samples: background warm start, outer alternation of fixed-background ECM phases and background updates, final
fixed-background ECM phase, final forward/backward pass + per-interval diagnostics (run_consenrich_batch, download=False),
everything resident in HBM.
Reports wall time, per-chromosome pass / iteration counts and the kernel-time breakdown."""
import sys, os, time, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests", "golden"))
import numpy as np
import bg_cases
from consenrich_amd import _lib as L
from consenrich_amd.batch import DeviceBatch, ModelParams
from consenrich_amd.driver import FitConfig, run_consenrich_batch
from consenrich_amd.sharding import hg38_chain_lengths

m = int(os.environ.get("M", "32"))
lengths = hg38_chain_lengths(200)
# OUTER: the cap on outer passes.  The reference CLI's default is 32 (constants.py:277, at least 3); "8" is what rounds 3-5 quoted
# as "CLI defaults" -- on this synthetic recipe the chromosome-sized chains are still moving at pass 8 AND at pass 32 (the
# background shift halves per pass and the objective keeps creeping), so both are reported: the price of N passes, not of a fit
# that met the stop rule.  OUTER="32,8" (default) runs both, each on a fresh batch.
outers = [int(v) for v in os.environ.get("OUTER", "32,8").split(",")]
xtol = int(os.environ.get("XTOL", "0"))
prof = os.environ.get("PROFILE", "1") != "0"


def run(outer):
    b = DeviceBatch(0, x_tol_ulps=xtol)
    b.configure(ModelParams(state_dim=2), m, lengths); b.synthesize(1234)
    cfg = FitConfig(penalties=bg_cases.penalties(750, 128.0), ecm_iters=int(os.environ.get("ECM_ITERS", "50")), ecm_rtol=1e-6,
                    inner_iters=5, outer_passes=outer, min_outer=3, patience=2, shift_rtol=5e-3,
                    seed_q=bool(os.environ.get("SEED_Q")))
    # DIAG=1: with the reference's per-phase run diagnostics for every chromosome (core_api.PassDiagnostics: per-bin tracks from the
    # device, O(n) summaries on a worker thread); PROFILE=0: without the per-kernel event timing (it costs ~0.15 s of wall time)
    passes = None
    if os.environ.get("DIAG"):
        from consenrich_amd.core_api import PassDiagnostics
        passes = PassDiagnostics(cfg, ModelParams(state_dim=2), 200)
    b.synchronize(); b.profile(prof)
    t = time.perf_counter()
    fits, _ = run_consenrich_batch(b, cfg, block_len_intervals=750, download=False, pass_diagnostics=passes)
    b.synchronize(); wall = time.perf_counter() - t
    if passes is not None:
        passes.close()
    kt = b.kernel_times() if prof else {}; b.profile(False)
    rs = b.run_stats()
    b.close()
    ecm_total = sum(sum(f.ecm_iters) for f in fits)
    reasons = {}
    for f in fits:
        reasons[f.outer_stop_reason] = reasons.get(f.outer_stop_reason, 0) + 1
    return {"outer_passes_cap": outer, "x_tol_ulps": rs["x_tol_ulps"], "seed_q": cfg.seed_q,
            "q0_first_chain": None if fits[0].q0 is None else [float(fits[0].q0[0, 0]), float(fits[0].q0[1, 1])],
            "workload": f"hg38 @200bp x {m}, {len(lengths)} chromosomes, {sum(lengths)} bins", "wall_s": round(wall, 3),
            "kernel_timing": prof, "per_phase_diagnostics": passes is not None,
            "phase_records_first_chain": len(fits[0].loop_diagnostics),
            "outer_passes": [f.passes for f in fits], "converged": [f.converged for f in fits],
            "chains_that_met_the_stop_rule": int(sum(bool(f.converged) for f in fits)), "stop_reasons": reasons,
            "stop_reason_per_chain": [f.outer_stop_reason for f in fits],
            "ecm_iterations_total_over_chains": ecm_total, "ecm_iters_first_chain": fits[0].ecm_iters,
            "final_ecm_iters": [f.final_ecm_iters for f in fits], "final_nll_first_chain": fits[0].final_nll,
            "shift_first_chain": [round(x, 6) for x in fits[0].shift],
            "objective_per_cell_first_chain": [round(o["penalized_objective_per_cell"], 7) for o in fits[0].objective],
            "objective_stable_first_chain": [o["stable"] for o in fits[0].objective],
            "pipeline_redos": rs["pipeline_redos"], "state_chain_bailouts": rs["sb_bailouts"],
            "kernel_ms": {k: round(v[1], 1) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][1])[:10]}}


for outer in outers:
    print(json.dumps(run(outer)))
    sys.stdout.flush()
