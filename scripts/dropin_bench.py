"""Wall time of the reference-shaped callables (host buffers, PCIe included) on a chr1-sized chain in both validation modes,
next to the CPU oracle (= the reference, bit for bit)."""
import sys, os, time, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests", "golden"))
import numpy as np
import cases
from consenrich_amd import cconsenrich as amd
from oracle import oracle as orc

n, m = 1244783, int(os.environ.get("M", "32"))
data, munc = cases.synth(n, m, 4242)
F = np.asarray(cases.F_TREND, np.float32); Q0 = np.diag([1e-3, 1e-4]).astype(np.float32)
bm = (np.arange(n) // 500).astype(np.int32)
def run(mod):
    xf, Pf, pn = np.zeros((n, 2), np.float32), np.zeros((n, 2, 2), np.float32), np.zeros((n, 2, 2), np.float32)
    t = time.perf_counter()
    mod.cforwardPass(matrixData=data, matrixPluginMuncInit=munc, matrixF=F, matrixQ0=Q0, intervalToBlockMap=bm,
                     blockCount=int(bm.max()) + 1, stateInit=0.0, stateCovarInit=1000.0, stateForward=xf,
                     stateCovarForward=Pf, pNoiseForward=pn, returnNLL=True, ECM_useObsPrecisionReweighting=False,
                     ECM_useProcessPrecisionReweighting=False)
    t1 = time.perf_counter()
    res = mod.cbackwardPass(matrixData=data, matrixF=F, stateForward=xf, stateCovarForward=Pf, pNoiseForward=pn)
    t2 = time.perf_counter()      # (the outputs are released after the clock stops: unmapping 209 MB costs ~10 ms for either module)
    del res
    return t1 - t, t2 - t1
def run_ecm(mod, iters=2):
    # what core._runFixedBackgroundECMPhase calls (core.py:3257-3290): kappa re-weighting on, lambda off, 5 inner sweeps
    t = time.perf_counter()
    r = mod.cfixedBackgroundECM(matrixData=data, matrixPluginMuncInit=munc, matrixF=F, matrixQ0=Q0, intervalToBlockMap=bm,
                                blockCount=int(bm.max()) + 1, stateInit=0.0, stateCovarInit=1000.0,
                                ECM_fixedBackgroundIters=iters, ECM_fixedBackgroundRtol=0.0,
                                ECM_useObsPrecisionReweighting=False, ECM_useProcessPrecisionReweighting=True,
                                procPrecisionMultiplierMin=5e-3, procPrecisionMultiplierMax=5e3, t_innerIters=5,
                                returnIntermediates=True, logIterations=False)
    dt = time.perf_counter() - t
    del r
    return dt
out = {}
for k in (0, 2):
    amd.set_validation(k); run(amd)
    f, b = min(run(amd) for _ in range(3))
    run_ecm(amd)
    out[f"gpu_xtol{k}"] = {"forward_s": round(f, 4), "backward_s": round(b, 4), "ecm_2_iters_s": round(min(run_ecm(amd) for _ in range(2)), 4)}
amd.set_validation(0)
f, b = run(orc)
out["cpu_oracle"] = {"forward_s": round(f, 3), "backward_s": round(b, 3), "ecm_2_iters_s": round(run_ecm(orc), 3)}
print(json.dumps({"chain_bins": n, "m": m, **out}))

# ---- the runConsenrich-signature entry (consenrich_amd.core_api): ONE chromosome per call, like the reference's CLI loop
# (consenrich.py:8809), against the same fits as ONE device-resident batch (driver.run_consenrich_batch) ----------------------------
if not os.environ.get("NO_CORE_API"):
    from consenrich_amd import core_api
    from consenrich_amd.batch import DeviceBatch, ModelParams
    from consenrich_amd.driver import run_consenrich_batch
    from consenrich_amd.sharding import hg38_chain_lengths

    lengths = hg38_chain_lengths(200)[: int(os.environ.get("CHAINS", "22"))]
    # the CLI's defaults (constants.py:266-281) with a fixed base process noise (what scripts/fit_bench.py runs)
    kw = dict(deltaF=1.0, minQ=1.0e-6, maxQ=1000.0, stateInit=0.0, stateCovarInit=1000.0, boundState=False, stateLowerBound=0.0,
              stateUpperBound=0.0, blockLenIntervals=750, pad=1.0e-4, ECM_fixedBackgroundIters=50, ECM_fixedBackgroundRtol=1.0e-6,
              t_innerIters=5, ECM_robustTNu=8.0, ECM_useObsPrecisionReweighting=False, ECM_useProcessPrecisionReweighting=True,
              ECM_useAPN=False, ECM_outerIters=int(os.environ.get("OUTER", "32")), ECM_minOuterIters=3, ECM_backgroundShiftRtol=5.0e-3, ECM_outerNLLRtol=5.0e-5,
              ECM_backgroundSmoothness=128.0, fitBackground=True, returnScales=True, returnBackground=True,
              initialProcessQ=np.diag([1e-3, 1e-4]).astype(np.float32), returnPrecisionDiagnostics=True,
              # ... and what the CLI itself asks for on every chromosome (consenrich.py:9216, 9241-9243): the run diagnostics with
              # their per-phase records (device tracks + O(n) host summaries per ECM phase, core_api.PassDiagnostics)
              intervalSizeBP=200, returnDiagnostics=True)
    with DeviceBatch(0) as gen:                     # the bench recipe's matrices, brought to the host once (not timed)
        gen.configure(ModelParams(state_dim=2), m, lengths)
        gen.synthesize(1234)
        host = [gen.download_inputs(c) for c in range(len(lengths))]

    def one_call(d_, v_, diagnostics=True):
        k = dict(kw, returnDiagnostics=diagnostics)
        t0 = time.perf_counter()
        plan = core_api.resolve_call(d_, v_, k.pop("deltaF"), k.pop("minQ"), k.pop("maxQ"), **k)
        t1 = time.perf_counter()
        fit, final = core_api.run_plan(plan, device=0)          # context + upload + resident fit + download of the final pass
        t2 = time.perf_counter()
        res = core_api.assemble_result(plan, fit, final)
        t3 = time.perf_counter()
        return res, fit, {"resolve_s": t1 - t0, "device_s": t2 - t1, "assemble_s": t3 - t2, "total_s": t3 - t0}

    one_call(*host[0])                              # warm-up (first-touch of the pinned staging buffers)
    res, fit, t_chr1 = one_call(*host[0])
    _, _, t_chr1_plain = one_call(*host[0], diagnostics=False)
    # split of the device part of one call: upload / fit / download, on a context of its own
    k = dict(kw)
    plan = core_api.resolve_call(host[0][0], host[0][1], k.pop("deltaF"), k.pop("minQ"), k.pop("maxQ"), **k)
    with DeviceBatch(0) as b1:
        t0 = time.perf_counter()
        b1.configure(plan.model, m, [lengths[0]]); b1.upload(0, plan.data, plan.munc); b1.synchronize()
        t1 = time.perf_counter()
        fits1, _ = run_consenrich_batch(b1, plan.cfg, block_len_intervals=750, model_q0=core_api._pad_q(plan.q0), download=False)
        b1.synchronize()
        t2 = time.perf_counter()
        got = [b1.download(0, name) for name in ("xs", "Ps", "resid", "D", "background", "Pf", "pnoise", "lambda", "kappa",
                                                  "sumGain0", "sumGain1", "effectiveQLevel", "effectiveQTrend", "muncTrace")]
        t3 = time.perf_counter()
        del got
    split = {"configure_upload_s": t1 - t0, "fit_s": t2 - t1, "download_s": t3 - t2}
    t = time.perf_counter()
    seq_passes = []
    for d_, v_ in host:
        r_, f_, _ = one_call(d_, v_)
        seq_passes.append(int(f_.passes))
        del r_
    genome_sequential = time.perf_counter() - t
    with DeviceBatch(0) as bb:
        bb.configure(plan.model, m, lengths)
        for c, (d_, v_) in enumerate(host):
            bb.upload(c, d_, v_)
        bb.synchronize()
        t = time.perf_counter()
        fitsb, _ = run_consenrich_batch(bb, plan.cfg, block_len_intervals=750, model_q0=core_api._pad_q(plan.q0), download=False)
        bb.synchronize()
        genome_batch = time.perf_counter() - t
    print(json.dumps({"core_api_runConsenrich": {
        "settings": "the CLI's settings (50 ECM iterations, rtol 1e-6, 5 inner sweeps, process re-weighting on, background fitted, at most "
                    "%d outer passes [OUTER; the CLI's default is 32, constants.py:277; rounds 3-5 quoted 8], at least 3), returnDiagnostics + "
                    "returnPrecisionDiagnostics like the CLI's call, fixed Q0 = diag(1e-3, 1e-4), hg38 @200bp x %d, library default (bit-exact) "
                    "mode" % (int(os.environ.get("OUTER", "32")), m),
        "outer_passes_cap": int(os.environ.get("OUTER", "32")),
        "chr1_call": {k_: round(v_, 4) for k_, v_ in t_chr1.items()},
        "chr1_call_returnDiagnostics_false": {k_: round(v_, 4) for k_, v_ in t_chr1_plain.items()},
        "chr1_ecm_phases_recorded": len(res[-1]["post_process_noise_fit"]["fixed_background_ecm"]), "chr1_device_part": {k_: round(v_, 4) for k_, v_ in split.items()},
        "chr1_outer_passes": int(fit.passes), "chr1_ecm_iterations": [int(v_) for v_ in fit.ecm_iters],
        "chromosomes": len(lengths), "sequential_calls_total_s": round(genome_sequential, 3), "sequential_outer_passes": seq_passes,
        "one_batch_fit_s": round(genome_batch, 3), "one_batch_outer_passes": [int(f_.passes) for f_ in fitsb]}}))
