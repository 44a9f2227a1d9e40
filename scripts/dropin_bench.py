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
