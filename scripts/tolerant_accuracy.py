"""Worst error of the smoothed state after a 6-iteration ECM in the tolerant mode (k = 2), in units of the parity tolerance
(1e-5 * max|row| + 2e-6), against the CPU oracle: cold speculative windows and warm-started ones (F = forward window, B =
smoother window; 0 = warm starting off).  OUTLIERS = fraction of outlier cells, M = samples.  Build container / GPU box."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "tests", "golden")): sys.path.insert(0, p)
import numpy as np
import cases
from oracle import oracle
oracle.lib()
from consenrich_amd import _lib as L
from consenrich_amd.batch import DeviceBatch, ModelParams
OUT = float(os.environ.get("OUTLIERS", "0.02")); n_list, m = [40000, 9000, 700], int(os.environ.get("M", "8"))
sets = [cases.synth(n, m, 5100 + i, outlier_frac=OUT) for i, n in enumerate(n_list)]
refs = []
for c, (d_, v_) in enumerate(sets):
    n = n_list[c]
    refs.append(oracle.cfixedBackgroundECM(matrixData=d_, matrixPluginMuncInit=v_, matrixF=np.asarray(cases.F_TREND, np.float32),
                                   matrixQ0=np.diag([1e-3, 1e-4]).astype(np.float32), intervalToBlockMap=np.zeros(n, np.int32), blockCount=1, stateInit=0.0,
                                   stateCovarInit=1000.0, ECM_fixedBackgroundIters=6, ECM_fixedBackgroundRtol=1e-7,
                                   ECM_useObsPrecisionReweighting=False, ECM_useProcessPrecisionReweighting=True,
                                   procPrecisionMultiplierMin=5e-3, procPrecisionMultiplierMax=5e3, t_innerIters=5,
                                   returnIntermediates=True, returnDiagnostics=False, logIterations=False))
def run(tag):
    with DeviceBatch(0, x_tol_ulps=2) as b:
        b.configure(ModelParams(state_dim=2), m, n_list)
        for c, (d_, v_) in enumerate(sets): b.upload(c, d_, v_)
        b.stats()
        outs, paths = b.ecm(max_iters=6, inner_iters=5, rtol=1e-7, use_lambda=False, use_kappa=True)
        b.export(L.EXPORT_SMOOTH | L.EXPORT_MULT)
        worst = 0.0
        for c in range(3):
            xs = b.download(c, "xs").astype(np.float64); r = refs[c]
            scale = np.abs(r[2]).max(axis=1, keepdims=True)
            worst = max(worst, float((np.abs(xs - r[2]) / (1e-5 * scale + 2e-6)).max()))
        rs = b.run_stats()
        print(tag, "worst xs error / tolerance %.3f" % worst, {k: rs[k] for k in ("local_repairs", "pipeline_redos", "reruns_p", "reruns_b", "ws_warm_f", "ws_warm_b")}, flush=True)
for ws in (0, 1):        # cold windows / warm-started sweeps (32-bin windows that widen themselves)
    os.environ["CONSENRICH_AMD_WARMSTART"] = str(ws)
    run("WARMSTART=%d" % ws)
