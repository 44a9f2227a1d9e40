"""BUILD CONTAINER ONLY (needs /root/reference + `make -C oracle ref`): the CPU baseline bench.py reports is the oracle's C
port of the reference loops (`cpu_baseline.kind = "port"`, the compiled reference cannot travel to the GPU box).  SURVEY
8(d) asks that the port be within +-10 % of the compiled Cython reference, single-threaded: this script times both on the same
chr21-sized and chr1-sized synthetic chains (forward store+NLL, backward+residuals; one ECM iteration) and writes the ratios.

  python scripts/cpu_port_vs_reference.py > profiles/r02_cpu_port_vs_reference.json
"""
import json
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

import cases  # noqa: E402
from consenrich_amd.sharding import hg38_chain_lengths  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from oracle import ref_loader  # noqa: E402

ref = ref_loader.load()
if ref is None:
    sys.exit("the compiled reference is not available here (make -C oracle ref in the build container)")
orc.lib()
m = int(os.environ.get("M", "32"))
out = {"m": m, "cores": 1, "cases": []}
for label, n in (("chr21-sized", hg38_chain_lengths(200)[20]), ("chr1-sized", hg38_chain_lengths(200)[0])):
    data, munc = cases.synth(n, m, 21)
    F = np.asarray(cases.F_TREND, np.float32)
    Q0 = np.diag([1e-3, 1e-4]).astype(np.float32)
    bm = (np.arange(n) // 500).astype(np.int32)

    def fb(mod):
        xf, Pf, pn = np.empty((n, 2), np.float32), np.empty((n, 2, 2), np.float32), np.empty((n, 2, 2), np.float32)
        D = np.empty(n, np.float32)
        mod.cforwardPass(matrixData=data, matrixPluginMuncInit=munc, matrixF=F, matrixQ0=Q0, intervalToBlockMap=bm,
                         blockCount=int(bm.max()) + 1, stateInit=0.0, stateCovarInit=1000.0, stateForward=xf,
                         stateCovarForward=Pf, pNoiseForward=pn, vectorD=D, returnNLL=True)
        mod.cbackwardPass(matrixData=data, matrixF=F, stateForward=xf, stateCovarForward=Pf, pNoiseForward=pn)

    def ecm(mod):
        mod.cfixedBackgroundECM(matrixData=data, matrixPluginMuncInit=munc, matrixF=F, matrixQ0=Q0, intervalToBlockMap=bm,
                                blockCount=int(bm.max()) + 1, stateInit=0.0, stateCovarInit=1000.0,
                                ECM_fixedBackgroundIters=1, ECM_fixedBackgroundRtol=0.0, procPrecisionMultiplierMin=5e-3,
                                procPrecisionMultiplierMax=5e3, ECM_useObsPrecisionReweighting=False, t_innerIters=5,
                                logIterations=False)

    def best(fn, mod, reps):
        fn(mod)
        t = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn(mod)
            t.append(time.perf_counter() - t0)
        return min(t)

    reps = 5 if n < 500000 else 3
    r_fb, p_fb = best(fb, ref, reps), best(fb, orc, reps)
    r_e, p_e = best(ecm, ref, 2), best(ecm, orc, 2)
    out["cases"].append({"chain": label, "bins": n,
                         "forward_nll_backward_ms": {"reference": 1e3 * r_fb, "port": 1e3 * p_fb, "port_over_reference": p_fb / r_fb},
                         "ecm_iteration_ms": {"reference": 1e3 * r_e, "port": 1e3 * p_e, "port_over_reference": p_e / r_e},
                         "reference_bins_per_s": n / r_fb, "port_bins_per_s": n / p_fb})
print(json.dumps(out, indent=1))
