"""Cost of the per-phase run diagnostics of one `core_api.runConsenrich` call (chr1-sized, hg38 @200 bp x 32, CLI defaults): the
same call with and without `returnDiagnostics` (what the reference's CLI passes, consenrich.py:9243)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from consenrich_amd import core_api
from consenrich_amd.batch import DeviceBatch, ModelParams
n, m = int(os.environ.get("N", "1244783")), int(os.environ.get("M", "32"))
with DeviceBatch(0) as gen:
    gen.configure(ModelParams(state_dim=2), m, [n]); gen.synthesize(1234); d, v = gen.download_inputs(0)
kw = dict(stateInit=0.0, stateCovarInit=1000.0, boundState=False, stateLowerBound=0.0, stateUpperBound=0.0, blockLenIntervals=750,
          pad=1.0e-4, ECM_fixedBackgroundIters=50, ECM_fixedBackgroundRtol=1.0e-6, t_innerIters=5, ECM_useObsPrecisionReweighting=False,
          ECM_outerIters=8, ECM_minOuterIters=3, ECM_backgroundShiftRtol=5.0e-3, ECM_outerNLLRtol=5.0e-5, ECM_backgroundSmoothness=128.0,
          initialProcessQ=np.diag([1e-3, 1e-4]).astype(np.float32), intervalSizeBP=200)
for diag in (False, True, False, True):
    t = time.perf_counter()
    out = core_api.runConsenrich(d, v, 1.0, 1.0e-6, 1000.0, returnDiagnostics=diag, trackOptimizationPath=diag, **kw)
    dt = time.perf_counter() - t
    extra = {}
    if diag:
        post = out[-1]["post_process_noise_fit"]
        extra = {"phases": len(post["fixed_background_ecm"]), "background_objective_per_cell": post["background_objective_per_cell"],
                 "relative_sign_change_per_kb": post["relative_sign_change_per_kb"]}
    print(json.dumps({"returnDiagnostics": diag, "seconds": round(dt, 3), **extra}), flush=True)
