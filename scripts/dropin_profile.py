"""cProfile of one `core_api.runConsenrich`-shaped call (chr1 x 32, the CLI's flags, OUTER outer passes): where the host time of the
per-call entry goes.   OUTER=8 python scripts/dropin_profile.py"""
import cProfile, os, pstats, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests", "golden"))
import numpy as np
from consenrich_amd import core_api
from consenrich_amd.batch import DeviceBatch, ModelParams
from consenrich_amd.sharding import hg38_chain_lengths

m = 32
n = hg38_chain_lengths(200)[0]
with DeviceBatch(0) as gen:
    gen.configure(ModelParams(state_dim=2), m, [n]); gen.synthesize(1234)
    data, munc = gen.download_inputs(0)
kw = dict(deltaF=1.0, minQ=1.0e-6, maxQ=1000.0, stateInit=0.0, stateCovarInit=1000.0, boundState=False, stateLowerBound=0.0,
          stateUpperBound=0.0, blockLenIntervals=750, pad=1.0e-4, ECM_fixedBackgroundIters=50, ECM_fixedBackgroundRtol=1.0e-6,
          t_innerIters=5, ECM_robustTNu=8.0, ECM_useObsPrecisionReweighting=False, ECM_useProcessPrecisionReweighting=True,
          ECM_useAPN=False, ECM_outerIters=int(os.environ.get("OUTER", "8")), ECM_minOuterIters=3, ECM_backgroundShiftRtol=5.0e-3,
          ECM_outerNLLRtol=5.0e-5, ECM_backgroundSmoothness=128.0, fitBackground=True, returnScales=True, returnBackground=True,
          initialProcessQ=np.diag([1e-3, 1e-4]).astype(np.float32), returnPrecisionDiagnostics=True, intervalSizeBP=200,
          returnDiagnostics=True)


def call():
    k = dict(kw)
    plan = core_api.resolve_call(data, munc, k.pop("deltaF"), k.pop("minQ"), k.pop("maxQ"), **k)
    fit, final = core_api.run_plan(plan, device=0)
    return core_api.assemble_result(plan, fit, final)


call(); call()
t = time.perf_counter(); call(); print("one call: %.3f s" % (time.perf_counter() - t))
pr = cProfile.Profile(); pr.enable(); call(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
