import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "tests", "golden"))
import numpy as np
import test_core_api as T
import twin_core
from consenrich_amd import core_api
name = sys.argv[1] if len(sys.argv) > 1 else "apn_smoke"
data, munc, kw = T.CASES[name]()
k = dict(kw)
plan = core_api.resolve_call(data, munc, k.pop("deltaF"), k.pop("minQ"), k.pop("maxQ"), **k)
fit, final = core_api.run_plan(plan)
tfit, tfinal = twin_core.twin_run(plan)
print("q0 dev", np.asarray(final["matrixQ0"]).ravel(), "twin", np.asarray(tfinal["matrixQ0"]).ravel())
print("ecm_iters", fit.ecm_iters, tfit.ecm_iters)
print("nll", fit.nll, tfit.nll)
print("passes", fit.passes, tfit.passes, fit.outer_stop_reason, tfit.outer_stop_reason)
print("final nll", fit.final_nll, tfit.final_nll, "final ecm iters", fit.final_ecm_iters, tfit.final_ecm_iters)
for key in ("stateSmoothed", "stateCovarSmoothed", "background", "NIS", "stateCovarForward", "pNoiseForward"):
    a, b = np.asarray(final[key], np.float64), np.asarray(tfinal[key], np.float64)
    print(key, a.shape, b.shape, "max abs diff", float(np.abs(a - b).max()), "max |ref|", float(np.abs(b).max()))
print("pn dev", np.asarray(final["pNoiseForward"])[:6].reshape(6, -1))
print("pn twin", np.asarray(tfinal["pNoiseForward"])[:6].reshape(6, -1))
