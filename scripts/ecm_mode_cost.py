import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "tests", "golden")): sys.path.insert(0, p)
import numpy as np, cases
from consenrich_amd import cconsenrich as product
from consenrich_amd import _lib as L
import ctypes as C
for (n, m) in ((233550, 8), (1244783, 32)):
    data, munc = cases.synth(n, m, 2121, outlier_frac=0.01)
    kw = dict(matrixData=data, matrixPluginMuncInit=munc, matrixF=np.asarray(cases.F_TREND, np.float32),
              matrixQ0=np.diag([1e-3, 1e-4]).astype(np.float32), intervalToBlockMap=np.zeros(n, np.int32), blockCount=1,
              stateInit=0.0, stateCovarInit=1000.0, ECM_fixedBackgroundIters=6, ECM_fixedBackgroundRtol=0.0,
              procPrecisionMultiplierMin=5e-3, procPrecisionMultiplierMax=5e3, ECM_useObsPrecisionReweighting=False,
              t_innerIters=5, returnIntermediates=True, logIterations=False)
    res = {}
    for xt in (2, 0, 2, 0):
        product.set_validation(xt)
        t = time.perf_counter(); r = product.cfixedBackgroundECM(**kw); dt = time.perf_counter() - t
        res[xt] = r
        print(f"n={n} m={m} xtol={xt}: {dt*1e3:.1f} ms total for 6 ECM iters (incl. PCIe) nll={r[1]:.6f}", flush=True)
    print("  max|dxs0| tol-vs-exact:", float(np.max(np.abs(res[2][2][:,0].astype(float)-res[0][2][:,0]))), " kappa max rel diff:", float(np.max(np.abs(res[2][7]-res[0][7])/res[0][7])))
