import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "tests", "golden")): sys.path.insert(0, p)
import numpy as np, cases, time
from consenrich_amd import cconsenrich as product
from oracle import oracle
n, m = int(os.environ.get("N", "1000000")), int(os.environ.get("M", "4"))
data, munc = cases.synth(n, m, 4242); lam, kap, qs = cases.multipliers(n, 4242)
F = np.asarray(cases.F_TREND, np.float32); Q0 = np.diag([1e-3, 1e-4]).astype(np.float32)
bm = (np.arange(n) // 500).astype(np.int32)
def run(mod):
    xf, Pf, pn = np.zeros((n, 2), np.float32), np.zeros((n, 2, 2), np.float32), np.zeros((n, 2, 2), np.float32)
    D = np.zeros(n, np.float32)
    t = time.time()
    r = mod.cforwardPass(matrixData=data, matrixPluginMuncInit=munc, matrixF=F, matrixQ0=Q0, intervalToBlockMap=bm, blockCount=int(bm.max()) + 1,
        stateInit=0.0, stateCovarInit=1000.0, stateForward=xf, stateCovarForward=Pf, pNoiseForward=pn, vectorD=D, returnNLL=True, processPrecExp=kap,
        procPrecisionMultiplierMin=5e-3, procPrecisionMultiplierMax=5e3)
    b = mod.cbackwardPass(matrixData=data, matrixF=F, stateForward=xf, stateCovarForward=Pf, pNoiseForward=pn)
    return r, xf, Pf, D, b, time.time() - t
ro = run(oracle)
for xtol in (0, 1, 2, 4):
    product.set_validation(xtol)
    rg = run(product); rg = run(product)
    xf, xo = rg[1], ro[1]; D, Do = rg[3].astype(np.float64), ro[3].astype(np.float64)
    xs, xso = rg[4][0], ro[4][0]; Ps, Pso = rg[4][1], ro[4][1]
    rel = np.abs(D - Do) / np.maximum(np.abs(Do), 1e-30)
    print(f"xtol={xtol} time {rg[5]:.3f}s  xf0 flips {np.mean(xf[:,0]!=xo[:,0]):.5f} max|dx0|/|x0| {np.max(np.abs(xf[:,0].astype(float)-xo[:,0])/np.maximum(np.abs(xo[:,0]),1e-3)):.2e}"
          f"  xs0 flips {np.mean(xs[:,0]!=xso[:,0]):.5f}  Pf flips {np.mean(rg[2]!=ro[2]):.6f} Ps00 maxrel {np.max(np.abs(Ps[:,0,0].astype(float)-Pso[:,0,0])/Pso[:,0,0]):.2e}"
          f"  D: frac>1e-5 {np.mean(rel>1e-5):.5f} max {rel.max():.2e}  dNLL rel {abs(rg[0][3]-ro[0][3])/abs(ro[0][3]):.2e}")
