"""CPU study of the tolerant validation mode (scripts/ubench/tolerant_emul.c): worst error of the smoothed state after a
6-iteration kappa-ECM of the speculative blocked recurrences against the sequential recursion, per acceptance rule.
Not product, not oracle.   python scripts/tolerant_emul.py [M] [OUTLIERS]"""
import ctypes as C, os, subprocess, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "tests", "golden"))
import cases

src = os.path.join(R, "scripts", "ubench", "tolerant_emul.c")
so = "/tmp/tolerant_emul.so"
subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", so, src, "-lm"])
lib = C.CDLL(so)
FP, DP = C.POINTER(C.c_float), C.POINTER(C.c_double)
lib.emul_ecm.argtypes = [C.c_int64, C.c_int64, FP, FP] + [C.c_double] * 7 + [C.c_int] * 7 + [C.c_double, FP, FP, FP, FP, DP]


def run(data, munc, B=0, Wf=0, Wb=0, rule=-1, k=0, budget=0.0, iters=6, inner=5, kappa_init=None, want_iter=False):
    m, n = data.shape
    xs, kap = np.empty((n, 2), np.float32), np.empty(n, np.float32)
    xi = np.empty((iters, n, 2), np.float32) if want_iter else None
    st = np.zeros(5)
    fp = lambda a: None if a is None else a.ctypes.data_as(FP)
    lib.emul_ecm(m, n, fp(data), fp(munc), 1.0, float(np.float32(1e-3)), float(np.float32(1e-4)), float(np.float32(1e-4)),
                 float(np.float32(5e-3)), float(np.float32(5e3)), 8.0, iters, inner, B, Wf, Wb, rule, k, budget,
                 fp(kappa_init), fp(xs), fp(kap), fp(xi), st.ctypes.data_as(DP))
    return xs, kap, xi, st


if __name__ == "__main__":
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    outl = float(sys.argv[2]) if len(sys.argv) > 2 else 0.03
    for i, n in enumerate([40000, 9000, 700]):
        d, v = cases.synth(n, m, 5100 + i, outlier_frac=outl)
        ref, kref, refi, _ = run(d, v, want_iter=True)
        scale = np.abs(ref.astype(np.float64)).max(axis=0, keepdims=True)
        tol = 1e-5 * scale + 2e-6
        print(f"n={n} m={m} outl={outl}: kappa min {kref.min():.4g} max {kref.max():.4g}  frac<=kmin {np.mean(kref <= 5.0001e-3):.4f}")
        for name, kw in [("spec only B32 W96/80", dict(B=32, Wf=96, Wb=80, rule=0)),
                         ("rule1 k=2 B32 W96/80", dict(B=32, Wf=96, Wb=80, rule=1, k=2)),
                         ("rule1 k=2 B32 W160/128", dict(B=32, Wf=160, Wb=128, rule=1, k=2)),
                         ("rule2 budget 4 B32 W96/80", dict(B=32, Wf=96, Wb=80, rule=2, k=2, budget=4.0)),
                         ("rule2 budget 8 B32 W96/80", dict(B=32, Wf=96, Wb=80, rule=2, k=2, budget=8.0)),
                         ("rule2 budget 16 B32 W96/80", dict(B=32, Wf=96, Wb=80, rule=2, k=2, budget=16.0)),
                         ("rule2 budget 8 B128 W96/80", dict(B=128, Wf=96, Wb=80, rule=2, k=2, budget=8.0))]:
            xs, kap, xi, st = run(d, v, want_iter=True, **kw)
            err = np.abs(xs.astype(np.float64) - ref) / tol
            per_it = [float((np.abs(xi[t].astype(np.float64) - refi[t]) / tol).max()) for t in range(xi.shape[0])]
            w = np.unravel_index(np.argmax(err), err.shape)
            print(f"  {name:28s} worst/tol {err.max():8.3f} at bin {w[0]} comp {w[1]}  per-iter {['%.2f' % e for e in per_it]}"
                  f"  reruns f {int(st[0])} b {int(st[1])} of {int(st[2])}  amp f {st[3]:.1f} b {st[4]:.1f}")
