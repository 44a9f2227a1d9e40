"""Golden-vector case table shared by make_golden.py (generator, build container only) and the parity tests.

A case is a plain dict of parameters; inputs are synthesised deterministically from it (seeded NumPy, the recipe
of SURVEY.md 8(d) / BASELINE.md 3) or are small literal matrices taken from the reference's own known-answer
tests (data only; /root/reference/tests/test_core.py:2914-2937, 3277-3297, 3356-3399).
`run_case(module, case)` drives ANY module that exposes the reference's consenrich.cconsenrich keyword interface
(the compiled reference, the C oracle, or the HIP product) and returns a dict of outputs.
"""
from __future__ import annotations

import numpy as np

F_TREND = [[1.0, 1.0], [0.0, 1.0]]


def synth(n: int, m: int, seed: int, mask_frac: float = 0.0, outlier_frac: float = 0.0):
    """SURVEY 8(d) synthetic chain: random-walk level + noise, log-normal munc."""
    rng = np.random.default_rng(seed)
    x = np.cumsum(rng.normal(0.0, 0.03, n))
    data = (x[None, :] + rng.normal(0.0, 0.5, (m, n))).astype(np.float32)
    munc = (0.25 * np.exp(rng.normal(0.0, 0.2, (m, n)))).astype(np.float32)
    if outlier_frac > 0:
        hit = rng.random((m, n)) < outlier_frac
        data = np.where(hit, data + rng.normal(0.0, 8.0, (m, n)).astype(np.float32), data).astype(np.float32)
    if mask_frac > 0:
        hit = rng.random((m, n)) < mask_frac
        munc = np.where(hit, np.float32(1.0e30), munc).astype(np.float32)  # core:2777 masked-cell encoding
    return np.ascontiguousarray(data), np.ascontiguousarray(munc)


def multipliers(n: int, seed: int):
    rng = np.random.default_rng(seed + 7919)
    lam = rng.uniform(0.1, 5.0, n).astype(np.float32)      # straddles the [0.25, 4] clamp
    kap = np.exp(rng.normal(0.0, 2.0, n)).astype(np.float32)  # straddles [5e-3, 5e3] rarely, [0.25,4] often
    qs = np.exp(rng.normal(0.0, 0.4, n)).astype(np.float32)
    if n:
        qs[0] = 1.0
    return lam, kap, qs


# literal matrices held by the reference's own tests (data)
LIT_A_DATA = [[0.2, 0.4, 0.5, 0.7, 0.1, -0.2, 0.0], [0.1, 0.3, 0.65, 0.5, 0.0, -0.1, 0.2]]
LIT_A_MUNC = [[0.40, 0.35, 0.30, 0.42, 0.38, 0.33, 0.31], [0.45, 0.37, 0.34, 0.40, 0.36, 0.35, 0.32]]
LIT_B_DATA = [[0.25, 0.10, 0.45, 0.75, 0.20, -0.10, 0.05], [0.15, 0.20, 0.35, 0.65, 0.10, -0.20, 0.15]]
LIT_B_MUNC = [[0.20, 0.25, 0.18, 0.22, 0.30, 0.28, 0.24], [0.27, 0.23, 0.31, 0.19, 0.29, 0.25, 0.33]]
LIT_B_LAM = [0.10, 0.50, 1.20, 3.50, 7.00, 0.80, 2.20]
LIT_B_KAP = [1.00, 0.20, 0.75, 2.50, 6.00, 1.25, 0.40]
LIT_B_QS = [1.00, 0.70, 1.80, 0.55, 1.20, 2.40, 0.90]
LIT_B_BLOCKS = [0, 0, 1, 1, 1, 2, 2]


def lit_c():
    """test_core.py:2914-2937 (_checkCFixedBackgroundPrecisionUpdates inputs)."""
    n = 8
    grid = np.linspace(-0.5, 0.8, n, dtype=np.float32)
    data = np.vstack([
        grid + np.asarray([0.0, 0.2, -0.1, 0.4, 0.0, -0.3, 0.1, 0.6], dtype=np.float32),
        grid + np.asarray([0.1, -0.2, 0.2, -0.1, 0.3, 0.1, -0.4, 0.2], dtype=np.float32),
    ]).astype(np.float32)
    munc = np.vstack([np.linspace(0.05, 0.20, n), np.linspace(0.12, 0.30, n)]).astype(np.float32)
    return data, munc


def _fb(name, d, n, m, seed, **kw):
    c = dict(kind="fb", name=name, d=d, n=n, m=m, seed=seed, Q0=[1e-3, 1e-4], init=0.0, cinit=1000.0, pad=1e-4,
             use_lam=False, use_kap=False, use_qs=False, nll=True, nll_in_d=False, apn=False,
             bounds=(0.25, 4.0, 5e-3, 5e3), mask=0.0, outl=0.0, F=F_TREND, block=500, literal=None)
    c.update(kw)
    return c


def _ecm(name, d, n, m, seed, **kw):
    c = dict(kind="ecm", name=name, d=d, n=n, m=m, seed=seed, Q0=[1e-3, 1e-4], init=0.0, cinit=1000.0, pad=1e-4,
             iters=3, rtol=1e-6, inner=2, nu=8.0, use_lam=True, use_kap=True, use_qs=False, apn=False,
             bounds=(0.25, 4.0, 5e-3, 5e3), mask=0.0, outl=0.0, F=F_TREND, block=500, literal=None,
             warm=False)
    c.update(kw)
    return c


def all_cases():
    cs = []
    for d in (2, 1):
        t = "trend" if d == 2 else "level"
        cs += [
            _fb(f"fb_{t}_n7_m2_plain", d, 7, 2, 11),
            _fb(f"fb_{t}_n1_m3", d, 1, 3, 12),
            _fb(f"fb_{t}_n2_m1", d, 2, 1, 13),
            _fb(f"fb_{t}_n64_m4_mult", d, 64, 4, 14, use_lam=True, use_kap=True, use_qs=True, nll_in_d=True),
            _fb(f"fb_{t}_n64_m4_nonll", d, 64, 4, 15, nll=False, use_kap=True),
            _fb(f"fb_{t}_n333_m5_mask", d, 333, 5, 16, mask=0.1, outl=0.02, use_lam=True),
            _fb(f"fb_{t}_n4096_m32", d, 4096, 32, 17, use_kap=True),
            _fb(f"fb_{t}_n4096_m4_smallq", d, 4096, 4, 18, Q0=[1e-5, 1e-6], use_lam=True, use_kap=True),
            _fb(f"fb_{t}_n3000_m8_apn", d, 3000, 8, 19, apn=True, outl=0.05),
            _fb(f"fb_{t}_n20000_m8", d, 20000, 8, 20, use_kap=True, use_qs=True),
            _fb(f"fb_{t}_n1500_m64", d, 1500, 64, 26),
            _ecm(f"ecm_{t}_n5_tiny", d, 5, 2, 21),
            _ecm(f"ecm_{t}_n200_m4", d, 200, 4, 22, iters=4, inner=3),
            _ecm(f"ecm_{t}_n4096_m8_defaults", d, 4096, 8, 23, use_lam=False, iters=6, inner=5),
            _ecm(f"ecm_{t}_n1000_m3_warm_qs", d, 1000, 3, 24, use_qs=True, warm=True, outl=0.03, mask=0.05,
                 iters=5, rtol=1e-4),
            _ecm(f"ecm_{t}_n600_m6_converge", d, 600, 6, 25, iters=50, rtol=1e-3, inner=2),
        ]
    # the reference tests' own literal inputs
    cs += [
        _fb("lit_levelA", 1, 7, 2, 0, literal="A", Q0=[0.06, 0.5], init=-0.1, cinit=0.8, pad=0.02, block=10**9),
        _fb("lit_levelB", 1, 7, 2, 0, literal="B", Q0=[0.045, 0.125], init=-0.15, cinit=0.7, pad=0.015,
            use_lam=True, use_kap=True, use_qs=True, nll_in_d=True, bounds=(0.25, 4.0, 0.25, 4.0)),
        _fb("lit_trendB_identityF", 2, 7, 2, 0, literal="B", Q0=[0.045, 0.125], init=-0.15, cinit=0.7, pad=0.015,
            use_lam=True, use_kap=True, use_qs=True, nll_in_d=True, bounds=(0.25, 4.0, 0.25, 4.0),
            F=[[1.0, 0.0], [0.0, 1.0]]),
        _ecm("lit_ecm_trendC", 2, 8, 2, 0, literal="C", Q0=[0.04, 0.02], init=0.0, cinit=1.0, pad=0.01, nu=5.0,
             iters=1, inner=1, rtol=0.0, bounds=(0.1, 10.0, 0.1, 10.0), F=[[1.0, 0.3], [0.0, 1.0]],
             block=10**9),
        _ecm("lit_ecm_levelC", 1, 8, 2, 0, literal="C", Q0=[0.04, 0.02], init=0.0, cinit=1.0, pad=0.01, nu=5.0,
             iters=1, inner=1, rtol=0.0, bounds=(0.1, 10.0, 0.1, 10.0), block=10**9),
    ]
    return cs


def inputs(case):
    lit = case.get("literal")
    if lit == "A":
        data, munc = np.asarray(LIT_A_DATA, np.float32), np.asarray(LIT_A_MUNC, np.float32)
    elif lit == "B":
        data, munc = np.asarray(LIT_B_DATA, np.float32), np.asarray(LIT_B_MUNC, np.float32)
    elif lit == "C":
        data, munc = lit_c()
    else:
        data, munc = synth(case["n"], case["m"], case["seed"], case["mask"], case["outl"])
    n = data.shape[1]
    if lit == "B":
        bm = np.asarray(LIT_B_BLOCKS, np.int32)
        lam, kap, qs = (np.asarray(v, np.float32) for v in (LIT_B_LAM, LIT_B_KAP, LIT_B_QS))
    else:
        bm = (np.arange(n) // case["block"]).astype(np.int32)
        lam, kap, qs = multipliers(n, case["seed"])
    return dict(data=data, munc=munc, bm=bm, bc=int(bm.max()) + 1 if n else 1, lam=lam, kap=kap, qs=qs)


def _q0(case):
    d = case["d"]
    q = np.zeros((2, 2), np.float32)
    q[0, 0], q[1, 1] = case["Q0"][0], case["Q0"][1]
    return q if d == 2 else q[:1, :1].copy()


def run_case(mod, case):
    """Drive `mod` (reference-compatible keyword interface) through the case; return {name: array/scalar}."""
    inp = inputs(case)
    d, (m, n) = case["d"], inp["data"].shape
    oMin, oMax, pMin, pMax = case["bounds"]
    F = np.asarray(case["F"], np.float32)
    Q0 = _q0(case)
    common = dict(matrixData=inp["data"], matrixPluginMuncInit=inp["munc"], matrixQ0=Q0,
                  intervalToBlockMap=inp["bm"], blockCount=inp["bc"], stateInit=case["init"],
                  stateCovarInit=case["cinit"], pad=case["pad"],
                  obsPrecisionMultiplierMin=oMin, obsPrecisionMultiplierMax=oMax,
                  procPrecisionMultiplierMin=pMin, procPrecisionMultiplierMax=pMax, ECM_useAPN=case["apn"],
                  processQScale=inp["qs"] if case["use_qs"] else None)
    if d == 2:
        common["matrixF"] = F
    out = {}
    if case["kind"] == "fb":
        xf = np.zeros((n, d), np.float32)
        Pf = np.zeros((n, d, d), np.float32)
        pn = np.zeros((n, d, d), np.float32)
        D = np.zeros(n, np.float32)
        fwd = mod.cforwardPass if d == 2 else mod.cforwardPassLevel
        r = fwd(**common, stateForward=xf, stateCovarForward=Pf, pNoiseForward=pn, vectorD=D,
                returnNLL=case["nll"], storeNLLInD=case["nll_in_d"],
                lambdaExp=inp["lam"] if case["use_lam"] else None,
                processPrecExp=inp["kap"] if case["use_kap"] else None,
                ECM_useObsPrecisionReweighting=True, ECM_useProcessPrecisionReweighting=True)
        out["phi"] = np.float64(r[0])
        if case["nll"]:
            out["nll"] = np.float64(r[3])
        out.update(D=D, xf=xf, Pf=Pf, pn=pn[: max(n - 1, 0)])
        if d == 2:
            b = mod.cbackwardPass(matrixData=inp["data"], matrixF=F, stateForward=xf, stateCovarForward=Pf,
                                  pNoiseForward=pn)
        else:
            b = mod.cbackwardPassLevel(matrixData=inp["data"], stateForward=xf, stateCovarForward=Pf,
                                       pNoiseForward=pn)
        out.update(xs=b[0], Ps=b[1], lag=b[2][: max(n - 1, 0)], resid=b[3])
        # a12's float64 follow-up (core:3451/3483) on the smoothed moments
        if d == 2:
            s = mod.cExpectedTransitionResidualSums(b[0].astype(np.float64), b[1].astype(np.float64),
                                                    b[2].astype(np.float64), F.astype(np.float64))
        else:
            s = mod.cExpectedTransitionResidualSumsLevel(b[0].astype(np.float64), b[1].astype(np.float64),
                                                         b[2].astype(np.float64))
        out["tsums"] = np.asarray(s, np.float64)
    else:
        ecm = mod.cfixedBackgroundECM if d == 2 else mod.cfixedBackgroundECMLevel
        lam0 = kap0 = None
        if case["warm"]:
            lam0, kap0 = inp["lam"], inp["kap"]
        r = ecm(**common, ECM_fixedBackgroundIters=case["iters"], ECM_fixedBackgroundRtol=case["rtol"],
                ECM_robustTNu=case["nu"], ECM_useObsPrecisionReweighting=case["use_lam"],
                ECM_useProcessPrecisionReweighting=case["use_kap"], t_innerIters=case["inner"],
                returnIntermediates=True, returnDiagnostics=True, lambdaExpInit=lam0,
                processPrecExpInit=kap0, logIterations=False)
        out["iters"] = np.int64(r[0])
        out["nll"] = np.float64(r[1])
        out.update(xs=r[2], Ps=r[3], lag=r[4][: max(n - 1, 0)], resid=r[5])
        if r[6] is not None:
            out["lam"] = r[6]
        if r[7] is not None:
            out["kap"] = r[7]
        diag = r[8]
        out["converged"] = np.int64(bool(diag["converged"]))
        out["skipped"] = np.int64(bool(diag["skipped"]))
    return {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in out.items()}


SAMPLE_FULL_MAX = 512


def sample_index(n: int) -> np.ndarray:
    """Rows kept in a compacted fixture: everything for small n, else head/tail + a strided sweep."""
    if n <= SAMPLE_FULL_MAX:
        return np.arange(n)
    idx = np.concatenate([np.arange(64), np.arange(n - 64, n), np.arange(0, n, max(n // 256, 1))])
    return np.unique(idx)


def compact(case, out):
    """Shrink outputs of large cases: keep sampled rows + float64 sum / abs-sum of every array."""
    n = case["n"]
    rec = {}
    for k, v in out.items():
        if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] >= n - 1 and n > SAMPLE_FULL_MAX:
            idx = sample_index(v.shape[0])
            rec[k + "__rows"] = v[idx]
            v64 = v.astype(np.float64)
            rec[k + "__sum"] = np.float64(v64.sum())
            rec[k + "__abs"] = np.float64(np.abs(v64).sum())
        else:
            rec[k] = v
    return rec


def compare(case, got, gold, rtol, atol, check=None):
    """Assert `got` (run_case output) matches a loaded fixture `gold` within rtol/atol. Returns max rel err seen."""
    worst = 0.0
    n = case["n"]
    keys = sorted({k.split("__")[0] for k in gold.keys()})
    for k in keys:
        if check is not None and k not in check:
            continue
        g = got[k]
        if k in gold:
            ref = gold[k]
            if np.asarray(ref).dtype.kind in "iu":
                assert int(g) == int(ref), f"{case['name']}:{k}: {g} != {ref}"
                continue
            np.testing.assert_allclose(np.asarray(g, np.float64), np.asarray(ref, np.float64), rtol=rtol, atol=atol,
                                       err_msg=f"{case['name']}:{k}")
            den = np.maximum(np.abs(np.asarray(ref, np.float64)), max(atol / max(rtol, 1e-300), 1e-300))
            if np.size(ref):
                with np.errstate(invalid="ignore", divide="ignore"):
                    worst = max(worst, float(np.nanmax(np.abs(np.asarray(g, np.float64) - ref) / den)))
        else:
            rows = gold[k + "__rows"]
            idx = sample_index(g.shape[0])
            np.testing.assert_allclose(g[idx].astype(np.float64), rows.astype(np.float64), rtol=rtol, atol=atol,
                                       err_msg=f"{case['name']}:{k} rows")
            g64 = g.astype(np.float64)
            scale = float(gold[k + "__abs"]) + atol * g64.size
            assert abs(g64.sum() - float(gold[k + "__sum"])) <= 4 * rtol * scale + atol, f"{case['name']}:{k} sum"
            assert abs(np.abs(g64).sum() - float(gold[k + "__abs"])) <= 4 * rtol * scale + atol, \
                f"{case['name']}:{k} abs-sum"
    return worst
