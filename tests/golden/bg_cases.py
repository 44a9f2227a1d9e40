"""Case table for the background-update natives: `csolveZeroCenteredBackground` (pyx:944-1096) and
`cbackgroundWeightedStatsWithSupport` (pyx:9700-9724).  Inputs are re-synthesised from the seed; the committed
bg_*.npz fixtures hold the REAL reference's outputs (tests/golden/make_golden.py)."""
from __future__ import annotations

import numpy as np

# literal vectors of the reference's own test (tests/test_core.py:2512-2520)
LIT_W = [0.7, 1.4, 2.2, 0.9, 3.1, 1.8, 0.6]
LIT_R = [0.3, -1.1, 2.4, 0.8, -0.7, 1.6, -0.2]


def penalties(span, smoothness):
    """core.py:7478-7491"""
    s = max(2.0, float(span))
    return float(max(1.0, smoothness * s * s / 4.0)), float(max(1.0, smoothness * s ** 4 / 16.0))


def solve_cases():
    cs = [dict(name="bg_lit7", n=7, seed=0, span=3, smooth=2.0, literal=True)]
    for n in (1, 2, 3, 4, 5, 9, 64, 1000, 5000, 12000):
        for span, smooth in ((5, 0.8), (100, 128.0), (750, 128.0)):
            if n > 1000 and span == 5:
                continue
            cs.append(dict(name=f"bg_n{n}_span{span}", n=n, seed=1000 + n + span, span=span, smooth=smooth, literal=False))
    cs.append(dict(name="bg_n3000_nopenalty", n=3000, seed=77, span=0, smooth=0.0, literal=False))
    return cs


def solve_inputs(case):
    if case["literal"]:
        return np.asarray(LIT_W, np.float64), np.asarray(LIT_R, np.float64)
    n = case["n"]
    rng = np.random.default_rng(case["seed"])
    w = 128.0 * np.exp(rng.normal(0.0, 0.3, n))
    if n > 8:
        w[rng.random(n) < 0.03] = 0.0                      # masked bins
        if n >= 1000:
            w[n // 3: n // 3 + n // 50] = 0.0               # a masked stretch
    base = 0.3 * np.sin(np.arange(n) / max(n / 7.0, 3.0)) + 0.05
    r = w * (base + rng.normal(0.0, 0.09, n))
    return w, r


def solve_lams(case):
    if case["span"] == 0:
        return 0.0, 0.0
    return penalties(case["span"], case["smooth"])


def run_solve(mod, case):
    w, r = solve_inputs(case)
    lam_first, lam = solve_lams(case)
    out = {}
    for zc in (False, True):
        key = "zc" if zc else "plain"
        try:
            out[key] = np.asarray(mod.csolveZeroCenteredBackground(w, r, lam, zc, lamFirst=lam_first))
        except RuntimeError as e:                       # pivot modification: the message is the expected output
            out[key + "_error"] = np.asarray(str(e))
    return out


def stats_inputs(seed=5, m=6, n=777):
    rng = np.random.default_rng(seed)
    res = rng.normal(0, 1, (m, n)).astype(np.float32)
    inv = np.abs(rng.normal(4, 2, (m, n))).astype(np.float32)
    inv[:, 11] = 0.0
    inv[2, 100:140] = 0.0
    return res, inv
