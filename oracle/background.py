"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's background update between ECM phases
(SURVEY.md 8(f) rank 1), on top of the oracle's natives (oracle.oracle.csolveZeroCenteredBackground /
cbackgroundWeightedStatsWithSupport, both pinned bit for bit to the compiled reference):

  * ``weight_rhs_tracks``    /root/reference/src/consenrich/core.py:5064-5083 (float32 inverse variances and residuals,
                             float64 sums);
  * ``penalties``            core.py:7478-7491;
  * ``solve_background``     core.py:8085-8233 `solveZeroCenteredBackground` (support / conditioning guard / dispatch);
  * ``_solve_nonnegative``   core.py:8236-8378 `_solveNonnegativeBackground` (asymmetric IRLS, <= 5 passes).

Pinning: `consenrich.core` itself cannot be imported in this image (third-party `itrigamma`, `structlog` absent, no
stand-ins written); the two natives carry the arithmetic and are pinned, the Python control flow restated here is
pinned by the expectations of the reference's own tests (tests/test_core.py:58-110, 2542-2581) reproduced in
tests/test_oracle_background.py.  Never imported by ``consenrich_amd``.
"""
from __future__ import annotations

import numpy as np

from . import oracle as _nat

EPS = float(np.finfo(np.float64).eps)


def penalties(span_intervals, smoothness):
    s = max(2.0, float(span_intervals))
    return float(max(1.0, smoothness * s * s / 4.0)), float(max(1.0, smoothness * s ** 4 / 16.0))


def weight_rhs_tracks(data, munc, xs_level, pad, lam=None, lam_bounds=(0.25, 4.0)):
    """float32 arithmetic exactly where the reference's NumPy expressions are float32"""
    munc = np.asarray(munc, np.float32)
    inv = (np.float32(1.0) / np.maximum(munc + np.float32(pad), np.float32(1.0e-8))).astype(np.float32)
    if lam is not None:
        inv = inv * np.clip(np.asarray(lam, np.float32).reshape(1, -1), np.float32(lam_bounds[0]), np.float32(lam_bounds[1]))
    res = np.asarray(data, np.float32) - np.asarray(xs_level, np.float32)[None, :]
    w = np.sum(inv, axis=0, dtype=np.float64)
    r = np.einsum("ij,ij->j", inv, res, dtype=np.float64)
    return w, r, inv, res


def _solve(w, r, lam_first, lam, zero_center):
    n = r.shape[0]
    if zero_center and n == 1:
        return np.zeros(1)
    return np.asarray(_nat.csolveZeroCenteredBackground(np.ascontiguousarray(w, np.float64), r, lam, bool(zero_center),
                                                        lamFirst=lam_first), np.float64)


def _solve_nonnegative(w, r, lam_first, lam, zero_center, multiplier, initial, max_passes=5):
    def finite(g):
        if not np.all(np.isfinite(g)):
            raise RuntimeError("solver returned non-finite values")
        return g

    plain = multiplier is None or not np.isfinite(multiplier) or multiplier <= 0.0
    pen = 0.0
    if not plain:
        pos = w[np.isfinite(w) & (w > 0.0)]
        scale = float(np.median(pos)) if pos.size else 1.0
        if not np.isfinite(scale) or scale <= 0.0:
            scale = 1.0
        pen = float(multiplier * scale)
        plain = not np.isfinite(pen) or pen <= 0.0
    if plain:
        return finite(_solve(w, r, lam_first, lam, zero_center)).astype(np.float32), 0
    prev = None
    if initial is not None:
        init = np.asarray(initial, np.float64).reshape(-1)
        if init.shape[0] != r.shape[0]:
            raise ValueError("initialBackground length must match interval count")
        prev = init < 0.0
        g = finite(_solve(w + pen * prev, r, lam_first, lam, zero_center))
    else:
        g = finite(_solve(w, r, lam_first, lam, zero_center))
    passes = 0
    for k in range(max_passes):
        neg = g < 0.0
        if prev is not None and np.array_equal(neg, prev):
            break
        if not neg.any():
            break
        prev = neg
        g = finite(_solve(w + pen * neg, r, lam_first, lam, zero_center))
        passes = k + 1
    return g.astype(np.float32), passes


def solve_background(w, r, span_intervals, smoothness=1.0, zero_center=False, use_nonnegative=True, multiplier=1.0,
                     initial=None, penalties_override=None, return_info=False):
    """core.py:8085-8233 with weightTrack / rhsTrack supplied (the path runConsenrich takes, core.py:5124-5136)"""
    w = np.ascontiguousarray(w, np.float64).reshape(-1)
    r = np.ascontiguousarray(r, np.float64).reshape(-1)
    n = r.shape[0]
    info = {"passes": 0, "roundoff_index": 0.0}
    if n < 1:
        return (np.zeros(0, np.float32), info) if return_info else np.zeros(0, np.float32)
    support = int(np.count_nonzero(w > 0.0))
    if support <= 0:
        out = np.zeros(n, np.float32)
        return (out, info) if return_info else out
    lam_first, lam = penalties_override if penalties_override is not None else penalties(span_intervals, smoothness)
    mean_pos = float(np.sum(w, dtype=np.float64) / float(support))
    ratio = float(1.0 + (4.0 * lam_first + 16.0 * lam) / mean_pos)
    if not np.isfinite(mean_pos) or mean_pos <= 0.0 or not np.isfinite(ratio) or ratio <= 0.0:
        raise RuntimeError("roughness-penalized LDL scale is invalid")
    info["roundoff_index"] = EPS * ratio
    if info["roundoff_index"] >= 1.0:
        raise RuntimeError("roughness-penalized LDL system exceeds float64 reliability")
    if use_nonnegative:
        out, info["passes"] = _solve_nonnegative(w, r, lam_first, lam, zero_center, multiplier, initial)
    else:
        out = _solve(w, r, lam_first, lam, zero_center).astype(np.float32)
    return (out, info) if return_info else out
