"""Device-resident multi-chain batch: the genome-level driver above the C ABI (include/consenrich_amd.h, level 2).

The reference processes chromosomes sequentially (consenrich.py:8809); here every chain of a rank is packed into one
batch so a single launch covers ~10^4-10^5 speculative blocks.  Inputs stay in HBM across sweeps.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L


@dataclass
class ModelParams:
    """Estimator parameters that reach the hot path (defaults: reference constants.py:140-153, 247-249)."""
    state_dim: int = 2
    F: tuple = ((1.0, 1.0), (0.0, 1.0))
    Q0: tuple = ((1.0e-3, 0.0), (0.0, 1.0e-4))
    state_init: float = 0.0
    state_covar_init: float = 1000.0
    pad: float = 1.0e-4
    lambda_bounds: tuple = (0.25, 4.0)
    kappa_bounds: tuple = (5.0e-3, 5.0e3)
    apn: tuple = (1.0e-4, 1000.0, 5.0, 10.0, 2.0)

    def to_c(self) -> L.Model:
        f32 = lambda v: float(np.float32(v))  # noqa: E731
        mdl = L.Model()
        mdl.state_dim = int(self.state_dim)
        F = np.asarray(self.F, np.float32)
        Q = np.asarray(self.Q0, np.float32)
        mdl.F[:] = [float(F[0, 0]), float(F[0, 1]), float(F[1, 0]), float(F[1, 1])]
        mdl.Q0[:] = [float(Q[0, 0]), float(Q[0, 1]), float(Q[1, 0]), float(Q[1, 1])]
        mdl.state_init, mdl.state_covar_init, mdl.pad = f32(self.state_init), f32(self.state_covar_init), f32(self.pad)
        mdl.w_min, mdl.w_max = f32(self.lambda_bounds[0]), f32(self.lambda_bounds[1])
        mdl.k_min, mdl.k_max = f32(self.kappa_bounds[0]), f32(self.kappa_bounds[1])
        mdl.apn_min_q, mdl.apn_max_q, mdl.apn_thresh, mdl.apn_scale, mdl.apn_pc = (f32(v) for v in self.apn)
        return mdl


_ARR = {"D": L.ARR_D, "xf": L.ARR_XF, "Pf": L.ARR_PF, "pnoise": L.ARR_PNOISE, "xs": L.ARR_XS, "Ps": L.ARR_PS,
        "lag": L.ARR_LAG, "resid": L.ARR_RESID, "lambda": L.ARR_LAMBDA, "kappa": L.ARR_KAPPA, "qscale": L.ARR_QSCALE,
        "sumGain0": L.ARR_SUMGAIN0, "sumGain1": L.ARR_SUMGAIN1, "effectiveQLevel": L.ARR_EFFQ_LEVEL,
        "effectiveQTrend": L.ARR_EFFQ_TREND, "muncTrace": L.ARR_MUNCTRACE, "background": L.ARR_BACKGROUND,
        "background_next": L.ARR_BACKGROUND_NEXT}


class DeviceBatch:
    """One GPU, many chains.  Thin, explicit wrapper: every method is one C-ABI call.

    `x_tol_ulps`: carry validation of the forward state chain (csr_set_validation).  None / 0 = the default, bit-exact
    sequential semantics; 2 = the opt-in throughput mode (a few float32 ulps per pass; not a contract through an ECM loop)."""

    def __init__(self, device: int = 0, block_len: int = 0, warm=(-1, -1, -1), x_tol_ulps=None):
        L.require_gpu()
        self._lib = L.lib()
        self._ctx = self._lib.csr_create(int(device))
        if not self._ctx:
            raise L.ConsenrichAMDError(L.last_error())
        L.check(self._lib.csr_set_tuning(self._ctx, int(block_len), int(warm[0]), int(warm[1]), int(warm[2])))
        if x_tol_ulps is not None:
            L.check(self._lib.csr_set_validation(self._ctx, int(x_tol_ulps)))
        self.chain_lens = []
        self.m = 0
        self.d = 2
        self._comms = []

    def _attach_comm(self, comm):
        self._comms.append(comm)

    def _detach_comm(self, comm):
        if comm in self._comms:
            self._comms.remove(comm)

    def set_validation(self, x_tol_ulps: int):
        L.check(self._lib.csr_set_validation(self._ctx, int(x_tol_ulps)))

    def close(self):
        if self._ctx:
            for comm in list(getattr(self, "_comms", [])):      # a communicator holds the context's device and stream
                comm.close()
            self._lib.csr_destroy(self._ctx)
            self._ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- setup ---------------------------------------------------------------------------------------------------
    def configure(self, model: ModelParams, m: int, chain_lens):
        lens = np.ascontiguousarray(chain_lens, dtype=np.int64)
        mdl = model.to_c()
        L.check(self._lib.csr_batch_configure(self._ctx, C.byref(mdl), int(m), int(lens.size),
                                              lens.ctypes.data_as(L.I64P)))
        self.chain_lens = [int(v) for v in lens]
        self.m, self.d = int(m), int(model.state_dim)
        self.model = model

    def set_model(self, model: ModelParams):
        mdl = model.to_c()
        L.check(self._lib.csr_batch_set_model(self._ctx, C.byref(mdl)))
        self.model = model

    def set_chain_q(self, q_list):
        """Per-chain base process noise: one (2,2) (or (1,1) for the level model) matrix per chain, values taken as
        float32 like the reference's matrixQ0; None = the model's Q0 for every chain again."""
        if q_list is None:
            L.check(self._lib.csr_batch_set_chain_q(self._ctx, None))
            return
        if len(q_list) != len(self.chain_lens):
            raise ValueError("one Q0 per chain")
        flat = np.zeros((len(q_list), 4))
        for i, q in enumerate(q_list):
            q = np.asarray(q, np.float32).astype(np.float64)
            if q.shape == (1, 1):
                flat[i, 0] = q[0, 0]
            elif q.shape == (2, 2):
                flat[i] = q.reshape(-1)
            else:
                raise ValueError("Q0 must be (2,2) or (1,1)")
        L.check(self._lib.csr_batch_set_chain_q(self._ctx, L.dp(flat)))

    def set_tuning(self, block_len=0, warm_p=-1, warm_x=-1, warm_b=-1):
        L.check(self._lib.csr_set_tuning(self._ctx, int(block_len), int(warm_p), int(warm_x), int(warm_b)))

    def upload(self, chain: int, data: np.ndarray, munc: np.ndarray):
        data = np.ascontiguousarray(data, np.float32)
        munc = np.ascontiguousarray(munc, np.float32)
        if data.shape != (self.m, self.chain_lens[chain]) or munc.shape != data.shape:
            raise ValueError("data/munc must have shape (m, chain_len)")
        L.check(self._lib.csr_batch_upload(self._ctx, chain, L.fp(data), L.fp(munc)))

    def upload_multipliers(self, chain: int, lam=None, kappa=None, qscale=None):
        arrs = [None if a is None else np.ascontiguousarray(a, np.float32) for a in (lam, kappa, qscale)]
        for a in arrs:
            if a is not None and a.shape != (self.chain_lens[chain],):
                raise ValueError("multiplier arrays must have shape (chain_len,)")
        L.check(self._lib.csr_batch_upload_multipliers(self._ctx, chain, *(L.fp(a) for a in arrs)))

    def synthesize(self, seed: int):
        L.check(self._lib.csr_batch_synthesize(self._ctx, int(seed)))

    def download_inputs(self, chain: int):
        """(data, munc) of one chain as resident on the device, (m, n) float32 each."""
        shape = (self.m, self.chain_lens[chain])
        data, munc = np.empty(shape, np.float32), np.empty(shape, np.float32)
        L.check(self._lib.csr_batch_download_inputs(self._ctx, int(chain), L.fp(data), L.fp(munc)))
        return data, munc

    # -- compute -------------------------------------------------------------------------------------------------
    def stats(self):
        L.check(self._lib.csr_batch_stats(self._ctx))

    def forward(self, flags: int = L.RETURN_NLL, want_sums: bool = True):
        nc = len(self.chain_lens)
        if not want_sums:
            L.check(self._lib.csr_batch_forward(self._ctx, int(flags), None, None))
            return None, None
        sd, sn = np.zeros(nc), np.zeros(nc)
        L.check(self._lib.csr_batch_forward(self._ctx, int(flags), L.dp(sd), L.dp(sn)))
        return sd, sn

    def backward(self):
        L.check(self._lib.csr_batch_backward(self._ctx))

    def forward_backward(self, flags: int = L.RETURN_NLL, want_sums: bool = True):
        """forward() + backward() as one pipeline (one host synchronisation; `_runForwardBackward`, core.py:4207)."""
        nc = len(self.chain_lens)
        if not want_sums:
            L.check(self._lib.csr_batch_forward_backward(self._ctx, int(flags), None, None))
            return None, None
        sd, sn = np.zeros(nc), np.zeros(nc)
        L.check(self._lib.csr_batch_forward_backward(self._ctx, int(flags), L.dp(sd), L.dp(sn)))
        return sd, sn

    def sums(self):
        """Per-chain (sumD, sumNLL) of the resident forward pass; synchronises (and validates) everything queued."""
        nc = len(self.chain_lens)
        sd, sn = np.zeros(nc), np.zeros(nc)
        L.check(self._lib.csr_batch_sums(self._ctx, L.dp(sd), L.dp(sn)))
        return sd, sn

    def ecm(self, max_iters=50, inner_iters=5, rtol=1.0e-6, nu=8.0, use_lambda=False, use_kappa=True,
            use_apn=False, use_qscale=False, chain_mask=None):
        """chain_mask: optional per-chain booleans; chains with False keep their resident results untouched."""
        nc = len(self.chain_lens)
        cfg = L.EcmCfg(int(max_iters), int(inner_iters), float(np.float32(rtol)), float(np.float32(nu)),
                       int(use_lambda), int(use_kappa), int(use_apn), 0)
        outs = (L.EcmOut * nc)()
        path = np.zeros(nc * max(int(max_iters), 1))
        mask = None if chain_mask is None else bytes(bytearray(int(bool(t)) for t in chain_mask))
        L.check(self._lib.csr_batch_ecm_masked(self._ctx, C.byref(cfg), L.USE_QSCALE if use_qscale else 0, mask, outs,
                                               L.dp(path)))
        return list(outs), path.reshape(nc, -1)

    def step(self, flags: int = L.RETURN_NLL, what: int = 0, want_sums: bool = True):
        """stats() + forward_backward(flags) + export(what) + sums() in one C-ABI call: same kernels, same results; in the
        bit-exact mode the tail of each chain (smoother, residuals) starts as soon as that chain's filtered state stands."""
        if not want_sums:
            L.check(self._lib.csr_batch_step(self._ctx, int(flags), int(what), None, None))
            return None, None
        nc = len(self.chain_lens)
        sd, sn = np.zeros(nc), np.zeros(nc)
        L.check(self._lib.csr_batch_step(self._ctx, int(flags), int(what), L.dp(sd), L.dp(sn)))
        return sd, sn

    def step_forward(self, flags: int = L.RETURN_NLL, what: int = L.EXPORT_FORWARD, want_sums: bool = True):
        """stats() + forward(flags) + export(what & EXPORT_FORWARD) + sums() in one C-ABI call: the forward filter alone
        (`cforwardPass` without `cbackwardPass`, pyx:6393-6632; BASELINE config 2)."""
        if not want_sums:
            L.check(self._lib.csr_batch_step_forward(self._ctx, int(flags), int(what), None, None))
            return None, None
        nc = len(self.chain_lens)
        sd, sn = np.zeros(nc), np.zeros(nc)
        L.check(self._lib.csr_batch_step_forward(self._ctx, int(flags), int(what), L.dp(sd), L.dp(sn)))
        return sd, sn

    def forward_masked(self, flags: int, chain_mask):
        """forward() for the chains with chain_mask[c] true only; returns (sum_d, sum_nll) (masked chains: stale)."""
        nc = len(self.chain_lens)
        sd, sn = np.zeros(nc), np.zeros(nc)
        mask = bytes(bytearray(int(bool(t)) for t in chain_mask))
        L.check(self._lib.csr_batch_forward_masked(self._ctx, int(flags), mask, L.dp(sd), L.dp(sn)))
        return sd, sn

    def phase_tracks(self, chain: int, pad: float, with_fit: bool = False, use_lambda: bool = False, out=None):
        """Per-bin float64 tracks of one chain behind the two per-phase diagnostics of `runConsenrich` that read the (m, n)
        matrices (csr_batch_phase_tracks): `rel` = smoothed level - weighted mean of the background-adjusted observations
        (core.py:2663-2697); with_fit (a background proposal is resident): `fit` = sum_j iv (r - proposal)^2 and `cnt` = its
        cell count (core.py:4546-4552, 4587-4596).  Returns (rel, fit or None, cnt or None)."""
        n = self.chain_lens[chain]
        if out is not None:                 # caller-owned (n,) float64 / float64 / int32 buffers (re-used: no first-touch faults)
            rel, fit, cnt = out[0], (out[1] if with_fit else None), (out[2] if with_fit else None)
            for a, t in ((rel, np.float64), (fit, np.float64), (cnt, np.int32)):
                if a is not None and (a.dtype != t or a.shape != (n,) or not a.flags.c_contiguous):
                    raise ValueError("phase_tracks: out buffers must be contiguous (n,) float64, float64, int32")
        else:
            rel = np.empty(n, np.float64)
            fit = np.empty(n, np.float64) if with_fit else None
            cnt = np.empty(n, np.int32) if with_fit else None
        L.check(self._lib.csr_batch_phase_tracks(self._ctx, int(chain), int(bool(use_lambda)), float(pad), L.dp(rel),
                                                 L.dp(fit) if with_fit else None,
                                                 cnt.ctypes.data_as(C.POINTER(C.c_int32)) if with_fit else None))
        return rel, fit, cnt

    def gain_summary(self, chain: int, pad: float, use_lambda: bool = False, lambda_bounds=(0.25, 4.0)) -> dict:
        """`_finalForwardReplicateGainContigSummary` (core.py:7671-7731) of one chain from the resident forward pass
        (csr_batch_gain_summary: moments and the six order statistics around the quartile positions per replicate, exact); the
        quartiles are interpolated here the way NumPy's 'linear' method does (np.median / np.quantile of the reference)."""
        out = np.empty((self.m, 9))
        L.check(self._lib.csr_batch_gain_summary(self._ctx, int(chain), int(bool(use_lambda)), float(pad), float(lambda_bounds[0]),
                                                 float(lambda_bounds[1]), L.dp(out)))
        cnt = out[:, 0].astype(np.int64)
        res = {"count": [int(v) for v in cnt], "mean": [], "median": [], "sd": [], "iqr": []}

        def lerp(a, b, t):          # numpy.lib._function_base_impl._lerp
            d = b - a
            v = a + d * t
            if t >= 0.5:
                v = b - d * (1.0 - t)
            return a if d == 0 else v

        for j in range(self.m):
            if cnt[j] == 0:
                for k in ("mean", "median", "sd", "iqr"):
                    res[k].append(float("nan"))
                continue
            q = []
            for i, frac in enumerate((0.25, 0.5, 0.75)):
                pos = (cnt[j] - 1) * frac
                t = pos - np.floor(pos)
                q.append(lerp(out[j, 3 + 2 * i], out[j, 4 + 2 * i], float(t)))
            lo, hi = out[j, 5], out[j, 6]
            res["mean"].append(float(out[j, 1]))
            res["median"].append(float(lo if cnt[j] % 2 else 0.5 * (lo + hi)))     # np.median: the mean of the two middle values
            res["sd"].append(float(out[j, 2]))
            res["iqr"].append(float(q[2] - q[0]))
        return res

    def objective_terms(self, nu: float, lam_first: float, lam: float, negative_penalty_multiplier=1.0, pad=1.0e-4,
                        use_lambda_penalty=False, use_kappa_penalty=True, use_lambda_weights=False, use_nonnegative=True):
        """Everything of the reference's penalised objective (core.py:4418-4538) except the forward NLL, per chain, from
        the resident variances / multipliers / current background; list of dicts."""
        nc = len(self.chain_lens)
        mult = float("nan") if negative_penalty_multiplier is None else float(negative_penalty_multiplier)
        cfg = L.ObjectiveCfg(float(nu), float(lam_first), float(lam), mult, float(pad), int(use_lambda_penalty),
                             int(use_kappa_penalty), int(use_lambda_weights), int(use_nonnegative))
        out = (L.ObjectiveTerms * nc)()
        L.check(self._lib.csr_batch_objective_terms(self._ctx, C.byref(cfg), out))
        return [{k: getattr(o, k) for k, _ in L.ObjectiveTerms._fields_} for o in out]

    def diagnostics(self, flags: int = 0):
        """Per-interval output diagnostics (core.py:7734-7878) of the resident forward pass; results are the arrays
        sumGain0, sumGain1, effectiveQLevel, effectiveQTrend, muncTrace (download()).  flags: USE_LAMBDA / USE_KAPPA /
        USE_QSCALE = which resident multipliers enter (None in the reference call otherwise)."""
        L.check(self._lib.csr_batch_diagnostics(self._ctx, int(flags)))

    # -- background update between ECM phases -----------------------------------------------------
    def background_update(self, lam_first: float, lam: float, zero_center=False, use_nonnegative=True,
                          negative_penalty_multiplier=1.0, use_lambda=False, use_initial=True, max_passes=5,
                          block_len=0, raise_on_error=True, zero_state=False):
        """core.py:5064-5137 + 8085-8378 for every chain, device-resident: weight / rhs tracks from the original data,
        munc and the smoothed level, conditioning guard, pentadiagonal solve with the asymmetric-IRLS wrapper.  The
        proposal is the array "background_next"; returns one dict per chain.  Errors the reference raises
        (pivot modification, float64 reliability, non-finite solution) raise RuntimeError here unless
        raise_on_error=False.  zero_state=True is the reference's background warm start from the weighted data
        (`_estimateBackgroundWarmStart`, core.py:2809-2910: residual = data, no fit needs to be resident)."""
        nc = len(self.chain_lens)
        cfg = L.BgCfg(float(lam_first), float(lam),
                      float("nan") if negative_penalty_multiplier is None else float(negative_penalty_multiplier),
                      int(bool(zero_center)), int(bool(use_nonnegative)), int(bool(use_lambda)),
                      (L.BG_INIT_FROM_CURRENT if use_initial else 0) | (L.BG_ZERO_STATE if zero_state else 0),
                      int(max_passes), int(block_len))
        outs = (L.BgOut * nc)()
        L.check(self._lib.csr_batch_background_update(self._ctx, C.byref(cfg), outs))
        res = [{k: getattr(o, k) for k, _ in L.BgOut._fields_} for o in outs]
        if raise_on_error:
            for c, o in enumerate(res):
                if o["status"] == L.BG_BAD_PIVOT:
                    raise RuntimeError("roughness-penalized LDL factorization required pivot modification at index "
                                       f"{o['bad_index']} (pivot={o['bad_value']:.6g}, floor={1.0e-12:.6g}). [chain {c}]")
                if o["status"] == L.BG_UNRELIABLE:
                    raise RuntimeError("roughness-penalized LDL system exceeds float64 reliability: "
                                       f"roundoffIndex={o['roundoff_index']:.6g} threshold=1 [chain {c}]")
                if o["status"] == L.BG_NONFINITE:
                    raise RuntimeError(f"solver returned non-finite values [chain {c}]")
        return res

    def background_apply(self, take=None):
        """current background := last proposal (for the chains with take[c] true; default all)."""
        buf = None if take is None else bytes(bytearray(int(bool(t)) for t in take))
        L.check(self._lib.csr_batch_background_apply(self._ctx, buf))

    def set_background(self, chain: int, background=None):
        a = None if background is None else np.ascontiguousarray(background, np.float32)
        if a is not None and a.shape != (self.chain_lens[chain],):
            raise ValueError("background must have shape (chain_len,)")
        L.check(self._lib.csr_batch_set_background(self._ctx, chain, L.fp(a)))

    def bedgraph_bytes(self, chain: int, name: str, chrom: str, start0: int, step: int, end_cap: int = 0, comp: int = 0,
                       transform=None) -> bytes:
        """bedGraph text (consenrich.py:9797-9805 format) of component `comp` of an exported array of one chain,
        formatted on the device.  transform: None, "round4" (state track) or "sqrt" (uncertainty from a variance)."""
        t = {None: 0, "none": 0, "round4": 1, "sqrt": 2}[transform]
        cn = str(chrom).encode("ascii")
        f = self._lib.csr_batch_format_bedgraph
        size = f(self._ctx, chain, _ARR[name], comp, t, cn, int(start0), int(step), int(end_cap), None, 0)
        if size < 0:
            raise L.ConsenrichAMDError(L.last_error())
        if size == 0:
            return b""
        buf = C.create_string_buffer(int(size))
        if f(self._ctx, chain, _ARR[name], comp, t, cn, int(start0), int(step), int(end_cap), buf, int(size)) != size:
            raise L.ConsenrichAMDError(L.last_error() or "bedGraph writer size mismatch")
        return buf.raw

    def bigwig_track(self, chain: int, name: str, chrom_id: int, start0: int, step: int, end_cap: int = 0, comp: int = 0,
                     transform=None, zoom_bases=()):
        """The fixed-record body of a bigWig file for component `comp` of an exported array of one chain, formatted on the
        device (io.py:530-790 is the reference's pyBigWig conversion of its bedGraph): data sections + total summary + one
        set of reduction records per entry of `zoom_bases` (bases per record, multiples of `step`).  Values = float32 of the
        "%.4f" text of the (transformed) track value, i.e. what the reference's file holds.  Returns a
        consenrich_amd.bigwig.TrackPiece for `bigwig.write_bigwig`."""
        from . import bigwig as BW

        t = {None: 0, "none": 0, "round4": 1, "sqrt": 2}[transform]
        f = self._lib.csr_batch_bigwig_sections
        args = (self._ctx, int(chain), _ARR[name], int(comp), t, int(chrom_id), int(start0), int(step), int(end_cap))
        size = f(*args, BW.ITEMS_PER_SECTION, None, 0, None)
        if size < 0:
            raise L.ConsenrichAMDError(L.last_error())
        buf = C.create_string_buffer(int(size))
        summ = L.BwSummary()
        if f(*args, BW.ITEMS_PER_SECTION, buf, int(size), C.byref(summ)) != size:
            raise L.ConsenrichAMDError(L.last_error() or "bigWig sections size mismatch")
        if summ.non_finite:
            raise ValueError(f"Non-finite bedGraph value in chain {chain} ({summ.non_finite} intervals)")     # io.py:713-716
        zooms = {}
        for zb in zoom_bases:
            if int(zb) % int(step):
                raise ValueError("zoom levels must be multiples of the interval step")
            g = self._lib.csr_batch_bigwig_zoom
            zsize = g(*args, int(zb) // int(step), None, 0)
            if zsize < 0:
                raise L.ConsenrichAMDError(L.last_error())
            zbuf = C.create_string_buffer(int(zsize))
            if g(*args, int(zb) // int(step), zbuf, int(zsize)) != zsize:
                raise L.ConsenrichAMDError(L.last_error() or "bigWig zoom size mismatch")
            zooms[int(zb)] = zbuf.raw
        return BW.TrackPiece(int(chrom_id), buf.raw, self.chain_lens[chain], int(summ.bases_covered), float(summ.min_val),
                             float(summ.max_val), float(summ.sum_data), float(summ.sum_squares), zooms)

    def make_fold(self, src: int, dst: int, block_len: int, fold: int, block_fold, reps_count, reps, pad: float,
                  rho: float = 0.0, masked_variance: float = 1.0e30):
        """Delete-block calibration fold as an extra chain (uncertainty.py:1370-1419, cuncertainty.pyx:160-305): chain `dst`
        receives chain `src`'s data and variances with the fold's deleted (replicate, block) cells masked; returns the
        (kept, heldout, h) information tracks.  Fit all chains afterwards (ecm / driver.fit_batch)."""
        n = self.chain_lens[src]
        bf = np.ascontiguousarray(block_fold, np.int32)
        rc = np.ascontiguousarray(reps_count, np.int64)
        rb = np.ascontiguousarray(reps, np.int64)
        kept, held, h = np.empty(n), np.empty(n), np.empty(n)
        L.check(self._lib.csr_batch_make_fold(self._ctx, int(src), int(dst), int(block_len), int(fold),
                                              bf.ctypes.data_as(C.POINTER(C.c_int32)), rc.ctypes.data_as(L.I64P),
                                              rb.ctypes.data_as(L.I64P), rb.shape[1], 0, float(pad), float(rho),
                                              float(masked_variance), L.dp(kept), L.dp(held), L.dp(h)))
        return kept, held, h

    def qseed(self, *, pad: float, stateModel: str = "levelTrend", minQ: float = 1.0e-6, maxQ: float = 1000.0,
              deltaF: float = 1.0, robustTNu=8.0, qSeedPriorLevel: float = 1.0e-5):
        """Initial process-noise seed of every chain (core.py:3621-3780 `_estimateInitialProcessNoiseFromData`) from the
        resident float32 data / variance matrices: list of (matrixQ float32 (2,2), diagnostics dict), one per chain."""
        from . import qseed as Q

        for name, v in (("minQ", minQ), ("minQ", qSeedPriorLevel)):
            if not np.isfinite(v) or v <= 0.0:
                raise ValueError(f"`{name}` must be positive and finite")
        cfg = Q.seed_config(pad=pad, stateModel=stateModel, minQ=minQ, maxQ=maxQ, deltaF=deltaF, robustTNu=robustTNu,
                            qSeedPriorLevel=qSeedPriorLevel)
        out = (L.QseedOut * len(self.chain_lens))()
        Q._call(self._lib.csr_batch_qseed(self._ctx, C.byref(cfg), out))
        return [Q.seed_result(o, float(minQ)) for o in out]

    def export(self, what: int):
        L.check(self._lib.csr_batch_export(self._ctx, int(what)))

    def synchronize(self):
        L.check(self._lib.csr_synchronize(self._ctx))

    # -- results -------------------------------------------------------------------------------------------------
    def download(self, chain: int, name: str, out=None) -> np.ndarray:
        n, d, m = self.chain_lens[chain], self.d, self.m
        shape = {"D": (n,), "xf": (n, d), "Pf": (n, d, d), "pnoise": (max(n - 1, 0), d, d), "xs": (n, d),
                 "Ps": (n, d, d), "lag": (max(n - 1, 0), d, d), "resid": (n, m)}.get(name, (n,))
        if out is None:
            out = np.empty(shape, np.float32)
        elif out.dtype != np.float32 or out.shape != shape or not out.flags.c_contiguous:
            raise ValueError(f"download({name!r}): out must be a contiguous float32 array of shape {shape}")
        L.check(self._lib.csr_batch_download(self._ctx, chain, _ARR[name], out.ctypes.data_as(C.c_void_p)))
        return out

    def device_array(self, name: str):
        ptr, cnt = C.c_void_p(), C.c_int64()
        L.check(self._lib.csr_batch_device_array(self._ctx, _ARR[name], C.byref(ptr), C.byref(cnt)))
        return ptr.value, int(cnt.value)

    def chain_offset(self, chain: int) -> int:
        return int(self._lib.csr_batch_chain_offset(self._ctx, chain))

    # -- instrumentation -----------------------------------------------------------------------------------------
    def profile(self, on: bool):
        L.check(self._lib.csr_profile_enable(self._ctx, int(on)))

    def kernel_times(self):
        buf = (L.KernelTime * 64)()
        n = C.c_int32()
        L.check(self._lib.csr_profile_read(self._ctx, buf, 64, C.byref(n)))
        return {buf[i].name.decode(): (int(buf[i].launches), float(buf[i].total_ms)) for i in range(min(n.value, 64))}

    def run_stats(self):
        rs = L.RunStats()
        L.check(self._lib.csr_get_run_stats(self._ctx, C.byref(rs)))
        return {k: getattr(rs, k) for k, _ in L.RunStats._fields_}
