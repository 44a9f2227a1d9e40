"""GPU tests of what stays RESIDENT between calls of one context (run with -m gpu): a pass may leave some of its results in the
reference layout only (round 5's byte cuts) -- every later call that reads or partly rewrites the blocked copies must find them.
Round-5 advisor findings: masked passes after a step, imported forward results after a lean forward pass, a mode toggle between
the statistics and the forward pass, conversions remembered per fit."""
import os

import numpy as np
import pytest

import cases
from conftest import gpu_available

pytestmark = pytest.mark.gpu

F = np.asarray(cases.F_TREND, np.float32)
Q0 = np.diag([1e-3, 1e-4]).astype(np.float32)
ARRS = ("xf", "Pf", "xs", "Ps", "lag")


@pytest.fixture(scope="module")
def product():
    if not gpu_available():
        pytest.fail("GPU tests selected but no HIP device / library: the product has no CPU fallback")
    from consenrich_amd import cconsenrich

    return cconsenrich


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as orc

    orc.lib()
    return orc


def _oracle_pass(oracle, d_, v_):
    n = d_.shape[1]
    xf, Pf, pn, D = np.zeros((n, 2), np.float32), np.zeros((n, 2, 2), np.float32), np.zeros((n, 2, 2), np.float32), np.zeros(n, np.float32)
    oracle.cforwardPass(matrixData=d_, matrixPluginMuncInit=v_, matrixF=F, matrixQ0=Q0, intervalToBlockMap=np.zeros(n, np.int32),
                        blockCount=1, stateInit=0.0, stateCovarInit=1000.0, stateForward=xf, stateCovarForward=Pf,
                        pNoiseForward=pn, vectorD=D, returnNLL=True)
    bw = oracle.cbackwardPass(matrixData=d_, matrixF=F, stateForward=xf, stateCovarForward=Pf, pNoiseForward=pn)
    return {"xf": xf, "Pf": Pf, "xs": bw[0], "Ps": bw[1], "lag": bw[2][: n - 1]}


def _close(got, ref, lvl, msg):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    if got.ndim == 2 and got.shape[1] == 2:
        assert np.all(np.abs(got - ref) <= 1e-5 * lvl + 2e-6), msg
    else:
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=2e-6, err_msg=msg)


@pytest.mark.parametrize("natin", ["1", "0"], ids=["natin", "blocked-smoother"])
@pytest.mark.parametrize("xtol", [0, 2], ids=["exact", "ulp2"])
@pytest.mark.parametrize("then", ["forward_masked", "ecm_masked", "stats+forward_masked", "stats+ecm_masked"])
def test_a_masked_pass_after_a_step_leaves_the_other_chains_resident(product, oracle, monkeypatch, xtol, natin, then):
    """csr_batch_step leaves xf / Pf (2-ulp mode) or Pf (default mode, pipelined tails) and the smoothed arrays in the reference
    layout only.  A masked forward pass / a masked ECM call rewrites the blocked copies of ITS chains; the chains outside the mask
    must come back from every later conversion exactly as the step left them (csr_batch_forward_masked / csr_batch_ecm_masked:
    'chain c is left exactly as it is') -- and equal to the oracle."""
    from consenrich_amd import _lib as L
    from consenrich_amd.batch import DeviceBatch, ModelParams

    monkeypatch.setenv("CONSENRICH_AMD_NATIN", natin)
    n_list, m = [70000, 33000, 4, 50001], 4
    sets = [cases.synth(n, m, 8800 + c, outlier_frac=0.01) for c, n in enumerate(n_list)]
    mask = [True, False, True, False]
    what = L.EXPORT_FORWARD | L.EXPORT_SMOOTH | L.EXPORT_RESID
    with DeviceBatch(0, x_tol_ulps=xtol) as b:
        b.configure(ModelParams(state_dim=2), m, n_list)
        for c, (d_, v_) in enumerate(sets):
            b.upload(c, d_, v_)
        b.step(L.RETURN_NLL, what)
        before = {(c, a): b.download(c, a) for c in range(len(n_list)) for a in ARRS}
        if then.startswith("stats+"):       # the driver's order inside an outer pass: new statistics, then the masked call
            b.stats()
            then = then[6:]
        if then == "forward_masked":
            b.forward_masked(L.RETURN_NLL, mask)
            b.backward()
            b.export(L.EXPORT_FORWARD | L.EXPORT_SMOOTH)
            check = range(len(n_list))          # same data, same flags: the masked chains are recomputed to the same values
        else:
            outs, _ = b.ecm(max_iters=2, inner_iters=2, rtol=0.0, use_kappa=True, chain_mask=mask)
            assert [int(o.skipped) for o in outs] == [0, 2, 1, 2]
            b.export(L.EXPORT_FORWARD | L.EXPORT_SMOOTH)
            check = [c for c in range(len(n_list)) if not mask[c]]
        for c in check:
            ref = _oracle_pass(oracle, *sets[c]) if n_list[c] > 8 else None
            for a in ARRS:
                got = b.download(c, a)
                # What was NOT recomputed must come back bit for bit: every array of a chain outside an ECM call's mask, the
                # filtered arrays of a chain outside a forward pass's mask.  What WAS recomputed (the masked chains' forward
                # results; after `forward_masked` the smoothed arrays of every chain, by `backward()`): the same bits in the exact
                # mode (one sequential recursion whatever the kernel form and window); in the 2-ulp mode another pass may accept
                # other carries -- within the mode's tolerance, the oracle gate below
                untouched = not mask[c] and (then == "ecm_masked" or a in ("xf", "Pf"))
                if untouched or xtol == 0:
                    assert np.array_equal(got, before[(c, a)]), (then, xtol, natin, c, a)
                if ref is not None:
                    lvl = np.maximum(np.abs(ref["xs"][:, :1].astype(np.float64)), 1.0)
                    _close(got[: ref[a].shape[0]], ref[a], lvl, (c, a))


@pytest.mark.parametrize("xtol", [2, 0], ids=["ulp2", "exact"])
def test_backward_pass_on_imported_forward_results_after_a_lean_forward_pass(product, oracle, xtol):
    """cforwardPass(A) in the 2-ulp mode leaves A's xf / Pf in the reference layout only; a following cbackwardPass with the
    forward results of ANOTHER chain -- of the same (m, n), then of a different n (a reconfiguration) -- must smooth what the
    caller passed, not what was resident."""
    m = 4
    product.set_validation(xtol)
    try:
        for n_a, n_b in ((30000, 30000), (30000, 21000)):
            dA, vA = cases.synth(n_a, m, 4100)
            dB, vB = cases.synth(n_b, m, 4200)
            kw = dict(matrixF=F, matrixQ0=Q0, stateInit=0.0, stateCovarInit=1000.0)
            product.cforwardPass(matrixData=dA, matrixPluginMuncInit=vA, intervalToBlockMap=np.zeros(n_a, np.int32), blockCount=1,
                                 **kw)
            ref = _oracle_pass(oracle, dB, vB)
            pn = np.zeros((n_b, 2, 2), np.float32)
            pn[:] = Q0
            got = product.cbackwardPass(matrixData=dB, matrixF=F, stateForward=ref["xf"], stateCovarForward=ref["Pf"],
                                        pNoiseForward=pn)
            lvl = np.maximum(np.abs(ref["xs"][:, :1].astype(np.float64)), 1.0)
            _close(got[0], ref["xs"], lvl, ("xs", n_a, n_b))
            _close(got[1], ref["Ps"], lvl, ("Ps", n_a, n_b))
            _close(got[2][: n_b - 1], ref["lag"], lvl, ("lag", n_a, n_b))
    finally:
        product.set_validation(0)


def test_statistics_of_the_throughput_mode_are_recomputed_when_the_context_turns_exact(product):
    """set_validation(2); stats(); set_validation(0); forward(): the resident {S2c, log R} are a float32 pair (a property of the
    2-ulp mode's statistics) -- the bit-exact mode promises float64 statistics and recomputes them: D and the NLL equal a
    context that was exact all along, bit for bit."""
    from consenrich_amd import _lib as L
    from consenrich_amd.batch import DeviceBatch, ModelParams

    n_list, m = [60000, 9000], 8
    sets = [cases.synth(n, m, 5100 + c, outlier_frac=0.01) for c, n in enumerate(n_list)]

    def run(toggle):
        # (one block length for both: the per-chain sums are reduced block by block)
        with DeviceBatch(0, block_len=64, x_tol_ulps=2 if toggle else 0) as b:
            b.configure(ModelParams(state_dim=2), m, n_list)
            for c, (d_, v_) in enumerate(sets):
                b.upload(c, d_, v_)
            b.stats()
            if toggle:
                b.set_validation(0)
            sd, sn = b.forward(L.RETURN_NLL)
            b.export(L.EXPORT_FORWARD)
            assert b.run_stats()["x_tol_ulps"] == 0
            return np.array(sd), np.array(sn), [b.download(c, "D") for c in range(len(n_list))]

    a, t = run(False), run(True)
    assert np.array_equal(a[0], t[0]) and np.array_equal(a[1], t[1])
    for x, y in zip(a[2], t[2]):
        assert np.array_equal(x, y)


def test_remembered_conversions_follow_the_resident_fit(product):
    """The conversions of the multipliers / the smoothed state into the reference layout are remembered per fit (the per-phase
    run diagnostics ask chain by chain): a second export of the same fit launches nothing, a new ECM call or an upload is seen."""
    from consenrich_amd import _lib as L
    from consenrich_amd.batch import DeviceBatch, ModelParams

    n_list, m = [20000, 7000], 4
    sets = [cases.synth(n, m, 6100 + c, outlier_frac=0.02) for c, n in enumerate(n_list)]
    with DeviceBatch(0) as b:
        b.configure(ModelParams(state_dim=2), m, n_list)
        for c, (d_, v_) in enumerate(sets):
            b.upload(c, d_, v_)
        b.stats()
        b.ecm(max_iters=1, inner_iters=2, rtol=0.0, use_kappa=True)
        b.export(L.EXPORT_MULT | L.EXPORT_SMOOTH)
        k1, x1 = b.download(0, "kappa"), b.download(0, "xs")
        b.profile(True)
        b.export(L.EXPORT_MULT | L.EXPORT_SMOOTH)
        rel, _, _ = b.phase_tracks(1, 1e-4)
        times = b.kernel_times()
        b.profile(False)
        assert "export_natural" not in times, times                 # nothing to convert: same fit, same multipliers
        assert np.array_equal(k1, b.download(0, "kappa")) and np.array_equal(x1, b.download(0, "xs"))
        b.ecm(max_iters=2, inner_iters=2, rtol=0.0, use_kappa=True)
        b.export(L.EXPORT_MULT | L.EXPORT_SMOOTH)
        k2, x2 = b.download(0, "kappa"), b.download(0, "xs")
        assert not np.array_equal(k1, k2) and not np.array_equal(x1, x2)
        new = np.full(n_list[0], 1.5, np.float32)
        b.upload_multipliers(0, kappa=new)
        b.export(L.EXPORT_MULT)
        assert np.array_equal(b.download(0, "kappa"), new)
        assert np.array_equal(b.download(1, "kappa"), b.download(1, "kappa")) and np.all(np.isfinite(rel))


@pytest.mark.parametrize("use_lambda", [False, True], ids=["plain", "lambda"])
def test_gain_summary_on_the_device_equals_the_reference_expression(product, use_lambda):
    """`DeviceBatch.gain_summary` (csr_batch_gain_summary: per-replicate moments + exact order statistics by radix select on the
    resident final pass) against `_finalForwardReplicateGainContigSummary` restated with NumPy on the downloaded arrays
    (core.py:7671-7731: np.mean / np.median / np.std / np.quantile of the finite float64 gains): counts, medians and
    inter-quartile ranges EQUAL (the gains are the same float64 values, the selection is exact), mean and sd to 1e-12."""
    from consenrich_amd import _lib as L
    from consenrich_amd.batch import DeviceBatch, ModelParams

    n_list, m = [300001, 1000, 5, 2, 1, 4096], 5
    mp = ModelParams(state_dim=2)
    sets = [cases.synth(n, m, 7700 + c, mask_frac=0.03, outlier_frac=0.01) for c, n in enumerate(n_list)]
    sets[3][1][:] = np.float32(1.0e30)              # a chain whose every cell is masked: gains ~1e-27, still finite
    flags = L.RETURN_NLL | (L.USE_LAMBDA if use_lambda else 0)
    with DeviceBatch(0) as b:
        b.configure(mp, m, n_list)
        for c, (d_, v_) in enumerate(sets):
            b.upload(c, d_, v_)
            if use_lambda:
                lam, _kap, _qs = cases.multipliers(n_list[c], 7800 + c)
                b.upload_multipliers(c, lam, None, None)
        b.stats()
        b.forward(flags)
        b.export(L.EXPORT_FORWARD | L.EXPORT_MULT)
        for c, n in enumerate(n_list):
            got = b.gain_summary(c, 1.0e-4, use_lambda=use_lambda, lambda_bounds=mp.lambda_bounds)
            p00 = np.maximum(b.download(c, "Pf").astype(np.float64)[:, 0, 0], 0.0)
            prec = np.clip(b.download(c, "lambda").astype(np.float64), *mp.lambda_bounds) if use_lambda else np.ones(n)
            for j in range(m):
                g = (p00 * prec) / np.maximum(sets[c][1][j].astype(np.float64) + 1.0e-4, 1.0e-12)
                g = g[np.isfinite(g)]
                assert got["count"][j] == g.size == n
                q25, q75 = np.quantile(g, [0.25, 0.75])
                assert got["median"][j] == float(np.median(g)), (c, j)
                assert got["iqr"][j] == float(q75 - q25), (c, j)
                assert got["mean"][j] == pytest.approx(float(np.mean(g)), rel=1e-12)
                assert got["sd"][j] == pytest.approx(float(np.std(g)), rel=1e-10, abs=1e-300)


@pytest.mark.parametrize("xtol", [0, 2], ids=["exact", "ulp2"])
def test_constant_process_noise_rows_are_written_once_and_follow_the_model(product, xtol):
    """The reference-layout process-noise rows of a pass with ONE constant Q are a fill; a following step with the same Q does not
    write them again (csr_ctx::pnFillValid).  They must still follow the model: another Q0, per-bin process noise (kappa) in
    between, a different Q0 again."""
    from consenrich_amd import _lib as L
    from consenrich_amd.batch import DeviceBatch, ModelParams

    n_list, m = [50000, 20000, 7], 4
    sets = [cases.synth(n, m, 3300 + c) for c, n in enumerate(n_list)]
    what = L.EXPORT_FORWARD | L.EXPORT_SMOOTH | L.EXPORT_RESID
    qa, qb = ((1e-3, 0.0), (0.0, 1e-4)), ((2e-3, 0.0), (0.0, 5e-5))

    def rows(b, q):
        want = np.asarray(q, np.float32)
        for c, n in enumerate(n_list):
            pn = b.download(c, "pnoise")
            assert pn.shape == (n - 1, 2, 2) and np.all(pn == want[None]), (c, q)

    with DeviceBatch(0, x_tol_ulps=xtol) as b:
        b.configure(ModelParams(state_dim=2, Q0=qa), m, n_list)
        for c, (d_, v_) in enumerate(sets):
            b.upload(c, d_, v_)
        b.step(L.RETURN_NLL, what)
        rows(b, qa)
        b.profile(True)
        b.step(L.RETURN_NLL, what)
        fills = b.kernel_times().get("export_natural", (0, 0.0))[0]
        b.profile(False)
        rows(b, qa)
        b.set_model(ModelParams(state_dim=2, Q0=qb))
        b.step(L.RETURN_NLL, what)
        rows(b, qb)
        for c, n in enumerate(n_list):                      # per-bin process noise in between: the rows are Q0 / kappa
            b.upload_multipliers(c, None, np.full(n, 2.0, np.float32), None)
        b.step(L.RETURN_NLL | L.USE_KAPPA, what)
        assert np.allclose(b.download(0, "pnoise")[5], np.asarray(qb, np.float32) / 2.0)
        b.step(L.RETURN_NLL, what)
        rows(b, qb)
    from test_gpu_parity import _default_switches

    if _default_switches():         # (under the suite's mode switches a step converts other arrays under the same profile name)
        assert fills == 0, fills    # the second step with the same constant Q launched no conversion / fill at all
