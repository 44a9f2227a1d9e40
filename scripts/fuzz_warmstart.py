"""One-off robustness run: ECM with warm-started sweeps against the same ECM with cold windows (tolerant mode) over random
ragged batches and chain masks: same iteration counts, NLL paths to 1e-7, smoothed state within the mode's
tolerance class."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "tests", "golden")): sys.path.insert(0, p)
import numpy as np
import cases
from consenrich_amd import _lib as L
from consenrich_amd.batch import DeviceBatch, ModelParams
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
def run(n_list, m, seed, env, mask):
    os.environ.pop("CONSENRICH_AMD_WARMSTART", None)
    os.environ.update(env)
    with DeviceBatch(0, x_tol_ulps=2) as b:
        b.configure(ModelParams(state_dim=2), m, n_list)
        for c, n in enumerate(n_list):
            d_, v_ = cases.synth(n, m, seed + c, outlier_frac=0.01)
            b.upload(c, d_, v_)
        b.stats()
        outs, paths = b.ecm(max_iters=4, inner_iters=5, rtol=1e-7, use_lambda=False, use_kappa=True, chain_mask=mask)
        b.export(L.EXPORT_SMOOTH | L.EXPORT_MULT)
        res = [(int(o.iters_done), np.asarray(paths[c]).copy(), b.download(c, "xs"), b.download(c, "kappa")) for c, o in enumerate(outs)]
        return res, b.run_stats()
worst = 0.0
for trial in range(int(os.environ.get("TRIALS", "10"))):
    n_list = [int(v) for v in rng.integers(6, 60000, size=int(rng.integers(1, 6)))]
    m = int(rng.choice([4, 16, 32])); seed = int(rng.integers(1, 10000))
    mask = None
    if len(n_list) > 1 and rng.integers(0, 2):
        mask = np.ones(len(n_list), np.uint8); mask[int(rng.integers(0, len(n_list)))] = 0
    cold, _ = run(n_list, m, seed, {"CONSENRICH_AMD_WARMSTART": "0"}, mask)
    warm, rs = run(n_list, m, seed, {}, mask)          # (the warm windows start at 32 bins and widen themselves; round 4 removed their switches)
    for c in range(len(n_list)):
        if mask is not None and mask[c] == 0: continue
        assert cold[c][0] == warm[c][0], (trial, c, cold[c][0], warm[c][0])
        k = cold[c][0]
        np.testing.assert_allclose(warm[c][1][:k], cold[c][1][:k], rtol=1e-7)
        scale = np.abs(cold[c][2]).max(axis=1, keepdims=True)
        e = float((np.abs(warm[c][2].astype(np.float64) - cold[c][2]) / (1e-5 * scale + 2e-6)).max())
        worst = max(worst, e)
    print("trial", trial, "ok: chains", n_list, "m", m, "mask", None if mask is None else mask.tolist(),
          "local repairs", rs["local_repairs"], "redos", rs["pipeline_redos"], "windows now", rs["ws_warm_f"], rs["ws_warm_b"], flush=True)
print("worst |warm - cold| / tolerance over all trials: %.3f" % worst)
