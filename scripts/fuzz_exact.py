"""One-off robustness run: the bit-exact mode on the superblock state chain against the sequential kernel (k_state_seq_trend),
bit for bit, over random ragged batches whose chain lengths sit on and around multiples of the superblock length, random
superblock / window settings, per-bin multipliers on and off, a general F; every batch also through csr_batch_step (twice), whose
tail is pipelined per chain."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "tests", "golden")): sys.path.insert(0, p)
import numpy as np
import cases
from consenrich_amd import _lib as L
from consenrich_amd.batch import DeviceBatch, ModelParams

rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
def run(n_list, m, seed, mult, F01, env):
    for k in ("CONSENRICH_AMD_SEQ_STATE", "CONSENRICH_AMD_SB_BINS", "CONSENRICH_AMD_TAIL_PCT"):
        os.environ.pop(k, None)
    os.environ.update({k: v for k, v in env.items() if k.startswith("CONSENRICH_")})
    mp = ModelParams(state_dim=2, F=((1.0, F01), (0.0, 1.0)) if F01 != "gen" else ((0.98, 0.7), (0.01, 0.97)), Q0=((1e-3, 0.0), (0.0, 1e-4)))
    out = {}
    with DeviceBatch(0, x_tol_ulps=0) as b:
        b.configure(mp, m, n_list)
        for c, n in enumerate(n_list):
            d_, v_ = cases.synth(n, m, seed + c, mask_frac=0.02, outlier_frac=0.01)
            b.upload(c, d_, v_)
            if mult:
                lam, kap, qs = cases.multipliers(n, seed + c)
                b.upload_multipliers(c, lam, kap, qs)
        fl = L.RETURN_NLL | ((L.USE_LAMBDA | L.USE_KAPPA | L.USE_QSCALE) if mult else 0)
        if env.get("USE_STEP"):      # one call: without multipliers the step pipelines its tail per chain (step_pipelined)
            for _ in range(2):
                sd, sn = b.step(fl, L.EXPORT_FORWARD | L.EXPORT_SMOOTH | L.EXPORT_RESID)
        else:
            b.stats()
            sd, sn = b.forward(fl); b.backward(); b.export(L.EXPORT_FORWARD | L.EXPORT_SMOOTH | L.EXPORT_RESID)
        out["sums"] = (np.asarray(sd).copy(), np.asarray(sn).copy())
        for c in range(len(n_list)):
            for name in ("D", "xf", "xs", "Ps", "resid"):
                out[(c, name)] = b.download(c, name)
        out["stats"] = b.run_stats()
    return out

cases_run = 0
for trial in range(int(os.environ.get("TRIALS", "12"))):
    sbb = int(rng.choice([64, 256, 1024, 8192]))
    sbw = int(rng.choice([0, 64, 448, 2048, 16384]))
    base = [sbb, sbb + 1, sbb - 1, 2 * sbb, 2 * sbb + 17, 1, 5, 63, 3 * sbb - 1]
    n_list = [int(v) for v in rng.choice(base, size=int(rng.integers(2, 7)))] + [int(rng.integers(1, 40000))]
    m = int(rng.choice([1, 3, 8])); mult = bool(rng.integers(0, 2)); F01 = [1.0, 0.5, "gen"][int(rng.integers(0, 3))]
    seed = int(rng.integers(1, 10000))
    ref = run(n_list, m, seed, mult, F01, {"CONSENRICH_AMD_SEQ_STATE": "1"})
    got = run(n_list, m, seed, mult, F01, {"CONSENRICH_AMD_SB_BINS": str(sbb)})
    for key, val in ref.items():
        if key == "stats": continue
        if key == "sums":
            assert np.array_equal(val[0], got[key][0]) and np.array_equal(val[1], got[key][1]), (trial, "sums")
        else:
            assert np.array_equal(val, got[key]), (trial, key, n_list, sbb, sbw, m, mult, F01)
    # the same batch through csr_batch_step (twice), thresholds that make small groups of finished chains
    stp = run(n_list, m, seed, mult, F01, {"CONSENRICH_AMD_SB_BINS": str(sbb),
                                           "CONSENRICH_AMD_TAIL_PCT": str(int(rng.choice([1, 10, 50]))) + ",1", "USE_STEP": "1"})
    for key, val in ref.items():
        if key == "stats": continue
        if key == "sums":
            assert np.array_equal(val[0], stp[key][0]) and np.array_equal(val[1], stp[key][1]), (trial, "sums (step)")
        else:
            assert np.array_equal(val, stp[key]), (trial, "step", key, n_list, sbb, sbw, m, mult, F01)
    cases_run += 1
    print("trial", trial, "ok: chains", n_list, "m", m, "mult", mult, "F01", F01, "superblock", sbb, "window", sbw, "reruns_x", got["stats"]["reruns_x"], "tail groups", stp["stats"]["tail_groups"], flush=True)
print("all", cases_run, "trials bit-identical")
