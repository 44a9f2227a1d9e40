"""SURVEY 8(f) rank 4 measurement: initial process-noise seed (core.py:3621-3780) for the 22 hg38 autosomes x 32 samples
from the device-resident matrices (one call, all chains) vs the CPU oracle (= the reference's natives bit for bit, driven
by the restated caller, which like the reference converts both matrices to float64 and builds the activity mask on the
host) on chr1."""
import sys, os, time, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
from consenrich_amd.batch import DeviceBatch, ModelParams
from consenrich_amd.sharding import hg38_chain_lengths
from oracle import qseed as oq

m = 32
lens = hg38_chain_lengths(200)
rng = np.random.default_rng(0)
n0 = lens[0]
x = np.cumsum(rng.normal(0, 0.03, n0))
data = (x[None, :] + rng.normal(0, 0.5, (m, n0))).astype(np.float32)
munc = (0.25 * np.exp(rng.normal(0, 0.2, (m, n0)))).astype(np.float32)
munc[rng.random((m, n0)) < 0.02] = np.float32(1e30)
kw = dict(pad=1e-4, stateModel="levelTrend", minQ=1e-6, maxQ=1000.0, deltaF=1.0, robustTNu=8.0)
b = DeviceBatch(0); b.configure(ModelParams(), m, lens); b.synthesize(1); b.upload(0, data, munc)
b.qseed(**kw); b.synchronize()
b.profile(True)
ts = []
for _ in range(5):
    t = time.perf_counter(); got = b.qseed(**kw); ts.append(time.perf_counter() - t)
kt = {k: (v[0], round(v[1], 3)) for k, v in b.kernel_times().items() if k.startswith("qseed")}
b.profile(False)
t = time.perf_counter(); Q, diag = oq.estimate_initial_process_noise(oq, matrixData=data, matrixMunc=munc, **kw); cpu = time.perf_counter() - t
assert np.array_equal(Q, got[0][0]), (Q, got[0][0])
rel = abs(diag["qSeedPosteriorMedianLevel"] - got[0][1]["qSeedPosteriorMedianLevel"]) / diag["qSeedPosteriorMedianLevel"]
assert rel <= 1e-12, rel      # device log / log1p / exp vs glibc (the grid posterior runs on the device since round 2)
print(json.dumps({"row": "8(f) rank 4 Q0 seed", "m": m, "chains": len(lens), "bins": int(sum(lens)),
                  "gpu_wall_ms_all_chains_best": round(min(ts) * 1e3, 2), "gpu_wall_ms_all_chains_median": round(sorted(ts)[2] * 1e3, 2),
                  "gpu_kernels_ms_total_(launches,ms)_over_5_calls": kt,
                  "cpu_oracle_ms_chr1_only": round(cpu * 1e3, 1), "chr1_bins": n0,
                  "chr1_Q": [float(Q[0, 0]), float(Q[1, 1])], "chr1_source": diag["qSeedSource"],
                  "identical_Q": True, "posterior_median_rel_diff_vs_cpu": rel}))
