"""SURVEY 8(f) rank 2 measurement: per-interval output diagnostics (core.py:7734-7878) on the bench workload.
GPU: csr_batch_diagnostics over hg38 @200bp x 32 (kernel time from HIP events; algorithmic bytes 4m + 52 per bin:
munc 4m, Pf 16, pNoise 16, five float32 outputs 20).  CPU: the NumPy oracle (vectorised restatement) on a
chr21-sized chain; the reference itself runs a per-bin Python loop there (core.py:7837-7865)."""
import sys, os, time, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
from consenrich_amd import _lib as L
from consenrich_amd.batch import DeviceBatch, ModelParams
from consenrich_amd.sharding import hg38_chain_lengths
from oracle import diagnostics as orc

m = int(os.environ.get("M", "32"))
lengths = hg38_chain_lengths(200)
b = DeviceBatch(0)
b.configure(ModelParams(state_dim=2), m, lengths); b.synthesize(1234)
b.stats(); b.forward(L.RETURN_NLL, True)
for _ in range(3): b.diagnostics(0)
b.synchronize(); b.profile(True)
K = 10
for _ in range(K): b.diagnostics(0)
b.synchronize(); kt = b.kernel_times(); b.profile(False)
nb = sum(lengths)
ms = kt["diagnostics"][1] / K
exp_ms = kt.get("export_natural", (0, 0.0))[1] / K
alg = (4 * m + 52) * nb
# CPU oracle on a chr21-sized chain
n = 233550
rng = np.random.default_rng(0)
covar = np.tile(np.asarray([[0.02, 0.001], [0.001, 0.01]], np.float32), (n, 1, 1))
munc = (0.25 * np.exp(rng.normal(0, 0.2, (m, n)))).astype(np.float32)
kw = dict(stateCovarForward=covar, matrixMunc=munc, matrixQ0=np.diag([1e-3, 1e-4]).astype(np.float32),
          matrixF=np.asarray([[1, 1], [0, 1]], np.float32), stateCovarInit=1000.0, state_dim=2)
orc.output_diagnostic_tracks(**kw)
t = time.perf_counter(); orc.output_diagnostic_tracks(**kw); cpu = time.perf_counter() - t
print(json.dumps({"row": "8(f) rank 2 diagnostics", "bins": nb, "m": m, "gpu_kernel_ms": round(ms, 4),
                  "gpu_export_forward_ms": round(exp_ms, 4), "gpu_bins_per_s": nb / (ms * 1e-3),
                  "alg_bytes_per_bin": 4 * m + 52, "achieved_GBps": alg / (ms * 1e-3) / 1e9, "frac_of_8TBps": alg / (ms * 1e-3) / 8e12,
                  "cpu_numpy_oracle_bins_per_s": n / cpu, "cpu_sample": f"{n} bins x {m}, 1 thread, vectorised NumPy restatement"}))
