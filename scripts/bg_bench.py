"""SURVEY 8(f) rank 1 measurement: the pentadiagonal background solve (pyx:944-1096) for all hg38 autosomes @200bp in
one device pass (default span 750 bins, smoothness 128) vs the CPU oracle (== the reference, bit for bit).
Kernel times: HIP events on the library's stream; the host-buffer entry point also pays 24 B/bin of PCIe."""
import sys, os, time, json, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests", "golden"))
import numpy as np
import bg_cases
from consenrich_amd import _lib as L, cconsenrich as amd
from consenrich_amd.sharding import hg38_chain_lengths
from oracle import oracle as orc

lengths = hg38_chain_lengths(200)
rng = np.random.default_rng(0)
lamF, lam = bg_cases.penalties(750, 128.0)
ws, rs = [], []
for n in lengths:
    w = 128.0 * np.exp(rng.normal(0, 0.3, n)); w[rng.random(n) < 0.02] = 0.0
    ws.append(w); rs.append(w * (0.4 * np.sin(np.arange(n) / 40000.0) + rng.normal(0, 0.09, n)))
lib = L.lib()
res = {}
for bl in (512, 1024, 2048, 4096):
    amd.solveBackgroundBatch(ws, rs, lam, False, lamF, blockLen=bl)
    L.check(lib.csr_profile_enable(None, 1))
    t = time.perf_counter()
    K = 3
    for _ in range(K): out = amd.solveBackgroundBatch(ws, rs, lam, False, lamF, blockLen=bl)
    wall = (time.perf_counter() - t) / K
    buf = (L.KernelTime * 64)(); nn = C.c_int32()
    L.check(lib.csr_profile_read(None, buf, 64, C.byref(nn)))
    kt = {buf[i].name.decode(): buf[i].total_ms / K for i in range(nn.value)}
    L.check(lib.csr_profile_enable(None, 0))
    res[bl] = {"kernels_ms": {k: round(v, 3) for k, v in kt.items() if k.startswith("bg_")},
               "kernel_total_ms": round(sum(v for k, v in kt.items() if k.startswith("bg_")), 3), "wall_ms_with_pcie": round(wall * 1e3, 1)}
# CPU oracle on chr21-sized chain
i21 = int(np.argmin(np.abs(np.asarray(lengths) - 233550)))
t = time.perf_counter(); ref = orc.csolveZeroCenteredBackground(ws[i21], rs[i21], lam, False, lamFirst=lamF); cpu = time.perf_counter() - t
err = float(np.abs(out[i21] - ref).max() / np.abs(ref).max())
best = min(res.values(), key=lambda r: r["kernel_total_ms"])
nb = sum(lengths)
print(json.dumps({"row": "8(f) rank 1 background solve", "bins": nb, "chains": len(lengths), "by_block_len": res,
                  "gpu_bins_per_s_kernels": nb / (best["kernel_total_ms"] * 1e-3),
                  "cpu_oracle_bins_per_s": lengths[i21] / cpu, "cpu_sample": f"chain of {lengths[i21]} bins, 1 thread",
                  "max_rel_diff_vs_oracle": err}))

# ---- device-resident update on the bench batch (hg38 @200bp x 32): everything stays in HBM -------------------------
from consenrich_amd.batch import DeviceBatch, ModelParams
b = DeviceBatch(0)
b.configure(ModelParams(state_dim=2), 32, lengths); b.synthesize(1234)
b.stats(); b.forward_backward(L.RETURN_NLL, True)
b.background_update(lamF, lam, negative_penalty_multiplier=1.0)
b.synchronize(); b.profile(True)
t = time.perf_counter(); info = b.background_update(lamF, lam, negative_penalty_multiplier=1.0); b.synchronize(); wall = time.perf_counter() - t
kt = b.kernel_times(); b.profile(False)
print(json.dumps({"row": "8(f) rank 1 device-resident update (hg38 x 32)", "wall_ms": round(wall * 1e3, 2),
                  "irls_passes_max": max(o["passes"] for o in info),
                  "kernels_ms": {k: round(v[1], 3) for k, v in kt.items() if k.startswith("bg_") or k == "export_natural"},
                  "bins_per_s": nb / wall}))
