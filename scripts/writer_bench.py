"""SURVEY 8(f) rank 3 measurement: bedGraph text of a genome-wide track (hg38 @200bp, 14.4 M rows) formatted on the GPU vs
the reference's pandas to_csv call (consenrich.py:9797-9805)."""
import sys, os, time, json, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
from consenrich_amd import _lib as L, writers as w
from consenrich_amd.sharding import hg38_chain_lengths
from oracle import writers as ow

lengths = hg38_chain_lengths(200)
rng = np.random.default_rng(0)
lib = L.lib()
vals = [rng.normal(0, 5, n).astype(np.float32) for n in lengths]
w.bedgraph_bytes_regular("chr1", 0, 200, vals[0])                      # warm-up (allocations)
L.check(lib.csr_profile_enable(None, 1))
t = time.perf_counter(); total = 0
for i, v in enumerate(vals):
    total += len(w.bedgraph_bytes_regular(f"chr{i+1}", 0, 200, v))
wall = time.perf_counter() - t
buf = (L.KernelTime * 64)(); nn = C.c_int32()
L.check(lib.csr_profile_read(None, buf, 64, C.byref(nn)))
kt = {buf[i].name.decode(): buf[i].total_ms for i in range(nn.value) if buf[i].name.decode().startswith("bedgraph")}
L.check(lib.csr_profile_enable(None, 0))
n_small = 1_000_000
s = np.arange(n_small, dtype=np.int64) * 200
t = time.perf_counter(); ref = ow.bedgraph_bytes("chr1", s, s + 200, vals[0][:n_small]); cpu = time.perf_counter() - t
assert w.bedgraph_bytes_regular("chr1", 0, 200, vals[0][:n_small]) == ref
rows = sum(lengths); kms = sum(kt.values())
print(json.dumps({"row": "8(f) rank 3 bedGraph writer", "rows": rows, "text_bytes": total, "gpu_kernels_ms": {k: round(v, 3) for k, v in kt.items()},
                  "gpu_rows_per_s_kernels": rows / (kms * 1e-3), "text_GBps_kernels": total / (kms * 1e-3) / 1e9,
                  "wall_s_incl_pcie_and_python": round(wall, 3), "rows_per_s_wall": rows / wall,
                  "pandas_rows_per_s": n_small / cpu, "pandas_sample": f"{n_small} rows, 1 thread", "byte_identical": True}))
