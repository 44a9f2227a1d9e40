import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
from consenrich_amd import _lib as L
from consenrich_amd.batch import DeviceBatch, ModelParams
from consenrich_amd.sharding import hg38_chain_lengths
lens = hg38_chain_lengths(1600); m = 32
b = DeviceBatch(0); b.configure(ModelParams(state_dim=2), m, lens); b.synthesize(1234); b.stats()
lib = L.lib()
lib.csr_debug_chain_step.argtypes=[C.c_void_p,C.c_int,C.c_int,C.c_int,C.c_uint32,C.c_int,C.POINTER(C.c_uint)]
lib.csr_debug_read.argtypes=[C.c_void_p,C.c_int,C.c_void_p,C.c_int64]
rs = b.run_stats(); B = rs["block_len"]; NB = rs["blocks"]
cnt = C.c_uint()
flags = L.RETURN_NLL
b.forward(flags, True)        # full forward so P outputs are final
lib.csr_debug_chain_step(b._ctx, 1, 0, 0, flags, 0, C.byref(cnt))   # X spec
cin = np.zeros((NB, 2), np.float32); out = np.zeros((NB, 2), np.float32)
lib.csr_debug_read(b._ctx, 0, cin.ctypes.data_as(C.c_void_p), cin.nbytes)
lib.csr_debug_read(b._ctx, 1, out.ctypes.data_as(C.c_void_p), out.nbytes)
# chain structure
starts = np.cumsum([0] + [(n + B - 1) // B for n in lens])
first = set(starts[:-1].tolist())
bad = [i for i in range(1, NB) if i not in first and (cin[i].view(np.uint32) != out[i-1].view(np.uint32)).any()]
print("B", B, "NB", NB, "bitwise mismatches", len(bad))
def ulps(a, b): return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
big = []
for i in bad:
    ch = np.searchsorted(starts, i, side="right") - 1
    d0 = abs(float(cin[i,0]) - float(out[i-1,0])); d1 = abs(float(cin[i,1]) - float(out[i-1,1]))
    u = np.spacing(np.float32(max(abs(cin[i,0]), abs(out[i-1,0]), 1.0)))
    if d0 + d1 > 2 * u: big.append((i, ch, i - starts[ch], float(out[i-1,0]), d0 / u, d1 / u))
print("beyond 2 ulp:", len(big)); [print("  block %d chain %d blockInChain %d level %.3f d0=%.2f ulp d1=%.2f ulp" % t) for t in big[:20]]
