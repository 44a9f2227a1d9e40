"""Time the residual kernel alone (no concurrent epilogue) on the bench workload."""
import sys, os, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from consenrich_amd import _lib as L
from consenrich_amd.batch import DeviceBatch, ModelParams
from consenrich_amd.sharding import hg38_chain_lengths
b = DeviceBatch(0)
b.configure(ModelParams(state_dim=2), 32, hg38_chain_lengths(200)); b.synthesize(1234)
b.step(L.RETURN_NLL, L.EXPORT_SMOOTH | L.EXPORT_RESID)
for _ in range(3):
    b.export(L.EXPORT_RESID)
b.synchronize(); b.profile(True)
for _ in range(10):
    b.export(L.EXPORT_RESID)
b.synchronize()
kt = b.kernel_times()
print(json.dumps({k: round(v[1] / max(v[0], 1), 4) for k, v in kt.items()}))
