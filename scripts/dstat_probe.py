"""Time the NIS/NLL epilogue kernel alone (forward pass without a concurrent smoother) on the bench workload."""
import sys, os, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from consenrich_amd import _lib as L
from consenrich_amd.batch import DeviceBatch, ModelParams
from consenrich_amd.sharding import hg38_chain_lengths
b = DeviceBatch(0)
b.configure(ModelParams(state_dim=2), 32, hg38_chain_lengths(200)); b.synthesize(1234)
b.stats()
for _ in range(3):
    b.forward(L.RETURN_NLL)
b.profile(True)
for _ in range(10):
    b.forward(L.RETURN_NLL)
kt = b.kernel_times()
print(json.dumps({k: round(v[1] / max(v[0], 1), 4) for k, v in kt.items()}))
