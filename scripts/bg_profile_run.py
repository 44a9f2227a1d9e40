import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
import bg_cases
from consenrich_amd import _lib as L
from consenrich_amd.batch import DeviceBatch, ModelParams
from consenrich_amd.sharding import hg38_chain_lengths
lengths = hg38_chain_lengths(200)
lamF, lam = bg_cases.penalties(750, 128.0)
b = DeviceBatch(0)
b.configure(ModelParams(state_dim=2), 32, lengths); b.synthesize(1234)
b.stats(); b.forward_backward(L.RETURN_NLL, True)
for _ in range(2): b.background_update(lamF, lam, negative_penalty_multiplier=1.0)
b.synchronize()
