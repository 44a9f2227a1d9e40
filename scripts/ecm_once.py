"""Two ECM iterations on the bench workload (for rocprofv3 counter passes)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from consenrich_amd.batch import DeviceBatch, ModelParams
from consenrich_amd.sharding import hg38_chain_lengths
b = DeviceBatch(0); b.configure(ModelParams(state_dim=2), 32, hg38_chain_lengths(200)); b.synthesize(1234); b.stats()
b.ecm(max_iters=2, inner_iters=5, rtol=0.0, use_lambda=False, use_kappa=True)
b.synchronize()
