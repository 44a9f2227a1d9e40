"""Run a few bench steps (for profiling under rocprofv3) in the library's default validation mode (bit-exact); XTOL=2: the opt-in
throughput mode."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from consenrich_amd import _lib as L
from consenrich_amd.batch import DeviceBatch, ModelParams
from consenrich_amd.sharding import hg38_chain_lengths
m = int(os.environ.get("M", "32"))
b = DeviceBatch(0, x_tol_ulps=int(os.environ["XTOL"]) if os.environ.get("XTOL") else None)
lengths = hg38_chain_lengths(int(os.environ.get("BINBP", "200")))
if os.environ.get("SHARD"):
    from consenrich_amd.sharding import lpt_assign
    w, r = map(int, os.environ["SHARD"].split(":")); lengths = [lengths[i] for i in lpt_assign(lengths, w)[r]]
b.configure(ModelParams(state_dim=2), m, lengths); b.synthesize(1234)
what = L.EXPORT_FORWARD | L.EXPORT_SMOOTH | L.EXPORT_RESID
for _ in range(int(os.environ.get("STEPS", "3"))):
    if os.environ.get("SPLIT_CALLS"):
        b.stats(); b.forward_backward(L.RETURN_NLL, False); b.export(what); b.sums()
    else:
        b.step(L.RETURN_NLL, what)          # what bench.py times
b.synchronize()
