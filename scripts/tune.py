"""Sweep speculative warm-up lengths on the bench workload; print per-kernel ms and rerun counts."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
from consenrich_amd import _lib as L
from consenrich_amd.batch import DeviceBatch, ModelParams
from consenrich_amd.sharding import hg38_chain_lengths
m = int(os.environ.get("M", "32")); B = int(os.environ.get("B", "0"))
lengths = hg38_chain_lengths(int(os.environ.get("BINBP", "200")))
if os.environ.get("SHARD"):      # "8:0" = the contigs rank 0 of 8 would own
    from consenrich_amd.sharding import lpt_assign
    w, r = map(int, os.environ["SHARD"].split(":")); lengths = [lengths[i] for i in lpt_assign(lengths, w)[r]]
b = DeviceBatch(0, block_len=B)
b.configure(ModelParams(state_dim=2), m, lengths); b.synthesize(int(os.environ.get("SEED", "1234")))
what = L.EXPORT_FORWARD | L.EXPORT_SMOOTH | L.EXPORT_RESID
def step():
    if os.environ.get("SPLIT_CALLS"):
        b.stats(); b.forward_backward(L.RETURN_NLL, False); b.export(what); b.sums()
    else:
        b.step(L.RETURN_NLL, what)
cfgs = [tuple(map(int, c.split(','))) for c in os.environ.get('CFGS', '256,256,128').split(';')]
for cfg in cfgs:
    b.set_tuning(0, *cfg)
    for _ in range(2): step()
    b.synchronize(); t=time.perf_counter()
    for _ in range(20): step()
    b.synchronize(); clean=(time.perf_counter()-t)/20
    r0 = b.run_stats()
    b.profile(True); t=time.perf_counter()
    for _ in range(5): step()
    b.synchronize(); dt=(time.perf_counter()-t)/5
    kt = b.kernel_times(); b.profile(False); r1=b.run_stats()
    sel = {k: round(v[1]/5,3) for k,v in kt.items() if k in ("fwd_chain","fwd_fix","fwd_dstat","fwd_cov_chain","fwd_state_chain","bwd_chain","fwd_state_fix","fwd_cov_fix","bwd_fix","stats","residuals","export_natural")}
    print("warm",cfg,"ms/step %.3f (profiled %.2f)"%(clean*1e3, dt*1e3), "reruns/step p,x,b", [(r1[k]-r0[k])/5 for k in ("reruns_p","reruns_x","reruns_b")], "fixl", (r1["fix_launches"]-r0["fix_launches"])/5, sel, flush=True)
