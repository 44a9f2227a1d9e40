"""Experiment: the genome as K independent sub-batches (LPT split of the chromosomes) on K streams of ONE GPU, steps issued
back to back without host synchronisation in between -- the latency-bound chain kernels of one sub-batch overlap the
streaming kernels of another."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from consenrich_amd import _lib as L
from consenrich_amd.batch import DeviceBatch, ModelParams
from consenrich_amd.sharding import hg38_chain_lengths, lpt_assign
m = 32
lengths = hg38_chain_lengths(200)
what = L.EXPORT_FORWARD | L.EXPORT_SMOOTH | L.EXPORT_RESID
for K in (1, 2, 3, 4):
    parts = lpt_assign(lengths, K)
    bs = []
    for r in range(K):
        b = DeviceBatch(0); b.configure(ModelParams(state_dim=2), m, [lengths[i] for i in parts[r]]); b.synthesize(1234 + r); bs.append(b)
    def step():
        for b in bs:
            b.stats(); b.forward_backward(L.RETURN_NLL, False); b.export(what)
        for b in bs:
            b.sums()
    for _ in range(3): step()
    for b in bs: b.synchronize()
    t = time.perf_counter()
    for _ in range(10): step()
    for b in bs: b.synchronize()
    print("sub-batches", K, "ms/step %.3f" % ((time.perf_counter() - t) / 10 * 1e3), flush=True)
    for b in bs: b.close()
