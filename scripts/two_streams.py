"""Experiment: a batch as K independent sub-batches (LPT split of its chromosomes) on K streams of ONE GPU, steps issued
back to back without host synchronisation in between -- the latency-bound chain kernels of one sub-batch overlap the
streaming kernels of another.  SHARD=8:6 restricts the batch to the chromosomes rank 6 of 8 would own; THREADS=1 issues
every sub-batch from its own host thread."""
import sys, os, time, threading
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from consenrich_amd import _lib as L
from consenrich_amd.batch import DeviceBatch, ModelParams
from consenrich_amd.sharding import hg38_chain_lengths, lpt_assign
m = 32
lengths = hg38_chain_lengths(200)
if os.environ.get("SHARD"):
    w, r = map(int, os.environ["SHARD"].split(":")); lengths = [lengths[i] for i in lpt_assign(lengths, w)[r]]
what = L.EXPORT_FORWARD | L.EXPORT_SMOOTH | L.EXPORT_RESID
threads = os.environ.get("THREADS", "0") == "1"
for K in (1, 2, 3, 4):
    if K > len(lengths): break
    parts = lpt_assign(lengths, K)
    bs = []
    for r in range(K):
        b = DeviceBatch(0); b.configure(ModelParams(state_dim=2), m, [lengths[i] for i in parts[r]]); b.synthesize(1234 + r); bs.append(b)
    def step_one(b, n):
        for _ in range(n): b.step(L.RETURN_NLL, what)
    def run(n):
        if threads and K > 1:
            ts = [threading.Thread(target=step_one, args=(b, n)) for b in bs]
            for t in ts: t.start()
            for t in ts: t.join()
        else:
            for _ in range(n):
                for b in bs:
                    b.stats(); b.forward_backward(L.RETURN_NLL, False); b.export(what)
                for b in bs:
                    b.sums()
    run(3)
    for b in bs: b.synchronize()
    t = time.perf_counter()
    run(20)
    for b in bs: b.synchronize()
    print("sub-batches", K, "threads" if threads else "one host thread", "ms/step %.3f" % ((time.perf_counter() - t) / 20 * 1e3), flush=True)
    for b in bs: b.close()
