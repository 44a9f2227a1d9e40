"""Experiment: the genome as K batches (LPT shards) in K contexts of ONE GPU, stepped concurrently from K host threads -- what
chain-group concurrency at the context level would buy the bit-exact mode.  K=1 is the plain step."""
import sys, os, time, threading
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from consenrich_amd import _lib as L
from consenrich_amd.batch import DeviceBatch, ModelParams
from consenrich_amd.sharding import hg38_chain_lengths, lpt_assign
m = int(os.environ.get("M", "32"))
lengths = hg38_chain_lengths(200)
what = L.EXPORT_FORWARD | L.EXPORT_SMOOTH | L.EXPORT_RESID
for K in [int(k) for k in os.environ.get("KS", "1,2,3,4").split(",")]:
    shards = lpt_assign(lengths, K)
    bs = []
    for r in range(K):
        b = DeviceBatch(0)
        b.configure(ModelParams(state_dim=2), m, [lengths[i] for i in shards[r]])
        b.synthesize(1234 + r)
        bs.append(b)
    def one(b, n, bar):
        for _ in range(n):
            bar.wait()
            b.step(L.RETURN_NLL, what)
    def run(n):
        bar = threading.Barrier(K)
        th = [threading.Thread(target=one, args=(b, n, bar)) for b in bs]
        t = time.perf_counter()
        for x in th: x.start()
        for x in th: x.join()
        return (time.perf_counter() - t) / n
    run(3)
    dt = run(20)
    itr = int(os.environ.get("ECM_ITERS", "3"))
    def ecm(b):
        b.stats(); b.ecm(max_iters=itr, inner_iters=5, rtol=0.0, use_lambda=False, use_kappa=True); b.synchronize()
    th = [threading.Thread(target=ecm, args=(b,)) for b in bs]
    t = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    de = (time.perf_counter() - t) / itr
    print(f"K={K}: step {dt*1e3:.3f} ms ({sum(lengths)/dt/1e9:.2f} G bins/s), ECM iteration {de*1e3:.2f} ms; superblocks per context "
          f"{[bb.run_stats()['blocks'] for bb in bs]}", flush=True)
    for b in bs: b.close()
