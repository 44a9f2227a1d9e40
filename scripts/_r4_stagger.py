"""Experiment: the genome as TWO contexts of one GPU -- the long chromosomes in one, the short ones in the other -- stepped from two
host threads, the second DELAYED so that its head (statistics, covariance chain, record conversion) runs underneath the first
one's state chain ("the head of a step per chain group", the cheap way)."""
import sys, os, time, threading
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from consenrich_amd import _lib as L
from consenrich_amd.batch import DeviceBatch, ModelParams
from consenrich_amd.sharding import hg38_chain_lengths
m = int(os.environ.get("M", "32"))
lengths = hg38_chain_lengths(200)
what = L.EXPORT_FORWARD | L.EXPORT_SMOOTH | L.EXPORT_RESID
order = sorted(range(len(lengths)), key=lambda i: -lengths[i])
total = sum(lengths)
for frac in [float(x) for x in os.environ.get("FRACS", "0.35,0.45,0.55").split(",")]:
    a, acc = [], 0
    for i in order:
        if acc < frac * total: a.append(i); acc += lengths[i]
    bsel = [i for i in order if i not in a]
    bs = []
    for r, sel in enumerate([a, bsel]):
        b = DeviceBatch(0)
        b.configure(ModelParams(state_dim=2), m, [lengths[i] for i in sel])
        b.synthesize(1234 + r)
        bs.append(b)
    for delay_us in [int(x) for x in os.environ.get("DELAYS", "0,300,500,700,900,1200").split(",")]:
        def one(b, n, bar, d):
            for _ in range(n):
                bar.wait()
                if d:
                    t = time.perf_counter() + d * 1e-6
                    while time.perf_counter() < t: pass
                b.step(L.RETURN_NLL, what)
        def run(n):
            bar = threading.Barrier(2)
            th = [threading.Thread(target=one, args=(bs[k], n, bar, delay_us if k == 1 else 0)) for k in range(2)]
            t = time.perf_counter()
            for x in th: x.start()
            for x in th: x.join()
            return (time.perf_counter() - t) / n
        run(3)
        dt = run(20)
        print(f"long share {frac:.2f} ({len(a)} chains, {acc} bins) delay {delay_us} us: step {dt*1e3:.3f} ms ({total/dt/1e9:.2f} G bins/s)", flush=True)
    for b in bs: b.close()
