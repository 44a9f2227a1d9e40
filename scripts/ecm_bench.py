"""ECM throughput on the bench workload (whole genome in one batch) + CPU oracle rate on a chr21-sized sample."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, "tests", "golden")): sys.path.insert(0, p)
import numpy as np
from consenrich_amd import _lib as L
from consenrich_amd.batch import DeviceBatch, ModelParams
from consenrich_amd.sharding import hg38_chain_lengths
m = int(os.environ.get("M", "32")); iters = int(os.environ.get("ITERS", "4")); inner = 5
lengths = hg38_chain_lengths(200)
if os.environ.get("SHARD"):      # "8:0" = the contigs rank 0 of 8 would own
    from consenrich_amd.sharding import lpt_assign
    w, r = map(int, os.environ["SHARD"].split(":")); lengths = [lengths[i] for i in lpt_assign(lengths, w)[r]]
b = DeviceBatch(0); b.configure(ModelParams(state_dim=2), m, lengths); b.synthesize(1234); b.stats(); b.synchronize()
prof = os.environ.get("PROFILE", "1") == "1"      # PROFILE=0: wall time without the per-kernel event pairs
for rep in range(3):
    b.profile(prof); t = time.perf_counter()
    outs, paths = b.ecm(max_iters=iters, inner_iters=inner, rtol=0.0, use_lambda=False, use_kappa=True)
    b.synchronize(); dt = time.perf_counter() - t
    kt = b.kernel_times(); b.profile(False)
sweeps = iters * (inner + 1)      # forward passes; backward passes = iters*inner
print(f"GPU ECM: {iters} iters x ({inner} fwd+bwd+E + 1 NLL fwd) over {sum(lengths)} bins x {m}: {dt*1e3:.1f} ms "
      f"-> {dt*1e3/iters:.2f} ms/iter, {sum(lengths)*iters*inner/dt/1e9:.2f} G bin-sweeps/s (fwd+bwd+E-step)")
print({k: (v[0], round(v[1], 2)) for k, v in kt.items()})
rs = b.run_stats()
print("run stats:", {k: rs[k] for k in ("reruns_p", "reruns_x", "reruns_b", "pipeline_redos", "warm_p", "block_len", "local_repairs",
                                     "ws_warm_f", "ws_warm_b")})
if os.environ.get("CPU", "1") == "1":
    import cases
    from oracle import oracle as orc
    n = lengths[20]
    data, munc = cases.synth(n, m, 21)
    t = time.perf_counter()
    orc.cfixedBackgroundECM(matrixData=data, matrixPluginMuncInit=munc, matrixF=np.asarray(cases.F_TREND, np.float32),
        matrixQ0=np.diag([1e-3, 1e-4]).astype(np.float32), intervalToBlockMap=np.zeros(n, np.int32), blockCount=1, stateInit=0.0,
        stateCovarInit=1000.0, ECM_fixedBackgroundIters=2, ECM_fixedBackgroundRtol=0.0, procPrecisionMultiplierMin=5e-3,
        procPrecisionMultiplierMax=5e3, ECM_useObsPrecisionReweighting=False, t_innerIters=inner, logIterations=False)
    dc = time.perf_counter() - t
    print(f"CPU oracle ECM: 2 iters on {n} bins x {m}: {dc:.2f} s -> {n*2*inner/dc/1e6:.2f} M bin-sweeps/s")
