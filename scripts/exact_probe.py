"""Per-kernel times of a bit-exact step (x_tol_ulps = 0): genome batch and a single chr1-sized chain; superblock
length / window from CONSENRICH_AMD_SB_BINS / CONSENRICH_AMD_SB_WARM."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from consenrich_amd import _lib as L
from consenrich_amd.batch import DeviceBatch, ModelParams
from consenrich_amd.sharding import hg38_chain_lengths
m = int(os.environ.get("M", "32"))
what = L.EXPORT_FORWARD | L.EXPORT_SMOOTH | L.EXPORT_RESID
for name, lengths in (("genome", hg38_chain_lengths(200)), ("chr1", [1244783])):
    with DeviceBatch(0, x_tol_ulps=0) as b:
        b.configure(ModelParams(state_dim=2), m, lengths); b.synthesize(1234)
        for _ in range(2): b.step(L.RETURN_NLL, what)
        b.synchronize(); t = time.perf_counter()
        for _ in range(5): b.step(L.RETURN_NLL, what)
        b.synchronize(); dt = (time.perf_counter() - t) / 5
        r0 = b.run_stats()
        b.profile(True)
        for _ in range(3): b.step(L.RETURN_NLL, what)
        b.synchronize(); kt = b.kernel_times(); b.profile(False); r1 = b.run_stats()
        print(name, "ms/step %.2f" % (dt * 1e3), "reruns_x/step", (r1["reruns_x"] - r0["reruns_x"]) / 3, "fix launches/step",
              (r1["fix_launches"] - r0["fix_launches"]) / 3, {k: (v[0] // 3, round(v[1] / 3, 3)) for k, v in kt.items()}, flush=True)
