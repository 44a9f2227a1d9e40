"""SURVEY 8(f) rank 2b measurement: fold mask + information tracks (cuncertainty.pyx:97-157, 160-305) of a chr1-sized
chromosome x 32 samples -- GPU kernels (HIP events) vs the CPU oracle (= the reference, bit for bit); and the device-side
creation of a fold chain inside a batch (no matrix crosses PCIe)."""
import sys, os, time, json, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests", "golden"))
import numpy as np
import unc_cases
from consenrich_amd import _lib as L, cuncertainty as amd
from consenrich_amd.batch import DeviceBatch, ModelParams
from oracle import oracle as orc

m, n, bl = 32, 1244783, 250
rng = np.random.default_rng(0)
munc = (np.abs(rng.normal(0.3, 0.1, (m, n))) + 0.01).astype(np.float32)
act = np.ones((m, n), np.uint8); lam = np.ones(n)
bf, rc, rb = unc_cases.fold_spec(m, n, bl, 2, 0.5, 3)
lib = L.lib()
def gpu():
    tot = amd.cobservationTotalInformation(munc, act, lam, False, 1e-4, 0.0)
    return amd.cmakeFoldMaskAndInformation(m, n, bl, 0, bf, rc, rb, munc, act, tot, lam, False, 1e-4, 0.0)
gpu()
L.check(lib.csr_profile_enable(None, 1)); t = time.perf_counter(); g = gpu(); wall = time.perf_counter() - t
buf = (L.KernelTime * 64)(); nn = C.c_int32(); L.check(lib.csr_profile_read(None, buf, 64, C.byref(nn)))
kt = {buf[i].name.decode(): buf[i].total_ms for i in range(nn.value) if buf[i].name.decode().startswith("fold")}
L.check(lib.csr_profile_enable(None, 0))
t = time.perf_counter(); tot = orc.cobservationTotalInformation(munc, act, lam, False, 1e-4, 0.0)
o = orc.cmakeFoldMaskAndInformation(m, n, bl, 0, bf, rc, rb, munc, act, tot, lam, False, 1e-4, 0.0); cpu = time.perf_counter() - t
assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(g, o))
b = DeviceBatch(0); b.configure(ModelParams(state_dim=2), m, [n, n]); b.synthesize(1)
b.make_fold(0, 1, bl, 0, bf, rc, rb, pad=1e-4); b.synchronize(); b.profile(True)
t = time.perf_counter(); b.make_fold(0, 1, bl, 0, bf, rc, rb, pad=1e-4); wall_b = time.perf_counter() - t
ktb = b.kernel_times(); b.profile(False)
kms = sum(kt.values())
print(json.dumps({"row": "8(f) rank 2b fold natives", "m": m, "bins": n, "gpu_kernels_ms": {k: round(v, 3) for k, v in kt.items()},
                  "gpu_cells_per_s_kernels": m * n / (kms * 1e-3), "host_api_wall_ms_incl_pcie": round(wall * 1e3, 1),
                  "cpu_oracle_ms": round(cpu * 1e3, 1), "bit_identical": True,
                  "batch_make_fold_kernel_ms": round(ktb.get("fold_make", (0, 0.0))[1], 3), "batch_make_fold_wall_ms": round(wall_b * 1e3, 2)}))
