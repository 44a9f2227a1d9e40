"""CPU: the C-ABI library loads and exports every symbol the header declares (no compute without a GPU), the product
fails loudly when it cannot run, the host-side mirror raises the reference's errors before any launch, and the
sharding logic is right."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from consenrich_amd import build

    build.build()
    from consenrich_amd import _lib

    _lib.lib()
    return _lib


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "consenrich_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(csr_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(L):
    import ctypes

    handle = ctypes.CDLL(L.LIB_PATH)
    declared = _header_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in include/consenrich_amd.h but not exported"
    assert set(declared) == set(L.SYMBOLS), "ctypes binding and header disagree"
    assert L.lib().csr_abi_version() == 6


def test_struct_layouts_match_header(L):
    import ctypes as C

    assert C.sizeof(L.Model) == 8 + 8 * (4 + 4 + 3 + 4 + 5)
    assert C.sizeof(L.FwdIO) == 8 * 2 + 8 * 5 + 8 + 8 * 4
    assert C.sizeof(L.EcmCfg) == 8 * 4 + 4 * 4
    assert C.sizeof(L.EcmOut) == 8 * 7 + 4 * 4
    assert C.sizeof(L.RunStats) == 8 * 5 + 4 * 6 + 8 + 4 * 2 + 8 + 8 + 8


def _has_gpu(L):
    return L.device_count() > 0


def test_product_fails_loudly_without_gpu(L):
    if _has_gpu(L):
        pytest.skip("a GPU is present")
    from consenrich_amd import cconsenrich
    from consenrich_amd.batch import DeviceBatch

    assert L.lib().csr_create(0) is None and "no HIP device" in L.last_error()
    with pytest.raises(L.ConsenrichAMDError):
        DeviceBatch(0)
    n, m = 16, 2
    with pytest.raises(L.ConsenrichAMDError):
        cconsenrich.cforwardPass(matrixData=np.zeros((m, n), np.float32), matrixPluginMuncInit=np.ones((m, n), np.float32),
                                 matrixF=np.eye(2, dtype=np.float32), matrixQ0=np.eye(2, dtype=np.float32),
                                 intervalToBlockMap=np.zeros(n, np.int32), blockCount=1, stateInit=0.0,
                                 stateCovarInit=1.0)


def test_product_sources_never_touch_the_oracle():
    pkg = os.path.join(ROOT, "consenrich_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f
                assert "libconsenrich_oracle" not in text, f


def _kw(n=12, m=2):
    return dict(matrixData=np.zeros((m, n), np.float32), matrixPluginMuncInit=np.full((m, n), 0.2, np.float32),
                matrixF=np.asarray([[1, 1], [0, 1]], np.float32), matrixQ0=np.diag([1e-3, 1e-4]).astype(np.float32),
                intervalToBlockMap=np.zeros(n, np.int32), blockCount=1, stateInit=0.0, stateCovarInit=1.0)


def test_reference_error_contract_is_enforced_before_any_launch(L):
    """Messages of pyx:6503-6561, 6624, 125-130, 143-151; all raised on the host, so they work without a GPU."""
    from consenrich_amd import cconsenrich as cc

    kw = _kw()
    with pytest.raises(ValueError, match="blockCount must be positive"):
        cc.cforwardPass(**{**kw, "blockCount": 0})
    with pytest.raises(ValueError, match="must match matrixData shape"):
        cc.cforwardPass(**{**kw, "matrixPluginMuncInit": np.ones((2, 5), np.float32)})
    with pytest.raises(ValueError, match="at least shape"):
        cc.cforwardPass(**{**kw, "matrixF": np.ones((1, 2), np.float32)})
    with pytest.raises(ValueError, match="observation precision multiplier bounds"):
        cc.cforwardPass(**kw, obsPrecisionMultiplierMin=0.0)
    with pytest.raises(ValueError, match="process precision multiplier bounds"):
        cc.cforwardPass(**kw, procPrecisionMultiplierMin=2.0, procPrecisionMultiplierMax=1.0)
    with pytest.raises(ValueError, match="intervalToBlockMap length"):
        cc.cforwardPass(**{**kw, "intervalToBlockMap": np.zeros(3, np.int32)})
    with pytest.raises(ValueError, match="out-of-range block id"):
        cc.cforwardPass(**{**kw, "intervalToBlockMap": np.full(12, 4, np.int32)})
    with pytest.raises(ValueError, match="lambdaExp length"):
        cc.cforwardPass(**kw, lambdaExp=np.ones(3, np.float32))
    with pytest.raises(ValueError, match="wrong number of dimensions"):
        cc.cforwardPass(**kw, lambdaExp=np.ones((2, 12), np.float32))
    with pytest.raises(ValueError, match="processQScale\\[0\\] must be 1.0"):
        cc.cforwardPass(**kw, processQScale=np.full(12, 2.0, np.float32))
    with pytest.raises(ValueError, match="positive finite"):
        cc.cforwardPass(**kw, processQScale=np.asarray([1.0] + [-1.0] * 11, np.float32))
    with pytest.raises(ValueError, match="vectorD length"):
        cc.cforwardPass(**kw, vectorD=np.empty(3, np.float32))
    with pytest.raises(ValueError, match="stateForward shape"):
        cc.cforwardPass(**kw, stateForward=np.empty((3, 2), np.float32), stateCovarForward=np.empty((12, 2, 2), np.float32),
                        pNoiseForward=np.empty((12, 2, 2), np.float32))
    with pytest.raises(ValueError, match="dtype mismatch"):
        cc.cforwardPass(**{**kw, "matrixData": np.zeros((2, 12), np.float64)})
    with pytest.raises(ValueError, match="not C-contiguous"):
        cc.cforwardPass(**{**kw, "matrixData": np.zeros((12, 2), np.float32).T})
    # caller-provided outputs are written in place by the device-to-host copy: a read-only buffer is refused up front (the
    # Cython original fails in its writable buffer acquisition; a copy into a read-only mapping would be a segfault)
    ro = np.empty(12, np.float32)
    ro.setflags(write=False)
    with pytest.raises(ValueError, match="read-only"):
        cc.cforwardPass(**kw, vectorD=ro)
    ro2 = np.empty((12, 2), np.float32)
    ro2.setflags(write=False)
    with pytest.raises(ValueError, match="read-only"):
        cc.cbackwardPass(matrixData=kw["matrixData"], matrixF=kw["matrixF"], stateForward=np.zeros((12, 2), np.float32),
                         stateCovarForward=np.zeros((12, 2, 2), np.float32), pNoiseForward=np.zeros((12, 2, 2), np.float32),
                         stateSmoothed=ro2)
    lv = {k: v for k, v in kw.items() if k != "matrixF"}
    with pytest.raises(ValueError, match=r"matrixQ0\[0, 0\] must be positive"):
        cc.cforwardPassLevel(**{**lv, "matrixQ0": np.zeros((1, 1), np.float32)})
    with pytest.raises(ValueError, match="matrixQ0 is singular"):
        cc.cfixedBackgroundECM(**{**kw, "matrixQ0": np.ones((2, 2), np.float32)}, logIterations=False)
    with pytest.raises(ValueError, match="lambdaExpInit length"):
        cc.cfixedBackgroundECM(**kw, lambdaExpInit=np.ones(3, np.float32), logIterations=False)
    with pytest.raises(ValueError, match="only finite values"):
        cc.cfixedBackgroundECM(**kw, processPrecExpInit=np.full(12, np.nan, np.float32), logIterations=False)
    with pytest.raises(ValueError, match=r"stateSmoothed must have shape \(n, 2\)"):
        cc.cExpectedTransitionResidualSums(np.zeros((4, 1)), np.zeros((4, 2, 2)), np.zeros((3, 2, 2)), np.eye(2))
    with pytest.raises(ValueError, match="lagCovSmoothed must have shape"):
        cc.cExpectedTransitionResidualSums(np.zeros((4, 2)), np.zeros((4, 2, 2)), np.zeros((1, 2, 2)), np.eye(2))


def test_empty_and_degenerate_inputs_follow_the_reference(L):
    from consenrich_amd import cconsenrich as cc

    kw = _kw(n=0)
    r = cc.cforwardPass(**kw, returnNLL=True)
    assert r[0] == 0.0 and r[1] == 0 and r[2].shape == (0,) and r[3] == 0.0
    assert cc.cExpectedTransitionResidualSums(np.zeros((1, 2)), np.zeros((1, 2, 2)), np.zeros((0, 2, 2)), np.eye(2)) == (0.0, 0.0, 0)
    assert cc.cExpectedTransitionResidualSumsLevel(np.zeros((1, 1)), np.zeros((1, 1, 1)), np.zeros((0, 1, 1))) == (0.0, 0.0, 0)
    out = cc.cfixedBackgroundECM(**kw, returnIntermediates=True, returnDiagnostics=True, logIterations=False)
    assert out[0] == 0 and out[1] == 0.0 and out[8]["skipped"] is True and out[8]["skip_reason"] == "empty_input"
    xs, Ps, lag, res = cc.cbackwardPass(matrixData=np.empty((2, 0), np.float32), matrixF=kw["matrixF"],
                                        stateForward=np.empty((0, 2), np.float32),
                                        stateCovarForward=np.empty((0, 2, 2), np.float32),
                                        pNoiseForward=np.empty((0, 2, 2), np.float32))
    assert xs.shape == (0, 2) and Ps.shape == (0, 2, 2) and lag.shape == (1, 2, 2) and res.shape == (0, 2)


def test_convergence_replay_matches_reference_bookkeeping():
    from consenrich_amd.cconsenrich import _replay_path

    rec = _replay_path("t", [100.0, 90.0, 89.99999, 89.99998, 89.9], 1e-6, False)
    assert [r["stable_iters"] for r in rec] == [0, 0, 1, 2, 0]
    assert rec[0]["reset_iteration"] and rec[0]["change"] is None
    assert rec[3]["converged"] and not rec[4]["converged"]
    assert rec[1]["relative_improvement"] == pytest.approx(10.0 / 100.0)


def test_lpt_sharding_of_hg38():
    from consenrich_amd.sharding import hg38_chain_lengths, lpt_assign, shard_bound

    lens = hg38_chain_lengths(200)
    assert len(lens) == 22 and sum(lens) == 14375018 and lens[0] == 1244783
    for w in (1, 2, 4, 8):
        owned = lpt_assign(lens, w)
        assert sorted(i for v in owned for i in v) == list(range(22))
    assert shard_bound(lens, 8) == pytest.approx(7.67, abs=0.01)
    assert shard_bound(lens, 4) == pytest.approx(3.89, abs=0.02)
    assert shard_bound(lens, 2) == pytest.approx(2.0, abs=0.01)
    assert sum(hg38_chain_lengths(50)) == 57500042


def test_header_is_plain_c_and_library_links_from_c(tmp_path):
    """The boundary is a C ABI: compile a strict-C99 translation unit against include/consenrich_amd.h, link it with the
    shared library, run it (no GPU needed) and compare the struct sizes it prints with the ctypes mirrors."""
    import shutil
    import subprocess

    from consenrich_amd import _lib as L

    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    L.lib()                                                        # builds the library if necessary
    exe = tmp_path / "abi_check"
    libdir = os.path.dirname(L.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "abi_check.c"), "-o", str(exe), "-L", libdir, "-lconsenrich_amd",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    fields = r.stdout.split()
    assert int(fields[1]) == L.lib().csr_abi_version()
    sizes = [int(v) for v in fields[5:11]]
    assert sizes == [C.sizeof(L.Model), C.sizeof(L.EcmCfg), C.sizeof(L.EcmOut), C.sizeof(L.BgCfg), C.sizeof(L.BgOut),
                     C.sizeof(L.RunStats)]


def test_qseed_posterior_fails_loudly_without_a_gpu():
    """The grid posterior of the Q0 seed (pyx:1905-2146) runs on the device since round 2 (csr_qseed_post.h; its golden
    vectors are checked under -m gpu, tests/test_gpu_qseed.py): without a GPU the call must raise, not compute on the host."""
    import numpy as np
    import pytest

    import qseed_cases as qc
    from conftest import gpu_available
    from consenrich_amd import _lib as L
    from consenrich_amd import qseed

    if gpu_available():
        pytest.skip("a GPU is visible: the device path is covered by tests/test_gpu_qseed.py")
    case = next(c for c in qc.native_cases() if c["kind"] == "post")
    with pytest.raises(L.ConsenrichAMDError, match="no CPU fallback"):
        qc.run_native(qseed, case)


def test_qseed_mirror_argument_validation_without_a_gpu():
    """the Python-level validation of the Q0-seed natives (pyx:1532-1546, 1977-1996) happens before any device call"""
    import math

    import numpy as np
    import pytest

    from consenrich_amd import cconsenrich as amd

    data, act = np.zeros((2, 9)), np.ones((2, 9), bool)
    with pytest.raises(ValueError, match="signalPanelSize must be nonnegative"):
        amd.cEstimateSameTrackProcessNoiseTransitions(data, np.ones((2, 9)), act, 0.95, 20.0, 0, 32000, -1)
    with pytest.raises(ValueError, match=r"precisionCapQuantile must be in \[0, 1\]"):
        amd.cEstimateSameTrackProcessNoiseTransitions(data, np.ones((2, 9)), act, 1.5, 20.0)
    with pytest.raises(ValueError, match="precisionCapMultiplier must be positive"):
        amd.cEstimateSameTrackProcessNoiseTransitions(data, np.ones((2, 9)), act, 0.95, 0.0)
    with pytest.raises(ValueError, match="obsVar shape must match matrixData"):
        amd.cEstimateSameTrackProcessNoiseTransitions(data, np.ones((2, 8)), act, 0.95, 20.0)
    with pytest.raises(ValueError, match="matrixData must be a 2D array"):
        amd.cEstimatePooledProcessNoiseTransitions(np.zeros(9), np.ones(9), np.ones(9, bool))
    empty = amd.cEstimateSameTrackProcessNoiseTransitions(np.zeros((2, 1)), np.ones((2, 1)), np.ones((2, 1), bool), 0.95, 20.0)
    assert empty[0].size == 0 and math.isnan(empty[3]["precisionCap"])
    assert all(a.size == 0 for a in amd.cEstimatePooledProcessNoiseTransitions(np.zeros((2, 1)), np.ones((2, 1)), np.ones((2, 1), bool)))
    one = np.ones(8)
    tail = (1.0e-5, 8, math.log(4.0), 8.0, 64)
    for kwargs, msg in (((one, one[:3], one, 1e-5, 1.0, 8.0, "s") + tail, "same length"),
                        ((one, one, one, 0.0, 1.0, 8.0, "s") + tail, "qFloor must be positive finite"),
                        ((one, one, one, 1e-5, -1.0, 8.0, "s") + tail, "qCap must be positive or infinite"),
                        ((one, one, one, 1e-5, 1.0, 8.0, "s", 1e-5, 0, math.log(4.0), 8.0, 64), "minTransitions must be positive"),
                        ((one, one, one, 1e-5, 1.0, 8.0, "s", 1e-5, 8, math.log(4.0), 8.0, 0), "gridSize must be positive")):
        with pytest.raises(ValueError, match=msg):
            amd.cQSeedPosteriorFromTransitions(*kwargs)
    # (the data-dependent checks -- "samplingVariances must be nonnegative finite", ... -- are made by the device kernels:
    # tests/test_gpu_qseed.py)


def test_bench_fails_loudly_without_a_gpu():
    """No silent CPU path behind the benchmark either: without a device bench.py ends with the library's error."""
    import subprocess
    import sys

    from conftest import gpu_available

    if gpu_available():
        pytest.skip("a GPU is visible")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                        "--no-extras"], capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "no CPU fallback" in r.stderr or "no HIP device" in r.stderr, r.stderr[-1500:]



def test_bench_line_helpers_read_the_committed_records():
    """`throughput_mode.parity` of a bench line comes from the committed full-size parity record of that config (which arrays the
    2-ulp mode holds to 1e-5 on the workload, which it does not); `expected` at N > 1 from the model fitted to the emulated shards."""
    import importlib
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    for cfg in ("c2", "c3", "c4", "c5"):
        rec = bench.throughput_mode_parity(cfg)
        assert rec is not None and rec["source"].startswith("profiles/r0") and rec["source"].endswith("_ulp2.json"), cfg
        assert all(v <= 1e-5 for v in rec["holds_1e-5"].values()) and "level (xs, xf)" in rec["holds_1e-5"], (cfg, rec)
        nis = rec["outside_1e-5"]["NIS"]
        if cfg != "c3":                 # (the c3 test records the two output tracks only)
            assert 0.0 < nis["fraction_of_bins"] < 0.1, (cfg, rec)       # NIS is the array the mode does not hold
    assert bench.throughput_mode_parity("no such config") is None
    from consenrich_amd.sharding import hg38_chain_lengths, lpt_assign

    lengths = hg38_chain_lengths(200)
    bins = [sum(lengths[i] for i in part) for part in lpt_assign(lengths, 8)]
    exp = bench.expected_speedup(bins, sum(lengths))
    assert 1.5 < exp["default"]["speedup_vs_1gpu"] < 3.0 and 4.5 < exp["ulp2"]["speedup_vs_1gpu"] < 7.67, exp
    assert os.path.isfile(os.path.join(root, bench.TRAFFIC_FILES[0])) and os.path.isfile(os.path.join(root, bench.TRAFFIC_FILES[2]))
