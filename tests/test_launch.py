"""CPU: the control plane of a multi-rank job (consenrich_amd/launch.py) and `bench.py --gpus N` started plainly: the parent
starts its own rank processes, relays rank 0's line and propagates failures; stale or foreign files in a job directory are
ignored, never read."""
import json
import os
import subprocess
import sys
import threading
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_job_files_ignore_stale_ids_and_stale_barrier_values(tmp_path):
    from consenrich_amd.launch import JobFiles

    d = str(tmp_path / "job")
    os.makedirs(d, mode=0o700)
    # leftovers of an earlier attempt that used the same directory: an RCCL id and barrier values under ANOTHER token, and a
    # file without any token line
    for name, body in (("rccl_id", b"old-token\n" + bytes(128)), ("v1_r1", b"old-token\n123.0"), ("v1_r0", b"7.0")):
        with open(os.path.join(d, name), "wb") as fh:
            fh.write(body)
    a, b = JobFiles(0, 2, directory=d, token="new"), JobFiles(1, 2, directory=d, token="new")
    with pytest.raises(TimeoutError):
        b.fetch("rccl_id", timeout_s=0.3, expect_len=128)                      # the stale id is not taken
    got = {}
    t = threading.Thread(target=lambda: got.setdefault("b", b.allreduce_max(2.0, timeout_s=20.0)))
    t.start()
    time.sleep(0.2)
    assert "b" not in got                                                       # rank 0's stale value 7.0 / 123.0 was not used
    got["a"] = a.allreduce_max(5.0, timeout_s=20.0)
    t.join(20.0)
    assert got == {"a": 5.0, "b": 5.0}
    a.publish("rccl_id", bytes(range(128)))
    assert b.fetch("rccl_id", timeout_s=5.0, expect_len=128) == bytes(range(128))


def test_job_directory_is_private(tmp_path):
    from consenrich_amd.launch import JobFiles

    d = str(tmp_path / "open")
    os.makedirs(d, mode=0o777)
    os.chmod(d, 0o777)
    JobFiles(0, 1, directory=d, token="t")
    assert (os.stat(d).st_mode & 0o077) == 0                                    # tightened, or the constructor raises
    link = str(tmp_path / "link")
    os.symlink(d, link)
    f = JobFiles(0, 1, directory=d, token="t")
    os.symlink("/etc/passwd", os.path.join(d, "rccl_id"))
    assert f._read("rccl_id") is None                                           # symbolic links are not followed


def test_derived_identity_changes_per_attempt(monkeypatch):
    from consenrich_amd import launch

    monkeypatch.delenv(launch.ENV_DIR, raising=False)
    monkeypatch.delenv(launch.ENV_TOKEN, raising=False)
    base = {"MASTER_PORT": "29500", "TORCHELASTIC_RESTART_COUNT": "0"}
    d0, t0 = launch.job_identity(base)
    d1, t1 = launch.job_identity(dict(base, TORCHELASTIC_RESTART_COUNT="1"))
    d2, t2 = launch.job_identity(dict(base, MASTER_PORT="29501"))
    assert len({t0, t1, t2}) == 3 and len({d0, d1, d2}) == 3
    assert str(os.getppid()) in t0 and launch._proc_start_time(os.getppid()) in t0     # pid AND its start time
    dx, tx = launch.job_identity({launch.ENV_DIR: "/x/y", launch.ENV_TOKEN: "abc"})
    assert (dx, tx) == ("/x/y", "abc")


@pytest.mark.timeout(120)
def test_bench_started_plainly_spawns_its_ranks_and_relays_rank_zero():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--fake-ranks"], env=env,
                       capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out == {"fake": True, "n_gpus": 3, "n_ranks_seen": 3, "max_rank": 2}


@pytest.mark.timeout(120)
def test_a_failing_rank_fails_the_job():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["CONSENRICH_AMD_FAKE_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--fake-ranks"], env=env,
                       capture_output=True, text=True, timeout=100)
    assert r.returncode == 5, (r.returncode, r.stderr)
    assert "rank 1 exited with status 5" in r.stderr
    assert json.loads(r.stdout.strip())["n_ranks_seen"] == 2                   # rank 0's line is still relayed


def test_bench_refuses_a_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True,
                       timeout=100)
    assert r.returncode == 2 and "WORLD_SIZE is 2" in r.stderr


def test_both_transports_offer_the_reductions_bench_py_uses():
    """bench.py's barrier, max-over-ranks and per-rank reductions go through `comm`, which is an RcclComm or, on the fallback
    path, the job's JobFiles: both must offer the same three calls (round 4: `per_rank` needs allreduce_sum on both)."""
    from consenrich_amd.launch import JobFiles
    from consenrich_amd.sharding import RcclComm

    for cls in (RcclComm, JobFiles):
        for name in ("barrier", "allreduce_max", "allreduce_sum"):
            assert callable(getattr(cls, name, None)), (cls.__name__, name)


def test_bench_configs_name_the_baseline_workloads():
    """`bench.py --config cN`: one line per BASELINE config (BASELINE.json `configs` 2-5); the default is config 4 and carries
    BASELINE.json's metric string verbatim; c2 is the forward filter alone with B_alg = 8 m + 60; the N > 1 `expected` block is the
    fitted model evaluated on the job's LPT table."""
    import importlib.util
    import json

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    with open(os.path.join(root, "BASELINE.json")) as fh:
        baseline = json.load(fh)
    a = bench.parse_args([])
    assert a.config == "c4" and a.samples == 32 and a.bin_bp == 200 and not a.forward_only
    assert a.metric.replace("x", "\u00d7") == baseline["metric"] or a.metric == baseline["metric"].replace("\u00d7", "x")
    c2 = bench.parse_args(["--config", "c2"])
    assert c2.forward_only and c2.single_chain == 1000000 and c2.samples == 4 and "forward filter only" in c2.metric
    assert bench.b_alg(4, True) == 92 and bench.b_alg(32) == 484 and bench.b_alg(8) == 196 and bench.b_alg(64) == 868
    c3, c5 = bench.parse_args(["--config", "c3"]), bench.parse_args(["--config", "c5"])
    assert (c3.samples, c3.bin_bp, c5.samples, c5.bin_bp) == (8, 200, 64, 50)
    assert bench.parse_args(["--samples", "16"]).metric.endswith("hg38 200bp x 16 samples")
    from consenrich_amd.sharding import hg38_chain_lengths, lpt_assign
    lens = hg38_chain_lengths(200)
    bins = [sum(lens[i] for i in r) for r in lpt_assign(lens, 8)]
    e = bench.expected_speedup(bins, sum(lens))
    assert 1.5 < e["default"]["speedup_vs_1gpu"] < 3.0 < e["ulp2"]["speedup_vs_1gpu"] < 7.67
    assert e["bound"].startswith("LPT makespan bound: 7.67")
