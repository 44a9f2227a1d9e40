"""The schedule of the barrier-free bit-exact state chain (k_sb_async, csr_device.h) as a CPU simulation
(scripts/ubench/sb_async_sim.c: the bench recipe, the state recursion of the oracle's levelTrend step in float32 carries, one
simulated wavefront per superblock with the kernel's rules -- re-run when the predecessor publishes a new carry, abandon a run
in flight for a newer one, leave early only where the new trajectory meets the stored one BEHIND the last seam an abandoned run
left).  Whatever the schedule, the fixed point must be the sequential recursion; the rule about seams is what makes the early
exit safe (an off-by-one there was found by exactly this simulation)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sim(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("sim") / "sb_async_sim")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "scripts", "ubench", "sb_async_sim.c"), "-lm"])
    return exe


@pytest.mark.parametrize("bins,scale", [(1024, 0.05), (2048, 0.02), (512, 0.01)])
def test_every_schedule_reaches_the_sequential_recursion(sim, bins, scale):
    out = subprocess.run([sim, str(bins), str(scale)], check=True, capture_output=True, text=True, timeout=300).stdout
    lines = [ln for ln in out.splitlines() if "result" in ln]
    # three schedules from the end of the walks + five with the walks on the clock (first superblocks split, walks abandoned)
    assert len(lines) == 8, out
    for ln in lines:
        assert ln.endswith("result == sequential"), ln
    assert "deadlock" not in out
    runs = {ln.split(":")[0]: int(re.search(r"(\d+) superblock runs", ln).group(1)) for ln in lines[:3]}
    aborted = int(re.search(r"(\d+) aborted", lines[2]).group(1))
    assert runs["synchronous passes"] > 0 and runs["asynchronous"] > 0
    if bins <= 1024:
        assert aborted > 0, lines[2]          # runs were abandoned, i.e. the seam rule was exercised
