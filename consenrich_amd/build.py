"""Build libconsenrich_amd.so for gfx950 with hipcc (cross-compiles without a GPU; ~10 s)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "csr_lib.hip")
DEPS = [SRC, os.path.join(os.path.dirname(HERE), "include", "consenrich_amd.h"),
        *sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith((".h", ".inl")))]
OUT_DIR = os.path.join(HERE, "lib")
OUT = os.path.join(OUT_DIR, "libconsenrich_amd.so")

# -ffp-contract=off: every fused multiply-add in the kernels is an explicit fma(); see csr_device.h header.
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC",
         "-Wall", "-Wno-unused-function",
         "-Wno-bitwise-instead-of-logical"]     # branch-free '&' of predicates in the chain policies is deliberate


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm): consenrich_amd has no CPU fallback and cannot be built without it")


def up_to_date() -> bool:
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(d) <= t for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and up_to_date():
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = [hipcc(), *FLAGS, "-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
