"""Build libconsenrich_amd.so for gfx950 with hipcc (cross-compiles without a GPU; ~60 s).

The library records what it was built from (`csr_build_id`: a hash of every source file and the compiler flags).  `build()`
rebuilds exactly when that record differs from the sources in the tree -- file times do not enter (a prebuilt library that
travelled to another machine with the tree is kept if, and only if, it IS a build of these sources)."""
from __future__ import annotations

import hashlib
import os
import re
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
# tuning aid (scripts/ only): extra -D switches for compile-time constants of the kernels; they enter the source hash
FLAGS += [f for f in os.environ.get("CONSENRICH_AMD_EXTRA_FLAGS", "").split() if f.startswith("-D")]
FLAG_TAG = "gfx950,O3,fp-contract=off" + "".join("," + f for f in FLAGS if f.startswith("-D"))


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm): consenrich_amd has no CPU fallback and cannot be built without it")


def source_hash() -> str:
    h = hashlib.sha256()
    for d in DEPS:
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:16]


def built_id() -> str:
    """csr_build_id() of the library on disk, read from the file itself ('' if there is none or it predates the record)."""
    if not os.path.exists(OUT):
        return ""
    with open(OUT, "rb") as fh:
        m = re.search(rb"CSR_BUILD_ID:(abi \d+ src [0-9a-f]{16} [ -~]*)\0", fh.read())
    return m.group(1).decode("ascii") if m else ""


def up_to_date() -> bool:
    return f" src {source_hash()} " in built_id() + " "


def build(force: bool = False, verbose: bool = False) -> str:
    want = source_hash()
    if not force and up_to_date():
        if verbose:
            print(f"consenrich_amd: found a build of these sources ({built_id()})", file=sys.stderr)
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = OUT + f".{os.getpid()}.tmp"
    cmd = [hipcc(), *FLAGS, f'-DCSR_SOURCE_HASH="{want}"', f'-DCSR_BUILD_FLAGS="{FLAG_TAG}"', "-o", tmp, SRC]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, OUT)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    if verbose:
        print(f"consenrich_amd: compiled {OUT} (src {want})", file=sys.stderr)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
