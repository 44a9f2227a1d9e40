"""Control plane of a multi-rank job on ONE node: where the ranks of a job meet, and how a job starts its own ranks.

The reference has no counterpart (one process, chromosomes in a sequential loop, consenrich.py:8809).  The data path of the
sharded fit needs no collective; what the ranks exchange is tiny and happens outside the timed region: rank 0's
128-byte RCCL unique id (`sharding.RcclComm`) and, as a fallback control plane, a few scalars (`JobFiles.allreduce_max`).

JOB DIRECTORY AND TOKEN.  Every file of a job lives in one directory and starts with the job's token:

  * the launcher names them explicitly: CONSENRICH_AMD_RDZV_DIR (a directory the launcher created, mode 0700) and
    CONSENRICH_AMD_JOB_TOKEN (random).  `spawn_ranks` -- what `bench.py --gpus N` uses when it is started plainly -- does
    exactly that, so ranks need not share a parent process;
  * under a launcher that only sets RANK / WORLD_SIZE / MASTER_PORT (`python -m torch.distributed.run`) the token is derived
    from what all ranks of ONE attempt have in common and no other attempt has: MASTER_PORT, the launcher's process id AND
    ITS START TIME (/proc/<ppid>/stat: a recycled pid does not collide), the elastic run id and the restart count -- a
    worker group restarted by the same agent gets a new token, a crashed earlier job on the same port a different one.  The
    directory name is derived from the token, under $XDG_RUNTIME_DIR, /dev/shm or /tmp, created 0700.

READERS NEVER TRUST WHAT THEY FIND: a file is accepted only if it belongs to this user, is a regular file and begins with
this job's token line; anything else (a leftover of a dead job in a re-used directory, a planted file) is ignored and the
reader keeps polling until its deadline.  Writers publish atomically (write to a private name, rename).
"""
from __future__ import annotations

import hashlib
import os
import secrets
import shutil
import socket
import stat
import subprocess
import sys
import tempfile
import time
from typing import List, Optional, Sequence, Tuple

ENV_DIR = "CONSENRICH_AMD_RDZV_DIR"
ENV_TOKEN = "CONSENRICH_AMD_JOB_TOKEN"


def _proc_start_time(pid: int) -> str:
    """Start time of a process in clock ticks since boot (field 22 of /proc/<pid>/stat), '' when unavailable."""
    try:
        with open(f"/proc/{pid}/stat", "rb") as fh:
            raw = fh.read().decode("ascii", "replace")
        return raw[raw.rindex(")") + 2:].split()[19]
    except (OSError, ValueError, IndexError):
        return ""


def _base_dir() -> str:
    for cand in (os.environ.get("XDG_RUNTIME_DIR"), "/dev/shm", tempfile.gettempdir()):
        if cand and os.path.isdir(cand) and os.access(cand, os.W_OK | os.X_OK):
            return cand
    return tempfile.gettempdir()


def job_identity(env=None) -> Tuple[str, str]:
    """(directory, token) of the job this process belongs to (see the module docstring)."""
    env = os.environ if env is None else env
    token = env.get(ENV_TOKEN)
    if not token:
        ppid = os.getppid()
        token = ":".join(["auto", env.get("MASTER_PORT", "0"), str(ppid), _proc_start_time(ppid),
                          env.get("TORCHELASTIC_RUN_ID", ""), env.get("TORCHELASTIC_RESTART_COUNT", "0")])
    directory = env.get(ENV_DIR)
    if not directory:
        digest = hashlib.sha256(token.encode()).hexdigest()[:20]
        directory = os.path.join(_base_dir(), f"consenrich_amd_{os.getuid()}_{digest}")
    return directory, token


class JobFiles:
    """The files of one job.  publish / fetch of named blobs, and a max-all-reduce / barrier built on them."""

    def __init__(self, rank: int, world: int, directory: Optional[str] = None, token: Optional[str] = None):
        d, t = job_identity()
        self.rank, self.world = int(rank), int(world)
        self.dir = directory or d
        self.token = (token or t).encode()
        self._k = 0
        try:
            os.makedirs(self.dir, mode=0o700, exist_ok=True)
        except OSError as exc:
            raise RuntimeError(f"cannot create the job directory {self.dir}: {exc}") from exc
        st = os.lstat(self.dir)
        if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid():
            raise RuntimeError(f"job directory {self.dir} is not a directory owned by this user")
        if st.st_mode & 0o077:
            try:
                os.chmod(self.dir, 0o700)
            except OSError as exc:
                raise RuntimeError(f"job directory {self.dir} is accessible to other users") from exc

    # -- blobs ---------------------------------------------------------------------------------------------------
    def publish(self, name: str, payload: bytes) -> None:
        tmp = os.path.join(self.dir, f".{name}.{os.getpid()}.{secrets.token_hex(4)}.tmp")
        with open(tmp, "wb") as fh:
            fh.write(self.token + b"\n" + payload)
        os.replace(tmp, os.path.join(self.dir, name))        # atomic: a reader sees the whole file or none

    def _read(self, name: str) -> Optional[bytes]:
        path = os.path.join(self.dir, name)
        try:
            fd = os.open(path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
        except OSError:
            return None
        try:
            st = os.fstat(fd)
            if not stat.S_ISREG(st.st_mode) or st.st_uid != os.getuid():
                return None
            with os.fdopen(fd, "rb", closefd=False) as fh:
                raw = fh.read()
        finally:
            os.close(fd)
        head, sep, payload = raw.partition(b"\n")
        if not sep or head != self.token:
            return None             # another job's leftover (or a planted file): not ours
        return payload

    def fetch(self, name: str, timeout_s: float = 300.0, expect_len: Optional[int] = None) -> bytes:
        deadline = time.monotonic() + timeout_s
        while True:
            payload = self._read(name)
            if payload is not None and (expect_len is None or len(payload) == expect_len):
                return payload
            if time.monotonic() > deadline:
                raise TimeoutError(f"rank {self.rank}: nothing valid at {os.path.join(self.dir, name)} after {timeout_s:.0f} s")
            time.sleep(0.002)

    # -- scalars -------------------------------------------------------------------------------------------------
    def allreduce_max(self, value: float, timeout_s: float = 600.0) -> float:
        self._k += 1
        self.publish(f"v{self._k}_r{self.rank}", repr(float(value)).encode())
        vals = [float(self.fetch(f"v{self._k}_r{r}", timeout_s).decode()) for r in range(self.world)]
        return max(vals)

    def allreduce_sum(self, value: float, timeout_s: float = 600.0) -> float:
        self._k += 1
        self.publish(f"s{self._k}_r{self.rank}", repr(float(value)).encode())
        return float(sum(float(self.fetch(f"s{self._k}_r{r}", timeout_s).decode()) for r in range(self.world)))

    def barrier(self) -> None:
        self.allreduce_max(0.0)

    def close(self, remove: bool = True) -> None:
        """Last barrier, then rank 0 removes the directory (unless the launcher owns it: `spawn_ranks` removes its own)."""
        try:
            self.barrier()
        except TimeoutError:
            pass
        if remove and self.rank == 0 and not os.environ.get(ENV_DIR):
            time.sleep(0.1)
            shutil.rmtree(self.dir, ignore_errors=True)


# ---------------------------------------------------------------------------------------------------------------------
# starting the ranks of a job from one plain process
# ---------------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def spawn_ranks(argv: Sequence[str], n: int, extra_env: Optional[dict] = None, grace_s: float = 20.0,
                stdout=None, stderr=None) -> int:
    """Run `argv` as n rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / job directory / job token in
    the environment), relay rank 0's standard output to ours (the other ranks' goes to standard error), wait for all of them and
    return the largest exit status.  When a rank fails the others get `grace_s` seconds, then are terminated (exactly the
    processes started here).  The caller must not have touched the GPU: the children are fresh processes, nothing is exec'ed
    over a GPU-initialised one."""
    if n < 1:
        raise ValueError("need at least one rank")
    stdout = sys.stdout if stdout is None else stdout
    stderr = sys.stderr if stderr is None else stderr
    job_dir = tempfile.mkdtemp(prefix=f"consenrich_amd_{os.getuid()}_job_", dir=_base_dir())
    os.chmod(job_dir, 0o700)
    base = dict(os.environ)
    base.update({"WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()),
                 ENV_DIR: job_dir, ENV_TOKEN: secrets.token_hex(16), "HSA_ENABLE_IPC_MODE_LEGACY": base.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    if extra_env:
        base.update({k: str(v) for k, v in extra_env.items()})
    procs: List[subprocess.Popen] = []
    try:
        for r in range(n):
            env = dict(base, RANK=str(r), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(n), GROUP_RANK="0")
            procs.append(subprocess.Popen(list(argv), env=env, stdout=subprocess.PIPE if r == 0 else stderr, stderr=stderr))
        out0 = procs[0].stdout
        first_fail = None
        pending = set(range(n))
        buf = b""
        os.set_blocking(out0.fileno(), False)
        while pending:
            try:
                chunk = out0.read()
            except (BlockingIOError, ValueError):
                chunk = None
            if chunk:
                buf += chunk
                *lines, buf = buf.split(b"\n")
                for ln in lines:
                    stdout.write(ln.decode("utf-8", "replace") + "\n")
                stdout.flush()
            for r in sorted(pending):
                rc = procs[r].poll()
                if rc is not None:
                    pending.discard(r)
                    if rc != 0 and first_fail is None:
                        first_fail = time.monotonic()
                        print(f"[consenrich_amd.launch] rank {r} exited with status {rc}", file=stderr)
            if first_fail is not None and pending and time.monotonic() - first_fail > grace_s:
                for r in pending:
                    print(f"[consenrich_amd.launch] terminating rank {r} (pid {procs[r].pid}) after a failed rank", file=stderr)
                    procs[r].terminate()
                first_fail = time.monotonic() + 1.0e9       # terminate once
            time.sleep(0.02)
        try:
            rest = out0.read()
        except (BlockingIOError, ValueError):
            rest = None
        buf += rest or b""
        if buf:
            stdout.write(buf.decode("utf-8", "replace"))
            stdout.flush()
        codes = [p.returncode for p in procs]
        return max((abs(c) if c is not None else 1) for c in codes)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(job_dir, ignore_errors=True)
