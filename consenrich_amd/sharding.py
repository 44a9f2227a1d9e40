"""Contig sharding across the GPUs of one node.

Chromosomes/contigs are fully independent fits (reference: sequential loop consenrich.py:8809, no cross-chromosome
state inside runConsenrich), so the path shards with NO data-path collective: every rank fits its own chains with its
own DeviceBatch.  The only communication is the final gather of the per-bin output tracks, done once per job over RCCL /
xGMI: `RcclComm` binds the library's csr_comm_* entry points (RCCL itself is dlopen'ed by libconsenrich_amd.so) -- the
tracks are packed on the device straight from the exported arrays and all-gathered, no host bounce.  The
rendezvous (rank 0's 128-byte unique id) goes through the job's directory on the node (`consenrich_amd.launch.JobFiles`:
token-stamped files, explicit directory / token from the launcher or derived per launch attempt).
The host-side bookkeeping (ownership, packed layout, re-assembly in genome order) is transport-agnostic (`pack_layout`,
`unpack_gathered`, `gather_tracks`) and is what the world-size-2 CPU test drives over gloo.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import numpy as np

# hg38 autosome lengths (bp), chr1..chr22 (UCSC hg38.chrom.sizes; the reference ships the same table as
# src/consenrich/data/hg38.sizes)
HG38_AUTOSOMES = [
    248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
    135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
    46709983, 50818468,
]


def hg38_chain_lengths(bin_bp: int = 200) -> List[int]:
    """Bins per autosome at the given resolution: ceil(size / bin) (14 375 018 bins in total at 200 bp)."""
    return [-(-s // bin_bp) for s in HG38_AUTOSOMES]


def lpt_assign(lengths: Sequence[int], world_size: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of chains to ranks (cost ~ bins).  Deterministic.

    Returns, for every rank, the list of chain indices it owns (ascending by index).
    """
    if world_size <= 0:
        raise ValueError("world_size must be positive")
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    loads = [0] * world_size
    owned: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda q: (loads[q], q))
        owned[r].append(i)
        loads[r] += int(lengths[i])
    return [sorted(v) for v in owned]


def shard_bound(lengths: Sequence[int], world_size: int) -> float:
    """Upper bound on strong-scaling speed-up from the LPT makespan (7.67x for hg38 autosomes on 8 ranks)."""
    owned = lpt_assign(lengths, world_size)
    makespan = max(sum(int(lengths[i]) for i in v) for v in owned)
    return float(sum(int(v) for v in lengths)) / float(makespan)


def pack_layout(lengths: Sequence[int], world_size: int):
    """(owned, share, cap): chains of every rank, bins of every rank, and the per-rank capacity (bins) of the gather buffers."""
    owned = lpt_assign(lengths, world_size)
    share = [sum(int(lengths[i]) for i in v) for v in owned]
    return owned, share, max(max(share), 1)


def unpack_gathered(buf: np.ndarray, lengths: Sequence[int], world_size: int, row_width: int) -> List[np.ndarray]:
    """Re-assembles genome order from the gathered buffer (world, cap * row_width): rank r's chains follow each other,
    unpadded, from the start of row r."""
    owned, _share, cap = pack_layout(lengths, world_size)
    buf = np.asarray(buf, np.float32).reshape(world_size, cap * row_width)
    out: List[np.ndarray] = [None] * len(lengths)  # type: ignore[list-item]
    for r in range(world_size):
        pos = 0
        for i in owned[r]:
            cnt = int(lengths[i]) * row_width
            out[i] = buf[r, pos:pos + cnt].reshape(int(lengths[i]), row_width).copy()
            pos += cnt
    return out


def gather_tracks(local: Dict[int, np.ndarray], lengths: Sequence[int], row_width: int, transport):
    """Host-array form of the final gather over any transport with `rank`, `world` and
    `all_gather(send: float32 (cap * row_width,)) -> float32 (world, cap * row_width)`: every rank contributes
    {chain index: float32 (n_c, row_width)} for the chains it owns; rank 0 returns the genome-ordered list (others None).
    (The product's multi-GPU path is RcclComm.gather_batch_tracks: same layout, packed on the device.)"""
    world, rank = int(transport.world), int(transport.rank)
    owned, _share, cap = pack_layout(lengths, world)
    send = np.zeros(cap * row_width, np.float32)
    pos = 0
    for i in owned[rank]:
        arr = np.ascontiguousarray(local[i], dtype=np.float32).reshape(-1)
        if arr.size != int(lengths[i]) * row_width:
            raise ValueError(f"chain {i}: expected {int(lengths[i]) * row_width} values, got {arr.size}")
        send[pos:pos + arr.size] = arr
        pos += arr.size
    recv = transport.all_gather(send)
    if rank != 0:
        return None
    return unpack_gathered(recv, lengths, world, row_width)


def exchange_unique_id(rank: int, make_id, files, timeout_s: float = 300.0) -> bytes:
    """Rank 0 creates the 128-byte id and publishes it in the job's directory (`launch.JobFiles`: atomic, token-stamped);
    every other rank polls for a VALID file -- one of this user, carrying this job's token, 128 bytes long.  A leftover of an
    earlier job in a re-used directory is ignored, not read."""
    if rank == 0:
        raw = make_id()
        if len(raw) != 128:
            raise ValueError("the unique id must have 128 bytes")
        files.publish("rccl_id", raw)
        return raw
    return files.fetch("rccl_id", timeout_s, expect_len=128)


class RcclComm:
    """One RCCL communicator per rank, on the device of a DeviceBatch's context.  Single node (xGMI)."""

    def __init__(self, batch, world: Optional[int] = None, rank: Optional[int] = None, timeout_s: float = 300.0, files=None):
        from . import _lib as L
        from .launch import JobFiles

        self._L = L
        self._lib = L.lib()
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self._batch = batch
        self._comm = None

        def make_id() -> bytes:
            uid = C.create_string_buffer(128)
            L.check(self._lib.csr_comm_unique_id(uid))
            return uid.raw

        if self.world == 1:
            raw = make_id()
        else:
            files = files if files is not None else JobFiles(self.rank, self.world)
            raw = exchange_unique_id(self.rank, make_id, files, timeout_s)
        uid = C.create_string_buffer(raw, 128)
        self._comm = self._lib.csr_comm_create(batch._ctx, uid, self.world, self.rank)
        if not self._comm:
            raise L.ConsenrichAMDError(L.last_error())
        batch._attach_comm(self)                    # the batch closes its communicators before its context goes away
        self.barrier()

    def ranks_seen(self) -> int:
        """Sum over ranks of 1.0 through RCCL: the number of ranks the communicator really spans."""
        v = C.c_double(1.0)
        self._L.check(self._lib.csr_comm_allreduce_sum(self._comm, C.byref(v)))
        return int(round(v.value))

    def allreduce_sum(self, value: float) -> float:
        v = C.c_double(float(value))
        self._L.check(self._lib.csr_comm_allreduce_sum(self._comm, C.byref(v)))
        return float(v.value)

    def barrier(self):
        self._L.check(self._lib.csr_comm_barrier(self._comm))

    def allreduce_max(self, value: float) -> float:
        v = C.c_double(float(value))
        self._L.check(self._lib.csr_comm_allreduce_max(self._comm, C.byref(v)))
        return float(v.value)

    def gather_batch_tracks(self, lengths: Sequence[int], to_host: bool = True):
        """Final gather of (smoothed level, its variance) of every chromosome: the batch's exported xs / Ps are packed on the
        device and all-gathered over RCCL; with to_host the gathered buffer is copied to the host and rank 0 returns the
        genome-ordered list of float32 (n_c, 2) arrays (other ranks, or to_host=False: None).  `lengths`: bins of ALL
        chromosomes of the genome, in genome order (this rank's batch holds the ones lpt_assign gives it, ascending)."""
        owned, share, cap = pack_layout(lengths, self.world)
        if [int(lengths[i]) for i in owned[self.rank]] != [int(v) for v in self._batch.chain_lens]:
            raise ValueError("the batch does not hold the chains lpt_assign gives this rank")
        host = np.empty((self.world, cap * 2), np.float32) if to_host else None
        self._L.check(self._lib.csr_batch_gather_tracks(self._batch._ctx, self._comm, int(cap), self._L.fp(host)))
        if not to_host or self.rank != 0:
            return None
        return unpack_gathered(host, lengths, self.world, 2)

    def close(self):
        if getattr(self, "_comm", None):
            self._lib.csr_comm_destroy(self._comm)
            self._comm = None
            self._batch._detach_comm(self)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
