"""Contig sharding across the GPUs of one node.

Chromosomes/contigs are fully independent fits (reference: sequential loop consenrich.py:8809, no cross-chromosome
state inside runConsenrich), so the path shards with NO data-path collective: every rank fits its own chains with its
own DeviceBatch.  The only communication is the final gather of the per-bin output tracks, done once per job with
torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).
torch is imported lazily and only here / in bench.py: it is plumbing for the process group, not part of the product.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

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


def gather_tracks(local: Dict[int, np.ndarray], lengths: Sequence[int], row_width: int, group=None, device=None):
    """Final track gather: every rank contributes {chain index: float32 array (n_c, row_width)} for the chains it
    owns; rank 0 returns the genome-ordered list of arrays (others return None).

    One padded all_gather of a flat float32 buffer per call (uneven shards are padded to the largest rank's share);
    there is no all-reduce and no exchange inside the estimator.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    owned = lpt_assign(lengths, world)
    share = [sum(int(lengths[i]) for i in v) * row_width for v in owned]
    cap = max(max(share), 1)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    send = torch.zeros(cap, dtype=torch.float32, device=dev)
    pos = 0
    for i in owned[rank]:
        arr = np.ascontiguousarray(local[i], dtype=np.float32).reshape(-1)
        if arr.size != int(lengths[i]) * row_width:
            raise ValueError(f"chain {i}: expected {int(lengths[i]) * row_width} values, got {arr.size}")
        send[pos:pos + arr.size] = torch.from_numpy(arr).to(dev)
        pos += arr.size
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send, group=group)
    if rank != 0:
        return None
    out: List[np.ndarray] = [None] * len(lengths)  # type: ignore[list-item]
    for r in range(world):
        buf = recv[r].cpu().numpy()
        pos = 0
        for i in owned[r]:
            cnt = int(lengths[i]) * row_width
            out[i] = buf[pos:pos + cnt].reshape(int(lengths[i]), row_width).copy()
            pos += cnt
    return out
