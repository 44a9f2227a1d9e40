"""bedGraph track writer on the GPU.

Byte-for-byte the text the reference emits with ``df.to_csv(path, sep="\\t", header=False, index=False,
float_format="%.4f", lineterminator="\\n")`` (/root/reference/src/consenrich/consenrich.py:9797-9805; ~0.7 M rows/s in
pandas, i.e. ~20 s per genome-wide track at 200 bp), produced by three HIP kernels through ``csr_format_bedgraph``.
A maintainer replaces the ``to_csv`` call by::

    from consenrich_amd.writers import append_bedgraph
    append_bedgraph(bedgraphPath, chromosome, df["Start"].to_numpy(), df["End"].to_numpy(), df[col].to_numpy(),
                    mode="w" if c_ == 0 else "a")

No CPU fallback: raises ConsenrichAMDError without the library or a GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

_TRANSFORMS = {None: 0, "none": 0, "round4": 1, "sqrt": 2}


def bedgraph_bytes(chrom: str, starts, ends, values, transform=None) -> bytes:
    """Text of one track.  starts / ends: int arrays (or both None with start0/step given through
    `bedgraph_bytes_regular`); values: float32; transform: None, "round4" (core.getPrimaryState) or "sqrt"."""
    v = np.ascontiguousarray(values, dtype=np.float32).reshape(-1)
    s = np.ascontiguousarray(starts, dtype=np.int64).reshape(-1)
    e = np.ascontiguousarray(ends, dtype=np.int64).reshape(-1)
    if s.shape != v.shape or e.shape != v.shape:
        raise ValueError("starts, ends and values must have the same length")
    return _run(chrom, v.shape[0], s, e, 0, 0, 0, v, transform)


def bedgraph_bytes_regular(chrom: str, start0: int, step: int, values, end_cap: int = 0, transform=None) -> bytes:
    """Regular intervals: start = start0 + k*step, end = min(start + step, end_cap) (end_cap <= 0: no cap)."""
    v = np.ascontiguousarray(values, dtype=np.float32).reshape(-1)
    return _run(chrom, v.shape[0], None, None, int(start0), int(step), int(end_cap), v, transform)


def append_bedgraph(path, chrom, starts, ends, values, mode="a", transform=None) -> int:
    data = bedgraph_bytes(chrom, starts, ends, values, transform)
    with open(path, mode + "b") as fh:
        fh.write(data)
    return len(data)


def _run(chrom, n, s, e, start0, step, end_cap, v, transform):
    if transform not in _TRANSFORMS:
        raise ValueError("transform must be None, 'round4' or 'sqrt'")
    name = str(chrom).encode("ascii")
    if not 1 <= len(name) <= 63:
        raise ValueError("chromosome name must have 1..63 ASCII characters")
    if n == 0:
        return b""
    L.require_gpu()
    f = L.lib().csr_format_bedgraph
    sp = None if s is None else s.ctypes.data_as(L.I64P)
    ep = None if e is None else e.ctypes.data_as(L.I64P)
    t = _TRANSFORMS[transform]
    size = f(name, n, sp, ep, start0, step, end_cap, L.fp(v), t, None, 0)
    if size < 0:
        raise L.ConsenrichAMDError(L.last_error())
    buf = C.create_string_buffer(int(size))
    got = f(name, n, sp, ep, start0, step, end_cap, L.fp(v), t, buf, int(size))
    if got != size:
        raise L.ConsenrichAMDError(L.last_error() or "bedGraph writer size mismatch")
    return buf.raw
