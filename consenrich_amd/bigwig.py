"""bigWig files of Consenrich tracks (second half).

The reference converts the bedGraph files it has written with pyBigWig (/root/reference/src/consenrich/io.py:530-631
`convertBedGraphToBigWig`, io.py:633-790 `_convertBedGraphToBigWigPyBigWig`: header from the chromosome sizes, then
`addEntries(chroms, starts, ends=..., values=...)` in chunks -- i.e. bedGraph-type sections whose values are the float32 of
the "%.4f" text).  pyBigWig / libBigWig are not in this image and are not needed: the format (Kent et al. 2010, "BigWig and
BigBed: enabling browsing of large distributed datasets", file-format supplement) is written directly.

  * The fixed-record BODY -- data sections, total summary, zoom records -- is byte work on the track and comes from the
    device (`DeviceBatch.bigwig_track`, kernels in csrc/csr_writers.h) or, for host arrays / an existing bedGraph file
    (`convert_bedgraph_to_bigwig`, the reference's entry point with its validation and messages), from NumPy records.
  * The ASSEMBLY here is O(sections): chromosome B+ tree (multi-level beyond 256 names), optional zlib of every block, R-tree
    index (cirTree), headers.

File layout written (all little endian): header 64 B | zoom headers 24 B each | total summary 40 B | chromosome tree |
data: section count (u64) + blocks | data index (R-tree, 256 slots per node, one item per block) | per zoom level: record
count (u32) + blocks of <= 512 records + index.

Pinning: no reference-produced bigWig exists here; the writer is checked by an INDEPENDENT reader written from the format
description (oracle/bigwig_reader.py): intervals read back == the rows of the byte-exact bedGraph text parsed as float32,
summary fields == their definitions, plus the expectations of the reference's own test
(tests/test_config.py:3196-3245: intervals and nBasesCovered / minVal / maxVal / sumData / sumSquared of a toy track).
"""
from __future__ import annotations

import os
import struct
import tempfile
import zlib
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

BIGWIG_MAGIC = 0x888FFC26
CHROM_TREE_MAGIC = 0x78CA8C91
RTREE_MAGIC = 0x2468ACE0
ITEMS_PER_SECTION = 1024            # bedGraph items per data block (bedGraphToBigWig's default itemsPerSlot)
ZOOM_RECORDS_PER_BLOCK = 512
RTREE_BLOCK = 256
SECTION_DTYPE = np.dtype([("start", "<u4"), ("end", "<u4"), ("value", "<f4")])
ZOOM_DTYPE = np.dtype([("chrom", "<u4"), ("start", "<u4"), ("end", "<u4"), ("valid", "<u4"), ("min", "<f4"),
                       ("max", "<f4"), ("sum", "<f4"), ("sumsq", "<f4")])


@dataclass
class TrackPiece:
    """The body of one chromosome's track: uncompressed sections back to back + what the index and the summaries need."""
    chrom_id: int
    sections: bytes                         # 24-byte header + 12-byte items, ITEMS_PER_SECTION items per section
    n_items: int
    bases_covered: int
    min_val: float
    max_val: float
    sum_data: float
    sum_squares: float
    zooms: Dict[int, bytes]                 # bases per zoom record -> the 32-byte records of this chromosome


def text4_values(values: np.ndarray) -> np.ndarray:
    """float32 of the decimal text "%.4f" % v: what pyBigWig receives from the reference's bedGraph rows (io.py:707)."""
    v = np.asarray(values, np.float32).astype(np.float64)
    r = np.rint(np.abs(v) * 10000.0) / 10000.0          # |v| * 1e4 is exact in double for float32 v
    return np.where(np.signbit(v), -r, r).astype(np.float32)


def piece_from_arrays(chrom_id: int, starts, ends, values, zoom_bases: Sequence[int] = (), step: Optional[int] = None,
                      apply_text4: bool = True) -> TrackPiece:
    """NumPy statement of what the device kernels produce (same records), for host arrays / parsed bedGraph files."""
    s = np.ascontiguousarray(starts, np.int64)
    e = np.ascontiguousarray(ends, np.int64)
    v = text4_values(values) if apply_text4 else np.ascontiguousarray(values, np.float32)
    n = int(v.shape[0])
    nsec = (n + ITEMS_PER_SECTION - 1) // ITEMS_PER_SECTION
    out = bytearray()
    items = np.empty(n, SECTION_DTYPE)
    items["start"], items["end"], items["value"] = s, e, v
    for j in range(nsec):
        a, b = j * ITEMS_PER_SECTION, min(n, (j + 1) * ITEMS_PER_SECTION)
        out += struct.pack("<IIIIIBBH", chrom_id, int(s[a]), int(e[b - 1]), 0, 0, 1, 0, b - a)
        out += items[a:b].tobytes()
    w = (e - s).astype(np.float64)
    dv = v.astype(np.float64)
    zooms = {}
    for zb in zoom_bases:
        if step is None or zb % step:
            raise ValueError("zoom levels of a host track need a fixed step that divides them")
        g = zb // step
        nrec = (n + g - 1) // g
        rec = np.zeros(nrec, ZOOM_DTYPE)
        idx = np.arange(nrec) * g
        rec["chrom"] = chrom_id
        rec["start"] = s[idx]
        rec["end"] = e[np.minimum(idx + g, n) - 1]
        rec["valid"] = np.add.reduceat(e - s, idx)
        rec["min"] = np.minimum.reduceat(dv, idx).astype(np.float32)
        rec["max"] = np.maximum.reduceat(dv, idx).astype(np.float32)
        rec["sum"] = np.add.reduceat(dv * w, idx).astype(np.float32)
        rec["sumsq"] = np.add.reduceat(dv * dv * w, idx).astype(np.float32)
        zooms[int(zb)] = rec.tobytes()
    return TrackPiece(chrom_id, bytes(out), n, int(w.sum()), float(dv.min()) if n else 0.0, float(dv.max()) if n else 0.0,
                      float((dv * w).sum()), float((dv * dv * w).sum()), zooms)


CHROM_TREE_BLOCK = 256


def _chrom_tree(chrom_sizes: Sequence[Tuple[str, int]]) -> bytes:
    """B+ tree of (name -> chromId, size), keys sorted bytewise.  Up to CHROM_TREE_BLOCK names: one leaf node whose block size
    is their count.  More (assemblies with thousands of scaffolds; a node's count is a 16-bit field): levels of
    CHROM_TREE_BLOCK-way nodes written root first, every node padded to the block size so that child offsets are closed-form
    relative to the start of the tree -- the offsets stored in internal nodes are FILE offsets, so the caller passes where the
    tree starts (`_chrom_tree.at`)."""
    return _chrom_tree_at(chrom_sizes, 0)


def _chrom_tree_at(chrom_sizes: Sequence[Tuple[str, int]], file_offset: int) -> bytes:
    n = len(chrom_sizes)
    if n >= (1 << 32):
        raise ValueError("too many chromosomes for a bigWig chromosome tree")
    names = [name.encode("ascii") for name, _ in chrom_sizes]
    key = max(len(b) for b in names)
    order = sorted(range(n), key=lambda i: names[i])
    leaf_item = lambda i: names[i].ljust(key, b"\0") + struct.pack("<II", i, int(chrom_sizes[i][1]))      # noqa: E731
    if n <= CHROM_TREE_BLOCK:
        out = struct.pack("<IIIIQQ", CHROM_TREE_MAGIC, max(n, 1), key, 8, n, 0)
        out += struct.pack("<BBH", 1, 0, n)
        return out + b"".join(leaf_item(i) for i in order)
    blk = CHROM_TREE_BLOCK
    # levels bottom-up: level 0 = the sorted items, level k = first keys of the nodes of level k - 1
    counts = [n]
    while counts[-1] > blk:
        counts.append((counts[-1] + blk - 1) // blk)
    depth = len(counts)                         # node levels: depth (leaves at level 0 of `counts`, root has counts[-1] items)
    leaf_bytes, inner_bytes = 4 + blk * (key + 8), 4 + blk * (key + 8)
    header = struct.pack("<IIIIQQ", CHROM_TREE_MAGIC, blk, key, 8, n, 0)
    # file offset of every level's first node, root first
    level_off, pos = [], file_offset + len(header)
    for lv in range(depth - 1, -1, -1):
        level_off.append(pos)
        nodes = (counts[lv] + blk - 1) // blk
        pos += nodes * (leaf_bytes if lv == 0 else inner_bytes)
    first_key = [[names[i] for i in order]]     # first_key[lv][j]: smallest key below item j of level lv
    for lv in range(1, depth):
        prev = first_key[-1]
        first_key.append([prev[j * blk] for j in range(counts[lv])])
    out = bytearray(header)
    for d, lv in enumerate(range(depth - 1, -1, -1)):
        nodes = (counts[lv] + blk - 1) // blk
        for node in range(nodes):
            lo, hi = node * blk, min(counts[lv], (node + 1) * blk)
            out += struct.pack("<BBH", 1 if lv == 0 else 0, 0, hi - lo)
            if lv == 0:
                out += b"".join(leaf_item(order[j]) for j in range(lo, hi))
            else:
                child_bytes = leaf_bytes if lv == 1 else inner_bytes
                for j in range(lo, hi):
                    out += first_key[lv][j].ljust(key, b"\0") + struct.pack("<Q", level_off[d + 1] + j * child_bytes)
            out += b"\0" * ((blk - (hi - lo)) * (key + 8))
    return bytes(out)


def _rtree(bounds: np.ndarray, offsets: np.ndarray, sizes: np.ndarray, index_offset: int, end_file_offset: int) -> bytes:
    """cirTree over blocks: bounds (n, 4) = (startChrom, startBase, endChrom, endBase) per block, one item per slot; levels
    written root first, every node padded to RTREE_BLOCK slots (so the node offsets are closed-form)."""
    n = int(bounds.shape[0])
    hdr = struct.pack("<IIQIIIIQII", RTREE_MAGIC, RTREE_BLOCK, n, int(bounds[0, 0]) if n else 0,
                      int(bounds[0, 1]) if n else 0, int(bounds[-1, 2]) if n else 0,
                      int(bounds[:, 3][bounds[:, 2] == bounds[-1, 2]].max()) if n else 0, end_file_offset, 1, 0)
    if n == 0:
        return hdr + struct.pack("<BBH", 1, 0, 0)
    # levels bottom-up: level 0 = leaves over the blocks
    levels = [(bounds, None)]
    cur = bounds
    while cur.shape[0] > RTREE_BLOCK:
        k = (cur.shape[0] + RTREE_BLOCK - 1) // RTREE_BLOCK
        up = np.zeros((k, 4), np.int64)
        for i in range(k):
            part = cur[i * RTREE_BLOCK:(i + 1) * RTREE_BLOCK]
            up[i, 0], up[i, 1] = part[0, 0], part[0, 1]
            up[i, 2] = part[-1, 2]
            up[i, 3] = part[:, 3][part[:, 2] == part[-1, 2]].max()
        levels.append((up, None))
        cur = up
    # node counts per level (top-down), offsets of the levels in the file
    depth = len(levels)
    node_counts = []
    for lv in range(depth - 1, -1, -1):
        items = levels[lv][0].shape[0]
        node_counts.append((items + RTREE_BLOCK - 1) // RTREE_BLOCK)
    leaf_node_bytes = 4 + 32 * RTREE_BLOCK
    inner_node_bytes = 4 + 24 * RTREE_BLOCK
    level_offset = []
    pos = index_offset + len(hdr)
    for d, cnt in enumerate(node_counts):
        level_offset.append(pos)
        pos += cnt * (leaf_node_bytes if d == depth - 1 else inner_node_bytes)
    out = bytearray(hdr)
    for d in range(depth):
        lv = depth - 1 - d
        b = levels[lv][0]
        leaf = d == depth - 1
        for node in range(node_counts[d]):
            part = b[node * RTREE_BLOCK:(node + 1) * RTREE_BLOCK]
            out += struct.pack("<BBH", 1 if leaf else 0, 0, part.shape[0])
            for j in range(part.shape[0]):
                item = node * RTREE_BLOCK + j
                if leaf:
                    out += struct.pack("<IIIIQQ", *(int(x) for x in part[j]), int(offsets[item]), int(sizes[item]))
                else:
                    child = level_offset[d + 1] + item * (leaf_node_bytes if d + 1 == depth - 1 else inner_node_bytes)
                    out += struct.pack("<IIIIQ", *(int(x) for x in part[j]), child)
            out += b"\0" * ((RTREE_BLOCK - part.shape[0]) * (32 if leaf else 24))
    return bytes(out)


def _compress_blocks(blocks: List[bytes], compress: bool, threads: int) -> List[bytes]:
    if not compress:
        return blocks
    with ThreadPoolExecutor(max_workers=max(1, threads)) as pool:      # zlib releases the GIL
        return list(pool.map(lambda b: zlib.compress(b, 6), blocks))


def write_bigwig(path: str, chrom_sizes: Sequence[Tuple[str, int]], pieces: Sequence[TrackPiece], compress: bool = True,
                 threads: int = 8) -> None:
    """Assemble the file from per-chromosome bodies (ascending chrom_id = the order of `chrom_sizes`, like the reference's
    sorted bedGraph).  Written to a temporary file in the target directory and renamed (io.py:659-667, 776-790)."""
    pieces = sorted((p for p in pieces if p.n_items > 0), key=lambda p: p.chrom_id)
    if not pieces:
        raise ValueError("No bedGraph intervals found")          # io.py:766-767
    sec_bytes = 24 + 12 * ITEMS_PER_SECTION
    blocks, bounds = [], []
    for p in pieces:
        nsec = (p.n_items + ITEMS_PER_SECTION - 1) // ITEMS_PER_SECTION
        for j in range(nsec):
            blk = p.sections[j * sec_bytes:(j + 1) * sec_bytes]
            cid, cs, ce = struct.unpack_from("<III", blk, 0)
            blocks.append(blk)
            bounds.append((cid, cs, cid, ce))
    zoom_levels = sorted(set().union(*(set(p.zooms) for p in pieces)))
    uncompress_buf = max([len(b) for b in blocks] + [32 * ZOOM_RECORDS_PER_BLOCK]) if compress else 0
    cblocks = _compress_blocks(blocks, compress, threads)

    n_z = len(zoom_levels)
    header_len = 64 + 24 * n_z
    summary_off = header_len
    chrom_tree_off = summary_off + 40
    chrom_tree = _chrom_tree_at(chrom_sizes, chrom_tree_off)
    data_off = chrom_tree_off + len(chrom_tree)
    pos = data_off + 8
    offsets, sizes = [], []
    for b in cblocks:
        offsets.append(pos)
        sizes.append(len(b))
        pos += len(b)
    index_off = pos
    rtree = _rtree(np.asarray(bounds, np.int64), np.asarray(offsets, np.int64), np.asarray(sizes, np.int64), index_off, index_off)
    pos = index_off + len(rtree)
    zoom_parts = []
    for zb in zoom_levels:
        recs = np.concatenate([np.frombuffer(p.zooms[zb], ZOOM_DTYPE) for p in pieces if zb in p.zooms])
        zblocks, zbounds = [], []
        start = 0
        while start < recs.shape[0]:
            # a block never spans two chromosomes
            stop = min(recs.shape[0], start + ZOOM_RECORDS_PER_BLOCK)
            same = np.nonzero(recs["chrom"][start:stop] != recs["chrom"][start])[0]
            if same.size:
                stop = start + int(same[0])
            part = recs[start:stop]
            zblocks.append(part.tobytes())
            zbounds.append((int(part["chrom"][0]), int(part["start"][0]), int(part["chrom"][-1]), int(part["end"][-1])))
            start = stop
        zc = _compress_blocks(zblocks, compress, threads)
        z_data_off = pos
        zpos = pos + 4
        zoff, zsz = [], []
        for b in zc:
            zoff.append(zpos)
            zsz.append(len(b))
            zpos += len(b)
        z_index_off = zpos
        ztree = _rtree(np.asarray(zbounds, np.int64), np.asarray(zoff, np.int64), np.asarray(zsz, np.int64), z_index_off,
                       z_index_off)
        zoom_parts.append((zb, z_data_off, z_index_off, int(recs.shape[0]), zc, ztree))
        pos = z_index_off + len(ztree)

    out_dir = os.path.dirname(os.path.abspath(path)) or "."
    fd, tmp = tempfile.mkstemp(prefix="consenrich_bigwig_", suffix=".bw", dir=out_dir)
    try:
        with os.fdopen(fd, "wb") as fh:
            fh.write(struct.pack("<IHHQQQHHQQIQ", BIGWIG_MAGIC, 4, n_z, chrom_tree_off, data_off, index_off, 0, 0, 0,
                                 summary_off, uncompress_buf, 0))
            for zb, zd, zi, _cnt, _zc, _zt in zoom_parts:
                fh.write(struct.pack("<IIQQ", zb, 0, zd, zi))
            fh.write(struct.pack("<Qdddd", sum(p.bases_covered for p in pieces), min(p.min_val for p in pieces),
                                 max(p.max_val for p in pieces), sum(p.sum_data for p in pieces),
                                 sum(p.sum_squares for p in pieces)))
            fh.write(chrom_tree)
            fh.write(struct.pack("<Q", len(cblocks)))
            for b in cblocks:
                fh.write(b)
            fh.write(rtree)
            for _zb, _zd, _zi, cnt, zc, ztree in zoom_parts:
                fh.write(struct.pack("<I", cnt))
                for b in zc:
                    fh.write(b)
                fh.write(ztree)
            fh.write(struct.pack("<I", BIGWIG_MAGIC))        # trailing magic (bbi files end with it)
        os.replace(tmp, path)
    except Exception:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise


def zoom_plan(step: int, n_bins_max: int, max_levels: int = 10) -> List[int]:
    """Bases per zoom record: 10 intervals per record at the first level, x4 per level (the reduction schedule of UCSC's
    bedGraphToBigWig), while a level still has more than one block of records."""
    out, g = [], 10
    while len(out) < max_levels and n_bins_max // g > ZOOM_RECORDS_PER_BLOCK:
        out.append(g * int(step))
        g *= 4
    return out


def read_chrom_sizes(path: str) -> List[Tuple[str, int]]:
    """Two-column chromosome sizes file -> [(name, size)] in file order.  Error contract of the reference's reader
    (io.py:601-630: the FIRST offending row decides, in the order malformed row / invalid size / non-positive size / duplicate).
    All rows are split at once; the checks are array masks over the rows, and only the first offending row is looked at again
    to word its message."""
    with open(path, "r", encoding="utf-8") as handle:
        raw = handle.read().split("\n")
    fields = [ln.split() for ln in raw]
    keep = [i for i, f in enumerate(fields) if f and not f[0].startswith("#")]
    if not keep:
        raise ValueError(f"No chromosome sizes found in {path}")
    row_no = np.asarray(keep, np.int64) + 1
    short = np.fromiter((len(fields[i]) < 2 for i in keep), bool, len(keep))
    names = [fields[i][0] for i in keep]
    size = np.zeros(len(keep), np.int64)
    unparsed = np.zeros(len(keep), bool)
    for k, i in enumerate(keep):
        if not short[k]:
            try:
                size[k] = int(fields[i][1])
            except ValueError:
                unparsed[k] = True
    nonpos = ~short & ~unparsed & (size <= 0)
    _first, first_idx = np.unique(np.asarray(names, object), return_index=True)
    dup = np.ones(len(keep), bool)
    dup[first_idx] = False
    bad = short | unparsed | nonpos | dup
    if bad.any():
        k = int(np.argmax(bad))
        if short[k]:
            raise ValueError(f"Malformed chromosome sizes row {int(row_no[k])} in {path}")
        if unparsed[k]:
            raise ValueError(f"Invalid chromosome size on row {int(row_no[k])} in {path}")
        if nonpos[k]:
            raise ValueError(f"Chromosome {names[k]} has non-positive size on row {int(row_no[k])}")
        raise ValueError(f"Duplicate chromosome {names[k]} in {path}")
    return [(names[k], int(size[k])) for k in range(len(keep))]


_SKIP_WORDS = ("track", "browser")


def _uniform_step(starts: np.ndarray, ends: np.ndarray) -> Optional[int]:
    """The fixed interval width of a Consenrich track (every interval `step` wide and adjacent, the last possibly clipped at the
    chromosome end), or None for any other bedGraph."""
    n = starts.shape[0]
    if n == 0:
        return None
    step = int(ends[0] - starts[0])
    if n == 1:
        return step
    if step <= 0 or np.any(np.diff(starts) != step) or np.any((ends - starts)[:-1] != step) or (ends[-1] - starts[-1]) > step:
        return None
    return step


class _BedGraphColumns:
    """Rows of a bedGraph file as columns, a chunk of lines at a time, validated with array masks.  The reference's contract
    (io.py:693-752) is that the FIRST offending row raises, with the first failing check of that row deciding the message; the
    masks find that row, `_word` words its error."""

    def __init__(self, bedgraph_path: str, sizes_label: str, sizes: Sequence[Tuple[str, int]]):
        self.path, self.label = bedgraph_path, sizes_label
        self.rank = {c: i for i, (c, _s) in enumerate(sizes)}
        self.size = np.asarray([s for _c, s in sizes], np.int64)
        self.names = [c for c, _s in sizes]
        self.prev: Optional[Tuple[int, int, int]] = None       # (rank, start, end) of the last accepted row

    def _word(self, line_number: int, parts: List[str]):
        """Raise the reference's error for one offending row (its checks in its order)."""
        if len(parts) != 4:
            raise ValueError(f"Malformed bedGraph row {line_number} in {self.path}: expected 4 columns")
        chrom = parts[0]
        if chrom not in self.rank:
            raise ValueError(f"Chromosome {chrom} on bedGraph row {line_number} is not present in {self.label}")
        try:
            start, end = int(parts[1]), int(parts[2])
        except ValueError as exc:
            raise ValueError(f"Invalid bedGraph coordinates on row {line_number} in {self.path}") from exc
        try:
            value = float(parts[3])
        except ValueError as exc:
            raise ValueError(f"Invalid bedGraph value on row {line_number} in {self.path}") from exc
        if not np.isfinite(value):
            raise ValueError(f"Non-finite bedGraph value on row {line_number} in {self.path}")
        if start < 0:
            raise ValueError(f"Negative start coordinate on bedGraph row {line_number}")
        if end <= start:
            raise ValueError(f"End coordinate must be greater than start on bedGraph row {line_number}")
        size = int(self.size[self.rank[chrom]])
        if end > size:
            raise ValueError(f"End coordinate {end} on bedGraph row {line_number} exceeds {chrom} size of {size}")
        return self.rank[chrom], start, end

    def chunk(self, lines: List[str], first_line_number: int):
        """(rank, start, end, value) arrays of the data rows of `lines`; raises the reference's error at the first bad row."""
        stripped = [ln.strip() for ln in lines]
        data_idx = [i for i, t in enumerate(stripped)
                    if t and t[0] != "#" and not any(t == w or t.startswith(w + " ") for w in _SKIP_WORDS)]
        if not data_idx:
            return None
        parts = [stripped[i].split() for i in data_idx]
        n = len(parts)
        line_no = np.asarray(data_idx, np.int64) + first_line_number
        bad = np.fromiter((len(p) != 4 for p in parts), bool, n)
        rank = np.fromiter((self.rank.get(p[0], -1) if len(p) == 4 else -1 for p in parts), np.int64, n)
        bad |= rank < 0
        start, end, value = np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros(n, np.float64)
        ok_idx = np.nonzero(~bad)[0]
        try:        # whole columns at once; a token NumPy cannot parse sends the chunk to the per-token path below
            start[ok_idx] = np.asarray([parts[k][1] for k in ok_idx], dtype="U").astype(np.int64)
            end[ok_idx] = np.asarray([parts[k][2] for k in ok_idx], dtype="U").astype(np.int64)
            value[ok_idx] = np.asarray([parts[k][3] for k in ok_idx], dtype="U").astype(np.float64)
        except (ValueError, OverflowError):
            for k in ok_idx:
                try:
                    start[k], end[k], value[k] = int(parts[k][1]), int(parts[k][2]), float(parts[k][3])
                except (ValueError, OverflowError):
                    bad[k] = True
        good = ~bad
        bad |= good & (~np.isfinite(value) | (start < 0) | (end <= start) | (end > self.size[np.maximum(rank, 0)]))
        # order within the file: rank non-decreasing, starts non-decreasing within a chromosome, no overlap
        pr, ps, pe = (np.int64(-1), np.int64(-1), np.int64(-1)) if self.prev is None else self.prev
        r0 = np.concatenate(([pr], rank[:-1]))
        s0 = np.concatenate(([ps], start[:-1]))
        e0 = np.concatenate(([pe], end[:-1]))
        has_prev = np.ones(n, bool)
        has_prev[0] = self.prev is not None
        unsorted = has_prev & ((rank < r0) | ((rank == r0) & (start < s0)))
        overlap = has_prev & (rank == r0) & (start < e0)
        first_bad = int(np.argmax(bad)) if bad.any() else n
        order_bad = unsorted | overlap
        first_order = int(np.argmax(order_bad)) if order_bad.any() else n
        # a row that is itself malformed is reported before any ordering complaint about it; rows before it were fine
        if first_bad <= first_order and first_bad < n:
            self._word(int(line_no[first_bad]), parts[first_bad])
            raise AssertionError("unreachable: the masks flagged a row its checks accept")
        if first_order < n:
            k = first_order
            if unsorted[k]:
                raise ValueError(f"bedGraph input is not sorted at row {int(line_no[k])}; sort by chromosome sizes order, "
                                 "then start/end")
            raise ValueError(f"Overlapping bedGraph interval at row {int(line_no[k])}")
        self.prev = (rank[-1], start[-1], end[-1])
        return rank, start, end, value


def convert_bedgraph_to_bigwig(bedgraph_path: str, chrom_sizes_file: str, bigwig_path: str, *, chrom_sizes=None,
                               compress: bool = True, chunk_lines: int = 200_000, max_zooms: int = 10) -> None:
    """The reference's bedGraph -> bigWig entry point (io.py:633-790 `_convertBedGraphToBigWigPyBigWig`: same validation, same
    messages, output renamed into place only on success) without pyBigWig.  The file is read `chunk_lines` lines at a time;
    a chunk becomes four NumPy columns that are validated with array masks, a chromosome's columns become its data sections as
    soon as the next chromosome begins (the input is sorted by chromosome), so memory is bounded by the largest chromosome.
    Tracks with a fixed interval width (every Consenrich track) also get zoom levels (`zoom_plan`, up to `max_zooms`:
    pyBigWig's addHeader default is 10 levels); other bedGraphs are written without."""
    sizes = read_chrom_sizes(chrom_sizes_file) if chrom_sizes is None else [(str(c), int(s)) for c, s in chrom_sizes]
    if len(sizes) == 0:
        raise ValueError(f"No chromosome sizes found in {chrom_sizes_file}")
    cols = _BedGraphColumns(bedgraph_path, chrom_sizes_file, sizes)
    done: List[Tuple[int, np.ndarray, np.ndarray, np.ndarray]] = []     # finished chromosomes (rank, starts, ends, values)
    cur_rank, cur = -1, []

    def finish():
        if cur:
            done.append((cur_rank, np.concatenate([c[0] for c in cur]), np.concatenate([c[1] for c in cur]),
                         np.concatenate([c[2] for c in cur])))

    with open(bedgraph_path, "r", encoding="utf-8") as handle:
        line_number = 1
        while True:
            lines = handle.readlines(chunk_lines * 32) if chunk_lines > 0 else handle.readlines()
            if not lines:
                break
            got = cols.chunk(lines, line_number)
            line_number += len(lines)
            if got is None:
                continue
            rank, start, end, value = got
            cuts = np.nonzero(np.diff(rank))[0] + 1
            for lo, hi in zip(np.concatenate(([0], cuts)), np.concatenate((cuts, [rank.shape[0]]))):
                r = int(rank[lo])
                if r != cur_rank:
                    finish()
                    cur_rank, cur = r, []
                cur.append((start[lo:hi], end[lo:hi], value[lo:hi].astype(np.float32)))
    finish()
    if not done:
        raise ValueError(f"No bedGraph intervals found in {bedgraph_path}")           # io.py:766-767
    # one fixed interval width over the whole track (a chromosome holding a single, possibly clipped, interval fits any wider step)
    multi = {_uniform_step(s, e) for _r, s, e, _v in done if s.shape[0] > 1}
    step = multi.pop() if len(multi) == 1 else None
    if step and any(s.shape[0] == 1 and int(e[0] - s[0]) > step for _r, s, e, _v in done):
        step = None
    zooms = zoom_plan(step, max(s.shape[0] for _r, s, _e, _v in done), max_zooms) if step else []
    pieces = [piece_from_arrays(r, s, e, v, zoom_bases=zooms, step=step, apply_text4=False) for r, s, e, v in done]
    write_bigwig(bigwig_path, sizes, pieces, compress=compress)
