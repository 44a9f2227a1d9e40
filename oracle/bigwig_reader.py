"""TEST INFRASTRUCTURE ONLY -- an independent bigWig READER, written from the published file-format description (Kent WJ,
Zweig AS, Barber G, Hinrichs AS, Karolchik D: "BigWig and BigBed: enabling browsing of large distributed datasets",
Bioinformatics 2010, supplementary tables: common header, zoom headers, total summary, chromosome B+ tree, data sections of
type 1 bedGraph / 2 varStep / 3 fixedStep, R-tree index) with `struct.unpack` only.  It shares no code with the writer
(consenrich_amd/bigwig.py builds NumPy records): what the writer emits must parse here into the same intervals the
reference's pyBigWig-written file would return from `bw.intervals(chrom)` and the same summary `bw.header()` reports
(/root/reference/tests/test_config.py:3187-3245).  pyBigWig itself is not in the image: "pinned by an independent reader
and the reference's own known-answer test", not by a reference-produced file.  Never imported by ``consenrich_amd``.
"""
from __future__ import annotations

import struct
import zlib


class BigWig:
    def __init__(self, path):
        with open(path, "rb") as fh:
            self.buf = fh.read()
        b = self.buf
        (magic, self.version, self.zoom_levels, self.chrom_tree_offset, self.full_data_offset, self.full_index_offset,
         self.field_count, self.defined_field_count, self.auto_sql_offset, self.total_summary_offset,
         self.uncompress_buf_size, self.extension_offset) = struct.unpack_from("<IHHQQQHHQQIQ", b, 0)
        if magic != 0x888FFC26:
            raise ValueError("not a little-endian bigWig file")
        if struct.unpack_from("<I", b, len(b) - 4)[0] != 0x888FFC26:
            raise ValueError("trailing magic missing")
        self.zooms = [struct.unpack_from("<IIQQ", b, 64 + 24 * i) for i in range(self.zoom_levels)]   # reduction, reserved, data, index
        self.chroms = self._read_chrom_tree()
        self.section_count = struct.unpack_from("<Q", b, self.full_data_offset)[0]

    # -- chromosome B+ tree ------------------------------------------------------------------------------------------
    def _read_chrom_tree(self):
        b, off = self.buf, self.chrom_tree_offset
        magic, block_size, key_size, val_size, item_count, _res = struct.unpack_from("<IIIIQQ", b, off)
        if magic != 0x78CA8C91 or val_size != 8:
            raise ValueError("bad chromosome tree")
        out = {}

        def node(pos):
            is_leaf, _r, count = struct.unpack_from("<BBH", b, pos)
            pos += 4
            for _ in range(count):
                key = b[pos:pos + key_size].rstrip(b"\0").decode("ascii")
                if is_leaf:
                    cid, size = struct.unpack_from("<II", b, pos + key_size)
                    out[key] = (cid, size)
                else:
                    node(struct.unpack_from("<Q", b, pos + key_size)[0])
                pos += key_size + 8

        node(off + 32)
        if len(out) != item_count:
            raise ValueError("chromosome tree item count mismatch")
        return out

    # -- R-tree ------------------------------------------------------------------------------------------------------
    def _blocks(self, index_offset, chrom_id, start, end):
        b = self.buf
        magic, block_size, item_count, s_chrom, s_base, e_chrom, e_base, end_off, per_slot, _res = \
            struct.unpack_from("<IIQIIIIQII", b, index_offset)
        if magic != 0x2468ACE0:
            raise ValueError("bad R-tree magic")
        hits = []

        def overlaps(sc, sb, ec, eb):
            return (sc, sb) < (chrom_id, end) and (ec, eb) > (chrom_id, start)

        def node(pos):
            is_leaf, _r, count = struct.unpack_from("<BBH", b, pos)
            pos += 4
            for _ in range(count):
                if is_leaf:
                    sc, sb, ec, eb, doff, dsize = struct.unpack_from("<IIIIQQ", b, pos)
                    if overlaps(sc, sb, ec, eb):
                        hits.append((doff, dsize))
                    pos += 32
                else:
                    sc, sb, ec, eb, child = struct.unpack_from("<IIIIQ", b, pos)
                    if overlaps(sc, sb, ec, eb):
                        node(child)
                    pos += 24

        node(index_offset + 48)
        return hits, item_count

    def _payload(self, off, size):
        raw = self.buf[off:off + size]
        if self.uncompress_buf_size:
            raw = zlib.decompress(raw)
            if len(raw) > self.uncompress_buf_size:
                raise ValueError("block larger than uncompressBufSize")
        return raw

    # -- public ------------------------------------------------------------------------------------------------------
    def intervals(self, chrom, start=0, end=None):
        """[(start, end, value float32 as Python float)] like pyBigWig's bw.intervals(chrom)."""
        cid, size = self.chroms[chrom]
        end = size if end is None else end
        hits, _n = self._blocks(self.full_index_offset, cid, start, end)
        out = []
        for off, sz in sorted(hits):
            raw = self._payload(off, sz)
            c, cs, ce, step, span, typ, _r, count = struct.unpack_from("<IIIIIBBH", raw, 0)
            if c != cid:
                continue
            pos = 24
            for i in range(count):
                if typ == 1:
                    s, e, v = struct.unpack_from("<IIf", raw, pos)
                    pos += 12
                elif typ == 2:
                    s, v = struct.unpack_from("<If", raw, pos)
                    e = s + span
                    pos += 8
                elif typ == 3:
                    (v,) = struct.unpack_from("<f", raw, pos)
                    s = cs + i * step
                    e = s + span
                    pos += 4
                else:
                    raise ValueError("unknown section type")
                if s < end and e > start:
                    out.append((s, e, v))
        return out

    def header(self):
        bases, mn, mx, sm, sq = struct.unpack_from("<Qdddd", self.buf, self.total_summary_offset)
        return {"version": self.version, "nLevels": self.zoom_levels, "nBasesCovered": bases, "minVal": mn, "maxVal": mx,
                "sumData": sm, "sumSquared": sq}

    def zoom_records(self, level, chrom):
        """[(start, end, validCount, min, max, sum, sumSquares)] of one zoom level for one chromosome."""
        reduction, _res, data_off, index_off = self.zooms[level]
        cid, size = self.chroms[chrom]
        count = struct.unpack_from("<I", self.buf, data_off)[0]
        hits, _n = self._blocks(index_off, cid, 0, size)
        out = []
        for off, sz in sorted(hits):
            raw = self._payload(off, sz)
            for k in range(len(raw) // 32):
                c, s, e, valid, mn, mx, sm, sq = struct.unpack_from("<IIIIffff", raw, 32 * k)
                if c == cid:
                    out.append((s, e, valid, mn, mx, sm, sq))
        return reduction, count, out
