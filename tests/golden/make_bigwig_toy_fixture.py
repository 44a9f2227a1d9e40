"""Expected bytes of the toy bigWig (the reference's known-answer track, /root/reference/tests/test_config.py:3196-3245),
UNCOMPRESSED (uncompressBufSize = 0), written out BY HAND from the bbi file-format tables (Kent et al. 2010, supplement:
"common header", "total summary", "chromosome B+ tree", "bedGraph section", "R tree index") with literal numbers only.
Nothing of consenrich_amd is imported: this is the independent side of tests/test_bigwig.py::test_toy_file_is_byte_identical_
to_the_hand_made_fixture.  Free choices of a writer that the format permits and this fixture fixes: sections of <= 1024
items (one per chromosome here), a chromosome tree of one leaf whose block size is the chromosome count, an R tree with 256
slots per node, one data block per slot, nodes padded to full size, the index's endFileOffset = the index offset.

  python tests/golden/make_bigwig_toy_fixture.py        -> tests/golden/bigwig_toy_uncompressed.bin (8555 bytes)
"""
import os
import struct

out = bytearray()
# ---- common header, 64 bytes -----------------------------------------------------------------------------------------
CHROM_TREE_AT = 64 + 40                       # no zoom headers; the total summary sits right behind the header
CHROM_TREE_LEN = 32 + 4 + 3 * (5 + 8)         # tree header, node header, 3 items of (5-byte key, chromId, chromSize)
DATA_AT = CHROM_TREE_AT + CHROM_TREE_LEN      # 179
BLOCKS_AT = DATA_AT + 8                       # behind the u64 section count: 187
BLOCK_LEN = [24 + 2 * 12, 24 + 12, 24 + 12]   # section header + items
INDEX_AT = BLOCKS_AT + sum(BLOCK_LEN)         # 307
out += struct.pack("<I", 0x888FFC26)          # magic
out += struct.pack("<H", 4)                   # version
out += struct.pack("<H", 0)                   # zoomLevels
out += struct.pack("<Q", CHROM_TREE_AT)       # chromosomeTreeOffset
out += struct.pack("<Q", DATA_AT)             # fullDataOffset
out += struct.pack("<Q", INDEX_AT)            # fullIndexOffset
out += struct.pack("<H", 0)                   # fieldCount (bigWig: 0)
out += struct.pack("<H", 0)                   # definedFieldCount
out += struct.pack("<Q", 0)                   # autoSqlOffset
out += struct.pack("<Q", 64)                  # totalSummaryOffset
out += struct.pack("<I", 0)                   # uncompressBufSize: 0 = not compressed
out += struct.pack("<Q", 0)                   # reserved / extensionOffset
assert len(out) == 64
# ---- total summary, 40 bytes -----------------------------------------------------------------------------------------
# rows: chr1 0-10 0.5 | chr1 10-20 2.25 | chr2 0-8 2.0 | chr10 0-5 10.0
out += struct.pack("<Q", 10 + 10 + 8 + 5)                                        # basesCovered = 33
out += struct.pack("<d", 0.5)                                                    # minVal
out += struct.pack("<d", 10.0)                                                   # maxVal
out += struct.pack("<d", 0.5 * 10 + 2.25 * 10 + 2.0 * 8 + 10.0 * 5)              # sumData = 93.5
out += struct.pack("<d", 0.25 * 10 + 5.0625 * 10 + 4.0 * 8 + 100.0 * 5)          # sumSquares = 585.125
assert len(out) == CHROM_TREE_AT
# ---- chromosome B+ tree ----------------------------------------------------------------------------------------------
out += struct.pack("<I", 0x78CA8C91)          # magic
out += struct.pack("<I", 3)                   # blockSize
out += struct.pack("<I", 5)                   # keySize = len("chr10")
out += struct.pack("<I", 8)                   # valSize
out += struct.pack("<Q", 3)                   # itemCount
out += struct.pack("<Q", 0)                   # reserved
out += struct.pack("<BBH", 1, 0, 3)           # leaf node, 3 items, keys in byte order: chr1 < chr10 < chr2
out += b"chr1\0" + struct.pack("<II", 0, 100)     # chromId = position in the chromosome sizes file
out += b"chr10" + struct.pack("<II", 2, 100)
out += b"chr2\0" + struct.pack("<II", 1, 100)
assert len(out) == DATA_AT
# ---- data: section count, then bedGraph-type sections ----------------------------------------------------------------
out += struct.pack("<Q", 3)
for chrom_id, rows in ((0, [(0, 10, 0.5), (10, 20, 2.25)]), (1, [(0, 8, 2.0)]), (2, [(0, 5, 10.0)])):
    out += struct.pack("<I", chrom_id)        # chromId
    out += struct.pack("<I", rows[0][0])      # chromStart
    out += struct.pack("<I", rows[-1][1])     # chromEnd
    out += struct.pack("<I", 0)               # itemStep (bedGraph: unused)
    out += struct.pack("<I", 0)               # itemSpan (bedGraph: unused)
    out += struct.pack("<B", 1)               # type 1 = bedGraph
    out += struct.pack("<B", 0)               # reserved
    out += struct.pack("<H", len(rows))       # itemCount
    for s, e, v in rows:
        out += struct.pack("<IIf", s, e, v)
assert len(out) == INDEX_AT
# ---- R tree index ----------------------------------------------------------------------------------------------------
out += struct.pack("<I", 0x2468ACE0)          # magic
out += struct.pack("<I", 256)                 # blockSize
out += struct.pack("<Q", 3)                   # itemCount
out += struct.pack("<IIII", 0, 0, 2, 5)       # startChromIx, startBase, endChromIx, endBase
out += struct.pack("<Q", INDEX_AT)            # endFileOffset
out += struct.pack("<I", 1)                   # itemsPerSlot
out += struct.pack("<I", 0)                   # reserved
out += struct.pack("<BBH", 1, 0, 3)           # one leaf node
pos = BLOCKS_AT
for (cid, cs, ce), ln in zip(((0, 0, 20), (1, 0, 8), (2, 0, 5)), BLOCK_LEN):
    out += struct.pack("<IIIIQQ", cid, cs, cid, ce, pos, ln)
    pos += ln
out += bytes((256 - 3) * 32)                  # the node is padded to its full size
out += struct.pack("<I", 0x888FFC26)          # a bbi file ends with its magic
assert len(out) == 8555, len(out)
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bigwig_toy_uncompressed.bin"), "wb") as fh:
    fh.write(bytes(out))
print(len(out))
