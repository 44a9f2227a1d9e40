"""bigWig writer (io.py:530-790) against the INDEPENDENT reader oracle/bigwig_reader.py (pyBigWig is not
in the image; see the header of consenrich_amd/bigwig.py for how this row is pinned).  CPU tests: the host path
(`convert_bedgraph_to_bigwig` = the reference's `_convertBedGraphToBigWigPyBigWig` with its validation and messages) and the
file assembly; GPU test: the device-formatted body of a resident track."""
import os

import numpy as np
import pytest

from conftest import gpu_available


def _toy(tmp_path):
    bg, cs = tmp_path / "toy.bedGraph", tmp_path / "toy.chrom.sizes"
    bg.write_text("\n".join(["track type=bedGraph name=toy", "browser position chr1:1-20", "chr1 0 10 0.5",
                             "chr1\t10\t20\t2.25", "chr2\t0\t8\t2.0", "chr10 0 5 10.0"]) + "\n", encoding="ascii")
    cs.write_text("chr1\t100\nchr2\t100\nchr10\t100\n", encoding="ascii")
    return bg, cs


@pytest.mark.parametrize("compress", [True, False])
def test_reference_known_answer_toy_track(tmp_path, compress):
    """The reference's own test of its pyBigWig conversion (tests/test_config.py:3196-3245): intervals per chromosome and the
    summary header (pyBigWig reports the float summary fields truncated to integers)."""
    from consenrich_amd import bigwig as bw
    from oracle.bigwig_reader import BigWig

    bg, cs = _toy(tmp_path)
    out = tmp_path / "toy.bw"
    bw.convert_bedgraph_to_bigwig(str(bg), str(cs), str(out), compress=compress)
    r = BigWig(str(out))
    assert {c: r.intervals(c) for c in ("chr1", "chr2", "chr10")} == {
        "chr1": [(0, 10, 0.5), (10, 20, 2.25)], "chr2": [(0, 8, 2.0)], "chr10": [(0, 5, 10.0)]}
    h = r.header()
    assert h["nBasesCovered"] == 33 and int(h["minVal"]) == 0 and int(h["maxVal"]) == 10
    assert int(h["sumData"]) == 93 and int(h["sumSquared"]) == 585
    assert r.chroms == {"chr1": (0, 100), "chr2": (1, 100), "chr10": (2, 100)}
    assert (r.uncompress_buf_size > 0) == compress


def test_reference_error_contract(tmp_path):
    """tests/test_config.py:3247-3281: out-of-bounds rows and empty inputs raise the reference's messages and leave no file."""
    from consenrich_amd import bigwig as bw

    cs = tmp_path / "c.sizes"
    cs.write_text("chr1\t100\n", encoding="ascii")
    bad = tmp_path / "bad.bedGraph"
    bad.write_text("chr1\t90\t101\t1.0\n", encoding="ascii")
    with pytest.raises(ValueError, match="exceeds chr1 size"):
        bw.convert_bedgraph_to_bigwig(str(bad), str(cs), str(tmp_path / "bad.bw"))
    assert not (tmp_path / "bad.bw").exists()
    empty = tmp_path / "empty.bedGraph"
    empty.write_text("track type=bedGraph name=empty\nbrowser position chr1:1-10\n", encoding="ascii")
    with pytest.raises(ValueError, match="No bedGraph intervals"):
        bw.convert_bedgraph_to_bigwig(str(empty), str(cs), str(tmp_path / "empty.bw"))
    assert not (tmp_path / "empty.bw").exists()
    for text, msg in (("chr1\t10\t5\t1.0\n", "End coordinate must be greater"), ("chr1\t0\t5\tnan\n", "Non-finite"),
                      ("chr1\t5\t9\t1\nchr1\t0\t4\t1\n", "not sorted"), ("chr1\t0\t9\t1\nchr1\t5\t12\t1\n", "Overlapping"),
                      ("chr7\t0\t9\t1\n", "not present"), ("chr1\t0\t9\n", "expected 4 columns")):
        p = tmp_path / "x.bedGraph"
        p.write_text(text, encoding="ascii")
        with pytest.raises(ValueError, match=msg):
            bw.convert_bedgraph_to_bigwig(str(p), str(cs), str(tmp_path / "x.bw"))
    assert [f for f in os.listdir(tmp_path) if f.endswith(".bw")] == []        # no temporary file left behind either


def test_multi_level_index_zoom_levels_and_text_values(tmp_path):
    """> 256 data blocks (a two-level R-tree), two chromosomes, zoom levels, values through the "%.4f" text: every interval
    read back through the index equals the parsed bedGraph row, range queries use the tree, zoom records obey their
    definition."""
    from consenrich_amd import bigwig as bw
    from oracle.bigwig_reader import BigWig

    rng = np.random.default_rng(5)
    step = 25
    sizes = [("chrA", 25 * 300001), ("chrB", 25 * 5000 - 7)]
    pieces, truth = [], {}
    zb = [10 * step, 40 * step]
    for cid, (name, size) in enumerate(sizes):
        n = -(-size // step)
        starts = np.arange(n, dtype=np.int64) * step
        ends = np.minimum(starts + step, size)
        vals = (rng.normal(size=n) * 3).astype(np.float32)
        pieces.append(bw.piece_from_arrays(cid, starts, ends, vals, zoom_bases=zb, step=step))
        truth[name] = (starts, ends, np.array([float("%.4f" % v) for v in vals.tolist()], np.float32))
    out = tmp_path / "big.bw"
    bw.write_bigwig(str(out), sizes, pieces, compress=True)
    r = BigWig(str(out))
    assert r.section_count == sum(-(-len(t[0]) // bw.ITEMS_PER_SECTION) for t in truth.values()) > 256
    for name, (s, e, v) in truth.items():
        got = r.intervals(name)
        assert len(got) == len(v)
        g = np.asarray(got)
        assert np.array_equal(g[:, 0], s) and np.array_equal(g[:, 1], e) and np.array_equal(g[:, 2].astype(np.float32), v)
    # a range query in the middle of chrA touches only the blocks the tree selects
    got = r.intervals("chrA", 25 * 150000, 25 * 150010)
    assert [x[0] for x in got] == [25 * k for k in range(150000, 150010)]
    h = r.header()
    w = np.concatenate([t[1] - t[0] for t in truth.values()]).astype(np.float64)
    vv = np.concatenate([t[2] for t in truth.values()]).astype(np.float64)
    assert h["nBasesCovered"] == int(w.sum()) and h["minVal"] == vv.min() and h["maxVal"] == vv.max()
    assert h["sumData"] == pytest.approx(float((vv * w).sum()), rel=1e-12)
    assert h["sumSquared"] == pytest.approx(float((vv * vv * w).sum()), rel=1e-12)
    assert [z[0] for z in r.zooms] == zb
    for lvl, bases in enumerate(zb):
        red, count, recs = r.zoom_records(lvl, "chrB")
        s, e, v = truth["chrB"]
        g = bases // step
        assert red == bases and len(recs) == -(-len(v) // g)
        k = 3
        sl = slice(k * g, (k + 1) * g)
        ww, dv = (e[sl] - s[sl]).astype(np.float64), v[sl].astype(np.float64)
        assert recs[k][0] == s[sl][0] and recs[k][1] == e[sl][-1] and recs[k][2] == int(ww.sum())
        assert recs[k][3] == np.float32(dv.min()) and recs[k][4] == np.float32(dv.max())
        assert recs[k][5] == pytest.approx(float((dv * ww).sum()), rel=1e-6)
        assert recs[k][6] == pytest.approx(float((dv * dv * ww).sum()), rel=1e-6)


@pytest.mark.gpu
def test_bigwig_of_resident_tracks_round_trips_to_the_bedgraph_text(tmp_path):
    """State and uncertainty tracks of a fitted batch: the device-formatted bigWig body (data sections, summary, zoom
    records), assembled and read back by the independent reader, equals (a) the rows of the byte-exact bedGraph text of the
    same tracks parsed the way pyBigWig receives them, (b) the NumPy statement of the same records bit for bit."""
    if not gpu_available():
        pytest.fail("GPU tests selected but no HIP device / library")
    import cases
    from consenrich_amd import _lib as L
    from consenrich_amd import bigwig as bw
    from consenrich_amd.batch import DeviceBatch, ModelParams
    from oracle.bigwig_reader import BigWig

    step = 50
    n_list, m = [300123, 4097, 1], 3
    names = ["chr1", "chr2", "chrM"]
    sizes = [(nm, step * n - 13 if n > 1 else 37) for nm, n in zip(names, n_list)]
    zb = bw.zoom_plan(step, max(n_list))
    assert len(zb) >= 1
    with DeviceBatch(0) as b:
        b.configure(ModelParams(state_dim=2), m, n_list)
        for c, n in enumerate(n_list):
            b.upload(c, *cases.synth(n, m, 8800 + c))
        b.step(L.RETURN_NLL, L.EXPORT_SMOOTH)
        for arr, transform, fname in (("xs", "round4", "state"), ("Ps", "sqrt", "uncertainty")):
            pieces, texts, host = [], {}, []
            for c, (nm, size) in enumerate(sizes):
                pieces.append(b.bigwig_track(c, arr, c, 0, step, end_cap=size, transform=transform, zoom_bases=zb))
                texts[nm] = b.bedgraph_bytes(c, arr, nm, 0, step, end_cap=size, transform=transform)
                raw = b.download(c, arr)
                v = raw[:, 0] if arr == "xs" else raw[:, 0, 0]
                v = np.round(v, 4) if transform == "round4" else np.sqrt(v)
                st = np.arange(n_list[c], dtype=np.int64) * step
                host.append(bw.piece_from_arrays(c, st, np.minimum(st + step, size), v.astype(np.float32), zoom_bases=zb, step=step))
            for dev, ref in zip(pieces, host):      # device records == their NumPy statement
                assert dev.sections == ref.sections and dev.bases_covered == ref.bases_covered
                assert dev.min_val == ref.min_val and dev.max_val == ref.max_val
                assert dev.sum_data == pytest.approx(ref.sum_data, rel=1e-12) and dev.sum_squares == pytest.approx(ref.sum_squares, rel=1e-12)
                for z in zb:
                    d = np.frombuffer(dev.zooms[z], bw.ZOOM_DTYPE)
                    r_ = np.frombuffer(ref.zooms[z], bw.ZOOM_DTYPE)
                    for f in ("chrom", "start", "end", "valid", "min", "max"):
                        assert np.array_equal(d[f], r_[f]), (z, f)
                    np.testing.assert_allclose(d["sum"], r_["sum"], rtol=2e-7)
                    np.testing.assert_allclose(d["sumsq"], r_["sumsq"], rtol=2e-7)
            out = tmp_path / f"{fname}.bw"
            bw.write_bigwig(str(out), sizes, pieces, compress=True)
            r = BigWig(str(out))
            for nm in names:
                rows = [ln.split("\t") for ln in texts[nm].decode("ascii").splitlines()]
                want = [(int(a), int(b_), float(np.float32(float(v)))) for _c, a, b_, v in rows]
                assert r.intervals(nm) == want, (fname, nm)
            assert r.header()["nBasesCovered"] == sum(s for _n, s in sizes)


def test_toy_file_is_byte_identical_to_the_hand_made_fixture(tmp_path):
    """tests/golden/bigwig_toy_uncompressed.bin was written out by hand from the bbi format tables with literal numbers
    (tests/golden/make_bigwig_toy_fixture.py imports nothing of the product): the writer's uncompressed file of the
    reference's toy track must be those 8555 bytes."""
    from consenrich_amd import bigwig as bw

    bg, cs = _toy(tmp_path)
    out = tmp_path / "toy.bw"
    bw.convert_bedgraph_to_bigwig(str(bg), str(cs), str(out), compress=False)
    want = open(os.path.join(os.path.dirname(__file__), "golden", "bigwig_toy_uncompressed.bin"), "rb").read()
    got = out.read_bytes()
    assert len(got) == len(want) == 8555
    diff = [i for i, (a, b) in enumerate(zip(got, want)) if a != b]
    assert not diff, diff[:8]


def test_fields_sit_at_the_offsets_the_format_fixes(tmp_path):
    """A two-chromosome fixed-step track with several sections and zoom levels, uncompressed, taken apart with struct at the byte
    offsets of the bbi tables (no reader class): header magic / version / offsets, zoom headers, chromosome-tree key and value
    sizes, section count, EVERY section header, R-tree block size and itemsPerSlot, the trailing magic."""
    import struct

    from consenrich_amd import bigwig as bw

    step = 50
    sizes = [("chrX", step * 12000 - 11), ("chrY", step * 1030)]
    lines, truth = [], []
    for name, size in sizes:
        n = -(-size // step)
        for k in range(n):
            lines.append(f"{name}\t{k * step}\t{min((k + 1) * step, size)}\t{(k % 97) * 0.25:.4f}")
        truth.append(n)
    bg, cs, out = tmp_path / "t.bedGraph", tmp_path / "t.sizes", tmp_path / "t.bw"
    bg.write_text("\n".join(lines) + "\n", encoding="ascii")
    cs.write_text("".join(f"{c}\t{s}\n" for c, s in sizes), encoding="ascii")
    bw.convert_bedgraph_to_bigwig(str(bg), str(cs), str(out), compress=False, chunk_lines=700)
    b = out.read_bytes()
    magic, version, nzoom, ct_off, data_off, index_off, fc, dfc, sql, summ_off, ubuf, ext = struct.unpack_from("<IHHQQQHHQQIQ", b, 0)
    assert (magic, version, fc, dfc, sql, ubuf, ext) == (0x888FFC26, 4, 0, 0, 0, 0, 0)
    assert nzoom >= 1 and summ_off == 64 + 24 * nzoom and ct_off == summ_off + 40
    reductions = [struct.unpack_from("<IIQQ", b, 64 + 24 * i) for i in range(nzoom)]
    assert [r[0] for r in reductions] == bw.zoom_plan(step, max(truth))[:nzoom] and all(r[1] == 0 for r in reductions)
    assert struct.unpack_from("<Q", b, summ_off)[0] == sum(s for _c, s in sizes)            # every base covered once
    tmagic, tblock, tkey, tval, titems, _ = struct.unpack_from("<IIIIQQ", b, ct_off)
    assert (tmagic, tblock, tkey, tval, titems) == (0x78CA8C91, 2, 4, 8, 2)
    assert data_off == ct_off + 32 + 4 + 2 * (4 + 8)
    nsec = struct.unpack_from("<Q", b, data_off)[0]
    assert nsec == sum(-(-n // bw.ITEMS_PER_SECTION) for n in truth)
    pos, seen = data_off + 8, []
    for cid, n in enumerate(truth):
        for j in range(-(-n // bw.ITEMS_PER_SECTION)):
            cnt = min(bw.ITEMS_PER_SECTION, n - j * bw.ITEMS_PER_SECTION)
            c, cs_, ce, istep, ispan, typ, res, count = struct.unpack_from("<IIIIIBBH", b, pos)
            first = j * bw.ITEMS_PER_SECTION
            assert (c, cs_, istep, ispan, typ, res, count) == (cid, first * step, 0, 0, 1, 0, cnt)
            assert ce == min((first + cnt) * step, sizes[cid][1])
            s0, e0, v0 = struct.unpack_from("<IIf", b, pos + 24)
            assert (s0, e0) == (first * step, min((first + 1) * step, sizes[cid][1])) and v0 == np.float32((first % 97) * 0.25)
            seen.append((pos, 24 + 12 * cnt))
            pos += 24 + 12 * cnt
    assert pos == index_off
    rmagic, rblock, ritems, sc, sb, ec, eb, endoff, per_slot, rres = struct.unpack_from("<IIQIIIIQII", b, index_off)
    assert (rmagic, rblock, ritems, sc, sb, ec, eb, per_slot, rres) == (0x2468ACE0, 256, nsec, 0, 0, 1, sizes[1][1], 1, 0)
    leaf, _r, count = struct.unpack_from("<BBH", b, index_off + 48)
    assert (leaf, count) == (1, nsec)
    for k, (off, ln) in enumerate(seen):
        assert struct.unpack_from("<IIIIQQ", b, index_off + 52 + 32 * k)[4:] == (off, ln)
    assert struct.unpack_from("<I", b, len(b) - 4)[0] == 0x888FFC26


def test_first_offending_row_decides_also_across_chunk_boundaries(tmp_path):
    """The reference's contract (io.py:693-752): rows are judged in order, the first offending row raises, its first failing
    check words the message.  Vectorised chunks of 3 lines here, so that predecessors sit in another chunk."""
    from consenrich_amd import bigwig as bw

    cs = tmp_path / "c.sizes"
    cs.write_text("chr1\t1000\nchr2\t1000\n", encoding="ascii")
    good = ["chr1\t0\t10\t1", "chr1\t10\t20\t1", "# comment", "chr1\t20\t30\t1", "chr1\t30\t40\t1"]
    cases = [
        (good + ["chr1\t35\t50\t1"], r"Overlapping bedGraph interval at row 6"),
        (good + ["chr1\t5\t8\t1"], r"not sorted at row 6"),
        (good + ["chr2\t0\t5\t1", "chr1\t50\t60\t1"], r"not sorted at row 7"),
        (good + ["chr1\t40\t50\t1\textra", "chr1\t0\t5\t1"], r"Malformed bedGraph row 6 .*expected 4 columns"),
        (good + ["chr1\t40\tx\t1"], r"Invalid bedGraph coordinates on row 6"),
        (good + ["chr1\t40\t50\tvalue"], r"Invalid bedGraph value on row 6"),
        (good + ["chr1\t-4\t-9\tinf"], r"Non-finite bedGraph value on row 6"),             # value is checked before coordinates' signs
        (good + ["chr1\t-4\t-9\t1"], r"Negative start coordinate on bedGraph row 6"),
        (good + ["chr1\t40\t1001\t1", "chrZ\t0\t1\t1"], r"End coordinate 1001 on bedGraph row 6 exceeds chr1 size of 1000"),
        (good + ["chr1\t10\t5\t1"], r"End coordinate must be greater than start on bedGraph row 6"),    # malformed before unsorted
        (["chr1\t0\t10\t1", "chr1\t0\t10\t1"], r"Overlapping bedGraph interval at row 2"),
    ]
    for rows, msg in cases:
        p = tmp_path / "x.bedGraph"
        p.write_text("\n".join(rows) + "\n", encoding="ascii")
        for chunk in (3, 200000):
            with pytest.raises(ValueError, match=msg):
                bw.convert_bedgraph_to_bigwig(str(p), str(cs), str(tmp_path / "x.bw"), chunk_lines=chunk)
    assert [f for f in os.listdir(tmp_path) if f.endswith(".bw")] == []


def test_chrom_sizes_reader_contract(tmp_path):
    from consenrich_amd import bigwig as bw

    p = tmp_path / "s.sizes"
    p.write_text("# header\nchr1\t100\textra\n\nchr2 200\n", encoding="ascii")
    assert bw.read_chrom_sizes(str(p)) == [("chr1", 100), ("chr2", 200)]
    for text, msg in (("chr1\n", "Malformed chromosome sizes row 1"), ("chr1\t10\nchr2\tten\n", "Invalid chromosome size on row 2"),
                      ("chr1\t0\n", "Chromosome chr1 has non-positive size on row 1"),
                      ("chr1\t5\nchr1\t6\n", "Duplicate chromosome chr1"), ("# nothing\n\n", "No chromosome sizes found"),
                      ("chr1\t5\nchr1\t6\nchrB\n", "Duplicate chromosome chr1"), ("chrB\nchr1\t5\nchr1\t6\n", "Malformed chromosome sizes row 1")):
        p.write_text(text, encoding="ascii")
        with pytest.raises(ValueError, match=msg):
            bw.read_chrom_sizes(str(p))


def test_file_path_writes_zoom_levels_and_a_multi_level_chromosome_tree(tmp_path):
    """pyBigWig's addHeader defaults to 10 zoom levels; the file path now writes the reduction levels of `zoom_plan` for a
    fixed-step track.  1 000 scaffolds (> 256: a two-level chromosome B+ tree) read back through the independent reader."""
    from consenrich_amd import bigwig as bw
    from oracle.bigwig_reader import BigWig

    step = 200
    sizes = [(f"scaffold_{k}", step * (7000 if k == 0 else 3) - (k % 5)) for k in range(1000)]
    lines = []
    for name, size in sizes:
        for k in range(-(-size // step)):
            lines.append(f"{name}\t{k * step}\t{min((k + 1) * step, size)}\t{((k * 7) % 13) * 0.5:.4f}")
    bg, cs, out = tmp_path / "t.bedGraph", tmp_path / "t.sizes", tmp_path / "t.bw"
    bg.write_text("\n".join(lines) + "\n", encoding="ascii")
    cs.write_text("".join(f"{c}\t{s}\n" for c, s in sizes), encoding="ascii")
    bw.convert_bedgraph_to_bigwig(str(bg), str(cs), str(out), chunk_lines=1500)
    r = BigWig(str(out))
    assert r.chroms == {c: (i, s) for i, (c, s) in enumerate(sizes)}
    assert [z[0] for z in r.zooms] == bw.zoom_plan(step, 7000) and len(r.zooms) >= 1
    assert len(r.intervals("scaffold_0")) == 7000 and r.intervals("scaffold_999") == [(0, 200, 0.0), (200, 400, 3.5), (400, 596, 0.5)]
    red, _count, recs = r.zoom_records(0, "scaffold_0")
    assert red == 10 * step and recs[0][:3] == (0, 10 * step, 10 * step)
