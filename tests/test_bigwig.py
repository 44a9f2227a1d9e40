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
