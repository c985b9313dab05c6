"""v3 segment directory reader: pinned by the index_map / metadata of a segment the reference built
(tests/golden/startree_airline/segment_meta.json) and by a write → load → query round trip."""
import json
import os

import numpy as np
import pytest

from pinot_amd import segment_dir
from pinot_amd.executor import NativeSegment
from pinot_amd.segment import build_segment
from tests.fixtures import synth_star_segment

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "startree_airline")


def test_reference_index_map_follows_the_assumed_layouts():
    """Every dictionary / forward-index entry of the reference-built airlineStats segment has exactly the size our readers'
    layouts imply: marker + cardinality x width (fixed-width dictionaries, STRING padded to lengthOfEachEntry),
    marker + ceil(N x bitsPerElement / 8) (FixedBitSVForwardIndexReaderV2), marker + 8 x cardinality (sorted index)."""
    sm = json.load(open(os.path.join(GOLDEN, "segment_meta.json")))
    imap = segment_dir.parse_index_map({k: [v] for k, v in sm["index_map"].items()})
    cols = segment_dir.column_metadata({k: [v] for k, v in sm["columns"].items()})
    total = int(sm["segment.total.docs"])
    assert sm["segment.index.version"] == "v3" and total == 313 and len(cols) > 70
    checked = {"dictionary": 0, "fixed_bit": 0, "sorted": 0}
    for name, m in cols.items():
        exp = segment_dir.expected_entry_sizes(m, total)
        if "dictionary" in exp:
            assert imap[(name, "dictionary")][1] == exp["dictionary"], name
            checked["dictionary"] += 1
        if exp.get("forward_index") is not None:
            assert imap[(name, "forward_index")][1] == exp["forward_index"], name
            checked["sorted" if m["isSorted"] == "true" else "fixed_bit"] += 1
    assert checked["dictionary"] > 70 and checked["fixed_bit"] > 40 and checked["sorted"] >= 5
    # entries are laid back to back in columns.psf: each start = previous start + size
    spans = sorted(imap.values())
    assert spans[0][0] == 0
    for (s0, n0), (s1, _) in zip(spans, spans[1:]):
        assert s0 + n0 == s1
    # dotted / dollar column names parse ($ts$DAY is a generated timestamp-index column)
    assert ("$ts$DAY", "range_index") in imap


def test_segment_dir_round_trip(tmp_path, oracle_api):
    rng = np.random.default_rng(3)
    n = 5000
    data = {"a": rng.integers(0, 50, n).astype(np.int32), "s": rng.choice(["x", "yy", "zzz", "wwww"], n),
            "t": np.sort(rng.integers(0, 20, n)).astype(np.int32), "raw.m": rng.integers(-1000, 1000, n).astype(np.int64),
            "f": rng.random(n).astype(np.float32)}
    host = build_segment("rt", {k: (v.tolist() if v.dtype.kind == "U" else v) for k, v in data.items()},
                         {"a": "INT", "s": "STRING", "t": "INT", "raw.m": "LONG", "f": "FLOAT"},
                         inverted_index_columns=["a", "s"], no_dictionary_columns=["raw.m"])
    segment_dir.write_segment_dir(host, str(tmp_path / "rt"))
    back = segment_dir.load_segment_dir(str(tmp_path / "rt"))
    assert back.total_docs == n and set(back.columns) == set(host.columns) and not back.skipped
    for name, c in host.columns.items():
        b = back.columns[name]
        np.testing.assert_array_equal(b.forward_index, c.forward_index)
        assert (b.fwd_encoding, b.has_dictionary, b.cardinality, b.bits_per_value, b.is_sorted) == \
            (c.fwd_encoding, c.has_dictionary, c.cardinality, c.bits_per_value, c.is_sorted)
        if c.has_dictionary:
            np.testing.assert_array_equal(b.dictionary, c.dictionary)
            assert list(b.dict_values) == list(c.dict_values)
        if c.inverted_index is not None:
            np.testing.assert_array_equal(b.inverted_index, c.inverted_index)
    q = "SELECT s, COUNT(*), SUM(a), MAX(f) FROM rt WHERE a IN (1, 2, 3, 40) AND t > 4 GROUP BY s"
    s0, s1 = NativeSegment(oracle_api, host), NativeSegment(oracle_api, back)
    assert s0.execute(q).rows() == s1.execute(q).rows()
    # a corrupted marker is detected like SingleFileIndexDirectory#validateMagicMarker
    psf = tmp_path / "rt" / "v3" / "columns.psf"
    raw = bytearray(psf.read_bytes())
    raw[0] ^= 0xFF
    psf.write_bytes(bytes(raw))
    with pytest.raises(ValueError, match="possibly corrupted"):
        segment_dir.load_segment_dir(str(tmp_path / "rt"))


def test_segment_dir_with_star_tree(tmp_path, oracle_api):
    host = synth_star_segment(6000, max_leaf_records=50, skip=())
    segment_dir.write_segment_dir(host, str(tmp_path / "st"))
    back = segment_dir.load_segment_dir(str(tmp_path / "st"))
    assert len(back.star_trees) == 1
    st0, st1 = host.star_trees[0], back.star_trees[0]
    assert (st1.num_docs, st1.dimensions, [p.name for p in st1.pairs]) == (st0.num_docs, st0.dimensions, [p.name for p in st0.pairs])
    np.testing.assert_array_equal(st1.star_tree, st0.star_tree)
    q = "SELECT h1, COUNT(*), SUM(m), DISTINCTCOUNTHLL(u) FROM t WHERE h2 > 3 GROUP BY h1"
    a, b = NativeSegment(oracle_api, host).execute(q), NativeSegment(oracle_api, back).execute(q)
    assert a.stats.star_tree_index == b.stats.star_tree_index == 0
    assert a.rows() == b.rows()


# ---- the same v3 directories through the GPU library (SURVEY.md §8f rank 1 on the product path) ---------------------------------
@pytest.mark.gpu
def test_gpu_queries_a_loaded_v3_segment_dir(tmp_path, gpu_api, oracle_api):
    """columns.psf + index_map + metadata.properties -> load_segment_dir -> libpinot_gpu: every index kind the reader hands over
    (dictionary + fixed-bit, sorted, raw chunks, inverted index, null value vector) answers like the oracle over the same bytes."""
    rng = np.random.default_rng(17)
    n = 70_003
    data = {"a": rng.integers(0, 50, n).astype(np.int32), "s": rng.choice(["x", "yy", "zzz", "wwww"], n),
            "t": np.sort(rng.integers(0, 200, n)).astype(np.int32), "raw.m": rng.integers(-1000, 1000, n).astype(np.int64),
            "f": rng.random(n).astype(np.float32), "d": rng.normal(0, 10, n)}
    host = build_segment("rt", {k: (v.tolist() if v.dtype.kind == "U" else v) for k, v in data.items()},
                         {"a": "INT", "s": "STRING", "t": "INT", "raw.m": "LONG", "f": "FLOAT", "d": "DOUBLE"},
                         inverted_index_columns=["a", "s"], no_dictionary_columns=["raw.m", "d"], range_index_columns=["raw.m", "f"])
    segment_dir.write_segment_dir(host, str(tmp_path / "rt"))
    back = segment_dir.load_segment_dir(str(tmp_path / "rt"))
    assert not back.skipped
    assert back.columns["raw.m"].range_index is not None and back.columns["f"].range_index is not None and back.columns["a"].range_index is None
    g, o = NativeSegment(gpu_api, back), NativeSegment(oracle_api, back)
    import math
    for q in ("SELECT s, COUNT(*), SUM(a), MAX(f) FROM rt WHERE a IN (1, 2, 3, 40) AND t > 4 GROUP BY s",
              "SELECT t, COUNT(*), MIN(raw.m), MAX(raw.m) FROM rt WHERE t BETWEEN 20 AND 90 AND s != 'yy' GROUP BY t LIMIT 1000",
              "SELECT COUNT(*), SUM(raw.m), AVG(a) FROM rt WHERE (a < 10 OR raw.m > 500) AND NOT s IN ('x')",
              "SELECT a, DISTINCTCOUNT(s), DISTINCTCOUNTHLL(t) FROM rt WHERE f < 0.5 GROUP BY a LIMIT 100",
              "SELECT s, SUM(d), SUM(f) FROM rt GROUP BY s"):
        gb, ob = g.execute(q), o.execute(q)
        gr, orr = gb.rows(), ob.rows()
        assert sorted(gr) == sorted(orr), q
        for k in orr:
            for x, y in zip(gr[k], orr[k]):
                if isinstance(y, float) and "SUM(d)" in q:      # floating SUM: exact on the GPU, sequential double in the oracle
                    assert math.isclose(x, y, rel_tol=1e-12, abs_tol=1e-9), (q, k, x, y)
                else:
                    assert x == y, (q, k, x, y)
        assert (gb.stats.num_docs_scanned, gb.stats.num_entries_scanned_in_filter, gb.stats.num_entries_scanned_post_filter) == \
            (ob.stats.num_docs_scanned, ob.stats.num_entries_scanned_in_filter, ob.stats.num_entries_scanned_post_filter), q
    dg, do = g.filter("SELECT COUNT(*) FROM rt WHERE t > 150 AND a = 7"), o.filter("SELECT COUNT(*) FROM rt WHERE t > 150 AND a = 7")
    np.testing.assert_array_equal(dg.doc_ids(), do.doc_ids())
    g.destroy()
    o.destroy()


@pytest.mark.gpu
def test_gpu_queries_a_loaded_star_tree_segment_dir(tmp_path, gpu_api, oracle_api):
    host = synth_star_segment(30_000, max_leaf_records=50, skip=())
    segment_dir.write_segment_dir(host, str(tmp_path / "st"))
    back = segment_dir.load_segment_dir(str(tmp_path / "st"))
    g, o = NativeSegment(gpu_api, back), NativeSegment(oracle_api, back)
    for q in ("SELECT h1, COUNT(*), SUM(m), DISTINCTCOUNTHLL(u) FROM t WHERE h2 > 3 GROUP BY h1",
              "SELECT h1, h2, h3, h4, COUNT(*), DISTINCTCOUNTHLL(u) FROM t GROUP BY h1, h2, h3, h4 LIMIT 20000",
              "SELECT g2, COUNT(*) FROM t GROUP BY g2"):
        gb, ob = g.execute(q), o.execute(q)
        assert gb.stats.star_tree_index == ob.stats.star_tree_index
        assert gb.rows() == ob.rows(), q
        assert gb.stats.num_docs_scanned == ob.stats.num_docs_scanned
    g.destroy()
    o.destroy()
