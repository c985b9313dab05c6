"""Star-tree index: byte formats and the builder, pinned to the reference's own star-tree fixture
(tests/golden/startree_airline: built by the reference over airlineStats, extracted by tests/golden/make_startree_airline.py).
"""
import json
import os
import struct

import numpy as np
import pytest

from pinot_amd import formats, startree

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "startree_airline")


def load_fixture():
    meta = json.load(open(os.path.join(GOLDEN, "meta.json")))
    blob = np.fromfile(os.path.join(GOLDEN, "star_tree_index"), dtype=np.uint8)
    im = meta["index_map"]

    def entry(col, kind):
        off, size = im[f"0.{col}.{kind}.OFFSET"], im[f"0.{col}.{kind}.SIZE"]
        return blob[off:off + size]
    return meta, entry


def decode_fixture():
    meta, entry = load_fixture()
    n = meta["total_docs"]
    names, nodes = formats.read_star_tree(entry("null", "STAR_TREE"))
    dims = np.stack([formats.unpack_fixed_bit(entry(d, "FORWARD_INDEX"), meta["columns"][d]["bitsPerElement"], n)
                     for d in meta["split_order"]], axis=1)
    cnt_buf, max_buf = entry("count__*", "FORWARD_INDEX"), entry("max__ArrDelay", "FORWARD_INDEX")
    h = formats.parse_raw_fixed_byte_chunk_header(cnt_buf)
    counts = np.frombuffer(bytes(cnt_buf), dtype=">i8", count=n, offset=h["raw_data_start"]).astype(np.int64)
    maxes = np.frombuffer(bytes(max_buf), dtype=">f8", count=n, offset=h["raw_data_start"]).astype(np.float64)
    return meta, entry, names, nodes, dims, counts, maxes


def test_fixture_tree_parses_like_offheap_star_tree():
    meta, entry, names, nodes, dims, counts, maxes = decode_fixture()
    assert names == meta["split_order"] == ["AirlineID", "Origin", "Dest"]
    assert nodes.shape == (666, 7)
    root = nodes[0]
    assert root[0] == -1 and root[1] == -1               # dimensionId, dimensionValue of the root (INVALID_ID)
    assert root[5] == 1                                  # BFS: the root's first child is node 1
    # raw forward indexes are FixedByteChunk v2, PASS_THROUGH, 1000 docs per chunk
    h = formats.parse_raw_fixed_byte_chunk_header(entry("count__*", "FORWARD_INDEX"))
    assert (h["version"], h["num_chunks"], h["docs_per_chunk"], h["size_of_entry"], h["total_docs"], h["compression"]) == \
        (2, 2, 1000, 8, 1004, 0)
    # children of every node are sorted by dimension value, star (-1) first; child ranges partition the parent's range
    for nd in nodes:
        if nd[5] < 0:
            continue
        kids = nodes[nd[5]:nd[6] + 1]
        assert np.all(np.diff(kids[:, 1]) > 0)
        assert np.all(kids[:, 0] == nd[0] + 1)
    # known answers: the root's aggregated doc holds the whole segment
    agg = root[4]
    assert counts[agg] == meta["segment_total_docs"] == 313
    assert maxes[agg] == float(meta["columns"]["ArrDelay"]["maxValue"]) == 343.0


def test_builder_reproduces_reference_fixture_byte_for_byte():
    """Stage 2-4 of the builder (constructStarTree / createAggregatedDocs / serializeTree) re-run over the fixture's base
    docs must give back the reference's star-tree: tree file and all five forward indexes."""
    meta, entry, names, nodes, dims, counts, maxes = decode_fixture()
    root_kids = nodes[nodes[0][5]:nodes[0][6] + 1]
    n_base = int(max(k[3] for k in root_kids if k[1] != -1))      # the non-star children of the root cover the base docs
    assert n_base == 306
    assert int(counts[:n_base].sum()) == 313
    cards = [meta["columns"][d]["cardinality"] for d in names]
    st = startree.build_from_base_records(names, cards, dims[:n_base], [("COUNT", "*"), ("MAX", "ArrDelay")],
                                          [counts[:n_base].tolist(), maxes[:n_base].tolist()],
                                          max_leaf_records=meta["max_leaf_records"])
    assert st.num_docs == meta["total_docs"] == 1004
    assert st.n_base_docs == n_base
    np.testing.assert_array_equal(st.star_tree, entry("null", "STAR_TREE"))
    for j, d in enumerate(names):
        np.testing.assert_array_equal(st.dimension_forward_indexes[j], entry(d, "FORWARD_INDEX"))
    np.testing.assert_array_equal(st.pairs[0].forward_index, entry("count__*", "FORWARD_INDEX"))
    np.testing.assert_array_equal(st.pairs[1].forward_index, entry("max__ArrDelay", "FORWARD_INDEX"))


def test_star_tree_file_round_trip():
    meta, entry, names, nodes, *_ = decode_fixture()
    np.testing.assert_array_equal(formats.write_star_tree(names, nodes), entry("null", "STAR_TREE"))
    bad = entry("null", "STAR_TREE").copy()
    bad[0] ^= 1
    with pytest.raises(ValueError, match="magic"):
        formats.read_star_tree(bad)


def test_var_byte_chunk_and_hll_round_trip():
    rng = np.random.default_rng(5)
    vals = [bytes(rng.integers(0, 256, rng.integers(0, 40), dtype=np.uint8)) for _ in range(2503)]
    for version in (2, 3):
        buf = formats.write_raw_var_byte_chunk(vals, version=version, docs_per_chunk=1000)
        assert formats.read_raw_var_byte_chunk(buf) == vals
    regs = rng.integers(0, 27, 256, dtype=np.uint8)
    blob = formats.serialize_hll(regs, 8)
    assert len(blob) == 180 and blob[:8] == bytes.fromhex("00000008000000ac")     # rawhllresults.txt header
    log2m, back = formats.deserialize_hll(blob)
    assert log2m == 8
    np.testing.assert_array_equal(back, regs)
    # the reference's serialized blobs decode and re-encode to themselves
    path = os.path.join(os.path.dirname(GOLDEN), "rawhllresults.txt")
    n = 0
    for line in open(path):
        for tok in line.replace(",", " ").split():
            if len(tok) == 360 and tok.startswith("00000008000000ac"):
                raw = bytes.fromhex(tok)
                lg, r = formats.deserialize_hll(raw)
                assert formats.serialize_hll(r, lg) == raw
                n += 1
    assert n > 0


def test_java_hashmap_iteration_order():
    # keys 0..12 + star: capacity 32; star (hash 0xFFFF0000) lands in bucket 0 behind key 0
    keys = list(range(13)) + [-1]
    order = [keys[i] for i in startree._java_hashmap_order(keys)]
    assert order == [0, -1] + list(range(1, 13))
    # capacity 16: 17 and 1 share bucket 1 in insertion order
    keys = [1, 5, 17]
    assert [keys[i] for i in startree._java_hashmap_order(keys)] == [1, 17, 5]


# ---- queries on the reference-built star-tree: star-tree path == non-star-tree path (BaseStarTreeV2Test.java:216-235) ------
from pinot_amd import capi  # noqa: E402
from pinot_amd.executor import NativeSegment  # noqa: E402
from pinot_amd.query import parse_sql  # noqa: E402
from tests.fixtures import airline_star_segment  # noqa: E402

AIRLINE_QUERIES = [
    # (sql, uses the star-tree?)
    ("SELECT COUNT(*), MAX(ArrDelay) FROM t", True),
    ("SELECT COUNT(*) FROM t GROUP BY AirlineID", True),
    ("SELECT COUNT(*), MAX(ArrDelay) FROM t GROUP BY Origin", True),
    ("SELECT MAX(ArrDelay) FROM t GROUP BY Dest", True),
    ("SELECT COUNT(*), MAX(ArrDelay) FROM t GROUP BY AirlineID, Dest", True),
    ("SELECT COUNT(*), MAX(ArrDelay) FROM t GROUP BY Dest, Origin, AirlineID", True),
    ("SELECT COUNT(*), MAX(ArrDelay) FROM t WHERE AirlineID = 3", True),
    ("SELECT COUNT(*), MAX(ArrDelay) FROM t WHERE AirlineID IN (1, 4, 9) GROUP BY Origin", True),
    ("SELECT COUNT(*) FROM t WHERE Origin BETWEEN 10 AND 60 GROUP BY AirlineID", True),
    ("SELECT COUNT(*), MAX(ArrDelay) FROM t WHERE Dest > 50", True),
    ("SELECT COUNT(*), MAX(ArrDelay) FROM t WHERE Dest > 50 AND Origin < 40", True),
    ("SELECT COUNT(*) FROM t WHERE Dest != 7 GROUP BY Dest", True),
    ("SELECT COUNT(*) FROM t WHERE NOT AirlineID IN (0, 2, 5)", True),
    ("SELECT COUNT(*), MAX(ArrDelay) FROM t WHERE Origin = 5 OR Origin > 80 GROUP BY AirlineID", True),
    ("SELECT COUNT(*) FROM t WHERE AirlineID NOT IN (0, 1) AND Dest IN (3, 30, 60, 90) AND Origin >= 20 GROUP BY Dest", True),
    ("SELECT COUNT(*), MAX(ArrDelay) FROM t WHERE AirlineID >= 0", True),      # always-true predicate is dropped
    ("SELECT COUNT(*) FROM t WHERE AirlineID >= 0", False),                    # FastFilteredCountOperator comes first
    ("SELECT COUNT(*), MAX(ArrDelay) FROM t WHERE AirlineID = 3 OR Dest = 4", False),   # OR over two columns
    ("SELECT COUNT(*), MAX(ArrDelay) FROM t WHERE ArrDelay > 0", False),       # predicate column is not a dimension
    ("SELECT SUM(ArrDelay) FROM t GROUP BY AirlineID", False),                # sum__ArrDelay is not in the tree
    ("SELECT COUNT(*) FROM t WHERE NOT (AirlineID = 3 AND Dest = 4)", False),  # AND nested under NOT
]


@pytest.fixture(scope="module")
def airline(oracle_api):
    host, meta = airline_star_segment()
    seg = NativeSegment(oracle_api, host)
    yield seg, host, meta
    seg.destroy()


def run_both(seg, sql):
    star = seg.execute(parse_sql(sql))
    qc = parse_sql(sql)
    qc.flags |= capi.QUERY_FLAG_SKIP_STAR_TREE
    plain = seg.execute(qc)
    return star, plain


def test_oracle_known_answers_on_reference_star_tree(airline):
    seg, host, meta = airline
    b = seg.execute("SELECT COUNT(*), MAX(ArrDelay) FROM t")
    assert b.aggregation_result() == [313, 343.0]
    st = b.stats
    # every dimension is starred: exactly the root's aggregated document is read
    assert (st.star_tree_index, st.num_docs_scanned, st.num_entries_scanned_in_filter, st.num_total_docs) == (0, 1, 0, 313)
    assert st.num_entries_scanned_post_filter == 2            # count__* and max__ArrDelay of one doc
    b = seg.execute("SELECT COUNT(*) FROM t GROUP BY AirlineID")
    assert sum(v[0] for v in b.rows().values()) == 313 and len(b.rows()) == meta["columns"]["AirlineID"]["cardinality"]


@pytest.mark.parametrize("sql,uses_star", AIRLINE_QUERIES)
def test_oracle_star_tree_equals_plain_path(airline, sql, uses_star):
    seg, host, meta = airline
    star, plain = run_both(seg, sql)
    assert plain.stats.star_tree_index == -1
    assert star.stats.star_tree_index == (0 if uses_star else -1)
    assert star.rows() == plain.rows()
    assert star.stats.num_total_docs == plain.stats.num_total_docs == 313
    if uses_star:
        assert star.stats.num_docs_scanned <= meta["total_docs"]


# ---- star-trees built by our builder over synthetic docs: star-tree path == plain path in the oracle ---------------------
from tests.fixtures import SYNTH_STAR_QUERIES, synth_star_segment  # noqa: E402


@pytest.fixture(scope="module", params=[(64, ("h3",)), (10_000, ()), (1, ("h1", "h4"))],
                ids=["leaf64-skip-h3", "leaf10000", "leaf1-skip-h1-h4"])
def synth_star(request, oracle_api):
    host = synth_star_segment(40_000, max_leaf_records=request.param[0], skip=request.param[1])
    seg = NativeSegment(oracle_api, host)
    yield seg, host
    seg.destroy()


@pytest.mark.parametrize("sql,uses_star", SYNTH_STAR_QUERIES)
def test_oracle_synthetic_star_tree_equals_plain_path(synth_star, sql, uses_star):
    seg, host = synth_star
    star, plain = run_both(seg, sql)
    if "h1 = 99" in sql:
        uses_star = False            # the regular filter is empty: AggregationFunctionUtils#buildAggregationInfo skips the star-tree
    assert star.stats.star_tree_index == (0 if uses_star else -1)
    assert plain.stats.star_tree_index == -1
    assert star.rows() == plain.rows()
    if uses_star:
        assert star.stats.num_docs_scanned <= host.star_trees[0].num_docs


def test_star_tree_shrinks_the_scan(synth_star):
    seg, host = synth_star
    star, plain = run_both(seg, "SELECT h1, COUNT(*), SUM(m) FROM gpuBench GROUP BY h1")
    assert plain.stats.num_docs_scanned == 40_000
    assert star.stats.num_docs_scanned < plain.stats.num_docs_scanned


def test_star_tree_avg_and_min_max_range_pairs(oracle_api):
    """avg__m / minMaxRange__m are BYTES pairs (AvgValueAggregator.java:28-83: AvgPair sum + count; MinMaxRangePair min + max): the
    star-tree answer equals the plain scan's (INT metric: the sums are exact), with the star-tree's ExecutionStatistics."""
    from pinot_amd import capi
    from pinot_amd.query import parse_sql
    from tests.fixtures import STAR_PAIR_QUERIES, synth_star_pairs_segment
    host = synth_star_pairs_segment()
    o = NativeSegment(oracle_api, host)
    for sql, uses_star in STAR_PAIR_QUERIES:
        b = o.execute(sql)
        assert b.stats.star_tree_index == (0 if uses_star else -1), sql
        qc = parse_sql(sql)
        qc.flags |= capi.QUERY_FLAG_SKIP_STAR_TREE
        plain = o.execute(qc)
        assert plain.stats.star_tree_index == -1
        assert b.rows() == plain.rows(), sql
        if uses_star:
            assert b.stats.num_docs_scanned < plain.stats.num_docs_scanned
    o.destroy()
