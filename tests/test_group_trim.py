"""Segment-level group trim of the oracle (oracle/po_query.c) against the model of tests/trim_model.py, and the order-by plumbing of the
query structs (pg_order_by / limit / min_segment_group_trim_size, ABI 4).  Reference: GroupByOperator.java:120-133."""
import numpy as np
import pytest

from pinot_amd import capi, synth
from pinot_amd.executor import NativeSegment
from pinot_amd.query import CQuery, parse_sql
from pinot_amd.segment import build_segment
from tests.trim_model import assert_valid_trim, trim_size

QUERIES = [
    "SELECT g1, g2, COUNT(*), SUM(m) FROM gpuBench GROUP BY g1, g2 ORDER BY SUM(m) DESC LIMIT 7",
    "SELECT g1, g2, COUNT(*), SUM(m) FROM gpuBench GROUP BY g1, g2 ORDER BY COUNT(*), g2 DESC, g1 LIMIT 10",
    "SELECT u, COUNT(*), MAX(m), AVG(m) FROM gpuBench WHERE g1 < 50 GROUP BY u ORDER BY AVG(m) DESC, u LIMIT 100",
    "SELECT u, MINMAXRANGE(m), MIN(m) FROM gpuBench GROUP BY u ORDER BY MINMAXRANGE(m) DESC, MIN(m), u DESC LIMIT 20",
    "SELECT g2, u, COUNT(*) FROM gpuBench GROUP BY g2, u ORDER BY u DESC, g2 LIMIT 50",
    # ordered by a distinct count's final value: the set's size, HyperLogLog#cardinality (TableResizer.java:406-445 -> extractFinalResult)
    "SELECT g1, g2, DISTINCTCOUNT(u), COUNT(*) FROM gpuBench GROUP BY g1, g2 ORDER BY DISTINCTCOUNT(u) DESC, g1, g2 LIMIT 9",
    "SELECT g1, g2, DISTINCTCOUNTHLL(u), SUM(m) FROM gpuBench GROUP BY g1, g2 ORDER BY DISTINCTCOUNTHLL(u), g2 DESC, g1 LIMIT 12",
]


@pytest.fixture(scope="module")
def seg(oracle_api):
    host = synth.generate_segment(150_001, segment_index=9, columns=["g1", "g2", "m", "u"], native=False)
    s = NativeSegment(oracle_api, host)
    yield s
    s.destroy()


@pytest.mark.parametrize("sql", QUERIES)
@pytest.mark.parametrize("min_trim", [1, 333])
def test_oracle_trim_is_a_valid_trim(seg, sql, min_trim):
    full = seg.execute(parse_sql(sql)).rows()          # minSegmentGroupTrimSize = -1 (the reference's default): never trimmed
    qc = parse_sql(sql)
    qc.min_segment_group_trim_size = min_trim
    block = seg.execute(qc)
    assert len(full) > trim_size(qc)
    assert_valid_trim(qc, full, block.rows())
    assert block.stats.num_docs_scanned == seg.execute(parse_sql(sql)).stats.num_docs_scanned


def test_no_trim_without_order_by_or_below_the_trim_size(seg):
    sql = "SELECT g1, g2, COUNT(*) FROM gpuBench GROUP BY g1, g2 LIMIT 3"
    qc = parse_sql(sql)
    qc.min_segment_group_trim_size = 5     # no ORDER BY: "the groups are not trimmed if there is no ordering specified" (:115-116)
    assert len(seg.execute(qc).rows()) == 5000
    qc = parse_sql(sql.replace(" LIMIT", " ORDER BY g1 LIMIT"))
    qc.min_segment_group_trim_size = 5000  # 5 000 groups are not MORE than trimSize
    assert len(seg.execute(qc).rows()) == 5000
    qc.min_segment_group_trim_size = 4999
    assert len(seg.execute(qc).rows()) == 4999


def test_raw_group_keys_order_by_value(oracle_api):
    rng = np.random.default_rng(2)
    n = 20_000
    data = {"k": rng.integers(-10**9, 10**9, n).astype(np.int64), "s": np.array([f"k{v:05d}" for v in rng.integers(0, 3000, n)], dtype=object),
            "v": rng.integers(0, 1000, n).astype(np.int32)}
    host = build_segment("rawk", data, {"k": "LONG", "s": "STRING", "v": "INT"}, no_dictionary_columns=["k", "s", "v"])
    s = NativeSegment(oracle_api, host)
    for sql in ("SELECT k, SUM(v) FROM rawk GROUP BY k ORDER BY k LIMIT 4",
                "SELECT s, COUNT(*), SUM(v) FROM rawk GROUP BY s ORDER BY SUM(v) DESC, s LIMIT 4"):
        full = s.execute(parse_sql(sql)).rows()
        qc = parse_sql(sql)
        qc.min_segment_group_trim_size = 1
        got = s.execute(qc).rows()
        assert len(got) == 20
        if "ORDER BY k" in sql:
            assert sorted(got) == sorted(full)[:20]
        else:
            want = sorted(full.items(), key=lambda kv: (-kv[1][1], kv[0]))[:20]
            assert sorted(got.items()) == sorted(want)
    s.destroy()


def test_raw_string_keys_order_as_java_strings(oracle_api):
    """String.compareTo orders UTF-16 code units (TableResizer's comparators): a supplementary character (surrogates D800..DFFF) sorts BEFORE
    U+E000..U+FFFF although its UTF-8 bytes (F0..) sort after theirs (EE / EF).  The trim under ORDER BY a raw STRING key keeps the first
    trimSize keys of THAT order."""
    keys = ["a", "a\ue000", "a\U0001f600", "a\uffff", "a\U00010000", "b", "a\ud7ff", "\uff5e", "\U0002f800", "zz"] + [f"k{i:03d}" for i in range(40)]
    rng = np.random.default_rng(5)
    n = 4000
    data = {"s": np.array([keys[i] for i in rng.integers(0, len(keys), n)], dtype=object), "v": rng.integers(0, 100, n).astype(np.int32)}
    host = build_segment("rawu", data, {"s": "STRING", "v": "INT"}, no_dictionary_columns=["s", "v"])
    s = NativeSegment(oracle_api, host)
    for sql, reverse in (("SELECT s, COUNT(*) FROM rawu GROUP BY s ORDER BY s DESC LIMIT 2", True), ("SELECT s, COUNT(*) FROM rawu GROUP BY s ORDER BY s LIMIT 9", False)):
        full = s.execute(parse_sql(sql)).rows()
        assert len(full) == len(keys)
        qc = parse_sql(sql)
        qc.min_segment_group_trim_size = 1
        got = s.execute(qc).rows()
        java_order = sorted((k[0] for k in full), key=lambda x: x.encode("utf-16-be"), reverse=reverse)   # UTF-16 big-endian bytes order as code units do
        assert sorted(k[0] for k in got) == sorted(java_order[:trim_size(qc)])
    s.destroy()


def test_order_by_resolution_and_c_structs():
    qc = parse_sql("SELECT a, b, COUNT(*), SUM(x) FROM t GROUP BY a, b ORDER BY SUM(x) DESC, b, count(*) LIMIT 12")
    assert qc.resolved_order_by() == [(capi.ORDER_BY_AGGREGATION, 1, False), (capi.ORDER_BY_GROUP_KEY, 1, True), (capi.ORDER_BY_AGGREGATION, 0, True)]
    qc.min_segment_group_trim_size = 77
    c = CQuery(qc).query
    assert (c.n_order_by, c.limit, c.min_segment_group_trim_size) == (3, 12, 77)
    assert [(c.order_by[i].kind, c.order_by[i].index, c.order_by[i].ascending) for i in range(3)] == [(1, 1, 0), (0, 1, 1), (1, 0, 1)]
    # an expression the ABI does not carry (post-aggregation): no ORDER BY travels, the segment is not trimmed
    qc = parse_sql("SELECT a, SUM(x), SUM(y) FROM t GROUP BY a ORDER BY a LIMIT 5")
    qc.order_by = [("SUM(x)+SUM(y)", True)]
    assert qc.resolved_order_by() is None and CQuery(qc).query.n_order_by == 0


def test_trim_under_null_handling_over_columns_without_nulls(seg, oracle_api):
    """no order-by value can be null there: the same trim (the null-aware comparator has tests/test_null_handling_trim.py)"""
    sql = "SELECT g1, COUNT(*) FROM gpuBench GROUP BY g1 ORDER BY COUNT(*), g1 LIMIT 1"
    qc, qn = parse_sql(sql), parse_sql(sql)
    qc.min_segment_group_trim_size = qn.min_segment_group_trim_size = 1
    qn.flags |= capi.QUERY_FLAG_NULL_HANDLING
    rows = seg.execute(qn).rows()
    assert len(rows) == 5 and rows == seg.execute(qc).rows()
