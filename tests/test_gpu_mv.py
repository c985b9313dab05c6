"""Multi-value columns on the GPU (pg_kernels_mv.hip) against the oracle — itself pinned by the brute force of
tests/test_oracle_mv.py: filters over multi-value columns (scan with applyMV, inverted index), multi-value group keys (Cartesian
expansion), aggregateGroupByMV of the single-value functions and the *MV functions.  Results, group keys and ExecutionStatistics
are compared bit for bit; shapes the library leaves to the Java plan must be refused, not approximated."""
import numpy as np
import os

import pytest

from pinot_amd import capi
from pinot_amd.executor import NativeSegment
from tests import mv_fixture as mv

pytestmark = pytest.mark.gpu

STATS = ("num_docs_scanned", "num_entries_scanned_in_filter", "num_entries_scanned_post_filter", "num_total_docs")

MV_AGGS = "COUNTMV(mv1), SUMMV(mv1), MINMV(mv3), MAXMV(mv3), AVGMV(mv1), MINMAXRANGEMV(mv3), DISTINCTCOUNTMV(mv2), DISTINCTCOUNTHLLMV(mv1), COUNT(*)"

QUERIES = [
    # ---- filters over multi-value columns ---------------------------------------------------------------------------------
    "SELECT COUNT(*), SUM(m) FROM mvTable WHERE mv2 = 'cat'",
    "SELECT COUNT(*), SUM(m) FROM mvTable WHERE mv2 IN ('ant', 'lynx', 'zebra')",
    "SELECT COUNT(*), SUM(m) FROM mvTable WHERE mv2 != 'cat'",
    "SELECT COUNT(*), MAX(m) FROM mvTable WHERE mv2 NOT IN ('ant', 'bee', 'cat')",
    "SELECT COUNT(*), MIN(m) FROM mvTable WHERE mv1 BETWEEN 10 AND 19",
    "SELECT COUNT(*), SUM(m) FROM mvTable WHERE mv3 > 4000000",
    "SELECT COUNT(*) FROM mvTable WHERE mv1 = 7",
    "SELECT COUNT(*), SUM(m) FROM mvTable WHERE mv1 IN (1, 2, 3)",
    "SELECT COUNT(*), SUM(m) FROM mvTable WHERE mv1 NOT IN (1, 2, 3)",
    "SELECT COUNT(*), SUM(m) FROM mvTable WHERE s1 = 3 AND mv2 = 'dog'",
    "SELECT COUNT(*), SUM(m) FROM mvTable WHERE mv1 = 5 AND mv2 != 'dog'",
    "SELECT COUNT(*), SUM(m) FROM mvTable WHERE mv2 = 'eel' AND m < 0 AND s1 IN (1, 2)",
    "SELECT COUNT(*), SUM(m) FROM mvTable WHERE mv1 = -2147483648",
    "SELECT COUNT(*), SUM(m) FROM mvTable WHERE mv2 = 'cat' OR mv1 BETWEEN 3 AND 4",          # a drained OR of scans: every entry of both columns
    "SELECT COUNT(*) FROM mvTable WHERE mv1 IN (1, 2) AND mv1 BETWEEN 20 AND 39 AND mv2 IN ('ant', 'bee')",   # index, then two multi-value scans
    # shapes whose numEntriesScannedInFilter depends on how the iterators drive each other (pg_filter_stats.cpp, over ENTRIES here)
    "SELECT COUNT(*), SUM(m) FROM mvTable WHERE NOT (mv2 = 'cat')",
    "SELECT COUNT(*), SUM(m) FROM mvTable WHERE mv2 = 'eel' AND m < 0",                            # leapfrog of a single-value and a multi-value scan
    "SELECT COUNT(*), SUM(m) FROM mvTable WHERE mv2 IN ('ant', 'bee') AND mv3 > 3000000 AND mv1 BETWEEN 5 AND 30",
    "SELECT s1, COUNT(*) FROM mvTable WHERE s1 = 2 AND (mv2 = 'cat' OR m > 500) GROUP BY s1 LIMIT 10",     # an OR of scans under an AND
    "SELECT COUNT(*) FROM mvTable WHERE NOT (mv1 BETWEEN 3 AND 9 OR mv2 != 'gnu')",
    # ---- multi-value group keys ----------------------------------------------------------------------------------------------
    "SELECT mv1, COUNT(*), SUM(m), MAX(m) FROM mvTable GROUP BY mv1 LIMIT 1000",
    "SELECT s1, mv2, COUNT(*), MIN(m) FROM mvTable WHERE s1 IN (0, 1, 2, 3) GROUP BY s1, mv2 LIMIT 1000",
    "SELECT mv2, s2, AVG(m), MINMAXRANGE(m) FROM mvTable GROUP BY mv2, s2 LIMIT 1000",
    "SELECT mv1, mv2, COUNT(*), SUM(m) FROM mvTable WHERE s1 IN (0, 1, 2, 3) GROUP BY mv1, mv2 LIMIT 10000",
    "SELECT mv3, s1, mv1, COUNT(*), DISTINCTCOUNT(s2) FROM mvTable WHERE mv2 = 'fox' GROUP BY mv3, s1, mv1 LIMIT 100000",
    "SELECT mv1, DISTINCTCOUNTHLL(m), DISTINCTCOUNTHLL(s1) FROM mvTable WHERE mv1 < 12 GROUP BY mv1 LIMIT 1000",
    # three and four multi-value keys (MultiValueRawQueriesTest.java:455-540 groups by three)
    "SELECT mv1, mv2, mv3, COUNT(*), SUM(m) FROM mvTable WHERE s1 < 3 GROUP BY mv1, mv2, mv3 LIMIT 1000000",
    "SELECT mv3, s1, mv2, mv1, COUNT(*), COUNTMV(mv3) FROM mvTable WHERE s1 IN (1, 4) GROUP BY mv3, s1, mv2, mv1 LIMIT 1000000",
    # ---- the *MV functions: no GROUP BY, single-value keys, multi-value keys ------------------------------------------------------
    f"SELECT {MV_AGGS} FROM mvTable WHERE s1 IN (1, 2, 3)",
    f"SELECT s1, {MV_AGGS} FROM mvTable WHERE mv1 NOT IN (3, 4) GROUP BY s1 LIMIT 100",
    f"SELECT mv2, {MV_AGGS} FROM mvTable GROUP BY mv2 LIMIT 100",
    f"SELECT mv1, s2, {MV_AGGS} FROM mvTable WHERE s1 = 2 GROUP BY mv1, s2 LIMIT 10000",
    "SELECT s2, SUMMV(mv3), AVGMV(mv3), COUNTMV(mv3) FROM mvTable GROUP BY s2 LIMIT 100",         # LONG entries
    # ---- key spaces beyond one LDS table: the dense HBM table (pg_mv_query_g), DISTINCTCOUNT states in HBM ------------------------------
    "SELECT mvh, s1, COUNT(*), SUM(m), MAXMV(mv1) FROM mvTable GROUP BY mvh, s1 LIMIT 1000000",
    "SELECT mvh, mv2, COUNT(*), COUNTMV(mv3) FROM mvTable WHERE s1 IN (1, 2, 3, 4) GROUP BY mvh, mv2 LIMIT 1000000",
    "SELECT mv1, s1, DISTINCTCOUNTMV(mvh), DISTINCTCOUNT(s2) FROM mvTable GROUP BY mv1, s1 LIMIT 100000",
    "SELECT DISTINCTCOUNTMV(mvh), DISTINCTCOUNTHLLMV(mvh), SUMMV(mvh) FROM mvTable WHERE mv2 != 'ant'",
    # ---- answered from the dictionaries (NonScanBasedAggregationOperator) -------------------------------------------------------
    "SELECT MINMV(mv3), MAXMV(mv3), MINMAXRANGEMV(mv1), DISTINCTCOUNTMV(mv2), COUNT(*) FROM mvTable",
]


@pytest.fixture(scope="module", params=[1, 300, 2049, 50_000])
def pair(request, gpu_api, oracle_api):
    host = mv.build(mv.make_rows(request.param, seed=request.param))
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    yield g, o
    g.destroy()
    o.destroy()


@pytest.mark.parametrize("sql", QUERIES)
def test_multi_value_queries_match_oracle(pair, sql):
    g, o = pair
    gb, ob = g.execute(sql), o.execute(sql)
    assert gb.rows() == ob.rows()
    for f in STATS:
        assert getattr(gb.stats, f) == getattr(ob.stats, f), f


# leapfrogged shapes over multi-value scans: counted in tiles on the device (entries, not docs: pg_filter_stats_tiles.h), the others by the host walk
MV_STATS_PATHS = [
    ("SELECT COUNT(*), SUM(m) FROM mvTable WHERE mv2 = 'eel' AND m < 0", 2),                                   # single-value AND multi-value scan
    ("SELECT COUNT(*), SUM(m) FROM mvTable WHERE mv2 IN ('ant', 'bee') AND mv3 > 3000000 AND mv1 BETWEEN 5 AND 30", 2),
    ("SELECT s1, COUNT(*) FROM mvTable WHERE s1 = 2 AND (mv2 = 'cat' OR m > 500) GROUP BY s1 LIMIT 10", 2),      # OR of scans under an AND
    ("SELECT COUNT(*) FROM mvTable WHERE (mv1 BETWEEN 3 AND 9 AND mv3 > 2000000) OR mv2 = 'gnu'", 2),           # drained OR over an AND and a scan
    ("SELECT COUNT(*) FROM mvTable WHERE m < 0 AND NOT (mv2 = 'cat')", 2),                                     # NOT over a multi-value scan under an AND: next() without batches
    ("SELECT COUNT(*) FROM mvTable WHERE s1 IN (1, 2) AND (m > 900 OR NOT (mv1 BETWEEN 3 AND 30))", 2),        # ... inside an OR
    ("SELECT COUNT(*) FROM mvTable WHERE m < 0 AND NOT (mv2 = 'cat' OR m > 5)", 2),                            # NOT over an OR of leaves under an AND (a multi-value scan among them)
    ("SELECT COUNT(*) FROM mvTable WHERE m < 0 AND NOT (mv2 = 'cat' AND m > 5)", 1),                           # NOT over an AND under an AND: the host walk
]


@pytest.mark.parametrize("sql,path", MV_STATS_PATHS)
def test_entries_scanned_in_filter_over_multi_value_scans(pair, gpu_knobs, sql, path):
    g, o = pair
    gb, ob = g.execute(sql), o.execute(sql)
    assert gb.rows() == ob.rows()
    assert gb.stats.stats_exact == 1
    assert gb.stats.num_entries_scanned_in_filter == ob.stats.num_entries_scanned_in_filter
    if g.host.total_docs >= 2049:
        assert gb.stats.filter_stats_path == path
    gpu_knobs(PG_FILTER_STATS_HOST=1)
    hb = g.execute(sql)
    assert hb.stats.num_entries_scanned_in_filter == ob.stats.num_entries_scanned_in_filter and hb.stats.filter_stats_path <= 1


def test_segment_trim_ordered_by_multi_value_functions(pair):
    """ORDER BY a *MV function with minSegmentGroupTrimSize: the final results of the single-value functions they extend order the groups
    (TableResizer.java:406-445); the order ends in the group keys — a total order — so the survivors are the oracle's"""
    from pinot_amd.query import parse_sql
    from tests.trim_model import assert_valid_trim
    g, o = pair
    for sql in ("SELECT mv1, s2, COUNTMV(mv3), SUMMV(mv1) FROM mvTable GROUP BY mv1, s2 ORDER BY COUNTMV(mv3) DESC, mv1, s2 LIMIT 3",
                "SELECT s1, s2, MAXMV(mv1), AVGMV(mv3), DISTINCTCOUNTMV(mv2) FROM mvTable GROUP BY s1, s2 ORDER BY DISTINCTCOUNTMV(mv2), AVGMV(mv3) DESC, s1, s2 LIMIT 2"):
        full = g.execute(parse_sql(sql)).rows()
        qc, qo = parse_sql(sql), parse_sql(sql)
        qc.min_segment_group_trim_size = qo.min_segment_group_trim_size = 1
        gb, ob = g.execute(qc), o.execute(qo)
        assert_valid_trim(qc, full, gb.rows())
        assert gb.rows() == ob.rows()


def test_the_multi_value_kernels_run_them(pair):
    g, _ = pair
    if g.total_docs < 2049:
        pytest.skip("tiny segments: literals missing from the dictionaries turn leaves into Empty / MatchAll")
    for sql, kernel in (("SELECT COUNT(*), SUM(m) FROM mvTable WHERE mv2 = 'cat'", "pg_mv_query_l"),
                        ("SELECT mvh, s1, COUNT(*), SUM(m) FROM mvTable GROUP BY mvh, s1 LIMIT 1000000", "pg_mv_query_g" if g.total_docs >= 50_000 else None),
                        ("SELECT mv1, COUNT(*) FROM mvTable GROUP BY mv1 LIMIT 1000", "pg_mv_query_l" if os.environ.get("PG_NO_MVG") else "pg_mv_group_4"),
                        ("SELECT s1, SUMMV(mv1) FROM mvTable WHERE s1 < 3 GROUP BY s1 LIMIT 1000", "pg_mv_query_l"),
                        ("SELECT COUNT(*), SUM(m) FROM mvTable WHERE mv1 IN (1, 2, 3)", None)):   # inverted index only: nothing multi-value is read
        k = g.execute(sql).stats.kernel.decode()
        if kernel is None and "mvh" in sql:
            continue
        assert (k == kernel) if kernel else not k.startswith("pg_mv_"), (sql, k)


def test_filter_only_api_over_a_multi_value_column(pair):
    g, o = pair
    for where in ("mv2 = 'cat'", "mv1 NOT IN (1, 2, 3) AND mv2 != 'dog'", "s1 = 1 AND mv3 > 2000000"):
        f = f"SELECT COUNT(*) FROM mvTable WHERE {where}"
        gs, os_ = g.filter(f), o.filter(f)
        assert np.array_equal(gs.doc_ids(), os_.doc_ids()), where
        assert gs.stats().num_entries_scanned_in_filter == os_.stats().num_entries_scanned_in_filter, where
        gs.free()
        os_.free()


def test_more_groups_than_num_groups_limit_is_left_to_the_java_plan(pair):
    g, o = pair
    if g.total_docs < 50_000:
        pytest.skip("needs more groups than the limit")
    from pinot_amd.query import parse_sql
    q = parse_sql("SELECT mvh, s1, COUNT(*) FROM mvTable GROUP BY mvh, s1 LIMIT 1000000")
    q.num_groups_limit = 5000
    with pytest.raises(capi.NativeError):
        g.execute(q)
    q = parse_sql("SELECT mvh, s1, COUNT(*) FROM mvTable GROUP BY mvh, s1 LIMIT 1000000")
    q.num_groups_limit = 100_000         # the key space (210 000) may exceed the limit as long as the groups found (80 000) do not
    assert g.execute(q).rows() == o.execute(q).rows()


UNSUPPORTED = [
    "SELECT s1, SUM(mv1) FROM mvTable GROUP BY s1 LIMIT 10",                               # single-value function over a multi-value column
    "SELECT s1, SUMMV(m) FROM mvTable GROUP BY s1 LIMIT 10",
]


@pytest.mark.parametrize("sql", UNSUPPORTED)
def test_shapes_left_to_the_java_plan_are_refused(pair, sql):
    g, _ = pair
    if g.total_docs < 300:
        pytest.skip("tiny segment")
    with pytest.raises(capi.NativeError):
        g.execute(sql)


def test_merge_of_multi_value_results(gpu_api, oracle_api):
    """Two segments of one table, each with its own dictionaries: merged by VALUE (GroupByCombineOperator over the intermediate
    results, the *MV functions merging like their single-value forms) and compared with the oracle over the whole table."""
    rows = mv.make_rows(6000, seed=11)
    whole = mv.build(rows)
    from pinot_amd.executor import GroupByCombineOperator
    a, b = mv.build(rows[:3000], "a"), mv.build(rows[3000:], "b")
    sql = "SELECT mv2, COUNT(*), SUMMV(mv1), DISTINCTCOUNTMV(mv3), MAXMV(mv3) FROM mvTable WHERE mv1 NOT IN (0, 1) GROUP BY mv2 LIMIT 100"
    blocks = []
    for h in (a, b):
        s = NativeSegment(gpu_api, h)
        blocks.append(s.execute(sql))
        s.destroy()
    w = NativeSegment(oracle_api, whole)
    want = GroupByCombineOperator([w.execute(sql)]).final()
    w.destroy()
    assert GroupByCombineOperator(blocks).final() == want


# ---- raw (no-dictionary) multi-value columns: FixedByteChunkMVForwardIndexReader (round 4) ------------------------------------------------
# (query over the raw twin, the same query over the dictionary column): equal rows — the criterion of MultiValueRawQueriesTest — and
# GPU == oracle on the raw query itself, ExecutionStatistics included.  Empty rows hold the default null value (-2^31 / -inf).
RAW_PAIRS = [
    ("SELECT COUNT(*), SUM(m) FROM mvTable WHERE r1 BETWEEN 10 AND 19", "SELECT COUNT(*), SUM(m) FROM mvTable WHERE mv1 BETWEEN 10 AND 19"),
    ("SELECT COUNT(*), SUM(m) FROM mvTable WHERE r3 > 4000000 AND r1 NOT IN (1, 2, 3)", "SELECT COUNT(*), SUM(m) FROM mvTable WHERE mv3 > 4000000 AND mv1 NOT IN (1, 2, 3)"),
    ("SELECT COUNT(*) FROM mvTable WHERE r1 = -2147483648 OR rd < 1.5", "SELECT COUNT(*) FROM mvTable WHERE mv1 = -2147483648 OR fd < 1.5"),
    ("SELECT r1, COUNT(*), SUM(m), MAX(m) FROM mvTable GROUP BY r1 LIMIT 1000", "SELECT mv1, COUNT(*), SUM(m), MAX(m) FROM mvTable GROUP BY mv1 LIMIT 1000"),
    ("SELECT s1, r3, COUNT(*), MIN(m) FROM mvTable WHERE s1 IN (0, 1, 2, 3) GROUP BY s1, r3 LIMIT 1000", "SELECT s1, mv3, COUNT(*), MIN(m) FROM mvTable WHERE s1 IN (0, 1, 2, 3) GROUP BY s1, mv3 LIMIT 1000"),
    ("SELECT r1, mv2, COUNT(*), SUM(m) FROM mvTable WHERE s1 < 4 GROUP BY r1, mv2 LIMIT 10000", "SELECT mv1, mv2, COUNT(*), SUM(m) FROM mvTable WHERE s1 < 4 GROUP BY mv1, mv2 LIMIT 10000"),
    ("SELECT rd, s2, COUNT(*) FROM mvTable GROUP BY rd, s2 LIMIT 10000", "SELECT fd, s2, COUNT(*) FROM mvTable GROUP BY fd, s2 LIMIT 10000"),
    ("SELECT s1, COUNTMV(r1), SUMMV(r1), MINMV(r3), MAXMV(r3), AVGMV(r1), MINMAXRANGEMV(r3), DISTINCTCOUNTHLLMV(r1), COUNT(*) FROM mvTable WHERE r1 NOT IN (3, 4) GROUP BY s1 LIMIT 100",
     "SELECT s1, COUNTMV(mv1), SUMMV(mv1), MINMV(mv3), MAXMV(mv3), AVGMV(mv1), MINMAXRANGEMV(mv3), DISTINCTCOUNTHLLMV(mv1), COUNT(*) FROM mvTable WHERE mv1 NOT IN (3, 4) GROUP BY s1 LIMIT 100"),
    ("SELECT COUNTMV(rf), MINMV(rf), MAXMV(rd), MINMAXRANGEMV(rd) FROM mvTable WHERE s1 > 2", "SELECT COUNTMV(fd), MINMV(fd), MAXMV(fd), MINMAXRANGEMV(fd) FROM mvTable WHERE s1 > 2"),
    ("SELECT rh, s1, COUNT(*), SUM(m), MAXMV(r1) FROM mvTable GROUP BY rh, s1 LIMIT 1000000", "SELECT mvh, s1, COUNT(*), SUM(m), MAXMV(mv1) FROM mvTable GROUP BY mvh, s1 LIMIT 1000000"),
    ("SELECT SUMMV(rh), DISTINCTCOUNTHLLMV(rh) FROM mvTable WHERE mv2 != 'ant'", "SELECT SUMMV(mvh), DISTINCTCOUNTHLLMV(mvh) FROM mvTable WHERE mv2 != 'ant'"),
    # raw multi-value STRING (VarByteChunkMVForwardIndexReader)
    ("SELECT COUNT(*), SUM(m) FROM mvTable WHERE rs = 'cat'", "SELECT COUNT(*), SUM(m) FROM mvTable WHERE mv2 = 'cat'"),
    ("SELECT COUNT(*), SUM(m) FROM mvTable WHERE rs IN ('ant', 'lynx', 'zebra') AND r1 < 30", "SELECT COUNT(*), SUM(m) FROM mvTable WHERE mv2 IN ('ant', 'lynx', 'zebra') AND mv1 < 30"),
    ("SELECT COUNT(*), MAX(m) FROM mvTable WHERE rs NOT IN ('ant', 'bee', 'cat')", "SELECT COUNT(*), MAX(m) FROM mvTable WHERE mv2 NOT IN ('ant', 'bee', 'cat')"),
    ("SELECT rs, COUNT(*), SUM(m) FROM mvTable GROUP BY rs LIMIT 100", "SELECT mv2, COUNT(*), SUM(m) FROM mvTable GROUP BY mv2 LIMIT 100"),
    ("SELECT s1, rs, COUNT(*), MIN(m) FROM mvTable WHERE s1 IN (0, 1, 2, 3) GROUP BY s1, rs LIMIT 1000", "SELECT s1, mv2, COUNT(*), MIN(m) FROM mvTable WHERE s1 IN (0, 1, 2, 3) GROUP BY s1, mv2 LIMIT 1000"),
    ("SELECT rs, r1, COUNT(*) FROM mvTable WHERE rs != 'dog' GROUP BY rs, r1 LIMIT 10000", "SELECT mv2, mv1, COUNT(*) FROM mvTable WHERE mv2 != 'dog' GROUP BY mv2, mv1 LIMIT 10000"),
    ("SELECT s2, COUNTMV(rs), COUNT(*) FROM mvTable GROUP BY s2 LIMIT 100", "SELECT s2, COUNTMV(mv2), COUNT(*) FROM mvTable GROUP BY s2 LIMIT 100"),
    # four multi-value keys, all raw: INT, STRING, LONG, DOUBLE
    ("SELECT r1, rs, r3, rd, COUNT(*), SUM(m) FROM mvTable WHERE s1 = 2 GROUP BY r1, rs, r3, rd LIMIT 1000000",
     "SELECT mv1, mv2, mv3, fd, COUNT(*), SUM(m) FROM mvTable WHERE s1 = 2 GROUP BY mv1, mv2, mv3, fd LIMIT 1000000"),
]


@pytest.fixture(scope="module", params=[1, 300, 2049, 30_000])
def raw_pair(request, gpu_api, oracle_api):
    host = mv.build_with_raw_twins(mv.make_rows(request.param, seed=7 + request.param))
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    yield g, o
    g.destroy()
    o.destroy()


@pytest.mark.parametrize("raw_sql,dict_sql", RAW_PAIRS)
def test_raw_multi_value_columns(raw_pair, raw_sql, dict_sql):
    g, o = raw_pair
    gb, ob = g.execute(raw_sql), o.execute(raw_sql)
    assert gb.rows() == ob.rows()
    for f in STATS:
        assert getattr(gb.stats, f) == getattr(ob.stats, f), f
    assert gb.rows() == g.execute(dict_sql).rows()          # raw == dictionary twin, on the GPU ...
    assert ob.rows() == o.execute(dict_sql).rows()          # ... and in the oracle


FLOATING_MV_SUMS = [   # SUMMV / AVGMV over FLOAT / DOUBLE entries: fixed-point digit accumulators, every entry cut into its digits
    ("SELECT s1, SUMMV(rd), AVGMV(rf), COUNT(*) FROM mvTable WHERE r1 NOT IN (3, 4) GROUP BY s1 LIMIT 100",
     "SELECT s1, SUMMV(fd), AVGMV(fd), COUNT(*) FROM mvTable WHERE mv1 NOT IN (3, 4) GROUP BY s1 LIMIT 100"),
    ("SELECT SUMMV(rf), AVGMV(rd), SUM(m) FROM mvTable WHERE s1 > 2", "SELECT SUMMV(fd), AVGMV(fd), SUM(m) FROM mvTable WHERE s1 > 2"),
    ("SELECT rs, SUMMV(rd), MAXMV(rd) FROM mvTable GROUP BY rs LIMIT 100", "SELECT mv2, SUMMV(fd), MAXMV(fd) FROM mvTable GROUP BY mv2 LIMIT 100"),
    ("SELECT mv1, mv2, SUMMV(fd), AVGMV(fd) FROM mvTable WHERE s1 < 4 GROUP BY mv1, mv2 LIMIT 10000", None),
]


@pytest.mark.parametrize("n", [300, 30_000])
def test_floating_summv(gpu_api, oracle_api, n):
    """No empty rows here: an empty FLOAT / DOUBLE row holds the default null value -inf, and a column with NaN / Inf keeps the reference's
    IEEE additions — SUMMV over it is left to the Java plan (asserted below on the fixture that has them)."""
    host = mv.build_with_raw_twins(mv.make_rows(n, seed=11 + n, empty_rows=False))
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    for raw_sql, dict_sql in FLOATING_MV_SUMS:
        gb, ob = g.execute(raw_sql), o.execute(raw_sql)
        assert gb.rows() == ob.rows(), raw_sql
        for f in STATS:
            assert getattr(gb.stats, f) == getattr(ob.stats, f), f
        if dict_sql:
            assert gb.rows() == g.execute(dict_sql).rows()
    g.destroy()
    o.destroy()


def test_floating_summv_over_non_finite_entries_is_left_to_the_java_plan(raw_pair):
    g, _ = raw_pair
    if g.total_docs < 300:
        pytest.skip("the one-doc table has no empty row")
    with pytest.raises(capi.NativeError) as e:
        g.execute("SELECT SUMMV(rd) FROM mvTable WHERE s1 = 1")
    assert e.value.status == capi.PG_ERR_UNSUPPORTED and "NaN / Inf" in str(e.value)


def test_raw_multi_value_without_filter_is_scanned_not_answered_from_a_dictionary(raw_pair):
    """MINMV / MAXMV over a dictionary column without a filter come from the dictionary (NonScanBasedAggregationOperator: no entry is
    read); the raw twin has no dictionary: the same values, a scan's statistics."""
    g, o = raw_pair
    raw, dic = "SELECT MINMV(r3), MAXMV(r3), COUNT(*) FROM mvTable", "SELECT MINMV(mv3), MAXMV(mv3), COUNT(*) FROM mvTable"
    gb, ob = g.execute(raw), o.execute(raw)
    assert gb.rows() == ob.rows() == g.execute(dic).rows()
    for f in STATS:
        assert getattr(gb.stats, f) == getattr(ob.stats, f), f
    assert gb.stats.num_entries_scanned_post_filter > 0 and g.execute(dic).stats.num_entries_scanned_post_filter == 0


def test_distinctcountmv_over_a_raw_column_is_left_to_the_java_plan(raw_pair):
    g, o = raw_pair
    for api_seg in (g, o):
        with pytest.raises(capi.NativeError) as e:
            api_seg.execute("SELECT DISTINCTCOUNTMV(r1) FROM mvTable WHERE s1 = 1")
        assert e.value.status == capi.PG_ERR_UNSUPPORTED


def test_the_doc_by_doc_walk_still_agrees(gpu_api, oracle_api, gpu_knobs):
    """PG_MV_NO_WINDOWS: the entry-parallel scan leaves are off — the doc-by-doc walk answers every query alike (it stays the path of partial
    tiles and of runs beyond 8 192 entries)."""
    gpu_knobs(PG_MV_NO_WINDOWS="1")
    host = mv.build(mv.make_rows(20_000, seed=77))
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    for sql in QUERIES:
        assert g.execute(sql).rows() == o.execute(sql).rows(), sql
    g.destroy()
    o.destroy()


def test_entries_beyond_the_dictionary_are_refused_at_registration(gpu_api):
    """A multi-value forward index whose width leaves room above the cardinality (6 bits, 41 values) can hold dictIds the dictionary does not
    have — a corrupt or mismatched file.  The entries index dictionaries, look-up tables and LDS group tables exactly as a single-value
    column's docs do, so they are checked the same way, once, at registration (pg_segment.cpp, check_dict_ids)."""
    import copy
    import ctypes as C
    from pinot_amd import capi
    from pinot_amd.segment import HostSegment
    host = mv.build(mv.make_rows(3000, seed=31))
    good = host.columns["mv1"]
    assert good.cardinality == 41 and good.bits_per_value == 6   # 40 values + the default null value of empty docs
    seg = NativeSegment(gpu_api, HostSegment("mv_ids", host.total_docs))
    bad = copy.copy(good)
    fi = np.array(good.forward_index, dtype=np.uint8, copy=True)
    fi[-10:-8] = 0xFF                          # two bytes of ones inside the packed entries: one 6-bit entry becomes 63, the dictionary has 41 values
    bad.forward_index = fi
    d = bad.desc()
    status = gpu_api.f("segment_add_column")(seg.handle, C.byref(d))
    assert status == capi.PG_ERR_INVALID_ARGUMENT and "dictId" in gpu_api.last_error(), (status, gpu_api.last_error())
    low = copy.copy(good)
    low.cardinality = 17                       # metadata that disagrees with the entries (ids up to 40) — and with the dictionary buffer
    d = low.desc()
    status = gpu_api.f("segment_add_column")(seg.handle, C.byref(d))
    assert status == capi.PG_ERR_INVALID_ARGUMENT, (status, gpu_api.last_error())
    d = good.desc()
    assert gpu_api.f("segment_add_column")(seg.handle, C.byref(d)) == capi.PG_OK
    seg.destroy()
