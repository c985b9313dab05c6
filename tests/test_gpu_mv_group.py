"""The two commonest multi-value shapes through kernels of their own (pg_kernels_mvg.hip).  GROUP BY one multi-value column: `SELECT mv, COUNT(*), SUM(m) … GROUP BY mv` —
every entry of the doc's multi-value column is a key of the doc, repeated entries repeat the key
(DictionaryBasedGroupKeyGenerator.java:357-368, 504-573).  pg_mv_group_4 requests four entries per doc up front (columns of at most four entries
per doc), pg_mv_group_8 eight and walks what is left one by one (mvC: up to 11 entries).  Against the oracle at sizes with fewer tiles than
workgroups, a ragged last tile and several tiles per wavefront; the interpreter-frame kernel (PG_NO_MVG) must return the same rows.
pg_mv_aggr_*: the *MV functions over one multi-value INT column grouped by single-value columns — all entries of a doc go into the doc's key
(SumMVAggregationFunction.java, CountMVAggregationFunction.java:62-96, MinMV / MaxMV / AvgMV / MinMaxRangeMV)."""
import os

import numpy as np
import pytest

from pinot_amd import capi
from pinot_amd.executor import NativeSegment
from pinot_amd.segment import HostSegment, build_column, build_mv_column

pytestmark = pytest.mark.gpu
STATS = ("num_docs_scanned", "num_entries_scanned_in_filter", "num_entries_scanned_post_filter", "num_total_docs")


def table(n, seed):
    rng = np.random.default_rng(seed)

    def mv(card, lo, hi, empty=False):
        lens = rng.integers(lo, hi + 1, n)
        if empty:
            lens[rng.random(n) < 0.05] = 0     # the segment creator stores the default null value for an empty row
        flat = rng.integers(0, card, int(lens.sum()))
        out, at = [], 0
        for k in lens.tolist():
            out.append(flat[at:at + k].tolist())
            at += k
        return out
    seg = HostSegment("mvg", n)
    seg.columns["mvA"] = build_mv_column("mvA", mv(20, 1, 3), "INT")               # <= 4 entries: pg_mv_group_4
    seg.columns["mvB"] = build_mv_column("mvB", mv(1000, 1, 6, empty=True), "INT")  # <= 8: pg_mv_group_8, never the tail
    seg.columns["mvC"] = build_mv_column("mvC", mv(50, 1, 11), "LONG")              # up to 11: pg_mv_group_8 + the tail loop
    seg.columns["mvD"] = build_mv_column("mvD", [[v * 7 - 20000 for v in row] for row in mv(6000, 1, 3)], "INT")   # > 4 096 values: the dictionary stays in global memory
    seg.columns["mvS"] = build_mv_column("mvS", [[f"k{v % 7}" for v in row] for row in mv(40, 1, 4)], "STRING")
    seg.columns["m"] = build_column("m", rng.integers(-(1 << 31), 1 << 31, n).astype(np.int64).tolist(), "INT", dictionary=False)
    seg.columns["md"] = build_column("md", rng.integers(0, 300, n).tolist(), "INT")                      # dictionary-encoded value: not this kernel's shape
    seg.columns["s1"] = build_column("s1", rng.integers(0, 5, n).tolist(), "INT")
    return seg


QUERIES = [
    ("SELECT mvA, COUNT(*), SUM(m) FROM mvg GROUP BY mvA LIMIT 100", "pg_mv_group_4"),
    ("SELECT mvA, COUNT(*) FROM mvg GROUP BY mvA LIMIT 100", "pg_mv_group_4"),
    ("SELECT mvA, MIN(m), MAX(m), SUM(m), COUNT(*), AVG(m) FROM mvg GROUP BY mvA LIMIT 100", "pg_mv_group_4"),
    ("SELECT mvS, SUM(m), MINMAXRANGE(m) FROM mvg GROUP BY mvS LIMIT 100", "pg_mv_group_4"),
    ("SELECT mvB, COUNT(*), MAX(m) FROM mvg GROUP BY mvB LIMIT 2000", "pg_mv_group_8"),
    ("SELECT mvB, SUM(m) FROM mvg GROUP BY mvB LIMIT 2000", "pg_mv_group_8"),
    ("SELECT mvC, COUNT(*), SUM(m), MIN(m) FROM mvg GROUP BY mvC LIMIT 100", "pg_mv_group_8"),
    # the neighbours stay where they were: a filter, a second group column, a dictionary-encoded value, a *MV function
    ("SELECT mvA, COUNT(*), SUM(m) FROM mvg WHERE s1 < 3 GROUP BY mvA LIMIT 100", "pg_mv_query_l"),
    ("SELECT mvA, s1, COUNT(*), SUM(m) FROM mvg GROUP BY mvA, s1 LIMIT 1000", "pg_mv_group_4"),        # ... next to one single-value dictionary column
    ("SELECT s1, mvB, MAX(m) FROM mvg GROUP BY s1, mvB LIMIT 100000", "pg_mv_group_8"),
    ("SELECT s1, mvC, COUNT(*) FROM mvg GROUP BY s1, mvC LIMIT 1000", "pg_mv_group_8"),
    ("SELECT mvA, mvS, COUNT(*) FROM mvg GROUP BY mvA, mvS LIMIT 1000", "pg_mv_query_l"),                # two multi-value keys
    ("SELECT mvA, SUM(md) FROM mvg GROUP BY mvA LIMIT 100", "pg_mv_query_l"),
    ("SELECT mvA, SUMMV(mvC) FROM mvg GROUP BY mvA LIMIT 100", "pg_mv_query_l"),
    # ---- the *MV functions over ONE multi-value INT column, single-value keys: pg_mv_aggr_* (the doc's entries reduced in registers) ----
    ("SELECT s1, SUMMV(mvA), COUNTMV(mvA), MAXMV(mvA), MINMV(mvA), COUNT(*) FROM mvg GROUP BY s1 LIMIT 100", "pg_mv_aggr_4"),
    ("SELECT s1, md, SUMMV(mvB), AVGMV(mvB), MINMAXRANGEMV(mvB) FROM mvg GROUP BY s1, md LIMIT 10000", "pg_mv_aggr_8"),
    ("SELECT md, MAXMV(mvB) FROM mvg GROUP BY md LIMIT 1000", "pg_mv_aggr_8"),
    ("SELECT s1, SUMMV(mvD), MINMV(mvD), MAXMV(mvD) FROM mvg GROUP BY s1 LIMIT 100", "pg_mv_aggr_4"),
    ("SELECT s1, COUNTMV(mvC), COUNT(*) FROM mvg GROUP BY s1 LIMIT 100", "pg_mv_aggr_4"),          # only the number of entries: nothing of the entries is read
    ("SELECT s1, SUMMV(mvC) FROM mvg GROUP BY s1 LIMIT 100", "pg_mv_query_l"),                     # LONG entries
    ("SELECT s1, SUMMV(mvA), SUM(m) FROM mvg GROUP BY s1 LIMIT 100", "pg_mv_query_l"),              # a single-value source next to it
    ("SELECT s1, SUMMV(mvA), COUNTMV(mvB) FROM mvg GROUP BY s1 LIMIT 100", "pg_mv_query_l"),        # two multi-value columns
]


@pytest.fixture(scope="module", params=[1, 300, 2049, 70_001, 600_011])
def pair(request, gpu_api, oracle_api):
    host = table(request.param, seed=request.param)
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    yield g, o, host
    g.destroy()
    o.destroy()


@pytest.mark.parametrize("sql,kernel", QUERIES)
def test_group_by_one_multi_value_column(pair, sql, kernel):
    g, o, _ = pair
    gb, ob = g.execute(sql), o.execute(sql)
    assert gb.rows() == ob.rows()
    for f in STATS:
        assert getattr(gb.stats, f) == getattr(ob.stats, f), f
    if not os.environ.get("PG_NO_MVG") and not os.environ.get("PG_FORCE_INTERPRETER"):
        if gb.stats.num_total_docs >= 300 and kernel:   # (a single doc: shorter rows than the column's longest, predicates folded into match-all / empty)
            assert gb.stats.kernel.decode() == kernel
    assert g.execute(sql).rows() == ob.rows()   # the cached plan


def test_the_interpreter_frame_returns_the_same_rows(pair, gpu_api, gpu_knobs):
    g, _, host = pair
    fast = [g.execute(sql).rows() for sql, k in QUERIES if k and not k.startswith("pg_mv_query")]
    gpu_knobs(PG_NO_MVG="1")
    g2 = NativeSegment(gpu_api, host)   # (plans are cached per segment: a new one sees the knob)
    for (sql, k), rows in zip([q for q in QUERIES if q[1] and not q[1].startswith("pg_mv_query")], fast):
        gb = g2.execute(sql)
        assert gb.stats.kernel.decode() == "pg_mv_query_l" and gb.rows() == rows, sql
    g2.destroy()


def test_kept_device_table_merges(pair, gpu_api):
    """two results of the kernel fold in the library like any dense table (pg_result_merge): the counts double"""
    g, o, _ = pair
    sql = "SELECT mvA, COUNT(*), SUM(m) FROM mvg GROUP BY mvA LIMIT 100"
    a, b = g.execute_native(sql), g.execute_native(sql)
    merged = a.merge(b).block().rows()
    a.free()
    b.free()
    once = o.execute(sql).rows()
    assert {k: [v[0] * 2, v[1] * 2] for k, v in once.items()} == {k: list(v) for k, v in merged.items()}
