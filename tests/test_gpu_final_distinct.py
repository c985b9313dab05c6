"""PG_QUERY_FLAG_FINAL_DISTINCT: DISTINCTCOUNT / DISTINCTCOUNTHLL returned as final values computed from two integers per group the device
folds the states into (pg_aux_summarize_kernel) — equal to AggregationFunction#extractFinalResult of the oracle's intermediates:
HyperLogLog#cardinality of the registers (stream-lib 2.9.8; pinned by the reference's goldens 5977 / 23825 / 1886 / 4492,
InterSegmentAggregationSingleValueQueriesTest.java:261-274) and the size of the value set."""
import numpy as np
import pytest

from pinot_amd import capi, synth
from pinot_amd.executor import NativeSegment, hll_cardinality
from pinot_amd.query import parse_sql
from tests.fixtures import SV_FILTER, sv_segment

pytestmark = pytest.mark.gpu


def finals(block):
    """{group key: [final value per aggregation]} of a block holding intermediates"""
    out = {}
    for k, vals in block.rows().items():
        row = []
        for v in vals:
            if isinstance(v, (bytes, bytearray)):
                row.append(hll_cardinality(v))
            elif isinstance(v, frozenset):
                row.append(len(v))
            else:
                row.append(v)
        out[k] = row
    return out


def check(g, o, sql, kernel=None):
    qf = parse_sql(sql)
    qf.flags |= capi.QUERY_FLAG_FINAL_DISTINCT
    gb, ob = g.execute(qf), o.execute(sql)
    assert gb.rows() == finals(ob), sql
    assert gb.stats.num_docs_scanned == ob.stats.num_docs_scanned
    if kernel:
        assert gb.stats.kernel.decode() == kernel
    # without the flag the intermediates still come back
    assert g.execute(sql).rows() == ob.rows()


def test_reference_goldens_as_final_values(gpu_api, oracle_api, sv_data):
    host = sv_segment(sv_data)
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    q = parse_sql("SELECT DISTINCTCOUNTHLL(column1), DISTINCTCOUNTHLL(column3), DISTINCTCOUNT(column1) FROM testTable")
    q.flags |= capi.QUERY_FLAG_FINAL_DISTINCT
    q2 = parse_sql("SELECT DISTINCTCOUNTHLL(column1), DISTINCTCOUNTHLL(column3), DISTINCTCOUNT(column1) FROM testTable" + SV_FILTER)
    q2.flags |= capi.QUERY_FLAG_FINAL_DISTINCT
    # InterSegmentAggregationSingleValueQueriesTest.java:261-274 (HLL) and :226-244 (DISTINCTCOUNT 6582 / 1872)
    assert g.execute(q2).aggregation_result() == [1886, 4492, 1872]
    for sql in ("SELECT column11, DISTINCTCOUNTHLL(column1), DISTINCTCOUNT(column17), COUNT(*) FROM testTable GROUP BY column11",
                "SELECT column9, DISTINCTCOUNTHLL(column3), SUM(column1) FROM testTable WHERE column6 < 900000000 GROUP BY column9 LIMIT 100000"):
        check(g, o, sql)
    g.destroy()
    o.destroy()


def test_config5_final_values(gpu_api, oracle_api):
    host = synth.generate_segment(400_003, segment_index=1, columns=synth.CFG5_COLUMNS, native=True)
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    check(g, o, synth.QUERY_CFG5, kernel="pg_part_group_by")
    check(g, o, "SELECT h1, DISTINCTCOUNTHLL(u), DISTINCTCOUNT(h3) FROM gpuBench WHERE h2 < 5 GROUP BY h1")
    check(g, o, "SELECT DISTINCTCOUNTHLL(u) FROM gpuBench WHERE h4 = 99")   # nothing matches: cardinality of an empty HyperLogLog is 0
    g.destroy()
    o.destroy()


def test_star_tree_route_final_values(gpu_api, oracle_api):
    from tests.fixtures import synth_star_segment
    host = synth_star_segment(60_000)
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    qf = parse_sql(synth.QUERY_CFG5)
    qf.flags |= capi.QUERY_FLAG_FINAL_DISTINCT
    gb = g.execute(qf)
    assert gb.stats.star_tree_index == 0
    assert gb.rows() == finals(o.execute(synth.QUERY_CFG5))
    g.destroy()
    o.destroy()


def test_small_range_estimate_without_an_empty_register(gpu_api, oracle_api):
    """HyperLogLog#cardinality, small-range branch with NO empty register: linearCounting(m, 0) = m * Math.log(m / 0.0) = Infinity and
    Math.round(Infinity) = Long.MAX_VALUE (stream-lib 2.9.8 HyperLogLog.java).  At log2m 4 about a quarter of the ~40-value sets get there: every
    register 1 or 2 keeps the estimate under 2.5 m.  The oracle, the Python mirror and the device's table of m + 1 values give that value."""
    from pinot_amd.segment import build_segment
    n = 4000
    data = {"k": (np.arange(n) // 50).astype(np.int32), "v": (np.arange(n) % 50 * 7919 + np.arange(n) // 50 * 104729).astype(np.int32)}
    host = build_segment("hll_small", data, {"k": "INT", "v": "INT"})
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    sql = "SELECT k, DISTINCTCOUNTHLL(v, 4) FROM t GROUP BY k LIMIT 1000"
    expected = finals(o.execute(sql))
    saturated = [k for k, v in expected.items() if v[0] == (1 << 63) - 1]
    assert saturated, "no group of this data reaches the branch: change the generator"
    qf = parse_sql(sql)
    qf.flags |= capi.QUERY_FLAG_FINAL_DISTINCT
    assert g.execute(qf).rows() == expected
    g.destroy()
    o.destroy()
