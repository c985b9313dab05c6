"""PG_QUERY_FLAG_NULL_HANDLING (QueryContext#isNullHandlingEnabled) where it cannot change the answer — no column the query reads holds a null
in the segment: the reference keeps its ordinary plan then (AggregationPlanNode.java:104-121 hasNullValues, StarTreeUtils.java:381-400), fast
paths and star-trees included — (what is refused: nulls in multi-value columns).  The null-aware filters,
aggregations and keys themselves: tests/test_null_handling_filters.py, tests/test_null_handling_aggregations.py."""
import pytest

from pinot_amd import capi
from pinot_amd.executor import NativeSegment
from pinot_amd.query import parse_sql
from tests.test_null_and_valid_docs import null_segment

# nulls live in d (10 % of the docs), r (run containers); m has an EMPTY null vector, g and s have none
TAKEN = [
    "SELECT g, COUNT(*), SUM(m), MAX(m) FROM nulls GROUP BY g LIMIT 1000",
    "SELECT g, SUM(m) FROM nulls WHERE s BETWEEN 50 AND 200 AND m < 900000 GROUP BY g LIMIT 1000",
    "SELECT COUNT(*), MIN(m), AVG(m) FROM nulls WHERE g IN (1, 2, 3)",
    "SELECT s, DISTINCTCOUNT(g), DISTINCTCOUNTHLL(m) FROM nulls WHERE m IS NOT NULL GROUP BY s LIMIT 1000",   # m's vector is empty
    "SELECT COUNT(*) FROM nulls",
    "SELECT MIN(g), MAX(g) FROM nulls",                        # NonScanBasedAggregationOperator stays (no nulls in g)
]
NULL_AWARE = [   # columns with nulls: answered in three-valued logic / with NULL results (compared with the oracle in the gpu test)
    "SELECT g, SUM(m) FROM nulls WHERE d IN (1, 2, 3) GROUP BY g LIMIT 1000",        # a filter column with nulls
    "SELECT d, COUNT(*) FROM nulls GROUP BY d LIMIT 1000",                            # a group-by column with nulls
    "SELECT g, SUM(r) FROM nulls GROUP BY g LIMIT 1000",                              # an aggregation argument with nulls
    "SELECT COUNT(*) FROM nulls WHERE d IS NULL",
    "SELECT g, COUNT(*) FROM nulls WHERE NOT (r < 500) GROUP BY g LIMIT 1000",
    "SELECT SUM(m), MAX(m) FROM nulls WHERE g > 1000",                                # no GROUP BY, nothing matches: null results
    "SELECT r, COUNT(*) FROM nulls WHERE g < 4 GROUP BY r LIMIT 100000",              # a no-dictionary group-by column with nulls
]
REFUSED = []   # (nulls in multi-value columns: tests/test_null_handling_aggregations.py)


def flagged(sql):
    q = parse_sql(sql)
    q.flags |= capi.QUERY_FLAG_NULL_HANDLING
    return q


def check(api_segment, plain_segment=None):
    plain_segment = plain_segment or api_segment
    for sql in TAKEN:
        assert api_segment.execute(flagged(sql)).rows() == plain_segment.execute(parse_sql(sql)).rows(), sql
    for sql in REFUSED:
        with pytest.raises(capi.NativeError) as e:
            api_segment.execute(flagged(sql))
        assert e.value.status == capi.PG_ERR_UNSUPPORTED and "enableNullHandling" in str(e.value), sql
        api_segment.execute(parse_sql(sql))   # the same query without the flag runs
    assert api_segment.execute(flagged("SELECT SUM(m), MAX(m) FROM nulls WHERE g > 1000")).aggregation_result() == [None, None]


def test_oracle_takes_and_refuses(oracle_api):
    host, *_ = null_segment(60_000)
    seg = NativeSegment(oracle_api, host)
    check(seg)
    # a group-by over nothing is an empty table with or without null handling
    assert seg.execute(flagged("SELECT g, SUM(m) FROM nulls WHERE g > 1000 GROUP BY g LIMIT 10")).rows() == {}
    seg.destroy()


@pytest.mark.gpu
def test_gpu_takes_and_refuses_like_the_oracle(gpu_api, oracle_api):
    host, *_ = null_segment(60_000)
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    check(g, o)
    for sql in TAKEN:
        gb, ob = g.execute(flagged(sql)), o.execute(flagged(sql))
        assert gb.rows() == ob.rows() and gb.stats.num_docs_scanned == ob.stats.num_docs_scanned
        assert gb.stats.num_entries_scanned_in_filter == ob.stats.num_entries_scanned_in_filter
    for sql in NULL_AWARE:
        assert g.execute(flagged(sql)).rows() == o.execute(flagged(sql)).rows(), sql
    assert g.execute(flagged("SELECT g, SUM(m) FROM nulls WHERE g > 1000 GROUP BY g LIMIT 10")).rows() == {}
    g.destroy()
    o.destroy()
