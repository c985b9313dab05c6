"""enableNullHandling filters: the operator tree's getTrues in three-valued logic — a column predicate is true where it holds and the value
is not null (BaseColumnFilterOperator.java:45-72); NOT matches where its child is FALSE: NOT(trues OR nulls) for a column leaf
(BaseFilterOperator.java:105-122), AND / OR falses from their children's trues and nulls (AndFilterOperator.java:62-90,
OrFilterOperator.java:61-87; compound children have no nulls of their own: BaseFilterOperator.java:98-100), NotFilterOperator.java:52-63;
an always-false predicate is EmptyFilterOperator and an always-true one the not-null bitmap (FilterOperatorUtils.java:76-88), both two-valued.
The expected masks are written by hand in numpy from those rules; oracle against them on the CPU, HIP path against both in the gpu tests."""
import numpy as np
import pytest

from pinot_amd import capi, formats
from pinot_amd.executor import NativeSegment
from pinot_amd.query import parse_sql
from pinot_amd.segment import build_segment

N = 150_000


def segment(n=N, seed=11, with_valid=False):
    rng = np.random.default_rng(seed)
    data = {
        "d": rng.integers(0, 40, n).astype(np.int32),            # dictionary + inverted index
        "s": np.sort(rng.integers(0, 300, n)).astype(np.int32),  # sorted
        "r": rng.integers(0, 1000, n).astype(np.int32),          # raw
        "x": rng.integers(0, 5000, n).astype(np.int32),          # raw with a range index
        "g": rng.integers(0, 60, n).astype(np.int32),            # dictionary, no nulls
        "m": rng.integers(0, 1 << 20, n).astype(np.int32),
    }
    schema = {k: "INT" for k in data}
    host = build_segment("nh_0", data, schema, inverted_index_columns=["d"], no_dictionary_columns=["r", "m", "x"], range_index_columns=["x"])
    nulls = {
        "d": np.flatnonzero(rng.random(n) < 0.15),
        "s": np.flatnonzero(rng.random(n) < 0.05),
        "r": np.concatenate([np.arange(n // 200, n * 30 // 100), np.arange(n * 3 // 4, n * 3 // 4 + 10)]),
        "x": np.flatnonzero(rng.random(n) < 0.2),
    }
    for c, ids in nulls.items():
        host.columns[c].null_vector = np.frombuffer(formats.serialize_roaring(ids), dtype=np.uint8)
    valid = np.flatnonzero(rng.random(n) < 0.7) if with_valid else None
    return host, data, nulls, valid


# (WHERE clause, mask expression over the columns d s r x g and the null masks nd ns nr nx)
CASES = [
    ("d = 5", "(d == 5) & ~nd"),
    ("NOT d = 5", "(d != 5) & ~nd"),
    ("d != 5", "(d != 5) & ~nd"),
    ("NOT d != 5", "(d == 5) & ~nd"),
    ("d IN (1, 2, 3)", "np.isin(d, [1, 2, 3]) & ~nd"),
    ("d NOT IN (1, 2, 3)", "~np.isin(d, [1, 2, 3]) & ~nd"),
    ("r BETWEEN 100 AND 600", "(r >= 100) & (r <= 600) & ~nr"),
    ("NOT (r BETWEEN 100 AND 600)", "~((r >= 100) & (r <= 600)) & ~nr"),
    ("s < 100", "(s < 100) & ~ns"),
    ("NOT s < 100", "(s >= 100) & ~ns"),
    ("x > 2500", "(x > 2500) & ~nx"),
    ("NOT x > 2500", "(x <= 2500) & ~nx"),
    ("x = 77", "(x == 77) & ~nx"),
    ("d = 5 AND r < 500", "(d == 5) & ~nd & (r < 500) & ~nr"),
    ("d = 5 OR r < 500", "((d == 5) & ~nd) | ((r < 500) & ~nr)"),
    # falses of AND / OR: every child contributes trues OR nulls
    ("NOT (d = 5 AND r < 500)", "~((((d == 5) & ~nd) | nd) & (((r < 500) & ~nr) | nr))"),
    ("NOT (d = 5 OR r < 500)", "~((((d == 5) & ~nd) | nd) | (((r < 500) & ~nr) | nr))"),
    ("NOT (NOT d = 5)", "(d == 5) & ~nd"),
    ("NOT (NOT (NOT d = 5))", "(d != 5) & ~nd"),
    # a compound child has no nulls of its own: its trues alone are not-false
    ("NOT (g < 30 AND (d = 5 OR r < 500))", "~((g < 30) & (((d == 5) & ~nd) | ((r < 500) & ~nr)))"),
    ("NOT (g < 30 OR (d = 5 AND r < 500))", "~((g < 30) | (((d == 5) & ~nd) & ((r < 500) & ~nr)))"),
    ("g < 30 AND NOT (d IN (1, 2) OR s > 250)", "(g < 30) & ~(((np.isin(d, [1, 2]) & ~nd) | nd) | (((s > 250) & ~ns) | ns))"),
    # EmptyFilterOperator / the not-null bitmap of an always-true predicate are two-valued
    ("d = 99", "np.zeros(len(d), bool)"),
    ("NOT d = 99", "np.ones(len(d), bool)"),
    ("d >= 0", "~nd"),
    ("NOT d >= 0", "nd"),
    ("d >= 0 AND r < 500", "~nd & (r < 500) & ~nr"),
    ("NOT (d >= 0 AND r < 500)", "~(~nd & (((r < 500) & ~nr) | nr))"),
    # IS NULL / IS NOT NULL are bitmap leaves (two-valued)
    ("d IS NULL OR NOT d IN (1, 2, 3)", "nd | (~np.isin(d, [1, 2, 3]) & ~nd)"),
    ("NOT (d IS NULL) AND NOT r < 500", "~nd & (r >= 500) & ~nr"),
    ("NOT (d IS NULL OR r < 500)", "~(nd | (((r < 500) & ~nr) | nr))"),
    # no nulls in g: nothing changes
    ("NOT g < 30", "g >= 30"),
    ("g < 30 OR NOT g < 50", "(g < 30) | (g >= 50)"),
]


def expected_mask(expr, data, nulls, n):
    env = {k: v.astype(np.int64) for k, v in data.items()}
    for c in ("d", "s", "r", "x"):
        env["n" + c] = np.isin(np.arange(n), nulls[c])
    env["np"] = np
    return eval(expr, env)   # noqa: S307 (test data)


def check_filters(api, with_valid):
    host, data, nulls, valid = segment(with_valid=with_valid)
    seg = NativeSegment(api, host)
    if valid is not None:
        seg.set_queryable_doc_ids(valid)   # FilterPlanNode.run: AND(filter, queryableDocIds)
    for where, expr in CASES:
        mask = expected_mask(expr, data, nulls, N)
        if valid is not None:
            mask = mask & np.isin(np.arange(N), valid)
        got = seg.filter(f"SELECT COUNT(*) FROM t WHERE {where}", null_handling=True).doc_ids()
        assert np.array_equal(got, np.flatnonzero(mask)), where
        # the aggregation over it (a GROUP BY: no FastFilteredCountOperator)
        q = parse_sql(f"SELECT g, COUNT(*), SUM(m) FROM t WHERE {where} GROUP BY g LIMIT 1000")
        q.flags |= capi.QUERY_FLAG_NULL_HANDLING
        rows = seg.execute(q).rows()
        exp = {}
        for gv in np.unique(data["g"][mask]):
            sel = mask & (data["g"] == gv)
            exp[(int(gv),)] = [int(sel.sum()), float(data["m"][sel].astype(np.int64).sum())]
        assert rows == exp, where
    seg.destroy()


def test_oracle_filters_in_three_valued_logic(oracle_api):
    check_filters(oracle_api, with_valid=False)


def test_oracle_filters_in_three_valued_logic_under_queryable_doc_ids(oracle_api):
    check_filters(oracle_api, with_valid=True)


def test_without_the_flag_nulls_are_their_default_values(oracle_api):
    host, data, nulls, _ = segment()
    seg = NativeSegment(oracle_api, host)
    assert np.array_equal(seg.filter("SELECT COUNT(*) FROM t WHERE NOT d = 5").doc_ids(), np.flatnonzero(data["d"] != 5))
    seg.destroy()


def test_fast_filtered_count_knows_no_nulls(oracle_api):
    """A lone COUNT(*) over an index-only filter is FastFilteredCountOperator: getNumMatchingDocs / getBitmaps of the operators, which do not
    subtract the nulls (AggregationPlanNode.java:104-108, InvertedIndexFilterOperator.java:103-131) — the reference's answer, restated as it is."""
    host, data, nulls, _ = segment()
    seg = NativeSegment(oracle_api, host)
    q = parse_sql("SELECT COUNT(*) FROM t WHERE d = 5")
    q.flags |= capi.QUERY_FLAG_NULL_HANDLING
    assert seg.execute(q).aggregation_result() == [int((data["d"] == 5).sum())]
    q = parse_sql("SELECT COUNT(*) FROM t WHERE d = 5 AND r < 500")   # a scan: AggregationOperator over the three-valued filter
    q.flags |= capi.QUERY_FLAG_NULL_HANDLING
    nd, nr = np.isin(np.arange(N), nulls["d"]), np.isin(np.arange(N), nulls["r"])
    assert seg.execute(q).aggregation_result() == [int(((data["d"] == 5) & ~nd & (data["r"] < 500) & ~nr).sum())]
    seg.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("with_valid", [False, True])
def test_gpu_filters_in_three_valued_logic(gpu_api, with_valid):
    check_filters(gpu_api, with_valid)


@pytest.mark.gpu
def test_gpu_fast_filtered_count_like_the_oracle(gpu_api, oracle_api):
    host, *_ = segment()
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    for where in ("d = 5", "d IN (1, 2) AND s < 100", "NOT d = 5", "d = 5 AND r < 500", "d IS NULL OR d = 3"):
        q = parse_sql(f"SELECT COUNT(*) FROM t WHERE {where}")
        q.flags |= capi.QUERY_FLAG_NULL_HANDLING
        assert g.execute(q).aggregation_result() == o.execute(q).aggregation_result(), where
    g.destroy(); o.destroy()
