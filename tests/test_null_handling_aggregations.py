"""enableNullHandling beyond the filter: aggregations skip the docs whose argument is null (NullableSingleInputAggregationFunction), COUNT(col)
counts the values, SUM / MIN / MAX / AVG / MINMAXRANGE over no value are NULL, a null is a group key of its own.

Pinned by the reference's own expectations: NullEnabledQueriesTest (pinot-core/src/test/java/org/apache/pinot/queries/NullEnabledQueriesTest.java:
94-120 the table — 1 000 records, `column` = base + i for even i and NULL for odd i, `key` = 1 / 2 for the even halves and NULL for odd i —
and :283-345, :470-495 the expected rows; its broker serves the segment 4 times, hence the factors of 4 there).  The oracle restates the
semantics doc at a time; the HIP path composes them from IS [NOT] NULL partitions (pg_nullaware.cpp) and is compared with the oracle."""
import numpy as np
import pytest

from pinot_amd import capi, formats
from pinot_amd.executor import NativeSegment
from pinot_amd.query import parse_sql
from pinot_amd.segment import build_segment

BASE = 0.25
NUM_RECORDS = 1000


def flagged(sql):
    q = parse_sql(sql)
    q.flags |= capi.QUERY_FLAG_NULL_HANDLING
    return q


def reference_table(dictionary=True):
    n = NUM_RECORDS
    col = np.array([BASE + i if i % 2 == 0 else 0.0 for i in range(n)])                       # the stored default of a null is irrelevant
    key = np.array([(1 if i < n // 2 else 2) if i % 2 == 0 else -2147483648 for i in range(n)], dtype=np.int32)
    host = build_segment("testTable_0", {"column": col, "key": key}, {"column": "DOUBLE", "key": "INT"},
                         no_dictionary_columns=[] if dictionary else ["column"])
    odd = np.arange(1, n, 2)
    for c in ("column", "key"):
        host.columns[c].null_vector = np.frombuffer(formats.serialize_roaring(odd), dtype=np.uint8)
    return host


def check_reference_expectations(seg):
    sum1 = sum(BASE + i for i in range(0, NUM_RECORDS // 2, 2))
    sum2 = sum(BASE + i for i in range(NUM_RECORDS // 2, NUM_RECORDS, 2))
    # :283-330 (one segment: the test's 4 * ... are this / 4)
    rows = seg.execute(flagged("SELECT key, SUM(column), MIN(column), MAX(column), COUNT(column) FROM testTable GROUP BY key LIMIT 10")).rows()
    assert rows == {(1,): [sum1, BASE, BASE + 498, 250], (2,): [sum2, BASE + 500, BASE + 998, 250], (None,): [None, None, None, 0]}
    # :331-350: count(*) counts every doc, count(col) the values
    assert seg.execute(flagged("SELECT COUNT(*), COUNT(column), MIN(column), MAX(column) FROM testTable")).aggregation_result() == \
        [1000, 500, BASE, BASE + 998]
    # :470-495
    got = seg.execute(flagged("SELECT COUNT(column), MIN(column), MAX(column), AVG(column), SUM(column) FROM testTable")).aggregation_result()
    assert got == [500, BASE, BASE + 998, (sum1 + sum2, 500), sum1 + sum2]
    # :524-553 COUNT(*) GROUP BY column: 500 values once each + the NULL group holding the 500 null docs
    rows = seg.execute(flagged("SELECT column, COUNT(*) FROM testTable GROUP BY column LIMIT 1000")).rows()
    assert len(rows) == 501 and rows[(None,)] == [500] and all(v == [1] for k, v in rows.items() if k != (None,))
    # :600-640: comparisons never match a null
    assert seg.execute(flagged(f"SELECT COUNT(*), SUM(column) FROM testTable WHERE column > {BASE + 69}")).aggregation_result()[0] == 465
    assert seg.execute(flagged(f"SELECT COUNT(*), MAX(column) FROM testTable WHERE column = {BASE + 68}")).aggregation_result() == [1, BASE + 68]
    assert seg.execute(flagged(f"SELECT COUNT(*), MAX(column) FROM testTable WHERE column = {BASE + 69}")).aggregation_result() == [0, None]
    # :709-740 MAX(column) GROUP BY column
    rows = seg.execute(flagged("SELECT column, MAX(column) FROM testTable GROUP BY column LIMIT 1000")).rows()
    assert rows[(None,)] == [None] and rows[(BASE + 4,)] == [BASE + 4]


@pytest.mark.parametrize("dictionary", [True, False])
def test_oracle_reproduces_null_enabled_queries_test(oracle_api, dictionary):
    seg = NativeSegment(oracle_api, reference_table(dictionary))
    check_reference_expectations(seg)
    seg.destroy()


def test_oracle_no_dictionary_argument(oracle_api):
    seg = NativeSegment(oracle_api, reference_table(False))
    sum_all = sum(BASE + i for i in range(0, NUM_RECORDS, 2))
    assert seg.execute(flagged("SELECT COUNT(*), COUNT(column), MIN(column), MAX(column), SUM(column) FROM testTable")).aggregation_result() == \
        [1000, 500, BASE, BASE + 998, sum_all]
    rows = seg.execute(flagged("SELECT key, SUM(column), COUNT(column) FROM testTable GROUP BY key LIMIT 10")).rows()
    assert rows[(None,)] == [None, 0] and rows[(1,)][1] == 250
    seg.destroy()


# ---- a random segment: HIP path == oracle ------------------------------------------------------------------------------------------------
N = 120_000


def random_segment(n=N, seed=3):
    rng = np.random.default_rng(seed)
    data = {
        "a": rng.integers(0, 30, n).astype(np.int32),            # dictionary group column with nulls
        "b": rng.integers(0, 12, n).astype(np.int32),            # dictionary group column with nulls
        "g": rng.integers(0, 50, n).astype(np.int32),            # no nulls
        "m": rng.integers(0, 1 << 20, n).astype(np.int32),       # raw metric with nulls
        "w": rng.integers(-1000, 1000, n).astype(np.int64),      # dictionary LONG metric with nulls
        "x": (rng.integers(0, 1 << 16, n) / 8.0),                # raw DOUBLE metric with nulls (exact sums)
        "z": rng.integers(0, 1 << 20, n).astype(np.int32),       # raw metric, no nulls
        "r": rng.integers(0, 1000, n).astype(np.int32),          # raw filter column with nulls
    }
    schema = {"a": "INT", "b": "INT", "g": "INT", "m": "INT", "w": "LONG", "x": "DOUBLE", "z": "INT", "r": "INT"}
    host = build_segment("nh_1", data, schema, inverted_index_columns=["a"], no_dictionary_columns=["m", "x", "z", "r"])
    nulls = {
        "a": np.flatnonzero(rng.random(n) < 0.10),
        "b": np.flatnonzero(rng.random(n) < 0.30),
        "m": np.flatnonzero(rng.random(n) < 0.25),
        "w": np.flatnonzero((np.arange(n) % 7 == 0) | (data["g"] == 3)),   # every doc of group g = 3 is null in w
        "x": np.flatnonzero(rng.random(n) < 0.5),
        "r": np.flatnonzero(rng.random(n) < 0.2),
    }
    for c, ids in nulls.items():
        host.columns[c].null_vector = np.frombuffer(formats.serialize_roaring(ids), dtype=np.uint8)
    return host, data, nulls


QUERIES = [
    "SELECT COUNT(*), COUNT(m), SUM(m), MIN(m), MAX(m), AVG(m), MINMAXRANGE(m) FROM t",
    "SELECT COUNT(*), SUM(m), SUM(w), SUM(x), SUM(z), COUNT(w), COUNT(x) FROM t WHERE r < 500",
    "SELECT SUM(m), MAX(w), COUNT(*) FROM t WHERE g > 1000",                                   # nothing matches: NULLs and a 0
    "SELECT SUM(w), COUNT(w), COUNT(*) FROM t WHERE g = 3",                                    # every argument is null
    "SELECT DISTINCTCOUNT(w), DISTINCTCOUNTHLL(w), DISTINCTCOUNTHLL(m), COUNT(*) FROM t WHERE g < 10",
    "SELECT g, COUNT(*), COUNT(m), SUM(m), MIN(w), MAX(x), AVG(x), SUM(z) FROM t GROUP BY g LIMIT 1000",
    "SELECT g, SUM(w), MINMAXRANGE(w), COUNT(w) FROM t WHERE NOT r < 500 GROUP BY g LIMIT 1000",   # g = 3: a group whose SUM(w) is NULL
    "SELECT g, DISTINCTCOUNT(w), DISTINCTCOUNTHLL(x), COUNT(*) FROM t GROUP BY g LIMIT 1000",
    "SELECT a, COUNT(*), SUM(z) FROM t GROUP BY a LIMIT 1000",                                 # a NULL key
    "SELECT a, b, COUNT(*), SUM(z), MAX(z) FROM t WHERE r BETWEEN 100 AND 800 GROUP BY a, b LIMIT 10000",   # four null partitions
    "SELECT a, g, COUNT(*), SUM(m), COUNT(m), MIN(x) FROM t GROUP BY a, g LIMIT 10000",        # NULL keys and NULL-skipping arguments
    "SELECT b, a, SUM(w), AVG(m), DISTINCTCOUNT(a) FROM t WHERE a != 5 GROUP BY b, a LIMIT 10000",
    "SELECT a, COUNT(*) FROM t WHERE a IS NULL GROUP BY a LIMIT 10",                            # only the NULL group
    "SELECT a, COUNT(*) FROM t WHERE a IS NOT NULL AND a < 3 GROUP BY a LIMIT 10",
    "SELECT b, SUM(m) FROM t WHERE g > 1000 GROUP BY b LIMIT 10",                               # no group at all
    "SELECT r, COUNT(*), SUM(z) FROM t WHERE g < 5 GROUP BY r LIMIT 100000",                    # a no-dictionary INT key with nulls
    "SELECT x, COUNT(*), MAX(m) FROM t WHERE g = 1 GROUP BY x LIMIT 100000",                    # a no-dictionary DOUBLE key with nulls
    "SELECT a, r, COUNT(*), SUM(w) FROM t WHERE g < 3 GROUP BY a, r LIMIT 100000",              # dictionary and no-dictionary NULL keys together
]


def test_oracle_random_segment_against_numpy(oracle_api):
    """the oracle's aggregations and keys against a hand evaluation (the filters have their own test)"""
    host, data, nulls = random_segment()
    isnull = {c: np.isin(np.arange(N), ids) for c, ids in nulls.items()}
    seg = NativeSegment(oracle_api, host)
    rows = seg.execute(flagged("SELECT a, g, COUNT(*), SUM(m), COUNT(m), MIN(x) FROM t GROUP BY a, g LIMIT 10000")).rows()
    exp = {}
    for i in range(N):
        k = (None if isnull["a"][i] else int(data["a"][i]), int(data["g"][i]))
        e = exp.setdefault(k, [0, None, 0, None])
        e[0] += 1
        if not isnull["m"][i]:
            e[1] = (e[1] or 0.0) + float(data["m"][i]); e[2] += 1
        if not isnull["x"][i]:
            e[3] = float(data["x"][i]) if e[3] is None else min(e[3], float(data["x"][i]))
    assert rows == exp
    got = seg.execute(flagged("SELECT SUM(w), COUNT(w), COUNT(*) FROM t WHERE g = 3")).aggregation_result()
    assert got == [None, 0, int((data["g"] == 3).sum())]
    seg.destroy()


def test_oracle_runs_every_query(oracle_api):
    host, *_ = random_segment(20_000)
    seg = NativeSegment(oracle_api, host)
    for sql in QUERIES:
        seg.execute(flagged(sql)).rows()
    seg.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("dictionary", [True, False])
def test_gpu_reproduces_null_enabled_queries_test(gpu_api, dictionary):
    seg = NativeSegment(gpu_api, reference_table(dictionary))
    check_reference_expectations(seg)
    seg.destroy()


@pytest.mark.gpu
def test_gpu_equals_oracle(gpu_api, oracle_api):
    host, *_ = random_segment()
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    for sql in QUERIES:
        a, b = g.execute(flagged(sql)), o.execute(flagged(sql))
        assert a.rows() == b.rows(), sql
        assert a.stats.num_docs_scanned == b.stats.num_docs_scanned, sql
    # numGroupsLimit: the groups are the first N keys in docId order over ALL the matching docs — the joined sub-queries (other doc sets) trim nothing
    for limit in (7, 20):
        for sql in ("SELECT g, SUM(m), DISTINCTCOUNT(a), COUNT(*) FROM t GROUP BY g LIMIT 1000", "SELECT g, z, MAX(w), COUNT(x) FROM t WHERE r < 900 GROUP BY g, z LIMIT 1000"):
            qa, qb = flagged(sql), flagged(sql)
            qa.num_groups_limit = qb.num_groups_limit = limit
            a, b = g.execute(qa), o.execute(qb)
            assert a.rows() == b.rows() and len(a.rows()) == limit, (sql, limit)
            assert a.stats.num_groups_limit_reached == b.stats.num_groups_limit_reached == 1
    # FINAL_DISTINCT travels through the joins
    q = flagged("SELECT g, DISTINCTCOUNT(w), DISTINCTCOUNTHLL(x), COUNT(*) FROM t GROUP BY g LIMIT 1000")
    q.flags |= capi.QUERY_FLAG_FINAL_DISTINCT
    from pinot_amd.executor import hll_cardinality
    inter = o.execute(flagged("SELECT g, DISTINCTCOUNT(w), DISTINCTCOUNTHLL(x), COUNT(*) FROM t GROUP BY g LIMIT 1000")).rows()
    final = g.execute(q).rows()
    assert final == {k: [len(v[0]), hll_cardinality(v[1]), v[2]] for k, v in inter.items()}
    g.destroy(); o.destroy()


@pytest.mark.gpu
def test_gpu_refuses_what_is_not_partitioned(gpu_api):
    host, *_ = random_segment(20_000)
    seg = NativeSegment(gpu_api, host)
    q = flagged("SELECT a, COUNT(*) FROM t GROUP BY a LIMIT 1000")
    q.num_groups_limit = 5
    with pytest.raises(capi.NativeError) as e:
        seg.execute(q)
    assert e.value.status == capi.PG_ERR_UNSUPPORTED
    seg.destroy()


# ---- NullHandlingEnabledQueriesTest (pinot-core/src/test/java/org/apache/pinot/queries/NullHandlingEnabledQueriesTest.java): its small tables
# with their expected rows.  Selection queries there; their filters' doc sets and the group-by tables are what is asserted here. -----------------
INT_MIN = -2147483648


def small_table(rows, names=("column1", "column2"), inverted=()):
    """rows of one or two INT columns, None = null (stored as the default null value of an INT dimension)"""
    cols = list(zip(*rows)) if isinstance(rows[0], tuple) else [tuple(rows)]
    data, nulls = {}, {}
    for name, values in zip(names, cols):
        data[name] = np.array([INT_MIN if v is None else v for v in values], dtype=np.int32)
        nulls[name] = np.array([i for i, v in enumerate(values) if v is None], dtype=np.int64)
    host = build_segment("testTable_0", data, {k: "INT" for k in data}, inverted_index_columns=list(inverted))
    for name, ids in nulls.items():
        if len(ids):
            host.columns[name].null_vector = np.frombuffer(formats.serialize_roaring(ids), dtype=np.uint8)
    return host


SEVEN = [(None, None), (None, 1), (1, -1), (-1, None), (-1, 1), (1, None), (None, -1)]   # testOrFiltering / testNotAndFiltering / testNotOrFiltering


def reference_small_cases(api):
    def docs(host, where):
        seg = NativeSegment(api, host)
        got = seg.filter(f"SELECT COUNT(*) FROM testTable WHERE {where}", null_handling=True).doc_ids().tolist()
        seg.destroy()
        return got
    assert docs(small_table(SEVEN), "column1 > 0 OR column2 < 0") == [2, 5, 6]                       # :987-1010: 3 rows
    assert docs(small_table([None, -1, 1], names=("column1",)), "NOT column1 = 1") == [1]             # :1013-1030: the row -1
    assert docs(small_table(SEVEN), "NOT (column1 > 0 AND column2 < 0)") == [1, 3, 4]                  # :1033-1056: 3 rows
    assert docs(small_table(SEVEN), "NOT (column1 > 0 OR column2 < 0)") == [4]                         # :1059-1083: the row (-1, 1)
    assert docs(small_table([None, INT_MIN], names=("column1",)), f"column1 = {INT_MIN}") == [1]      # :967-984: not the null
    assert docs(small_table([None, 0, 1], names=("column1",), inverted=["column1"]), "column1 = 0") == [1]   # :903-922 (BOOLEAN false = 0, inverted index)
    assert docs(small_table([-1, None], names=("column1",)), "column1 < 0") == [0]                   # :947-964
    # :153-176 GROUP BY column COUNT(*): {2: 2, 1: 1, null: 3}
    seg = NativeSegment(api, small_table([None, None, None, 1, 2, 2], names=("column1",)))
    assert seg.execute(flagged("SELECT column1, COUNT(*) FROM testTable GROUP BY column1 LIMIT 10")).rows() == {(2,): [2], (1,): [1], (None,): [3]}
    seg.destroy()
    # :777-804 two columns: five groups, Integer.MIN_VALUE is not null
    seg = NativeSegment(api, small_table([(None, None), (None, 1), (None, 1), (1, 1), (1, None), (1, INT_MIN)]))
    assert seg.execute(flagged("SELECT column1, column2, COUNT(*) FROM testTable GROUP BY column1, column2 LIMIT 10")).rows() == \
        {(None, None): [1], (None, 1): [2], (1, 1): [1], (1, None): [1], (1, INT_MIN): [1]}
    seg.destroy()
    # :544-597 DISTINCTCOUNT skips the null
    seg = NativeSegment(api, small_table([(None, 7), (1, 7)]))
    assert seg.execute(flagged("SELECT DISTINCTCOUNT(column1) FROM testTable")).aggregation_result() == [frozenset({1})]
    assert seg.execute(flagged("SELECT column2, DISTINCTCOUNT(column1) FROM testTable GROUP BY column2 LIMIT 10")).rows() == {(7,): [frozenset({1})]}
    seg.destroy()


def test_oracle_reproduces_null_handling_enabled_queries_test(oracle_api):
    reference_small_cases(oracle_api)


@pytest.mark.gpu
def test_gpu_reproduces_null_handling_enabled_queries_test(gpu_api):
    reference_small_cases(gpu_api)


# ---- AllNullQueriesTest (pinot-core/src/test/java/org/apache/pinot/queries/AllNullQueriesTest.java:86-97 the table — 1 000 records whose one
# column is null everywhere — and :336-363, :443-468, :470-497, :538-582, :584-601 the expectations; its broker serves the segment 4 times) -----
def all_null_cases(api, data_type, dictionary):
    n = 1000
    values = np.zeros(n, dtype={"INT": np.int32, "LONG": np.int64, "FLOAT": np.float32, "DOUBLE": np.float64}[data_type])
    host = build_segment("testTable_0", {"column": values}, {"column": data_type}, no_dictionary_columns=[] if dictionary else ["column"])
    host.columns["column"].null_vector = np.frombuffer(formats.serialize_roaring(np.arange(n)), dtype=np.uint8)
    seg = NativeSegment(api, host)
    assert seg.execute(flagged("SELECT COUNT(*), COUNT(column), MIN(column), MAX(column) FROM testTable")).aggregation_result() == [1000, 0, None, None]
    assert seg.execute(flagged("SELECT COUNT(column), MIN(column), MAX(column), AVG(column), SUM(column) FROM testTable")).aggregation_result() == \
        [0, None, None, None, None]
    assert seg.execute(flagged("SELECT column, COUNT(*) FROM testTable GROUP BY column LIMIT 1000")).rows() == {(None,): [1000]}
    assert seg.execute(flagged("SELECT column, AVG(column), MAX(column) FROM testTable GROUP BY column LIMIT 20")).rows() == {(None,): [None, None]}
    assert seg.execute(flagged("SELECT COUNT(column), MIN(column), MAX(column), SUM(column) FROM testTable WHERE column = 69")).aggregation_result() == \
        [0, None, None, None]
    assert seg.filter("SELECT COUNT(*) FROM testTable WHERE column IS NULL", null_handling=True).cardinality() == 1000
    assert seg.filter("SELECT COUNT(*) FROM testTable WHERE column IS NOT NULL", null_handling=True).cardinality() == 0
    assert seg.filter("SELECT COUNT(*) FROM testTable WHERE column > 69", null_handling=True).cardinality() == 0
    # (not in the reference's test: NOT over a no-dictionary predicate leaves out the nulls; over a dictionary whose values cannot match, the
    # predicate is EmptyFilterOperator — two-valued — and its NOT matches every doc: FilterOperatorUtils.java:76-78,186-196)
    assert seg.filter("SELECT COUNT(*) FROM testTable WHERE NOT column > 69", null_handling=True).cardinality() == (1000 if dictionary else 0)
    seg.destroy()


@pytest.mark.parametrize("data_type", ["INT", "LONG", "FLOAT", "DOUBLE"])
@pytest.mark.parametrize("dictionary", [True, False])
def test_oracle_reproduces_all_null_queries_test(oracle_api, data_type, dictionary):
    all_null_cases(oracle_api, data_type, dictionary)


@pytest.mark.gpu
@pytest.mark.parametrize("data_type", ["INT", "LONG", "FLOAT", "DOUBLE"])
@pytest.mark.parametrize("dictionary", [True, False])
def test_gpu_reproduces_all_null_queries_test(gpu_api, data_type, dictionary):
    all_null_cases(gpu_api, data_type, dictionary)
