"""Multi-value columns in the CPU oracle: FixedBitMVForwardIndexReader, MVScanDocIdIterator + applyMV,
DictionaryBasedGroupKeyGenerator#processMultiValue (Cartesian expansion, repeats kept) and the *MV aggregation functions, checked
against a brute force over the rows (DictionaryBasedGroupKeyGeneratorTest.java:163-200 style) on random data.  The reference's own
multi-value numbers (MultiValueRawQueriesTest's formulaic table: COUNTMV / SUMMV / MINMV / MAXMV / AVGMV with and without filters and
multi-value group keys) are asserted in tests/test_mv_reference_goldens.py; filters over multi-value columns, DISTINCTCOUNTMV and the
entries-scanned statistics have no reference number and stay pinned by this brute force."""
import numpy as np
import pytest

from pinot_amd.executor import NativeSegment
from tests import mv_fixture as mv

N = 3000


@pytest.fixture(scope="module")
def table(oracle_api):
    rows = mv.make_rows(N)
    host = mv.build(rows)
    seg = NativeSegment(oracle_api, host)
    yield rows, seg
    seg.destroy()


def any_in(col, values):
    return lambda r: any(v in values for v in mv.values_of(r, col))


def none_in(col, values):
    return lambda r: all(v not in values for v in mv.values_of(r, col))


def entries(rows, col, candidates=lambda r: True):
    return sum(len(mv.values_of(r, col)) for r in rows if candidates(r))


# (WHERE clause, row predicate, numEntriesScannedInFilter as a function of the rows)
FILTERS = [
    # MVScanDocIdIterator: every doc evaluated, every value of it counted, any value matching is enough
    ("mv2 = 'cat'", any_in("mv2", {"cat"}), lambda rows: entries(rows, "mv2")),
    ("mv2 IN ('ant', 'lynx', 'zebra')", any_in("mv2", {"ant", "lynx"}), lambda rows: entries(rows, "mv2")),
    # exclusive predicates: every value has to pass
    ("mv2 != 'cat'", none_in("mv2", {"cat"}), lambda rows: entries(rows, "mv2")),
    ("mv2 NOT IN ('ant', 'bee', 'cat')", none_in("mv2", {"ant", "bee", "cat"}), lambda rows: entries(rows, "mv2")),
    # a RANGE never takes the inverted index: scan over dictIds
    ("mv1 BETWEEN 10 AND 19", lambda r: any(10 <= v <= 19 for v in mv.values_of(r, "mv1")), lambda rows: entries(rows, "mv1")),
    ("mv3 > 4000000", lambda r: any(v > 4000000 for v in mv.values_of(r, "mv3")), lambda rows: entries(rows, "mv3")),
    # inverted index on a multi-value column: one bitmap per dictId holding every doc with the value; NOT IN flips the union
    ("mv1 = 7", any_in("mv1", {7}), lambda rows: 0),
    ("mv1 IN (1, 2, 3)", any_in("mv1", {1, 2, 3}), lambda rows: 0),
    ("mv1 NOT IN (1, 2, 3)", none_in("mv1", {1, 2, 3}), lambda rows: 0),
    # index first, then the multi-value scan over the survivors only
    ("s1 = 3 AND mv2 = 'dog'", lambda r: r["s1"] == 3 and "dog" in r["mv2"], lambda rows: entries(rows, "mv2", lambda r: r["s1"] == 3)),
    ("mv1 = 5 AND mv2 != 'dog'", lambda r: 5 in r["mv1"] and "dog" not in r["mv2"], lambda rows: entries(rows, "mv2", lambda r: 5 in r["mv1"])),
    # a single-value scan is ordered before a multi-value scan (FilterOperatorUtils.java:253-265), whatever the query says
    ("mv2 = 'eel' AND m < 0 AND s1 IN (1, 2)", lambda r: "eel" in r["mv2"] and r["m"] < 0 and r["s1"] in (1, 2),
     lambda rows: sum(1 for r in rows if r["s1"] in (1, 2)) + entries(rows, "mv2", lambda r: r["s1"] in (1, 2) and r["m"] < 0)),
    # the default null value of an empty entry is a value like any other
    ("mv1 = -2147483648", lambda r: not r["mv1"], lambda rows: 0),
]


@pytest.mark.parametrize("where,pred,scanned", FILTERS)
def test_filters_over_multi_value_columns(table, where, pred, scanned):
    rows, seg = table
    b = seg.execute(f"SELECT COUNT(*), SUM(m) FROM mvTable WHERE {where}")
    want = [r for r in rows if pred(r)]
    assert b.aggregation_result() == [len(want), float(sum(r["m"] for r in want))]
    assert b.stats.num_docs_scanned == len(want)
    assert b.stats.num_entries_scanned_in_filter == scanned(rows)


def check(seg, rows, sql, where, group_by, aggs):
    got = seg.execute(sql).rows()
    want = mv.brute_force(rows, where, group_by, aggs)
    assert set(got) == set(want), sql
    for k, vals in want.items():
        for (fn, _), g, w in zip(aggs, got[k], vals):
            if fn in ("COUNT", "COUNTMV"):
                assert g == w, (sql, k, fn)
            elif fn in ("AVG", "AVGMV"):
                assert g[1] == w[1] and g[0] == pytest.approx(w[0], rel=1e-12), (sql, k, fn)
            elif fn in ("MINMAXRANGE", "MINMAXRANGEMV", "DISTINCTCOUNT", "DISTINCTCOUNTMV"):
                assert g == w, (sql, k, fn)
            else:
                assert g == pytest.approx(w, rel=1e-12), (sql, k, fn)


def test_group_by_a_multi_value_column(table):
    rows, seg = table
    # one multi-value column: the doc's dictIds are its keys (repeats aggregate the doc again)
    check(seg, rows, "SELECT mv1, COUNT(*), SUM(m), MAX(m) FROM mvTable GROUP BY mv1 LIMIT 1000", lambda r: True, ["mv1"],
          [("COUNT", None), ("SUM", "m"), ("MAX", "m")])
    # single-value x multi-value, both orders (the raw key is built from the last column down)
    check(seg, rows, "SELECT s1, mv2, COUNT(*), MIN(m) FROM mvTable WHERE m >= -500 GROUP BY s1, mv2 LIMIT 1000", lambda r: r["m"] >= -500,
          ["s1", "mv2"], [("COUNT", None), ("MIN", "m")])
    check(seg, rows, "SELECT mv2, s2, AVG(m), MINMAXRANGE(m) FROM mvTable GROUP BY mv2, s2 LIMIT 1000", lambda r: True, ["mv2", "s2"],
          [("AVG", "m"), ("MINMAXRANGE", "m")])
    # two multi-value columns: every combination
    check(seg, rows, "SELECT mv1, mv2, COUNT(*), SUM(m) FROM mvTable WHERE s1 < 4 GROUP BY mv1, mv2 LIMIT 10000", lambda r: r["s1"] < 4,
          ["mv1", "mv2"], [("COUNT", None), ("SUM", "m")])
    check(seg, rows, "SELECT mv3, s1, mv1, COUNT(*), DISTINCTCOUNT(s2) FROM mvTable WHERE mv2 = 'fox' GROUP BY mv3, s1, mv1 LIMIT 100000",
          lambda r: "fox" in r["mv2"], ["mv3", "s1", "mv1"], [("COUNT", None), ("DISTINCTCOUNT", "s2")])


def test_multi_value_aggregation_functions(table):
    rows, seg = table
    aggs = [("COUNTMV", "mv1"), ("SUMMV", "mv1"), ("MINMV", "mv3"), ("MAXMV", "mv3"), ("AVGMV", "mv1"), ("MINMAXRANGEMV", "mv3"),
            ("DISTINCTCOUNTMV", "mv2"), ("COUNT", None)]
    select = "COUNTMV(mv1), SUMMV(mv1), MINMV(mv3), MAXMV(mv3), AVGMV(mv1), MINMAXRANGEMV(mv3), DISTINCTCOUNTMV(mv2), COUNT(*)"
    # no GROUP BY (AggregationOperator), single-value keys (aggregateGroupBySV), multi-value keys (aggregateGroupByMV)
    check(seg, rows, f"SELECT {select} FROM mvTable WHERE m > 0", lambda r: r["m"] > 0, [], aggs)
    check(seg, rows, f"SELECT s1, {select} FROM mvTable WHERE mv1 NOT IN (3, 4) GROUP BY s1 LIMIT 100", none_in("mv1", {3, 4}), ["s1"], aggs)
    check(seg, rows, f"SELECT mv2, {select} FROM mvTable GROUP BY mv2 LIMIT 100", lambda r: True, ["mv2"], aggs)
    check(seg, rows, f"SELECT mv1, s2, {select} FROM mvTable WHERE s1 = 2 GROUP BY mv1, s2 LIMIT 10000", lambda r: r["s1"] == 2, ["mv1", "s2"], aggs)


def test_distinct_count_hll_mv_offers_every_value(table, oracle_api):
    rows, seg = table
    got = seg.execute("SELECT s1, DISTINCTCOUNTHLLMV(mv1) FROM mvTable GROUP BY s1 LIMIT 100").rows()
    want = mv.brute_force(rows, lambda r: True, ["s1"], [("DISTINCTCOUNTMV", "mv1")])
    for k, (vals,) in want.items():
        v = np.array(sorted(vals), dtype=np.int64)
        regs = np.zeros(256, dtype=np.uint8)
        oracle_api.lib.po_hll_registers_for_values(v.ctypes.data, len(v), 1, 8, regs.ctypes.data)
        assert got[k][0] == bytes(regs), k


def test_dictionary_answers_without_a_scan(table):
    """NonScanBasedAggregationOperator: MINMV / MAXMV / MINMAXRANGEMV / DISTINCTCOUNTMV over a dictionary column, no filter, no GROUP BY."""
    rows, seg = table
    b = seg.execute("SELECT MINMV(mv3), MAXMV(mv3), MINMAXRANGEMV(mv1), DISTINCTCOUNTMV(mv2), COUNT(*) FROM mvTable")
    all3 = [v for r in rows for v in mv.values_of(r, "mv3")]
    all1 = [v for r in rows for v in mv.values_of(r, "mv1")]
    assert b.aggregation_result() == [float(min(all3)), float(max(all3)), (float(min(all1)), float(max(all1))),
                                      frozenset(v for r in rows for v in r["mv2"]), len(rows)]
    assert b.stats.num_entries_scanned_post_filter == 0


def test_mismatched_functions_are_rejected(table):
    rows, seg = table
    with pytest.raises(Exception):
        seg.execute("SELECT SUM(mv1) FROM mvTable WHERE m > 0")
    with pytest.raises(Exception):
        seg.execute("SELECT SUMMV(m) FROM mvTable WHERE m > 0")


def test_reader_context_paths():
    """FixedBitMVForwardIndexReader#getDictIdMV: sequential docs, forward jumps inside a chunk, jumps across chunks, backward jumps —
    every path of the context logic lands on the same entries (the brute force above reads them through sparse filters)."""
    from pinot_amd.segment import build_mv_column, decode_mv_column
    rng = np.random.default_rng(3)
    for n, hi in ((1, 3), (5, 2), (4097, 6), (20000, 12)):
        rows = [rng.integers(0, 300, rng.integers(1, hi)).tolist() for _ in range(n)]
        col = build_mv_column("x", rows, "INT")
        dec = decode_mv_column(col, n)
        assert [[col.dict_values[i] for i in d] for d in dec] == rows


# ---- raw (no-dictionary) multi-value columns: FixedByteChunkMVForwardIndexReader's format in the oracle ------------------------------------
def test_raw_multi_value_reader_matches_the_brute_force():
    """The oracle over raw twins of the multi-value columns (every ChunkCompressionType of the var-byte chunk layout) against the brute
    force over the rows: filters with their numEntriesScannedInFilter, multi-value group keys, the *MV functions."""
    from tests.oracle_binding import load_oracle
    rows = mv.make_rows(N, seed=23)
    for r in rows:
        r["r1"], r["r3"], r["rh"] = r["mv1"], r["mv3"], r["mvh"]
    seg = NativeSegment(load_oracle(), mv.build_with_raw_twins(rows))
    b = seg.execute("SELECT COUNT(*), SUM(m) FROM mvTable WHERE r1 BETWEEN 10 AND 19")
    match = [r for r in rows if any(10 <= v <= 19 for v in mv.values_of(r, "r1"))]
    assert b.rows()[()] == [len(match), float(sum(r["m"] for r in match))]
    assert b.stats.num_entries_scanned_in_filter == sum(len(mv.values_of(r, "r1")) for r in rows)
    want = mv.brute_force(rows, lambda r: r["s1"] < 4, ["r1", "s1"], [("COUNT", None), ("SUMMV", "r3"), ("MAXMV", "rh")])
    got = seg.execute("SELECT r1, s1, COUNT(*), SUMMV(r3), MAXMV(rh) FROM mvTable WHERE s1 < 4 GROUP BY r1, s1 LIMIT 100000").rows()
    assert got == want
    seg.destroy()
