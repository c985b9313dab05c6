"""The reference's filter-operator unit tests as goldens: AndFilterOperatorTest, OrFilterOperatorTest, NotFilterOperatorTest
(pinot-core/src/test/java/org/apache/pinot/core/operator/filter/) give explicit docId lists for AND / OR / NOT combinations of
child operators, including nested AND-in-AND, OR-in-AND and OR-in-OR trees that the SQL front end would flatten.  Each child list
becomes a 0/1 flag column here, in three physical forms (inverted-index leaf, dictionary scan leaf, raw scan leaf), so that every
leaf kind must produce the reference's docIds — on the oracle (CPU) and on the HIP path."""
import itertools

import numpy as np
import pytest

from pinot_amd.executor import NativeSegment
from pinot_amd.query import FilterContext, Predicate, QueryContext
from pinot_amd.segment import build_segment

L1 = [2, 3, 6, 10, 15, 16, 28]
L1S = [2, 3, 10, 15, 16, 28]            # the two-list tests use a shorter first list
L2 = [3, 6, 8, 20, 28]
L3 = [1, 2, 3, 6, 30]
NOT_LIST = [2, 3, 10, 15, 16, 17, 18, 21, 22, 23, 24, 26, 28]


def leaf(i, form):
    return FilterContext.pred(Predicate("EQ", f"{form}{i}", ["1"]))


def cases(form_of):
    """(name, numDocs, lists, tree builder, expected docIds) — file:line of the golden in the comment"""
    f = lambda i: leaf(i, form_of(i))   # noqa: E731
    return [
        ("and_two", 40, [L1S, L2], lambda: FilterContext.and_([f(1), f(2)]), [3, 28]),                                   # AndFilterOperatorTest.java:36-50
        ("and_three", 40, [L1, L2, L3], lambda: FilterContext.and_([f(1), f(2), f(3)]), [3, 6]),                         # :53-69
        ("and_nested", 40, [L1, L2, L3], lambda: FilterContext.and_([FilterContext.and_([f(1), f(2)]), f(3)]), [3, 6]),  # :72-92 testComplex
        ("and_of_or", 40, [L1, L2, L3], lambda: FilterContext.and_([FilterContext.or_([f(3), f(2)]), f(1)]), [2, 3, 6, 28]),  # :141-163 testComplexWithOr
        ("or_two", 40, [L1S, L2], lambda: FilterContext.or_([f(1), f(2)]), sorted(set(L1S) | set(L2))),                  # OrFilterOperatorTest.java:38-57
        ("or_three", 40, [L1, L2, L3], lambda: FilterContext.or_([f(1), f(2), f(3)]), sorted(set(L1) | set(L2) | set(L3))),   # :60-82
        ("or_nested", 40, [L1, L2, L3], lambda: FilterContext.or_([FilterContext.or_([f(1), f(2)]), f(3)]), sorted(set(L1) | set(L2) | set(L3))),  # :85-111
        ("or_null_handling_disabled", 10, [[1, 2, 3], [0, 1, 2]], lambda: FilterContext.or_([f(1), f(2)]), [0, 1, 2, 3]),     # :130-143 getTrues
        ("not", 30, [NOT_LIST], lambda: FilterContext.not_(f(1)), [0, 1, 4, 5, 6, 7, 8, 9, 11, 12, 13, 14, 19, 20, 25, 27, 29]),   # NotFilterOperatorTest.java:35-45
        ("not_of_or_is_the_falses", 10, [[1, 2, 3], [0, 1, 2]], lambda: FilterContext.not_(FilterContext.or_([f(1), f(2)])), [4, 5, 6, 7, 8, 9]),  # OrFilterOperatorTest.java:142 getFalses
    ]


def segment_for(num_docs, lists):
    data, schema, inv, raw = {}, {}, [], []
    for i, ids in enumerate(lists, start=1):
        flag = np.zeros(num_docs, dtype=np.int32)
        flag[ids] = 1
        for form in ("inv", "dic", "raw"):
            data[f"{form}{i}"] = flag
            schema[f"{form}{i}"] = "INT"
        inv.append(f"inv{i}")
        raw.append(f"raw{i}")
    return build_segment("filterGoldens_0", data, schema, inverted_index_columns=inv, no_dictionary_columns=raw)


FORMS = [("inv",) * 3, ("dic",) * 3, ("raw",) * 3, ("inv", "raw", "dic"), ("raw", "inv", "inv"), ("dic", "dic", "inv")]


def run_all(api):
    for forms in FORMS:
        for name, n, lists, build, expected in cases(lambda i: forms[i - 1]):
            seg = NativeSegment(api, segment_for(n, lists))
            q = QueryContext(table="t", filter=build())
            d = seg.filter(q)
            assert d.doc_ids().tolist() == expected, (name, forms)
            assert d.cardinality() == len(expected)
            seg.destroy()


def test_reference_filter_operator_goldens_oracle(oracle_api):
    run_all(oracle_api)


def test_and_doc_id_set_reordering_golden(oracle_api):
    """AndFilterOperatorTest#testAndDocIdSetReordering (:94-138): four bitmaps (multiples of 2, 3, 4, 5 below 10 000) in either
    child order give 0, 60, 120, 180, ..."""
    n = 10_000
    data = {f"m{k}": (np.arange(n) % k == 0).astype(np.int32) for k in (2, 3, 4, 5)}
    host = build_segment("reorder_0", data, {c: "INT" for c in data}, inverted_index_columns=list(data))
    seg = NativeSegment(oracle_api, host)
    for order in ((2, 3, 4, 5), (5, 4, 3, 2)):
        q = QueryContext(table="t", filter=FilterContext.and_([FilterContext.pred(Predicate("EQ", f"m{k}", ["1"])) for k in order]))
        ids = seg.filter(q).doc_ids()
        assert ids[:4].tolist() == [0, 60, 120, 180] and ids.tolist() == list(range(0, n, 60))
    seg.destroy()


@pytest.mark.gpu
def test_reference_filter_operator_goldens_gpu(gpu_api):
    run_all(gpu_api)
    n = 10_000
    data = {f"m{k}": (np.arange(n) % k == 0).astype(np.int32) for k in (2, 3, 4, 5)}
    host = build_segment("reorder_0", data, {c: "INT" for c in data}, inverted_index_columns=list(data))
    seg = NativeSegment(gpu_api, host)
    for order in itertools.permutations((2, 3, 4, 5)):
        q = QueryContext(table="t", filter=FilterContext.and_([FilterContext.pred(Predicate("EQ", f"m{k}", ["1"])) for k in order]))
        assert seg.filter(q).doc_ids().tolist() == list(range(0, n, 60))
    seg.destroy()


# ---- BitmapCollectionTest (pinot-core/src/test/.../operator/filter/BitmapCollectionTest.java:31-128 and :130-227): and / or
# cardinalities of two bitmaps under inversion, numDocs = 10 — what FastFilteredCountOperator answers from
# (BaseFilterOperator#getNumMatchingDocs → BitmapCollection).  A bitmap is the null value vector of a column here: inverted =
# IS NOT NULL (BitmapBasedFilterOperator exclusive), plain = IS NULL.  (numDocs, left, leftInverted, right, rightInverted, expected)
AND_CARDINALITY = [(10, [0, 5], False, [0, 4], False, 1), (10, [0, 5], False, [1, 4], False, 0), (10, [0, 5], False, [], False, 0), (10, [], False, [0, 5], False, 0), (10, [], False, [], False, 0), (10, [0, 5], True, [0, 4], False, 1), (10, [0, 5], True, [1, 4], False, 2), (10, [0, 5], True, [], False, 0), (10, [], True, [0, 5], False, 2), (10, [], True, [], False, 0), (10, [0, 5], False, [0, 4], True, 1), (10, [0, 5], False, [1, 4], True, 2), (10, [0, 5], False, [], True, 2), (10, [], False, [], True, 0), (10, [], False, [0, 5], True, 0), (10, [0, 5], True, [0, 4], True, 7), (10, [0, 5], True, [1, 4], True, 6), (10, [0, 5], True, [], True, 8), (10, [], True, [0, 5], True, 8), (10, [], True, [], True, 10)]
OR_CARDINALITY = [(10, [0, 5], False, [0, 4], False, 3), (10, [0, 5], False, [1, 4], False, 4), (10, [0, 5], False, [], False, 2), (10, [], False, [0, 5], False, 2), (10, [], False, [], False, 0), (10, [0, 5], True, [0, 4], False, 9), (10, [0, 5], True, [1, 4], False, 8), (10, [0, 5], True, [], False, 8), (10, [], True, [0, 5], False, 10), (10, [], True, [], False, 10), (10, [0, 5], False, [0, 4], True, 9), (10, [0, 5], False, [1, 4], True, 8), (10, [0, 5], False, [], True, 10), (10, [], False, [0, 5], True, 8), (10, [], False, [], True, 10), (10, [0, 5], True, [0, 4], True, 9), (10, [0, 5], True, [1, 4], True, 10), (10, [0, 5], True, [], True, 10), (10, [], True, [0, 5], True, 10), (10, [], True, [], True, 10)]


def bitmap_collection_cases(api):
    from pinot_amd import formats
    for op, table in (("AND", AND_CARDINALITY), ("OR", OR_CARDINALITY)):
        for n, left, l_inv, right, r_inv, expected in table:
            host = build_segment("bc_0", {"a": np.arange(n, dtype=np.int32), "b": np.arange(n, dtype=np.int32)}, {"a": "INT", "b": "INT"},
                                 no_dictionary_columns=["a", "b"])
            host.columns["a"].null_vector = np.frombuffer(formats.serialize_roaring(np.array(left, dtype=np.int64)), dtype=np.uint8)
            host.columns["b"].null_vector = np.frombuffer(formats.serialize_roaring(np.array(right, dtype=np.int64)), dtype=np.uint8)
            seg = NativeSegment(api, host)
            where = f"a IS {'NOT ' if l_inv else ''}NULL {op} b IS {'NOT ' if r_inv else ''}NULL"
            b = seg.execute(f"SELECT COUNT(*) FROM t WHERE {where}")
            assert b.aggregation_result() == [expected], (where, left, right)
            assert b.stats.num_entries_scanned_in_filter == 0 and b.stats.num_docs_scanned == expected   # FastFilteredCountOperator
            assert seg.filter(f"SELECT COUNT(*) FROM t WHERE {where}").cardinality() == expected
            seg.destroy()


def test_bitmap_collection_cardinality_goldens_oracle(oracle_api):
    bitmap_collection_cases(oracle_api)


@pytest.mark.gpu
def test_bitmap_collection_cardinality_goldens_gpu(gpu_api):
    bitmap_collection_cases(gpu_api)


# ---- RangePredicateWithSortedInvertedIndexTest#testInnerSegmentQuery (pinot-core/src/test/.../queries/, :150-188): 30 000 rows,
# INT_COL = row index (sorted, dictionary → SortedIndexBasedFilterOperator), INT_COL_RAW = row index (sorted, no dictionary → scan);
# (filter, matching docs, docId ranges)
SORTED_RANGE_CASES = [
    ("INT_COL >= 20000", 10000, [(20000, 29999)]),
    ("INT_COL >= 20000 AND INT_COL_RAW >= 20000", 10000, [(20000, 29999)]),
    ("INT_COL >= 20000 AND INT_COL <= 23666", 3667, [(20000, 23666)]),
    ("INT_COL >= 20000 AND INT_COL <= 23666 AND INT_COL_RAW <= 23666", 3667, [(20000, 23666)]),
    ("INT_COL <= 20000", 20001, [(0, 20000)]),
    ("INT_COL_RAW = 20000", 1, [(20000, 20000)]),
    ("(INT_COL >= 15000 AND INT_COL <= 16665) OR (INT_COL >= 18000 AND INT_COL <= 19887)", 3554, [(15000, 16665), (18000, 19887)]),
]


def sorted_range_cases(api):
    n = 30000
    rng = np.random.default_rng(12)
    data = {"INT_COL": np.arange(n, dtype=np.int32), "INT_COL_RAW": np.arange(n, dtype=np.int32),
            "LONG_COL": rng.integers(-(1 << 62), 1 << 62, n)}
    host = build_segment("sortedRange_0", data, {"INT_COL": "INT", "INT_COL_RAW": "INT", "LONG_COL": "LONG"},
                         no_dictionary_columns=["INT_COL_RAW"])
    assert host.columns["INT_COL"].is_sorted
    seg = NativeSegment(api, host)
    for where, count, ranges in SORTED_RANGE_CASES:
        d = seg.filter(f"SELECT COUNT(*) FROM t WHERE {where}")
        expected = np.concatenate([np.arange(lo, hi + 1) for lo, hi in ranges])
        assert d.cardinality() == count == len(expected), where
        np.testing.assert_array_equal(d.doc_ids(), expected, err_msg=where)
        got = seg.execute(f"SELECT COUNT(*), MIN(INT_COL), MAX(INT_COL_RAW) FROM t WHERE {where}").aggregation_result()
        assert got == [count, float(ranges[0][0]), float(ranges[-1][1])], where
    # :190-226: the sorted range ANDed with a scan over an unsorted LONG column
    pivot = int(data["LONG_COL"][rng.integers(0, n)])
    want = np.flatnonzero((data["LONG_COL"] >= pivot) & (np.arange(n) >= 15000) & (np.arange(n) <= 16665))
    d = seg.filter(f"SELECT COUNT(*) FROM t WHERE INT_COL >= 15000 AND INT_COL <= 16665 AND LONG_COL >= {pivot}")
    np.testing.assert_array_equal(d.doc_ids(), want)
    seg.destroy()


def test_range_predicate_with_sorted_index_goldens_oracle(oracle_api):
    sorted_range_cases(oracle_api)


@pytest.mark.gpu
def test_range_predicate_with_sorted_index_goldens_gpu(gpu_api):
    sorted_range_cases(gpu_api)


# ---- NotOperatorQueriesTest#testRangePredicates / #testCompositePredicates (pinot-core/src/test/.../queries/, :169-188): 1 024 rows,
# FIRST_INT_COL = i, SECOND_INT_COL = 1000 + i; per-segment COUNT(*) (the LIKE / REGEXP_LIKE cases are outside the path)
NOT_OPERATOR_CASES = [
    ("NOT FIRST_INT_COL = 5", 1023), ("NOT FIRST_INT_COL < 5", 1019), ("NOT FIRST_INT_COL > 5", 6),
    ("FIRST_INT_COL NOT BETWEEN 10 AND 20", 1013), ("NOT FIRST_INT_COL BETWEEN 10 AND 20", 1013),
    ("NOT (FIRST_INT_COL > 5 AND SECOND_INT_COL < 1009)", 1021), ("NOT FIRST_INT_COL > 5 OR NOT SECOND_INT_COL < 1009", 1021),
    ("NOT (FIRST_INT_COL < 5 OR SECOND_INT_COL > 2000)", 996), ("NOT FIRST_INT_COL < 5 AND NOT SECOND_INT_COL > 2000", 996),
]


def not_operator_cases(api):
    i = np.arange(1024, dtype=np.int32)
    for raw in ((), ("SECOND_INT_COL",), ("FIRST_INT_COL", "SECOND_INT_COL")):   # the reference's layout, then raw forms of the same data
        host = build_segment("notOperator_0", {"FIRST_INT_COL": i, "SECOND_INT_COL": i + 1000},
                             {"FIRST_INT_COL": "INT", "SECOND_INT_COL": "INT"}, no_dictionary_columns=list(raw))
        seg = NativeSegment(api, host)
        for where, expected in NOT_OPERATOR_CASES:
            assert seg.execute(f"SELECT COUNT(*) FROM testTable WHERE {where}").aggregation_result() == [expected], (where, raw)
            assert seg.filter(f"SELECT COUNT(*) FROM testTable WHERE {where}").cardinality() == expected
        seg.destroy()


def test_not_operator_goldens_oracle(oracle_api):
    not_operator_cases(oracle_api)


@pytest.mark.gpu
def test_not_operator_goldens_gpu(gpu_api):
    not_operator_cases(gpu_api)


# ---- OrDocIdSet.iterator() (OrDocIdSet.java:62-125) hands an enclosing AND a bitmap only when it merged >= 2 sorted children:
# only then are the AND's scans applied to the OR's docs (numEntriesScannedInFilter); any other OR is leapfrogged.
def or_iterator_typing(api, gpu=False):
    n = 50_000
    rng = np.random.default_rng(21)
    data = {"so": np.sort(rng.integers(0, 200, n)).astype(np.int32), "d": rng.integers(0, 10, n).astype(np.int32),
            "e": rng.integers(0, 10, n).astype(np.int32), "r": rng.integers(0, 1000, n).astype(np.int32)}
    host = build_segment("orTyping_0", data, {c: "INT" for c in data}, inverted_index_columns=["d", "e"], no_dictionary_columns=["r"])
    seg = NativeSegment(api, host)
    so, d, e, r = (data[c] for c in ("so", "d", "e", "r"))
    cases = [   # (filter, docs the scan over r is applied to)
        ("(so < 20 OR so > 150) AND r < 500", (so < 20) | (so > 150)),                 # two sorted leaves: merged bitmap
        ("(so < 20 OR so > 150 OR d = 3) AND r < 500", (so < 20) | (so > 150) | (d == 3)),   # ... bitmap children ORed in
        ("(so < 20 OR so > 150) AND d = 3 AND r < 500", ((so < 20) | (so > 150)) & (d == 3)),
        ("(d = 3 OR e = 4) AND so < 100 AND r < 500", so < 100),                       # no merge: the OR is leapfrogged
        ("(so < 20 OR d = 3) AND e = 4 AND r < 500", e == 4),                          # one sorted child: no merge either
    ]
    results = []
    for where, cand in cases:
        b = seg.execute(f"SELECT COUNT(*), SUM(r) FROM t WHERE {where}")
        results.append((b.aggregation_result(), b.stats.num_entries_scanned_in_filter, b.stats.stats_exact))
        if not gpu:
            assert b.stats.num_entries_scanned_in_filter == int(cand.sum()), where
    seg.destroy()
    return results


def test_or_iterator_typing_oracle(oracle_api):
    or_iterator_typing(oracle_api)


@pytest.mark.gpu
def test_or_iterator_typing_gpu(gpu_api, oracle_api):
    for (gr, ge, gx), (orr, oe, _) in zip(or_iterator_typing(gpu_api, gpu=True), or_iterator_typing(oracle_api)):
        assert gr == orr
        assert gx and ge == oe
