"""GPU parity: libpinot_gpu.so (HIP kernels, through the C ABI) vs the CPU oracle and the reference's golden numbers.

Bit-exact for counts, docId sets, group keys, MIN/MAX and integer-valued SUMs (all sums here stay below 2^53, where the
reference's sequential double accumulation is exact and order independent — SURVEY.md §7 "Floating SUM parity").
"""
import numpy as np
import pytest

from pinot_amd import synth
from pinot_amd.executor import NativeSegment
from pinot_amd.segment import build_segment
from tests.fixtures import SV_FILTER, sv_segment
from tests.test_oracle_goldens import AGGREGATION_QUERY, FFC_CASES, RANGE_CASES, check_agg

pytestmark = pytest.mark.gpu


def both(gpu_api, oracle_api, host):
    return NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)


def assert_same_block(g, o, check_stats=True):
    assert sorted(g.rows().keys()) == sorted(o.rows().keys())
    gr, orr = g.rows(), o.rows()
    for k in orr:
        assert gr[k] == orr[k], (k, gr[k], orr[k])
    assert g.stats.num_docs_scanned == o.stats.num_docs_scanned
    assert g.stats.num_total_docs == o.stats.num_total_docs
    assert g.stats.num_entries_scanned_post_filter == o.stats.num_entries_scanned_post_filter
    if check_stats:
        assert g.stats.stats_exact == 1
        assert g.stats.num_entries_scanned_in_filter == o.stats.num_entries_scanned_in_filter


# ---- the reference's inner-segment goldens, on the GPU ------------------------------------------------------------------
@pytest.fixture(scope="module")
def sv(gpu_api, oracle_api, sv_data):
    host = sv_segment(sv_data)
    g, o = both(gpu_api, oracle_api, host)
    yield g, o
    g.destroy()
    o.destroy()


def test_golden_aggregation_only(sv):
    g, _ = sv
    b = g.execute(AGGREGATION_QUERY)
    check_agg(b.aggregation_result(), 30000, 32317185437847, 2147419555, 1689277, 28175373944314, 30000)
    st = b.execution_statistics()
    assert (st.num_docs_scanned, st.num_entries_scanned_in_filter, st.num_entries_scanned_post_filter,
            st.num_total_docs) == (30000, 0, 120000, 30000)
    b = g.execute(AGGREGATION_QUERY + SV_FILTER)
    check_agg(b.aggregation_result(), 6129, 6875947596072, 999813884, 1980174, 4699510391301, 6129)
    st = b.execution_statistics()
    # InnerSegmentAggregationSingleValueQueriesTest.java:43-60: (numDocsScanned, numEntriesScannedInFilter, numEntriesScannedPostFilter,
    # numTotalDocs) — 63064 is what the reference's iterators scan for the OR (column6 scan, column11 inverted) leapfrogged by the AND
    assert (st.num_docs_scanned, st.num_entries_scanned_in_filter, st.num_entries_scanned_post_filter, st.num_total_docs) == (6129, 63064, 24516, 30000)
    assert b.stats.stats_exact == 1


def test_golden_group_by(sv):
    g, _ = sv
    b = g.execute(AGGREGATION_QUERY + " GROUP BY column9")
    check_agg(b.rows()[(11270,)], 1, 815409257, 1215316262, 1328642550, 788414092, 1)
    b = g.execute(AGGREGATION_QUERY + SV_FILTER + " GROUP BY column9")
    check_agg(b.rows()[(242920,)], 3, 4348938306, 407993712, 296467636, 5803888725, 3)
    b = g.execute(AGGREGATION_QUERY + " GROUP BY column9, column11, column12")
    check_agg(b.rows()[(1813102948, "P", "HEuxNvH")], 4, 2062187196, 1988589001, 394608493, 4782388964, 4)
    b = g.execute(AGGREGATION_QUERY + SV_FILTER + " GROUP BY column9, column11, column12")
    check_agg(b.rows()[(1176631727, "P", "KrNxpdycSiwoRohEiTIlLqDHnx")], 1, 716185211, 489993380, 371110078,
              487714191, 1)


SV_QUERIES = [
    AGGREGATION_QUERY,
    AGGREGATION_QUERY + SV_FILTER,
    AGGREGATION_QUERY + " GROUP BY column9",
    AGGREGATION_QUERY + SV_FILTER + " GROUP BY column9",
    AGGREGATION_QUERY + " GROUP BY column9, column11, column12",
    AGGREGATION_QUERY + SV_FILTER + " GROUP BY column11, column12",
    "SELECT column11, SUM(column1) FROM testTable GROUP BY column11",
    "SELECT column11, column12, SUM(column1), MIN(column3), MAX(column17), COUNT(*) FROM testTable GROUP BY column11, column12",
    "SELECT COUNT(*) FROM testTable WHERE column11 NOT IN ('t', 'P')",
    "SELECT COUNT(*), SUM(column18) FROM testTable WHERE column6 < 500000000 OR column11 NOT IN ('t', 'P')",
    "SELECT COUNT(*), MAX(column1) FROM testTable WHERE NOT (column7 IN (1, 2, 3, 5, 8) OR column17 = 635553468)",
    "SELECT MINMAXRANGE(column6), AVG(column18) FROM testTable WHERE column17 > 1000 AND column18 <= 1000000000 GROUP BY column7",
    "SELECT COUNT(*) FROM testTable WHERE daysSinceEpoch = 126164076 AND column11 = 'P' AND column1 > 1500000000",
    "SELECT COUNT(*) FROM testTable WHERE column5 = 'gFuH'",
    "SELECT COUNT(*) FROM testTable WHERE column5 = 'nope'",
    "SELECT SUM(column1) FROM testTable WHERE column9 = -1",
]


NON_SCAN_QUERIES = [
    "SELECT COUNT(*), MAX(column3), MIN(column6), MINMAXRANGE(column1), DISTINCTCOUNT(column1), DISTINCTCOUNTHLL(column3) FROM testTable",
    "SELECT MAX(column17) FROM testTable",
    "SELECT DISTINCTCOUNT(column11), DISTINCTCOUNTHLL(column12), COUNT(*) FROM testTable",
]


@pytest.mark.parametrize("q", NON_SCAN_QUERIES)
def test_non_scan_based_aggregation_matches_oracle(sv, q):
    g, o = sv
    gb, ob = g.execute(q), o.execute(q)
    assert_same_block(gb, ob)
    assert (gb.stats.num_docs_scanned, gb.stats.num_entries_scanned_post_filter) == (30000, 0)
    assert gb.stats.kernel == b""                      # answered from the dictionaries on the host: no kernel ran


DISTINCT_QUERIES = [
    "SELECT DISTINCTCOUNTHLL(column1), DISTINCTCOUNTHLL(column3) FROM testTable",
    "SELECT DISTINCTCOUNTHLL(column1), DISTINCTCOUNTHLL(column3) FROM testTable" + SV_FILTER,
    "SELECT DISTINCTCOUNT(column1), DISTINCTCOUNT(column3) FROM testTable",
    "SELECT DISTINCTCOUNT(column1), DISTINCTCOUNT(column3), COUNT(*), SUM(column6) FROM testTable" + SV_FILTER,
    "SELECT DISTINCTCOUNT(column11), DISTINCTCOUNTHLL(column12), DISTINCTCOUNTHLL(column5) FROM testTable GROUP BY column7",
    "SELECT column11, DISTINCTCOUNT(column17), DISTINCTCOUNTHLL(column18), MAX(column1) FROM testTable WHERE column6 < 500000000 GROUP BY column11",
    "SELECT DISTINCTCOUNTHLL(column1), COUNT(*) FROM testTable GROUP BY column9, column11",
    "SELECT DISTINCTCOUNT(column9) FROM testTable WHERE column9 = -5",
]


def test_golden_distinct_counts(sv):
    """InterSegmentAggregationSingleValueQueriesTest.java:261-274 (HLL) and the DISTINCTCOUNT goldens, on the GPU."""
    from pinot_amd.executor import extract_final
    g, _ = sv
    b = g.execute(DISTINCT_QUERIES[0])
    assert [extract_final("DISTINCTCOUNTHLL", v) for v in b.aggregation_result()] == [5977, 23825]
    b = g.execute(DISTINCT_QUERIES[1])
    assert [extract_final("DISTINCTCOUNTHLL", v) for v in b.aggregation_result()] == [1886, 4492]
    b = g.execute(DISTINCT_QUERIES[2])
    assert [extract_final("DISTINCTCOUNT", v) for v in b.aggregation_result()] == [6582, 21910]
    b = g.execute("SELECT DISTINCTCOUNT(column1), DISTINCTCOUNT(column3) FROM testTable" + SV_FILTER)
    assert [extract_final("DISTINCTCOUNT", v) for v in b.aggregation_result()] == [1872, 4556]


@pytest.mark.parametrize("q", DISTINCT_QUERIES)
def test_distinct_queries_match_oracle(sv, q):
    g, o = sv
    assert_same_block(g.execute(q), o.execute(q))


@pytest.mark.parametrize("q", SV_QUERIES)
def test_sv_queries_match_oracle(sv, q):
    g, o = sv
    assert_same_block(g.execute(q), o.execute(q))


@pytest.mark.parametrize("q", SV_QUERIES)
def test_sv_filters_match_oracle(sv, q):
    g, o = sv
    dg, do = g.filter(q), o.filter(q)
    assert dg.cardinality() == do.cardinality()
    np.testing.assert_array_equal(dg.words(), do.words())
    np.testing.assert_array_equal(dg.doc_ids(), do.doc_ids())


# ---- FastFilteredCountTest / RangeQueriesTest fixtures on the GPU ---------------------------------------------------------
@pytest.fixture(scope="module")
def ffc(gpu_api):
    i = np.arange(1000)
    data = {"class": (i % 8).astype(np.int32), "sorted": i.astype(np.int32), "intRangeCol": (1000 - i).astype(np.int32)}
    host = build_segment("FastFilteredCountTest", data, {"class": "INT", "sorted": "INT", "intRangeCol": "INT"},
                         inverted_index_columns=["class"])
    s = NativeSegment(gpu_api, host)
    yield s
    s.destroy()


@pytest.mark.parametrize("flt,expected", FFC_CASES)
def test_fast_filtered_count_gpu(ffc, flt, expected):
    b = ffc.execute("select count(*) from testTable" + flt)
    assert b.aggregation_result()[0] == expected, flt
    st = b.execution_statistics()
    assert st.num_docs_scanned == expected and st.num_total_docs == 1000


@pytest.fixture(scope="module")
def rng_seg(gpu_api):
    i = np.arange(1000, dtype=np.int64)
    v = ((100000 + 500) - i * 100) % 100000
    data = {"dictionarized": v.astype(np.int32), "rawInt": v.astype(np.int32), "rawLong": v.astype(np.int64),
            "rawFloat": v.astype(np.float32), "rawDouble": v.astype(np.float64)}
    schema = {"dictionarized": "INT", "rawInt": "INT", "rawLong": "LONG", "rawFloat": "FLOAT", "rawDouble": "DOUBLE"}
    host = build_segment("RangeQueriesTest", data, schema,
                         no_dictionary_columns=["rawInt", "rawLong", "rawFloat", "rawDouble"])
    s = NativeSegment(gpu_api, host)
    yield s, v
    s.destroy()


@pytest.mark.parametrize("col", ["dictionarized", "rawInt", "rawLong", "rawFloat", "rawDouble"])
@pytest.mark.parametrize("case", range(len(RANGE_CASES)))
@pytest.mark.parametrize("bounds", [(250, 500), (0, 99900), (-1, 100000), (20000, 20300), (450, 450)])
def test_range_queries_gpu(rng_seg, col, case, bounds):
    seg, v = rng_seg
    tmpl, fn = RANGE_CASES[case]
    lo, hi = bounds
    where = tmpl.format(c=col, lo=lo, hi=hi)
    m = fn(v, lo, hi)
    b = seg.execute(f"SELECT COUNT(*), SUM({col}), MIN({col}), MAX({col}) FROM testTable WHERE {where}")
    r = b.aggregation_result()
    assert r[0] == int(m.sum()), where
    if m.any():
        assert r[1] == float(v[m].sum()) and r[2] == float(v[m].min()) and r[3] == float(v[m].max())
    else:
        assert r[1] == 0.0 and r[2] == float("inf") and r[3] == float("-inf")
    np.testing.assert_array_equal(seg.filter(f"SELECT COUNT(*) FROM t WHERE {where}").doc_ids(), np.flatnonzero(m))


# ---- synthetic gpuBench segments: BASELINE.json configs at oracle-checkable sizes ------------------------------------------
SYNTH_QUERIES = [
    synth.QUERY_CFG2,
    synth.QUERY_CFG3,
    synth.QUERY_NORTH_STAR,
    "SELECT COUNT(*) FROM gpuBench WHERE c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1)",
    "SELECT COUNT(*), SUM(m), MIN(m), MAX(m), AVG(r_int) FROM gpuBench WHERE c_inv1 = 5 AND r_int < 1000",
    "SELECT g2, COUNT(*), MIN(r_int) FROM gpuBench WHERE c_inv1 NOT IN (1, 7) OR g1 = 99 GROUP BY g2",
    "SELECT h1, h2, h3, h4, COUNT(*), SUM(m) FROM gpuBench WHERE g1 BETWEEN 10 AND 19 GROUP BY h1, h2, h3, h4",
    "SELECT g1, SUM(g2), MAX(u) FROM gpuBench WHERE u < 5000 GROUP BY g1",
    "SELECT COUNT(*) FROM gpuBench WHERE NOT (r_int BETWEEN 10 AND 999989) AND c_inv2 != 3",
    "SELECT h1, h2, h3, h4, COUNT(*), DISTINCTCOUNTHLL(u) FROM gpuBench GROUP BY h1, h2, h3, h4",
    "SELECT g1, DISTINCTCOUNT(g2), DISTINCTCOUNTHLL(m), DISTINCTCOUNTHLL(r_int) FROM gpuBench WHERE c_inv1 IN (1, 2) GROUP BY g1",
    "SELECT DISTINCTCOUNT(u), DISTINCTCOUNTHLL(u), DISTINCTCOUNTHLL(m) FROM gpuBench WHERE r_int < 300000",
]


@pytest.fixture(scope="module", params=[1, 63, 64, 65, 16383, 16384, 16385, 65537, 300_001, 2_500_000])
def synth_pair(request, gpu_api, oracle_api):
    host = synth.generate_segment(request.param)
    g, o = both(gpu_api, oracle_api, host)
    yield g, o
    g.destroy()
    o.destroy()


@pytest.mark.parametrize("q", SYNTH_QUERIES)
def test_synth_queries_match_oracle(synth_pair, q):
    g, o = synth_pair
    assert_same_block(g.execute(q), o.execute(q))


@pytest.mark.parametrize("q", [synth.QUERY_CFG2, synth.QUERY_CFG3, SYNTH_QUERIES[5], SYNTH_QUERIES[8]])
def test_synth_docid_sets_match_oracle(synth_pair, q):
    g, o = synth_pair
    dg, do = g.filter(q), o.filter(q)
    np.testing.assert_array_equal(dg.words(), do.words())
    np.testing.assert_array_equal(dg.doc_ids(), do.doc_ids())
    assert dg.stats().stats_exact == 1
    assert dg.stats().num_entries_scanned_in_filter == do.stats().num_entries_scanned_in_filter


# ---- array / run / bitmap Roaring containers and exclusive postings ------------------------------------------------------------
def test_container_kinds(gpu_api, oracle_api):
    n = 200_000
    rng = np.random.default_rng(11)
    runs = (np.arange(n) // 5000) % 7            # long runs → run containers
    sparse = rng.integers(0, 5000, n)            # ~40 docs per value → array containers
    dense = rng.integers(0, 3, n)                # bitmap containers
    data = {"runs": runs.astype(np.int32), "sparse": sparse.astype(np.int32), "dense": dense.astype(np.int32),
            "v": rng.integers(0, 1000, n).astype(np.int32)}
    host = build_segment("containers", data, {k: "INT" for k in data}, inverted_index_columns=["runs", "sparse", "dense"],
                         no_dictionary_columns=["v"])
    g, o = both(gpu_api, oracle_api, host)
    qs = ["SELECT COUNT(*), SUM(v) FROM t WHERE runs = 3",
          "SELECT COUNT(*), SUM(v) FROM t WHERE runs IN (1, 2, 6) AND dense = 1",
          "SELECT COUNT(*), SUM(v) FROM t WHERE sparse IN (7, 77, 777, 4999) OR runs = 0",
          "SELECT COUNT(*), SUM(v) FROM t WHERE sparse NOT IN (1, 2, 3) AND dense != 2 AND v < 500",
          "SELECT dense, COUNT(*), MAX(v) FROM t WHERE sparse = 42 OR sparse = 43 GROUP BY dense"]
    for q in qs:
        assert_same_block(g.execute(q), o.execute(q))
        np.testing.assert_array_equal(g.filter(q).doc_ids(), o.filter(q).doc_ids())
    g.destroy()
    o.destroy()


def test_unsupported_and_errors(gpu_api, sv):
    g, _ = sv
    from pinot_amd.capi import NativeError, PG_ERR_NOT_FOUND, PG_ERR_INVALID_ARGUMENT
    with pytest.raises(NativeError) as e:
        g.execute("SELECT COUNT(*) FROM t WHERE nosuchcolumn = 1")
    assert e.value.status == PG_ERR_NOT_FOUND
    with pytest.raises(NativeError) as e:
        g.execute("SELECT COUNT(*) FROM t WHERE column1 = 'abc'")
    assert e.value.status == PG_ERR_INVALID_ARGUMENT and "NumberFormatException" in e.value.message


def test_exclusive_bound_at_infinity_is_an_invalid_range_gpu(gpu_api):
    """RangePredicateEvaluatorFactory.java:449-456: checkArgument(nextUp(lower) > lower) — the planner refuses what the reference refuses."""
    from pinot_amd import capi
    from tests.test_oracle_goldens import INVALID_RANGES, invalid_range_segment
    seg = invalid_range_segment(gpu_api)
    for where in INVALID_RANGES:
        with pytest.raises(capi.NativeError) as e:
            seg.execute(f"SELECT COUNT(*) FROM inf WHERE {where}")
        assert e.value.status == capi.PG_ERR_INVALID_ARGUMENT and "Invalid range" in e.value.message, where
    assert seg.execute("SELECT COUNT(*) FROM inf WHERE d >= 'Infinity'").aggregation_result() == [0]
    assert seg.execute("SELECT COUNT(*) FROM inf WHERE f >= '-Infinity'").aggregation_result() == [1000]
    seg.destroy()


# ---- numGroupsLimit: the reference admits the first `limit` distinct keys in docId order -------------------------------------
LIMIT_CASES = [
    ("SELECT COUNT(*), SUM(column1), MAX(column3) FROM testTable GROUP BY column9", 50),
    ("SELECT COUNT(*), MIN(column6) FROM testTable GROUP BY column9, column11", 700),
    ("SELECT COUNT(*), SUM(column18) FROM testTable WHERE column6 < 900000000 GROUP BY column7", 7),
    ("SELECT DISTINCTCOUNT(column11), DISTINCTCOUNTHLL(column1), COUNT(*) FROM testTable GROUP BY column17", 20),
    ("SELECT COUNT(*) FROM testTable GROUP BY column11", 5),        # limit == number of groups: reached, nothing dropped
    ("SELECT COUNT(*) FROM testTable GROUP BY column11", 4),
]


@pytest.mark.parametrize("q,limit", LIMIT_CASES)
def test_num_groups_limit_matches_oracle(sv, q, limit):
    from pinot_amd.query import parse_sql
    g, o = sv
    qg, qo = parse_sql(q), parse_sql(q)
    qg.num_groups_limit = qo.num_groups_limit = limit
    gb, ob = g.execute(qg), o.execute(qo)
    assert len(ob.rows()) <= limit
    assert_same_block(gb, ob)
    assert gb.stats.num_groups_limit_reached == ob.stats.num_groups_limit_reached == 1


def test_num_groups_limit_config5_shape(gpu_api, oracle_api):
    from pinot_amd.query import parse_sql
    host = synth.generate_segment(150_000, segment_index=2, columns=synth.CFG5_COLUMNS, native=False)
    g, o = both(gpu_api, oracle_api, host)
    for limit in (1000, 12_799, 12_800, 100_000):
        qg, qo = parse_sql(synth.QUERY_CFG5), parse_sql(synth.QUERY_CFG5)
        qg.num_groups_limit = qo.num_groups_limit = limit
        gb, ob = g.execute(qg), o.execute(qo)
        assert_same_block(gb, ob)
        assert gb.stats.num_groups_limit_reached == ob.stats.num_groups_limit_reached == (1 if limit <= 12_800 else 0)
    g.destroy()
    o.destroy()


# ---- key spaces beyond one LDS table: range-partitioned LDS aggregation (PG_AGG_LDS_PART) and the dense HBM table ------------
PART_QUERIES = [
    "SELECT g1, g2, c_inv1, COUNT(*), SUM(m) FROM gpuBench GROUP BY g1, g2, c_inv1 LIMIT 100000",                 # 40 000 keys
    "SELECT g1, g2, c_inv1, SUM(m), MAX(m), MIN(r_int) FROM gpuBench WHERE r_int < 125000 GROUP BY g1, g2, c_inv1 LIMIT 100000",
    "SELECT g1, g2, c_inv1, c_inv2, COUNT(*), AVG(m) FROM gpuBench WHERE c_inv2 IN (0, 3) GROUP BY g1, g2, c_inv1, c_inv2 LIMIT 200000",
    "SELECT g2, g1, c_inv2, DISTINCTCOUNT(c_inv1), COUNT(*) FROM gpuBench WHERE m > 500000 GROUP BY g2, g1, c_inv2 LIMIT 100000",
    "SELECT g1, g2, c_inv1, c_inv2, COUNT(*) FROM gpuBench WHERE c_inv1 = 2 AND g1 BETWEEN 10 AND 20 GROUP BY g1, g2, c_inv1, c_inv2 LIMIT 200000",
]


@pytest.fixture(scope="module", params=[1_500, 70_001, 600_000])
def bench_seg(request, gpu_api, oracle_api):
    host = synth.generate_segment(request.param, segment_index=5, columns=synth.CFG3_COLUMNS, native=False)
    g, o = both(gpu_api, oracle_api, host)
    yield g, o
    g.destroy()
    o.destroy()


@pytest.mark.parametrize("q", PART_QUERIES)
def test_partitioned_group_by_matches_oracle(bench_seg, q):
    g, o = bench_seg
    assert_same_block(g.execute(q), o.execute(q))


def test_partitioned_group_by_with_limit(bench_seg):
    from pinot_amd.query import parse_sql
    g, o = bench_seg
    qg, qo = parse_sql(PART_QUERIES[0]), parse_sql(PART_QUERIES[0])
    qg.num_groups_limit = qo.num_groups_limit = 1234
    assert_same_block(g.execute(qg), o.execute(qo))


# ---- one worker thread per segment task, many queries at once (BaseCombineOperator.java:97-142) -----------------------------------
def test_concurrent_queries_from_many_threads(gpu_api, oracle_api, sv_data):
    """The ABI is thread-safe and re-entrant: 8 host threads hammer two segments with different queries (each thread gets its own
    HIP stream and workspaces, plans are cached per segment); every result must equal the oracle's."""
    import threading
    host_a = sv_segment(sv_data)
    host_b = synth.generate_segment(300_000, segment_index=9, columns=synth.CFG3_COLUMNS, native=False)
    ga, oa = both(gpu_api, oracle_api, host_a)
    gb, ob = both(gpu_api, oracle_api, host_b)
    work = [(ga, oa, q) for q in SV_QUERIES[:8] + DISTINCT_QUERIES[:4]] + \
           [(gb, ob, q) for q in (synth.QUERY_CFG2, synth.QUERY_CFG3, synth.QUERY_NORTH_STAR, PART_QUERIES[0], PART_QUERIES[3])]
    expected = [(o.execute(q).rows(), o.execute(q).stats.num_docs_scanned) for _, o, q in work]
    errors = []

    def worker(tid):
        try:
            for it in range(40):
                k = (tid * 7 + it * 3) % len(work)
                g, _, q = work[k]
                b = g.execute(q)
                if b.rows() != expected[k][0] or b.stats.num_docs_scanned != expected[k][1]:
                    errors.append((tid, it, q))
        except Exception as e:  # noqa
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
    for s in (ga, oa, gb, ob):
        s.destroy()


def test_plain_c_caller_of_the_abi():
    """examples/abi_smoke.c: a C99 program builds Pinot-format bytes by hand (dictionary, fixed-bit forward index, RoaringBitmap
    inverted index, raw chunk) and drives the C ABI end to end — no Python in the loop."""
    import subprocess
    from tests.test_host_formats import _abi_smoke_binary
    out = subprocess.run([_abi_smoke_binary()], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "abi smoke ok" in out.stdout and "d=20 count=250" in out.stdout and "d=30 count=250" in out.stdout


# ---- LDS aggregation over wide group columns and 64-bit sources (pg_fast_none_w / pg_fast_multi_w, general aggregator) ----------
@pytest.fixture(scope="module")
def wide_seg(gpu_api, oracle_api):
    rng = np.random.default_rng(21)
    n = 180_003
    data = {
        "k": rng.integers(0, 2000, n).astype(np.int32),            # 11-bit dictionary column
        "k2": rng.integers(0, 7, n).astype(np.int32),
        "lm": rng.integers(-10**12, 10**12, n).astype(np.int64),    # raw LONG metric
        "dm": (rng.integers(-10**6, 10**6, n) * 0.25).astype(np.float64),   # raw DOUBLE, exactly representable sums
        "fm": (rng.integers(-1000, 1000, n) * 0.5).astype(np.float32),      # raw FLOAT
        "ld": rng.integers(0, 300, n).astype(np.int64) * 10**10,     # dictionary-encoded LONG
        "r": rng.integers(0, 1000, n).astype(np.int32),
        "inv": rng.integers(0, 5, n).astype(np.int32),
    }
    host = build_segment("wide", data, {"k": "INT", "k2": "INT", "lm": "LONG", "dm": "DOUBLE", "fm": "FLOAT", "ld": "LONG",
                                         "r": "INT", "inv": "INT"},
                         inverted_index_columns=["inv"], no_dictionary_columns=["lm", "dm", "fm", "r"])
    g, o = both(gpu_api, oracle_api, host)
    yield g, o
    g.destroy()
    o.destroy()


WIDE_QUERIES = [
    "SELECT k, SUM(lm), MIN(lm), MAX(lm), COUNT(*) FROM wide GROUP BY k LIMIT 5000",
    "SELECT k, SUM(dm), MAX(fm), AVG(ld) FROM wide WHERE r BETWEEN 100 AND 700 GROUP BY k LIMIT 5000",
    "SELECT k2, SUM(lm), SUM(dm), MINMAXRANGE(fm) FROM wide WHERE inv IN (1, 3) AND r < 900 AND lm > 0 GROUP BY k2",
    "SELECT SUM(lm), MIN(dm), MAX(ld), COUNT(*) FROM wide WHERE inv = 2",
    "SELECT k2, k, SUM(ld) FROM wide WHERE r < 300 AND fm > 0 GROUP BY k2, k LIMIT 20000",
    "SELECT k, MAX(lm) FROM wide WHERE dm BETWEEN -1000 AND 1000 AND lm < 0 GROUP BY k LIMIT 5000",
]


@pytest.mark.parametrize("q", WIDE_QUERIES)
def test_wide_aggregations_match_oracle(wide_seg, q):
    g, o = wide_seg
    gb, ob = g.execute(q), o.execute(q)
    assert_same_block(gb, ob)
    assert gb.stats.kernel.decode().startswith(("pg_fast_", "pg_pipe_w", "pg_generic_", "pg_radix_", "pg_part_"))


# ---- COUNT(*) behind an index-only filter of dense postings: the bitmap stream (pg_dense_count_*) --------------------------------------
@pytest.mark.parametrize("n", [151_072, 151_073, 151_103, 151_104, 151_105, 151_199, 151_200, 217_001])
def test_dense_count_stream(gpu_api, oracle_api, n):
    """FastFilteredCountOperator's shape over bitmap containers: IN / NOT IN / AND of two columns, every tail length of the last 128-doc
    group; the last 2^16-doc chunk holds > 4096 docs per dictId so that its containers are bitmaps too (else the leaf is not 'dense')."""
    rng = np.random.default_rng(n)
    data = {"a": rng.integers(0, 4, n).astype(np.int32), "b": rng.integers(0, 3, n).astype(np.int32), "r": rng.integers(0, 100, n).astype(np.int32)}
    host = build_segment("dc", data, {"a": "INT", "b": "INT", "r": "INT"}, inverted_index_columns=["a", "b"], no_dictionary_columns=["r"])
    g, o = both(gpu_api, oracle_api, host)
    for where, dense in (("a IN (0, 2)", True), ("a = 1", True), ("a NOT IN (3)", True), ("a IN (0, 1, 2) AND b = 1", True), ("a != 0 AND b != 2", True),
                         ("a = 1 AND r < 50", False)):
        q = f"SELECT COUNT(*) FROM dc WHERE {where}"
        gb, ob = g.execute(q), o.execute(q)
        assert_same_block(gb, ob)
        assert gb.stats.num_entries_scanned_in_filter == ob.stats.num_entries_scanned_in_filter
        assert (gb.stats.kernel.decode() == "pg_dense_count") == dense, (where, gb.stats.kernel)
    g.destroy()
    o.destroy()


@pytest.mark.parametrize("card", [2, 3, 5, 12, 20, 50, 100, 200])
def test_dict_count_stream(gpu_api, oracle_api, card):
    """COUNT(*) behind ONE scan of a dictionary column of 1 .. 8 bits (no inverted index): the bit stream, 32 docs per thread — EQ / range /
    IN / NOT IN, ragged tails, nothing matching."""
    rng = np.random.default_rng(card)
    for n in (1, 31, 32, 33, 2047, 70_001):
        data = {"d": rng.integers(0, card, n).astype(np.int32) * 3, "r": rng.integers(0, 100, n).astype(np.int32)}
        host = build_segment("dd", data, {"d": "INT", "r": "INT"}, no_dictionary_columns=["r"])
        g, o = both(gpu_api, oracle_api, host)
        top = 3 * (card - 1)
        for where, stream in ((f"d = {top}", True), ("d BETWEEN 3 AND 30", True), (f"d IN (0, 6, {top})", True), (f"d NOT IN (3, {top})", True),
                              ("d > 100000", True), ("d >= 3 AND r < 50", False)):
            q = f"SELECT COUNT(*) FROM dd WHERE {where}"
            gb, ob = g.execute(q), o.execute(q)
            assert_same_block(gb, ob)
            assert gb.stats.num_entries_scanned_in_filter == ob.stats.num_entries_scanned_in_filter, where
            if n >= 2047 and 0 < gb.stats.num_docs_scanned < n:   # (a predicate no / every dictionary value satisfies never reaches a scan kernel)
                assert (gb.stats.kernel.decode() == "pg_dict_count") == stream, (where, n, gb.stats.kernel)
        g.destroy()
        o.destroy()


# ---- the wide pipeline (pg_pipe_w_*): raw LONG / INT values, group columns of up to 16 bits, behind no filter / dense index / range scan ----
PIPE_WIDE_QUERIES = [
    ("SELECT k, SUM(lm), MIN(lm), MAX(lm), COUNT(*) FROM wide GROUP BY k LIMIT 5000", "pg_pipe_w64_none"),
    ("SELECT k, COUNT(*) FROM wide GROUP BY k LIMIT 5000", "pg_pipe_w0_none"),                              # no value column at all
    ("SELECT k2, SUM(lm), MAX(lm) FROM wide GROUP BY k2", "pg_pipe_w64_none"),                               # narrow key, LONG values
    ("SELECT k2, SUM(lm) FROM wide WHERE inv IN (1, 3) GROUP BY k2", "pg_pipe_w64_index"),
    ("SELECT SUM(lm), MIN(lm), COUNT(*) FROM wide WHERE inv = 2", "pg_pipe_w64_index"),                      # no GROUP BY
    ("SELECT SUM(lm), MAX(lm) FROM wide", "pg_pipe_w64_none"),
    ("SELECT k, SUM(r), COUNT(*) FROM wide WHERE r BETWEEN 100 AND 700 GROUP BY k LIMIT 5000", "pg_pipe_w32_scan"),   # INT values, 11-bit key
    ("SELECT k, SUM(lm), MIN(lm) FROM wide WHERE r BETWEEN 100 AND 700 GROUP BY k LIMIT 5000", "pg_pipe_w64_scan"),
    ("SELECT k2, MAX(lm), COUNT(*) FROM wide WHERE inv NOT IN (0, 4) AND r < 500 GROUP BY k2", "pg_pipe_w64_index_scan"),
    ("SELECT k, SUM(dm), MIN(dm), MAX(dm), COUNT(*) FROM wide GROUP BY k LIMIT 5000", "pg_pipe_wd_none"),      # raw DOUBLE: digit sums, order-key MIN / MAX
    ("SELECT k2, AVG(dm) FROM wide GROUP BY k2", "pg_pipe_wd_none"),
    ("SELECT k2, SUM(dm) FROM wide WHERE inv IN (1, 3) AND r < 900 GROUP BY k2", "pg_pipe_wd_index_scan"),
    ("SELECT SUM(dm), MAX(dm), COUNT(*) FROM wide WHERE r BETWEEN 100 AND 700", "pg_pipe_wd_scan"),             # no GROUP BY: lane-held digit sums
    ("SELECT k, MINMAXRANGE(dm) FROM wide WHERE inv = 4 GROUP BY k LIMIT 5000", "pg_pipe_wd_index"),
    ("SELECT k, k2, SUM(lm) FROM wide WHERE inv = 2 AND r >= 250 GROUP BY k, k2 LIMIT 20000", None),          # 14 000 groups: whichever table mode
    ("SELECT k, SUM(lm) FROM wide WHERE r > 5000 GROUP BY k LIMIT 5000", None),                                # nothing matches
]


@pytest.mark.parametrize("q,kernel", PIPE_WIDE_QUERIES)
def test_wide_pipeline_matches_oracle(wide_seg, q, kernel):
    g, o = wide_seg
    gb, ob = g.execute(q), o.execute(q)
    assert_same_block(gb, ob)
    if kernel and gb.stats.num_docs_scanned > 0:
        assert gb.stats.kernel.decode() == kernel


@pytest.mark.parametrize("n", [1, 5, 1023, 1025, 2047, 2048, 2049, 4096, 6145, 40_000])
def test_wide_pipeline_sizes(gpu_api, oracle_api, n):
    """Segments of fewer tiles than the pipeline keeps in flight, ragged last halves."""
    rng = np.random.default_rng(n)
    data = {"k": rng.integers(0, 600, n).astype(np.int32), "k2": rng.integers(0, 3, n).astype(np.int32),
            "lm": rng.integers(-10**11, 10**11, n).astype(np.int64), "r": rng.integers(0, 1000, n).astype(np.int32),
            "inv": rng.integers(0, 3, n).astype(np.int32),   # (sums stay below 2^53: the domain where the reference's double sums are exact)
            "dd": (rng.integers(-10**6, 10**6, n) * 0.25).astype(np.float64),
            "dn": np.where(rng.random(n) < 0.2, np.nan, rng.integers(-50, 50, n) * 1.5).astype(np.float64)}   # NaN never replaces a MIN / MAX holder
    data["dn"][::7] = np.inf
    host = build_segment("ws", data, {"k": "INT", "k2": "INT", "lm": "LONG", "r": "INT", "inv": "INT", "dd": "DOUBLE", "dn": "DOUBLE"},
                         inverted_index_columns=["inv"], no_dictionary_columns=["lm", "r", "dd", "dn"])
    g, o = both(gpu_api, oracle_api, host)
    for q in ("SELECT k, SUM(lm), MIN(lm), MAX(lm), COUNT(*) FROM ws GROUP BY k LIMIT 5000",
              "SELECT k, k2, SUM(lm) FROM ws WHERE r < 700 GROUP BY k, k2 LIMIT 5000",
              "SELECT k2, SUM(r), MAX(r) FROM ws WHERE inv = 1 AND r >= 100 GROUP BY k2",
              "SELECT MIN(lm), COUNT(*) FROM ws WHERE inv != 0",
              "SELECT k, SUM(dd), MIN(dd), COUNT(*) FROM ws WHERE r < 800 GROUP BY k LIMIT 5000",
              "SELECT SUM(dd), MAX(dd) FROM ws WHERE inv = 2",
              "SELECT k2, MIN(dn), MAX(dn), COUNT(*) FROM ws GROUP BY k2",
              "SELECT MIN(dn), MAX(dn) FROM ws WHERE r >= 300"):
        assert_same_block(g.execute(q), o.execute(q))
    g.destroy()
    o.destroy()


@pytest.mark.parametrize("card", [300, 1000, 2000, 4000, 8000, 16000])
def test_wide_pipeline_key_widths(gpu_api, oracle_api, card):
    """Group columns of 9 .. 14 bits (three-dword windows, every misalignment of a quad's 4 x bits inside them); with a second, narrow
    column in front and behind."""
    rng = np.random.default_rng(card)
    n = 50_021
    data = {"k": rng.integers(0, card, n).astype(np.int32), "k2": rng.integers(0, 3, n).astype(np.int32),
            "lm": rng.integers(-10**9, 10**9, n).astype(np.int64), "r": rng.integers(0, 1000, n).astype(np.int32)}
    host = build_segment("wk", data, {"k": "INT", "k2": "INT", "lm": "LONG", "r": "INT"}, no_dictionary_columns=["lm", "r"])
    g, o = both(gpu_api, oracle_api, host)
    gb = g.execute("SELECT k, MAX(r) FROM wk GROUP BY k LIMIT 100000")
    assert_same_block(gb, o.execute("SELECT k, MAX(r) FROM wk GROUP BY k LIMIT 100000"))
    assert gb.stats.kernel.decode() == "pg_pipe_w32_none"
    qs = ["SELECT k, MIN(r), MAX(r) FROM wk WHERE r BETWEEN 100 AND 899 GROUP BY k LIMIT 100000"]
    if card <= 2000:
        qs += ["SELECT k, k2, SUM(lm), COUNT(*) FROM wk GROUP BY k, k2 LIMIT 100000", "SELECT k2, k, MAX(lm) FROM wk WHERE r < 500 GROUP BY k2, k LIMIT 100000"]
    for q in qs:
        assert_same_block(g.execute(q), o.execute(q))
    g.destroy()
    o.destroy()


def test_wide_pipeline_knob(gpu_api, oracle_api, gpu_knobs):
    """PG_NO_PIPE_WIDE: the same plans on the 16-wavefront walk (pg_fast_none_w / pg_fast_multi_w), the A/B knob of the variants table."""
    gpu_knobs(PG_NO_PIPE_WIDE="1")
    rng = np.random.default_rng(4)
    n = 30_001
    data = {"k": rng.integers(0, 900, n).astype(np.int32), "lm": rng.integers(-10**9, 10**9, n).astype(np.int64)}
    host = build_segment("wk", data, {"k": "INT", "lm": "LONG"}, no_dictionary_columns=["lm"])
    g, o = both(gpu_api, oracle_api, host)
    gb = g.execute("SELECT k, SUM(lm) FROM wk GROUP BY k LIMIT 5000")
    assert_same_block(gb, o.execute("SELECT k, SUM(lm) FROM wk GROUP BY k LIMIT 5000"))
    assert gb.stats.kernel.decode() == "pg_fast_none_w"
    g.destroy()
    o.destroy()


# ---- radix-partitioned group-by (PG_AGG_RADIX): one visit per doc, tuples bucketed by key range, LDS aggregation per bucket ---
RADIX_QUERIES = [
    ("SELECT u, COUNT(*) FROM gpuBench GROUP BY u LIMIT 2000000", None),                       # 1 M keys, numGroupsLimit 100 000 bites
    ("SELECT u, COUNT(*), SUM(h1) FROM gpuBench WHERE h2 IN (1, 2) GROUP BY u LIMIT 2000000", 1_000_000),
    ("SELECT u, h1, COUNT(*) FROM gpuBench WHERE h2 = 3 AND h3 > 4 GROUP BY u, h1 LIMIT 20000000", 20_000_000),   # 16 M keys
    ("SELECT h4, u, MAX(h2), MIN(h3), AVG(h1) FROM gpuBench WHERE u < 300000 GROUP BY h4, u LIMIT 100", 50),
]


@pytest.mark.parametrize("q,limit", RADIX_QUERIES)
def test_radix_group_by_matches_oracle(gpu_api, oracle_api, q, limit):
    from pinot_amd.query import parse_sql
    host = synth.generate_segment(260_000, segment_index=4, columns=synth.CFG5_COLUMNS, native=False)
    g, o = both(gpu_api, oracle_api, host)
    qg, qo = parse_sql(q), parse_sql(q)
    if limit:
        qg.num_groups_limit = qo.num_groups_limit = limit
    gb, ob = g.execute(qg), o.execute(qo)
    assert_same_block(gb, ob)
    assert gb.stats.num_groups_limit_reached == ob.stats.num_groups_limit_reached
    g.destroy()
    o.destroy()


# ---- key spaces beyond any dense table (> 64 M keys, the reference's LongMapBasedHolder): hash-partitioned LDS hash tables ------
HASH_QUERIES = [
    ("SELECT u, h1, h2, COUNT(*), SUM(h3) FROM gpuBench GROUP BY u, h1, h2 LIMIT 10000000", 10_000_000),     # 160 M keys
    ("SELECT u, h1, h2, COUNT(*) FROM gpuBench GROUP BY u, h1, h2 LIMIT 10000000", None),                    # default numGroupsLimit trims
    ("SELECT h2, u, h1, MIN(h3), MAX(h4), AVG(h3) FROM gpuBench WHERE h4 IN (1, 5) AND u >= 500000 GROUP BY h2, u, h1 LIMIT 10000000", 10_000_000),
    ("SELECT u, h1, h2, h3, SUM(h4) FROM gpuBench WHERE u = 123456 OR u < 50 GROUP BY u, h1, h2, h3 LIMIT 100", 1_000_000),   # 1.6 G keys
    ("SELECT u, h1, h2, COUNT(*) FROM gpuBench WHERE h1 = 99 GROUP BY u, h1, h2 LIMIT 10", 1000),           # nothing matches
]


@pytest.mark.parametrize("q,limit", HASH_QUERIES)
def test_hashed_group_by_matches_oracle(gpu_api, oracle_api, q, limit):
    from pinot_amd.query import parse_sql
    host = synth.generate_segment(260_000, segment_index=6, columns=synth.CFG5_COLUMNS, native=False)
    g, o = both(gpu_api, oracle_api, host)
    qg, qo = parse_sql(q), parse_sql(q)
    if limit:
        qg.num_groups_limit = qo.num_groups_limit = limit
    gb, ob = g.execute(qg), o.execute(qo)
    assert_same_block(gb, ob)
    assert gb.stats.num_groups_limit_reached == ob.stats.num_groups_limit_reached
    g.destroy()
    o.destroy()


# ---- robustness of the boundary (round-1 advisor findings) ------------------------------------------------------------------------
def test_corrupt_sorted_index_is_refused(gpu_api):
    """SortedIndexReaderImpl pairs come from a file: a range outside the segment, descending or overlapping ranges, and a
    bits_per_value too small for the cardinality return PG_ERR_INVALID_ARGUMENT instead of writing past a host buffer."""
    import ctypes as C
    from pinot_amd import capi
    n = 5000
    vals = np.sort(np.random.default_rng(1).integers(0, 20, n)).astype(np.int32)
    host = build_segment("s", {"t": vals}, {"t": "INT"})
    col = host.columns["t"]
    assert col.fwd_encoding == capi.FWD_DICT_SORTED
    good = col.forward_index.copy()
    pairs = np.frombuffer(bytes(good), dtype=">i4").reshape(-1, 2).copy()
    cases = []
    p = pairs.copy(); p[3, 1] = n + 100; cases.append(p)           # end beyond the segment
    p = pairs.copy(); p[2, 0] = -5; cases.append(p)                # negative start
    p = pairs.copy(); p[5, 0] = p[4, 0]; cases.append(p)           # overlaps the previous range
    p = pairs.copy(); p[6] = p[6][::-1]; cases.append(p)           # end < start
    for p in cases:
        col.forward_index = np.frombuffer(p.astype(">i4").tobytes(), dtype=np.uint8).copy()
        with pytest.raises(capi.NativeError) as e:
            NativeSegment(gpu_api, host)
        assert e.value.status == capi.PG_ERR_INVALID_ARGUMENT and "sorted index" in e.value.message
    col.forward_index = good
    col.bits_per_value = 2                                         # 20 dictIds do not fit 2 bits
    with pytest.raises(capi.NativeError) as e:
        NativeSegment(gpu_api, host)
    assert e.value.status == capi.PG_ERR_INVALID_ARGUMENT
    col.bits_per_value = 5
    seg = NativeSegment(gpu_api, host)
    assert seg.execute("SELECT COUNT(*) FROM s WHERE t < 7").aggregation_result()[0] == int((vals < 7).sum())
    seg.destroy()


def test_plan_cache_keys_do_not_collide(gpu_api, oracle_api):
    """two range queries whose bounds concatenate to the same text ("a,b" / "c" vs "a" / "b,c") must not share a cached plan"""
    rng = np.random.default_rng(4)
    words = ["a", "a,b", "a,c", "b", "b,c", "c", "d"]
    data = {"s": rng.choice(words, 20_000), "g": rng.integers(0, 3, 20_000).astype(np.int32)}
    host = build_segment("k", {"s": data["s"].tolist(), "g": data["g"]}, {"s": "STRING", "g": "INT"})
    g, o = both(gpu_api, oracle_api, host)
    q1 = "SELECT g, COUNT(*) FROM k WHERE s BETWEEN 'a,b' AND 'c' GROUP BY g"
    q2 = "SELECT g, COUNT(*) FROM k WHERE s BETWEEN 'a' AND 'b,c' GROUP BY g"
    r1, r2 = g.execute(q1), g.execute(q2)
    assert_same_block(r1, o.execute(q1))
    assert_same_block(r2, o.execute(q2))
    assert r1.rows() != r2.rows()
    g.destroy()
    o.destroy()
