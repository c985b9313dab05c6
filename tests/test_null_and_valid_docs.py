"""IS_NULL / IS_NOT_NULL leaves over the null value vector (FilterPlanNode.java:298-312) and the upsert queryableDocIds snapshot
that FilterPlanNode.run ANDs into every filter (:88-106) — both BitmapBasedFilterOperators over a RoaringBitmap the segment holds.
Oracle vs numpy on the CPU; HIP path (posting leaf over the uploaded bitmap) vs oracle in the gpu tests."""
import numpy as np
import pytest

from pinot_amd import formats
from pinot_amd.executor import NativeSegment
from pinot_amd.segment import build_segment
from tests.fixtures import sv_segment

N = 200_000


def null_segment(n=N, seed=5):
    rng = np.random.default_rng(seed)
    data = {
        "d": rng.integers(0, 40, n).astype(np.int32),          # dictionary + inverted index
        "s": np.sort(rng.integers(0, 300, n)).astype(np.int32),  # sorted
        "r": rng.integers(0, 1000, n).astype(np.int32),        # raw
        "g": rng.integers(0, 60, n).astype(np.int32),
        "m": rng.integers(0, 1 << 20, n).astype(np.int32),
    }
    schema = {k: "INT" for k in data}
    host = build_segment("nulls_0", data, schema, inverted_index_columns=["d"], no_dictionary_columns=["r", "m"])
    nulls = {
        "d": np.flatnonzero(rng.random(n) < 0.1),                                   # array + bitmap containers
        "r": np.concatenate([np.arange(n // 200, n * 35 // 100), np.arange(n * 3 // 4, n * 3 // 4 + 10)]),  # run containers
        "m": np.array([], dtype=np.int64),                                          # a vector with no nulls
    }
    for c, ids in nulls.items():
        host.columns[c].null_vector = np.frombuffer(formats.serialize_roaring(ids), dtype=np.uint8)
    valid = np.flatnonzero(rng.random(n) < 0.7)
    return host, data, nulls, valid


QUERIES = [
    "SELECT COUNT(*) FROM nulls WHERE d IS NULL",
    "SELECT COUNT(*) FROM nulls WHERE d IS NOT NULL",
    "SELECT COUNT(*) FROM nulls WHERE m IS NULL",
    "SELECT COUNT(*), SUM(m) FROM nulls WHERE m IS NOT NULL",
    "SELECT COUNT(*) FROM nulls WHERE g IS NULL",                  # no null value vector: EmptyFilterOperator
    "SELECT COUNT(*), MAX(m) FROM nulls WHERE g IS NOT NULL",      # MatchAllFilterOperator
    "SELECT g, COUNT(*), SUM(m) FROM nulls WHERE r IS NOT NULL AND d IN (1, 2, 3) GROUP BY g LIMIT 1000",
    "SELECT g, COUNT(*), MIN(m) FROM nulls WHERE d IS NULL AND r BETWEEN 100 AND 600 GROUP BY g LIMIT 1000",
    "SELECT g, COUNT(*) FROM nulls WHERE r IS NULL OR d = 7 GROUP BY g LIMIT 1000",
    "SELECT COUNT(*), SUM(m) FROM nulls WHERE NOT (d IS NULL) AND s < 100",
    "SELECT d, COUNT(*) FROM nulls WHERE r IS NULL AND d IS NOT NULL AND m > 500000 GROUP BY d LIMIT 1000",
    "SELECT COUNT(*) FROM nulls",
    "SELECT g, SUM(m), MAX(m) FROM nulls GROUP BY g LIMIT 1000",
    "SELECT g, SUM(m) FROM nulls WHERE r BETWEEN 100 AND 600 GROUP BY g LIMIT 1000",
    "SELECT g, SUM(m) FROM nulls WHERE d IN (1, 2, 3) AND r BETWEEN 100 AND 600 GROUP BY g LIMIT 1000",
    "SELECT g, SUM(m) FROM nulls WHERE d IN (1, 2, 3) AND s BETWEEN 50 AND 200 AND r > 300 AND m < 900000 GROUP BY g LIMIT 1000",
    "SELECT g, SUM(m) FROM nulls WHERE d = 5 OR r < 50 GROUP BY g LIMIT 1000",
    "SELECT MIN(d), MAX(g) FROM nulls",                            # not NonScanBased any more: the filter is the bitmap
    "SELECT COUNT(*) FROM nulls WHERE d = 99",                     # EmptyFilterOperator stays empty
]


def numpy_mask(data, nulls, q, n):
    """hand evaluation of the WHERE clauses above"""
    isnull = {c: np.isin(np.arange(n), nulls.get(c, [])) for c in ("d", "r", "m", "g")}
    env = {**{k: v.astype(np.int64) for k, v in data.items()}, "null": isnull, "np": np}
    where = {
        QUERIES[0]: "null['d']", QUERIES[1]: "~null['d']", QUERIES[2]: "null['m']", QUERIES[3]: "~null['m']",
        QUERIES[4]: "null['g']", QUERIES[5]: "~null['g']",
        QUERIES[6]: "~null['r'] & np.isin(d, [1, 2, 3])",
        QUERIES[7]: "null['d'] & (r >= 100) & (r <= 600)",
        QUERIES[8]: "null['r'] | (d == 7)",
        QUERIES[9]: "~null['d'] & (s < 100)",
        QUERIES[10]: "null['r'] & ~null['d'] & (m > 500000)",
        QUERIES[11]: "np.ones(len(d), bool)", QUERIES[12]: "np.ones(len(d), bool)",
        QUERIES[13]: "(r >= 100) & (r <= 600)",
        QUERIES[14]: "np.isin(d, [1, 2, 3]) & (r >= 100) & (r <= 600)",
        QUERIES[15]: "np.isin(d, [1, 2, 3]) & (s >= 50) & (s <= 200) & (r > 300) & (m < 900000)",
        QUERIES[16]: "(d == 5) | (r < 50)",
        QUERIES[17]: "np.ones(len(d), bool)",
        QUERIES[18]: "d == 99",
    }[q]
    return eval(where, env)


@pytest.mark.parametrize("with_valid", [False, True])
def test_oracle_null_and_valid_docs_match_numpy(oracle_api, with_valid):
    n = 30_000
    host, data, nulls, valid = null_segment(n)
    o = NativeSegment(oracle_api, host)
    vmask = np.ones(n, bool)
    if with_valid:
        o.set_queryable_doc_ids(valid)
        vmask = np.isin(np.arange(n), valid)
    for q in QUERIES:
        mask = numpy_mask(data, nulls, q, n) & vmask
        b = o.execute(q)
        assert b.stats.num_docs_scanned == int(mask.sum()), q
        if " GROUP BY g" in q:
            rows = b.rows()
            assert sorted(k[0] for k in rows) == sorted(set(data["g"][mask].tolist())), q
        elif "GROUP BY" not in q:
            res = b.aggregation_result()
            if q.startswith("SELECT COUNT(*)"):
                assert res[0] == int(mask.sum()), q
            if "SUM(m)" in q:
                assert res[1] == float(data["m"][mask].astype(np.int64).sum()), q
    o.set_queryable_doc_ids(None)   # cleared: back to the plain filter
    assert o.execute("SELECT COUNT(*) FROM nulls").aggregation_result() == [n]
    o.destroy()


def test_oracle_valid_docs_restrict_the_scans(oracle_api):
    """AND(filter, validDocIds): a lone scan is applied to the valid docs only (AndDocIdSet: bitmap + scan → applyAnd), while the
    scans of a nested AND see the docs their own index leaves kept."""
    n = 30_000
    host, data, nulls, valid = null_segment(n)
    o = NativeSegment(oracle_api, host)
    o.set_queryable_doc_ids(valid)
    b = o.execute("SELECT COUNT(*) FROM nulls WHERE r BETWEEN 100 AND 600")
    assert b.stats.num_entries_scanned_in_filter == len(valid)
    b = o.execute("SELECT COUNT(*) FROM nulls WHERE d IN (1, 2, 3) AND r BETWEEN 100 AND 600")
    assert b.stats.num_entries_scanned_in_filter == int(np.isin(data["d"], [1, 2, 3]).sum())
    o.destroy()


def test_golden_is_not_null_without_null_vector(oracle_api, sv_data):
    """test_data-sv.avro holds no nulls: `COUNT(*) FILTER(WHERE column1 IS NOT NULL)` is 30000 in the reference's
    InnerSegmentAggregationSingleValueQueriesTest#testFilteredAggregations (:64-82)."""
    o = NativeSegment(oracle_api, sv_segment(sv_data))
    assert o.execute("SELECT COUNT(*) FROM testTable WHERE column1 IS NOT NULL").aggregation_result() == [30000]
    assert o.execute("SELECT COUNT(*) FROM testTable WHERE column1 IS NULL").aggregation_result() == [0]
    o.destroy()


def test_segment_dir_round_trips_null_vectors(tmp_path):
    from pinot_amd import segment_dir
    host, _, nulls, _ = null_segment(5_000)
    segment_dir.write_segment_dir(host, str(tmp_path))
    back = segment_dir.load_segment_dir(str(tmp_path))
    for c in ("d", "r", "m"):
        assert formats.deserialize_roaring(bytes(back.columns[c].null_vector)).tolist() == nulls[c].tolist()
    assert back.columns["g"].null_vector is None


# ---- HIP path ----------------------------------------------------------------------------------------------------------------
def _same(gb, ob, q):
    if "GROUP BY" in q:
        gr, orr = gb.rows(), ob.rows()
        assert sorted(gr) == sorted(orr), q
        for k in orr:
            assert gr[k] == orr[k], (q, k, gr[k], orr[k])
    else:
        assert gb.aggregation_result() == ob.aggregation_result(), q
    assert gb.stats.num_docs_scanned == ob.stats.num_docs_scanned, q
    assert gb.stats.num_entries_scanned_post_filter == ob.stats.num_entries_scanned_post_filter, q
    assert gb.stats.stats_exact == 1
    assert gb.stats.num_entries_scanned_in_filter == ob.stats.num_entries_scanned_in_filter, q


@pytest.mark.gpu
@pytest.mark.parametrize("snapshot", ["none", "random70", "dense_runs", "sparse", "empty"])
def test_gpu_null_and_valid_docs_match_oracle(gpu_api, oracle_api, snapshot):
    host, data, nulls, valid = null_segment()
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    ids = {"none": None, "random70": valid,
           "dense_runs": np.concatenate([np.arange(0, 66_000), np.arange(70_000, 190_000, 2), np.arange(195_000, N)]),
           "sparse": np.arange(17, N, 997), "empty": np.array([], dtype=np.int64)}[snapshot]
    if snapshot != "none":
        g.set_queryable_doc_ids(ids)
        o.set_queryable_doc_ids(ids)
    for q in QUERIES:
        _same(g.execute(q), o.execute(q), q)
    g.destroy()
    o.destroy()


@pytest.mark.gpu
def test_gpu_valid_docs_stats_are_exact_and_snapshots_replace(gpu_api, oracle_api):
    host, data, nulls, valid = null_segment()
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    qs = ["SELECT COUNT(*) FROM nulls WHERE r BETWEEN 100 AND 600",
          "SELECT g, SUM(m) FROM nulls WHERE d IN (1, 2, 3) AND r BETWEEN 100 AND 600 GROUP BY g LIMIT 1000",
          "SELECT g, SUM(m) FROM nulls WHERE d IN (1, 2, 3) AND s BETWEEN 50 AND 200 AND r > 300 AND m < 900000 GROUP BY g LIMIT 1000"]
    for ids in (valid, valid[::3], None, np.arange(0, N, 2)):
        g.set_queryable_doc_ids(ids)
        o.set_queryable_doc_ids(ids)
        for q in qs:
            gb, ob = g.execute(q), o.execute(q)
            _same(gb, ob, q)
            assert gb.stats.stats_exact, q      # the nested AND of FilterPlanNode.run keeps every scan count exact
            if ids is not None and "d IN" in q:   # [index leaves][scans][queryableDocIds]: the chain kernels, not the interpreter
                assert gb.stats.kernel.decode().startswith(("pg_fast_multi", "pg_pipe_")), (q, gb.stats.kernel)
            assert gb.stats.num_entries_scanned_in_filter == ob.stats.num_entries_scanned_in_filter, q
    g.destroy()
    o.destroy()
