"""DISTINCTCOUNT over raw (no-dictionary) columns (VERDICT r5 missing #4): the reference keeps a typed VALUE set per group for such a column —
IntOpenHashSet / LongOpenHashSet / FloatOpenHashSet / DoubleOpenHashSet (BaseDistinctAggregateAggregationFunction.java:325-380) — so the
intermediate result is the set of values (pg_result_kind PG_RESULT_VALUE_SET: pg_result_set_sizes + pg_result_set_values_long / _double).
The oracle against a numpy brute force here; the GPU library (the column's virtual dictionary, pg_vdict.hip, then the dictId-set kernels)
against the oracle under `-m gpu`."""
import numpy as np
import pytest

from pinot_amd import capi
from pinot_amd.executor import NativeSegment
from pinot_amd.query import parse_sql
from pinot_amd.segment import HostSegment, build_column


def table(n, seed):
    rng = np.random.default_rng(seed)
    d = {
        "ri": rng.integers(-60, 60, n).astype(np.int64),
        "rl": (rng.integers(-40, 40, n) * (1 << 40)).astype(np.int64),
        "rf": (rng.integers(-25, 25, n) / 4).astype(np.float32),
        "rd": rng.integers(-30, 30, n) / 8,
        "rw": rng.integers(0, 1 << 31, n).astype(np.int64) % max(1, min(n, 200_000)),   # many distinct values
        "g": rng.integers(0, 6, n).astype(np.int64),
        "h": rng.integers(0, 40, n).astype(np.int64),
        "di": rng.integers(0, 33, n).astype(np.int64),
    }
    if n > 3:
        d["rd"][1] = -0.0          # -0.0 and 0.0 are two elements of a DoubleOpenHashSet (bit-wise equality)
        d["rd"][2] = 0.0
    seg = HostSegment("t", n)
    seg.columns["ri"] = build_column("ri", d["ri"].tolist(), "INT", dictionary=False)
    seg.columns["rl"] = build_column("rl", d["rl"].tolist(), "LONG", dictionary=False)
    seg.columns["rf"] = build_column("rf", d["rf"].tolist(), "FLOAT", dictionary=False)
    seg.columns["rd"] = build_column("rd", d["rd"].tolist(), "DOUBLE", dictionary=False)
    seg.columns["rw"] = build_column("rw", d["rw"].tolist(), "INT", dictionary=False)
    seg.columns["g"] = build_column("g", d["g"].tolist(), "INT")
    seg.columns["h"] = build_column("h", d["h"].tolist(), "INT")
    seg.columns["di"] = build_column("di", d["di"].tolist(), "INT", inverted=True)
    return seg, d


QUERIES = [
    "SELECT DISTINCTCOUNT(ri), DISTINCTCOUNT(rl), DISTINCTCOUNT(rf), DISTINCTCOUNT(rd) FROM t",
    "SELECT g, DISTINCTCOUNT(ri), DISTINCTCOUNT(rl), COUNT(*) FROM t GROUP BY g LIMIT 100",
    "SELECT g, h, DISTINCTCOUNT(rd), DISTINCTCOUNT(rf), SUM(ri) FROM t WHERE di IN (1, 2, 3, 4, 5, 6, 7, 8) AND ri > -50 GROUP BY g, h LIMIT 1000",
    "SELECT DISTINCTCOUNT(rw), COUNT(*) FROM t WHERE di < 20",
    "SELECT g, DISTINCTCOUNT(rw) FROM t GROUP BY g LIMIT 100",
    "SELECT g, DISTINCTCOUNT(ri), DISTINCTCOUNT(di) FROM t GROUP BY g LIMIT 100",     # a raw and a dictionary column side by side
]


def brute(d, sql):
    """the two shapes the test checks by hand: no GROUP BY / GROUP BY g, no filter"""
    out = {}
    keys = [()] if "GROUP BY" not in sql else [(int(k),) for k in np.unique(d["g"])]
    for k in keys:
        m = np.ones(len(d["g"]), dtype=bool) if not k else d["g"] == k[0]
        out[k] = m
    return out


def value_set_of(api, seg, sql, dtype):
    """the first aggregation's (ungrouped) value set through the C ABI: pg_result_kind_of, pg_result_set_sizes, pg_result_set_values_<dtype>"""
    import ctypes as C
    from pinot_amd.query import CQuery
    cq = CQuery(parse_sql(sql))
    h = C.c_void_p()
    api.call("query_exec", seg.handle, cq.ptr(), C.byref(h))
    try:
        kind = C.c_int32()
        api.call("result_kind_of", h, 0, C.byref(kind))
        assert kind.value == capi.RESULT_VALUE_SET
        sizes = np.zeros(1, dtype=np.int32)
        api.call("result_set_sizes", h, 0, sizes.ctypes.data, 1)
        n = int(sizes[0])
        vals = np.zeros(max(n, 1), dtype=np.float64 if dtype == "double" else np.int64)
        api.call("result_set_values_" + dtype, h, 0, vals.ctypes.data, n)
        with pytest.raises(capi.NativeError):   # a DOUBLE column's set has no long values, and the other way round
            api.call("result_set_values_" + ("long" if dtype == "double" else "double"), h, 0, vals.ctypes.data, n)
        with pytest.raises(capi.NativeError):   # ... and no dictIds
            api.call("result_set_dict_ids", h, 0, vals.ctypes.data, n)
        return vals[:n]
    finally:
        api.call("result_free", h)


def value_bits(api, seg, sql):
    return value_set_of(api, seg, sql, "double").view(np.int64).tolist()


@pytest.mark.parametrize("n", [1, 300, 5000])
def test_oracle_value_sets_against_numpy(oracle_api, n):
    host, d = table(n, seed=n)
    o = NativeSegment(oracle_api, host)
    rows = o.execute(QUERIES[0]).rows()
    assert rows[()] == [frozenset(d["ri"].tolist()), frozenset(d["rl"].tolist()), frozenset(float(x) for x in d["rf"]), frozenset(d["rd"].tolist())]
    rows = o.execute(QUERIES[1]).rows()
    for k, m in brute(d, QUERIES[1]).items():
        assert rows[k] == [frozenset(d["ri"][m].tolist()), frozenset(d["rl"][m].tolist()), int(m.sum())]
    # the elements by BITS through the C ABI (frozenset equality lets -0.0 pass for 0.0): sizes, then the ascending values
    assert set(value_bits(oracle_api, o, "SELECT DISTINCTCOUNT(rd) FROM t")) == set(np.unique(d["rd"].view(np.int64)).tolist())
    o.destroy()


def test_unsupported_raw_shapes_are_refused(oracle_api):
    host, _ = table(300, seed=3)
    host.columns["rs"] = build_column("rs", [f"s{i % 7}" for i in range(300)], "STRING", dictionary=False)
    o = NativeSegment(oracle_api, host)
    with pytest.raises(capi.NativeError) as e:
        o.execute("SELECT DISTINCTCOUNT(rs) FROM t")
    assert e.value.status == capi.PG_ERR_UNSUPPORTED
    q = parse_sql("SELECT DISTINCTCOUNT(ri) FROM t")
    q.flags |= capi.QUERY_FLAG_NULL_HANDLING
    with pytest.raises(capi.NativeError) as e:
        o.execute(q)
    assert e.value.status == capi.PG_ERR_UNSUPPORTED
    o.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 300, 2049, 70_001, 400_003])
def test_gpu_value_sets_match_oracle(gpu_api, oracle_api, n):
    host, d = table(n, seed=n)
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    for sql in QUERIES:
        gb, ob = g.execute(sql), o.execute(sql)
        assert gb.rows() == ob.rows(), sql
        for f in ("num_docs_scanned", "num_entries_scanned_in_filter", "num_entries_scanned_post_filter", "num_total_docs"):
            assert getattr(gb.stats, f) == getattr(ob.stats, f), (sql, f)
        assert not gb.stats.kernel.decode().startswith("pg_generic_query_g") or n < 300
    # the sets' ELEMENTS by bits: the same on both sides, ascending LONG values through the other accessor
    assert value_bits(gpu_api, g, "SELECT DISTINCTCOUNT(rd) FROM t") == value_bits(oracle_api, o, "SELECT DISTINCTCOUNT(rd) FROM t")
    assert value_set_of(gpu_api, g, "SELECT DISTINCTCOUNT(rl) FROM t", "long").tolist() == sorted(set(d["rl"].tolist()))
    # final values on the device (PG_QUERY_FLAG_FINAL_DISTINCT): the sets' sizes
    q = parse_sql("SELECT g, DISTINCTCOUNT(rl), DISTINCTCOUNT(rw) FROM t GROUP BY g LIMIT 100")
    sets = o.execute(q).rows()
    q.flags |= capi.QUERY_FLAG_FINAL_DISTINCT
    assert g.execute(q).rows() == {k: [len(value) for value in v] for k, v in sets.items()}
    q = parse_sql("SELECT DISTINCTCOUNT(rd) FROM t")    # -0.0 and 0.0 are two elements (a Python frozenset would fold them)
    q.flags |= capi.QUERY_FLAG_FINAL_DISTINCT
    assert g.execute(q).rows()[()] == [len(np.unique(d["rd"].view(np.int64)))]
    q = parse_sql("SELECT DISTINCTCOUNT(ri) FROM t")
    q.flags |= capi.QUERY_FLAG_NULL_HANDLING
    with pytest.raises(capi.NativeError) as e:
        g.execute(q)
    assert e.value.status == capi.PG_ERR_UNSUPPORTED
    g.destroy()
    o.destroy()
