"""BASELINE.json's full sizes (10^9-doc segment, config 3 / north star) and the format's maximum (2^31 - 1 docs) on the GPU,
checked through size-independent properties (and, for the headline queries, against the oracle itself: test_full_size_equals_oracle):
  * linearity      Q over a doc-partitioning pair of filters merges (SUM / COUNT add, MAX is max) into Q over their union
  * complement     COUNT(F) + COUNT(NOT F) = totalDocs
  * marginals      the north-star table (g1, g2) summed over g2 is the config-3 table (g1)
  * idempotence    the same query twice gives identical bytes
  * two routes     BETWEEN (one range scan, specialised kernel) == the same bounds as two comparisons (scan chain kernel)
plus an empty segment (totalDocs = 0).  PG_TEST_FULL_DOCS shrinks the big segment for local debugging."""
import os

import numpy as np
import pytest

from pinot_amd import synth
from pinot_amd.executor import NativeSegment
from pinot_amd.segment import HostSegment, build_segment

FULL_DOCS = int(os.environ.get("PG_TEST_FULL_DOCS", "1000000000"))
MAX_DOCS = int(os.environ.get("PG_TEST_MAX_DOCS", str((1 << 31) - 1)))
PRE = "c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1) AND "


def stream_segment(api, num_docs, columns):
    """pins the columns one at a time so that the host never holds more than one of them"""
    seg = NativeSegment(api, HostSegment("gpuBench_0", num_docs))
    for name in columns:
        one = synth.generate_segment(num_docs, segment_index=0, columns=[name])
        seg.add_column(one.columns[name], keep_host_buffers=False)
        del one
    return seg


def merged(a, b):
    out = {}
    for rows in (a, b):
        for k, (s, mx) in rows.items():
            if k in out:
                out[k] = [out[k][0] + s, max(out[k][1], mx)]
            else:
                out[k] = [s, mx]
    return out


@pytest.mark.gpu
def test_full_size_segment_properties(gpu_api):
    seg = stream_segment(gpu_api, FULL_DOCS, ["c_inv1", "c_inv2", "r_int", "g1", "g2", "m"])
    n = FULL_DOCS
    q3 = "SELECT g1, SUM(m), MAX(m) FROM gpuBench WHERE " + PRE + "r_int BETWEEN {} AND {} GROUP BY g1 LIMIT 1000"
    whole = seg.execute(synth.QUERY_CFG3)
    assert whole.stats.num_total_docs == n
    rows = whole.rows()
    assert len(rows) == 100
    # idempotence
    again = seg.execute(synth.QUERY_CFG3)
    assert again.rows() == rows and again.stats.num_docs_scanned == whole.stats.num_docs_scanned
    # linearity over a partition of the range predicate
    lo, hi = seg.execute(q3.format(250000, 499999)), seg.execute(q3.format(500000, 749999))
    assert merged(lo.rows(), hi.rows()) == rows
    assert lo.stats.num_docs_scanned + hi.stats.num_docs_scanned == whole.stats.num_docs_scanned
    # the same bounds through the scan-chain kernel
    two = seg.execute("SELECT g1, SUM(m), MAX(m) FROM gpuBench WHERE " + PRE + "r_int >= 250000 AND r_int <= 749999 GROUP BY g1 LIMIT 1000")
    assert two.rows() == rows
    # marginals of the north-star table
    ns = seg.execute(synth.QUERY_NORTH_STAR).rows()
    assert len(ns) == 5000
    marg = {}
    for (k1, _k2), (s,) in ns.items():
        marg[(k1,)] = marg.get((k1,), 0.0) + s
    assert marg == {k: v[0] for k, v in rows.items()}
    # complement
    c_in = seg.execute(synth.QUERY_CFG2).aggregation_result()[0]
    c_out = seg.execute("SELECT COUNT(*) FROM gpuBench WHERE NOT (r_int BETWEEN 250000 AND 749999)").aggregation_result()[0]
    assert c_in + c_out == n
    assert abs(c_in / n - 0.5) < 1e-3          # r_int is uniform over [0, 10^6)
    # every doc belongs to exactly one (c_inv1, c_inv2) posting pair
    per_pair = seg.execute("SELECT c_inv1, c_inv2, COUNT(*) FROM gpuBench GROUP BY c_inv1, c_inv2 LIMIT 100").rows()
    assert len(per_pair) == 32 and sum(v[0] for v in per_pair.values()) == n
    seg.destroy()


@pytest.mark.gpu
def test_full_size_equals_oracle(gpu_api, oracle_api):
    """The headline configurations at BASELINE.json's full size against the oracle itself (about 4 s of CPU per 10^9-row query):
    config 3, the north-star 2-key variant and config 2's predicate — group keys, SUM / MAX values and ExecutionStatistics."""
    host = synth.generate_segment(FULL_DOCS, segment_index=0, columns=["c_inv1", "c_inv2", "r_int", "g1", "g2", "m"])
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    stats = ("num_docs_scanned", "num_entries_scanned_in_filter", "num_entries_scanned_post_filter", "num_total_docs")
    oracle_blocks = {}
    for q in (synth.QUERY_CFG3, synth.QUERY_NORTH_STAR, synth.QUERY_CFG2):
        ob = oracle_blocks[q] = o.execute(q)
        # twice: a plan's first execution chooses its kernel by the candidate rate the postings' cardinalities give at plan time, the
        # following ones by the rate the kernels counted — pg_fast_i32range_s both times for config 3 and the north star (25 %)
        for run in range(2):
            gb = g.execute(q)
            assert gb.rows() == ob.rows(), (q, run)
            for f in stats:
                assert getattr(gb.stats, f) == getattr(ob.stats, f), (q, run, f)
            assert gb.stats.num_total_docs == FULL_DOCS
            if q != synth.QUERY_CFG2 and not os.environ.get("PG_NO_WAVE_SPECIALISED") and FULL_DOCS >= 700_001:
                assert gb.stats.kernel.decode() == "pg_fast_i32range_s", (q, run)
    # the same docs in Pinot's default encoding (r_int_d / m_d: 20-bit dictId streams, identity dictionaries): the dictionary-encoded queries
    # return the oracle's rows and statistics of the raw-column queries (pg_fast_dictrange_s_a)
    for name in ("r_int_d", "m_d"):
        one = synth.generate_segment(FULL_DOCS, segment_index=0, columns=[name])
        g.add_column(one.columns[name], keep_host_buffers=False)
        del one
    for raw, enc in ((synth.QUERY_CFG3, synth.QUERY_CFG3_DICT), (synth.QUERY_NORTH_STAR, synth.QUERY_NORTH_STAR_DICT)):
        ob = oracle_blocks[raw]
        for run in range(2):
            gb = g.execute(enc)
            assert gb.rows() == ob.rows(), (enc, run)
            for f in stats:
                assert getattr(gb.stats, f) == getattr(ob.stats, f), (enc, run, f)
            if not os.environ.get("PG_NO_SPECD") and FULL_DOCS >= 700_001:
                assert gb.stats.kernel.decode() in (("pg_fast_dictrange_w_a",) if os.environ.get("PG_SPECW") else ("pg_fast_dictrange_s_a", "pg_fast_dictrange_s_a_dma")), (enc, run)
    g.destroy()
    o.destroy()


@pytest.mark.gpu
def test_maximum_segment_size(gpu_api):
    """2^31 - 1 docs (docIds are Java ints): doc / byte / bit offsets beyond 32 bits in every kernel that walks the columns"""
    seg = stream_segment(gpu_api, MAX_DOCS, ["r_int", "g1", "c_inv2"])
    n = MAX_DOCS
    c_in = seg.execute(synth.QUERY_CFG2).aggregation_result()[0]
    c_out = seg.execute("SELECT COUNT(*) FROM gpuBench WHERE r_int < 250000 OR r_int > 749999").aggregation_result()[0]
    assert c_in + c_out == n and abs(c_in / n - 0.5) < 1e-3
    g = seg.execute("SELECT g1, COUNT(*), MAX(r_int) FROM gpuBench GROUP BY g1 LIMIT 1000").rows()
    assert len(g) == 100 and sum(v[0] for v in g.values()) == n
    assert all(abs(v[0] / n - 0.01) < 1e-4 for v in g.values())
    lo = seg.execute("SELECT g1, COUNT(*), MAX(r_int) FROM gpuBench WHERE c_inv2 IN (0, 1) GROUP BY g1 LIMIT 1000").rows()
    hi = seg.execute("SELECT g1, COUNT(*), MAX(r_int) FROM gpuBench WHERE c_inv2 NOT IN (0, 1) GROUP BY g1 LIMIT 1000").rows()
    assert merged(lo, hi) == g
    # the last docs of the segment are reachable: the docIds of a selective filter end near numDocs
    d = seg.filter("SELECT COUNT(*) FROM gpuBench WHERE r_int = 123456 AND c_inv2 = 3")
    ids = d.doc_ids()
    assert len(ids) == d.cardinality() and (np.diff(ids) > 0).all() and ids[-1] < n and ids[-1] > n - 50_000_000
    seg.destroy()


@pytest.mark.gpu
def test_empty_segment(gpu_api, oracle_api):
    """totalDocs = 0.  The reference never plans such a segment (empty segments are pruned first, and their dictionaries hold no
    value): raw columns run and give the identities; an empty dictionary is refused at registration, not crashed on."""
    from pinot_amd import capi
    data = {"r": np.array([], dtype=np.int32), "m": np.array([], dtype=np.int64)}
    host = build_segment("empty_0", data, {"r": "INT", "m": "LONG"}, no_dictionary_columns=["r", "m"])
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    for q in ("SELECT COUNT(*), SUM(r), MIN(r), MAX(m), AVG(m), MINMAXRANGE(r) FROM t", "SELECT COUNT(*) FROM t WHERE r > 5",
              "SELECT r, COUNT(*) FROM t GROUP BY r LIMIT 10", "SELECT m, SUM(r) FROM t WHERE r BETWEEN 1 AND 2 GROUP BY m LIMIT 10"):
        gb, ob = g.execute(q), o.execute(q)
        assert gb.rows() == ob.rows(), q
        assert gb.stats.num_docs_scanned == ob.stats.num_docs_scanned == 0
    assert g.filter("SELECT COUNT(*) FROM t WHERE r < 3").cardinality() == 0
    g.destroy()
    o.destroy()
    host = build_segment("empty_1", {"a": np.array([], dtype=np.int32)}, {"a": "INT"})
    with pytest.raises(capi.NativeError) as e:
        NativeSegment(gpu_api, host)
    assert e.value.status == capi.PG_ERR_INVALID_ARGUMENT


@pytest.mark.gpu
def test_full_size_config5_equals_doc_sharded_oracle(gpu_api, oracle_api):
    """BASELINE config 5 (flat) at full size against the oracle: the oracle needs ~4 minutes of one core for 10^9 docs,
    so the doc space is cut into shards that run on the host's cores at once and merge the way GroupByCombineOperator merges segments —
    COUNT adds (CountAggregationFunction#merge), HyperLogLog registers take the maximum (HyperLogLog#addAll,
    DistinctCountHLLAggregationFunction#merge).  Every group key, every count and every one of the 12 800 x 256 registers must match;
    the GPU side runs the pruned-offer passes (pg_kernels_oct.hip)."""
    from concurrent.futures import ThreadPoolExecutor
    from pinot_amd.executor import merge_intermediate
    n = FULL_DOCS
    seg = stream_segment(gpu_api, n, synth.CFG5_COLUMNS)
    gb = seg.execute(synth.QUERY_CFG5)
    got = gb.rows()
    assert gb.stats.num_docs_scanned == n and gb.stats.num_total_docs == n
    if n >= (1 << 20):
        assert gb.stats.kernel.decode() == "pg_oct_pruned_group_by"
    seg.destroy()
    shards = max(4, min(32, (os.cpu_count() or 8)))
    bounds = [n * k // shards for k in range(shards + 1)]

    def shard(k):
        host = synth.generate_doc_range(bounds[k], bounds[k + 1] - bounds[k], segment_index=0, columns=synth.CFG5_COLUMNS, threads=1)
        o = NativeSegment(oracle_api, host)
        rows = o.execute(synth.QUERY_CFG5).rows()
        o.destroy()
        return rows
    with ThreadPoolExecutor(max_workers=shards) as pool:
        parts = list(pool.map(shard, range(shards)))
    fns = ["COUNT", "DISTINCTCOUNTHLL"]
    expect = {}
    for rows in parts:
        for k, vals in rows.items():
            expect[k] = [merge_intermediate(f, a, b) for f, a, b in zip(fns, expect[k], vals)] if k in expect else list(vals)
    assert sorted(got) == sorted(expect)
    assert sum(v[0] for v in got.values()) == n
    bad = [k for k in expect if got[k] != expect[k]]
    assert not bad, (len(bad), bad[:3])
