"""The two specialisations of the headline shape (SURVEY.md §8a rows a1-a4 + a9-a12 fused: dense inverted-index postings AND a raw-INT
range scan, LDS-table GROUP BY): pg_fast_i32range_p (software pipeline, 8 wavefronts per workgroup: pg_kernels_pipe.hip) and
pg_fast_i32range_d (dense index program, 16 wavefronts: pg_kernels_dense.hip).  Which one the planner picks is part of the contract
(pg_exec_stats.kernel), results and ExecutionStatistics equal the oracle's at every segment size around the pipeline's depth: the
pipelined kernel keeps three tiles in flight per wavefront, so segments with fewer tiles than wavefronts (700 001 docs), with two
or three tiles per wavefront (9 030 011 docs: 4 410 tiles over 2 048 wavefronts) and with a ragged last tile exercise its prologue
and its clamped tail; the smallest sizes keep array containers (CSR postings) and take the interpreted index leaves."""
import os

import pytest

from pinot_amd import capi, synth
from pinot_amd.executor import NativeSegment
from pinot_amd.query import parse_sql

pytestmark = pytest.mark.gpu

COLUMNS = ["c_inv1", "c_inv2", "r_int", "g1", "g2", "m"]
PIPE = "pg_fast_i32range_p"
DENSE = "pg_fast_i32range_d"
SCAN = "pg_fast_i32range_fp"      # BASELINE config 2: raw-INT range over the whole segment, COUNT(*) (pg_kernels_scan.hip)
knobs_off = not (os.environ.get("PG_NO_PIPE") or os.environ.get("PG_NO_DENSE_FUSED"))

QUERIES = [
    (synth.QUERY_CFG2, SCAN),
    ("SELECT COUNT(*) FROM gpuBench WHERE r_int > 999990", SCAN),
    (synth.QUERY_CFG3, PIPE),
    (synth.QUERY_NORTH_STAR, PIPE),
    ("SELECT g1, COUNT(*), MIN(m), MAX(m), SUM(m) FROM gpuBench WHERE c_inv1 NOT IN (0, 7) AND c_inv2 = 1 "
     "AND r_int BETWEEN 100 AND 900000 GROUP BY g1 ORDER BY g1 LIMIT 1000", PIPE),
    ("SELECT g2, g1, COUNT(*), SUM(m) FROM gpuBench WHERE c_inv1 IN (1, 2, 3, 4, 5) AND r_int < 10 GROUP BY g2, g1 "
     "ORDER BY g2, g1 LIMIT 10000", PIPE),
    # an empty range after the postings: every tile is skipped
    ("SELECT g1, SUM(m) FROM gpuBench WHERE c_inv2 IN (0, 1, 2) AND r_int BETWEEN 2000000 AND 3000000 GROUP BY g1 LIMIT 1000", PIPE),
    # accumulators over two value columns, or COUNT only: the dense kernel
    ("SELECT g1, SUM(m), MAX(r_int) FROM gpuBench WHERE c_inv1 IN (0, 1) AND r_int > 500000 GROUP BY g1 ORDER BY g1 LIMIT 1000", DENSE),
    ("SELECT g1, COUNT(*) FROM gpuBench WHERE c_inv1 IN (0, 1) AND r_int > 500000 GROUP BY g1 ORDER BY g1 LIMIT 1000", DENSE),
    # AVG keeps a DOUBLE sum next to the count: a floating accumulator
    ("SELECT g1, AVG(m) FROM gpuBench WHERE c_inv1 IN (0, 1) AND r_int > 500000 GROUP BY g1 ORDER BY g1 LIMIT 1000", None),
    # the pipeline's other filter shapes (pg_pipe_*, round 3): no filter, a lone range scan, inverted-index leaves only
    ("SELECT g1, SUM(m), MAX(m) FROM gpuBench GROUP BY g1 LIMIT 1000", "pg_pipe_none"),
    ("SELECT g1, g2, COUNT(*), MIN(m) FROM gpuBench GROUP BY g1, g2 LIMIT 10000", "pg_pipe_none"),
    ("SELECT g1, g2, SUM(m) FROM gpuBench WHERE r_int BETWEEN 250000 AND 749999 GROUP BY g1, g2 LIMIT 10000", "pg_pipe_scan"),
    ("SELECT g1, SUM(m), COUNT(*) FROM gpuBench WHERE r_int < 7 GROUP BY g1 LIMIT 1000", "pg_pipe_scan"),
    ("SELECT g1, SUM(m) FROM gpuBench WHERE c_inv1 IN (0, 1, 2, 3) AND c_inv2 IN (0, 1) GROUP BY g1 LIMIT 1000", "pg_pipe_index"),
    ("SELECT g2, g1, MAX(m), COUNT(*) FROM gpuBench WHERE c_inv1 NOT IN (3, 4) GROUP BY g2, g1 LIMIT 10000", "pg_pipe_index2"),
    ("SELECT g1, SUM(m) FROM gpuBench WHERE c_inv1 IN (0, 1, 2, 3) GROUP BY g1 LIMIT 1000", "pg_pipe_index"),
    # a second range scan on the aggregated column itself: tested on the value quads as they arrive (no load of its own)
    ("SELECT g1, SUM(m), COUNT(*) FROM gpuBench WHERE r_int BETWEEN 250000 AND 749999 AND m < 500000 GROUP BY g1 LIMIT 1000", "pg_pipe_scan_vscan"),
    ("SELECT g1, g2, MAX(m), MIN(m) FROM gpuBench WHERE r_int > 900000 AND m BETWEEN 1000 AND 2000 GROUP BY g1, g2 LIMIT 10000", "pg_pipe_scan_vscan"),
    ("SELECT g1, SUM(m) FROM gpuBench WHERE r_int < 500000 AND m > 5000000 GROUP BY g1 LIMIT 1000", "pg_pipe_scan_vscan"),     # empty second range
    ("SELECT g1, SUM(m), MAX(m) FROM gpuBench WHERE c_inv1 IN (0, 1, 2, 3) AND c_inv2 IN (0, 1) AND r_int BETWEEN 250000 AND 749999 "
     "AND m >= 524288 GROUP BY g1 LIMIT 1000", "pg_pipe_index_scan_vscan"),
    ("SELECT g2, g1, COUNT(*), SUM(m) FROM gpuBench WHERE c_inv1 NOT IN (3) AND r_int < 100 AND m < 1000000 GROUP BY g2, g1 LIMIT 10000",
     "pg_pipe_index_scan_vscan"),
    # no GROUP BY, no filter, integer accumulators over one / two raw INT columns: streamed with the accumulators in registers (pg_kernels_scan.hip)
    ("SELECT SUM(m), MIN(m), MAX(m), COUNT(*) FROM gpuBench", "pg_nogroup_s1"),
    ("SELECT MINMAXRANGE(r_int) FROM gpuBench", "pg_nogroup_s1"),
    ("SELECT SUM(m), MAX(r_int), MIN(r_int), MINMAXRANGE(m), COUNT(*) FROM gpuBench", "pg_nogroup_s2"),
]


@pytest.fixture(scope="module", params=[1, 2047, 2048, 2049, 4096, 3 * 2048 + 5, 17 * 2048 + 1999, 700_001, 9_030_011])
def pair(request, gpu_api, oracle_api):
    host = synth.generate_segment(request.param, segment_index=3, columns=COLUMNS)
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    yield g, o
    g.destroy()
    o.destroy()


@pytest.mark.parametrize("sql,kernel", QUERIES)
def test_headline_specialisations_match_oracle(pair, sql, kernel):
    g, o = pair
    qc = parse_sql(sql)
    # an AND of scans with no index leaf is a leapfrog of the scan iterators: above 2^22 docs the library only walks it on the host
    # (numEntriesScannedInFilter's exact value) when the caller asks for it
    qc.flags |= capi.QUERY_FLAG_EXACT_FILTER_STATS
    gb, ob = g.execute(qc), o.execute(sql)
    assert gb.rows() == ob.rows()
    for f in ("num_docs_scanned", "num_entries_scanned_in_filter", "num_entries_scanned_post_filter", "num_total_docs"):
        assert getattr(gb.stats, f) == getattr(ob.stats, f), f
    if kernel in (SCAN, "pg_nogroup_s1", "pg_nogroup_s2") and not os.environ.get("PG_NO_SCAN_PIPE"):
        assert gb.stats.kernel.decode() == kernel                   # no index involved: every segment size takes it
    elif kernel and knobs_off and gb.stats.num_total_docs >= 65536:   # small segments keep sparse (CSR) postings: the interpreted leaves
        # (the headline shape: the loader / consumer kernel where the index program lets >= 15 % of the docs through — known at plan time from
        # the postings' cardinalities since round 6 —, the pipelined one below that)
        assert gb.stats.kernel.decode() in ((kernel, "pg_fast_i32range_s") if kernel == PIPE else (kernel,))


# ---- behind an upsert queryableDocIds snapshot (FilterPlanNode.run's outer AND): the same pipeline with the snapshot's bitmap ANDed in after
# the scan — the scan's candidates, and with them numEntriesScannedInFilter, stay the reference's ----------------------------------------
UPSERT_QUERIES = [
    (synth.QUERY_CFG3, "pg_pipe_index_scan_tail"),
    (synth.QUERY_NORTH_STAR, "pg_pipe_index_scan_tail"),
    ("SELECT g1, SUM(m), MAX(m) FROM gpuBench GROUP BY g1 LIMIT 1000", "pg_pipe_index2"),         # the snapshot is the only (index) leaf
    ("SELECT g1, SUM(m) FROM gpuBench WHERE r_int BETWEEN 250000 AND 749999 GROUP BY g1 LIMIT 1000", None),
    ("SELECT g1, COUNT(*), SUM(m) FROM gpuBench WHERE c_inv2 = 1 GROUP BY g1 LIMIT 1000", None),
    ("SELECT SUM(m), MIN(m), MAX(r_int), COUNT(*) FROM gpuBench", None),                            # no GROUP BY: not the unfiltered stream kernel
]


@pytest.mark.parametrize("n", [2049, 700_001, 3_000_017])
def test_pipeline_behind_an_upsert_snapshot(gpu_api, oracle_api, n):
    import numpy as np
    host = synth.generate_segment(n, segment_index=5, columns=COLUMNS)
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    rng = np.random.default_rng(n)
    for keep in (0.9, 0.5, 0.02):
        ids = np.flatnonzero(rng.random(n) < keep)
        g.set_queryable_doc_ids(ids)
        o.set_queryable_doc_ids(ids)
        for sql, kernel in UPSERT_QUERIES:
            gb, ob = g.execute(sql), o.execute(sql)
            assert gb.rows() == ob.rows(), (sql, keep)
            assert not gb.stats.kernel.decode().startswith("pg_nogroup"), sql   # (the stream kernels know no filter)
            for f in ("num_docs_scanned", "num_entries_scanned_in_filter", "num_entries_scanned_post_filter", "num_total_docs"):
                assert getattr(gb.stats, f) == getattr(ob.stats, f), (f, sql, keep)
            if kernel and knobs_off and n >= 65536 and keep >= 0.5:   # dense snapshots are bitmap containers: the arithmetic (dense) form
                # (a plan that has run before may have moved to the loader / consumer kernel: the candidate rate of its last execution decides)
                assert gb.stats.kernel.decode() in (kernel, "pg_fast_i32range_st"), (sql, keep, gb.stats.kernel)
    g.destroy()
    o.destroy()


# ---- pg_fast_i32range_s: the same plans with loader / consumer wavefronts (pg_kernels_spec.hip, PG_WAVE_SPECIALISED) ------------------------
@pytest.mark.parametrize("n", [2049, 70_001, 700_001, 9_030_011])
def test_wave_specialised_variant_matches_oracle(gpu_api, oracle_api, gpu_knobs, n):
    """Every query pg_fast_i32range_p takes, through the 4-loader / 8-consumer kernel: results, statistics and the tile walk's edges — fewer
    tiles than workgroups, an odd and an even number of stages per workgroup (9 030 011 docs: 4 410 tiles over 256 workgroups, 17 or 18
    each), a ragged last tile."""
    gpu_knobs(PG_WAVE_SPECIALISED="1")
    host = synth.generate_segment(n, segment_index=3, columns=COLUMNS)
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    ran = 0
    for sql, kernel in QUERIES:
        if kernel != PIPE:
            continue
        qc = parse_sql(sql)
        qc.flags |= capi.QUERY_FLAG_EXACT_FILTER_STATS
        gb, ob = g.execute(qc), o.execute(sql)
        assert gb.rows() == ob.rows(), sql
        for f in ("num_docs_scanned", "num_entries_scanned_in_filter", "num_entries_scanned_post_filter", "num_total_docs"):
            assert getattr(gb.stats, f) == getattr(ob.stats, f), (f, sql)
        if knobs_off and n >= 700_001:   # (smaller segments: the planner keeps other kernels; two stage buffers beside a large table: pg_fast_i32range_p)
            assert gb.stats.kernel.decode() in ("pg_fast_i32range_s", PIPE), sql
            ran += gb.stats.kernel.decode() == "pg_fast_i32range_s"
    assert n < 700_001 or ran >= 3
    # ... and behind an upsert snapshot (pg_fast_i32range_st): the snapshot masks the matches, not the scan's candidates
    import numpy as np
    rng = np.random.default_rng(n)
    for keep in (0.7, 0.03):
        ids = np.flatnonzero(rng.random(n) < keep)
        g.set_queryable_doc_ids(ids)
        o.set_queryable_doc_ids(ids)
        for sql in (synth.QUERY_CFG3, synth.QUERY_NORTH_STAR):
            gb, ob = g.execute(sql), o.execute(sql)
            assert gb.rows() == ob.rows(), (sql, keep)
            for f in ("num_docs_scanned", "num_entries_scanned_in_filter", "num_entries_scanned_post_filter", "num_total_docs"):
                assert getattr(gb.stats, f) == getattr(ob.stats, f), (f, sql, keep)
            if knobs_off and n >= 700_001 and keep >= 0.5 and sql == synth.QUERY_CFG3:
                assert gb.stats.kernel.decode() == "pg_fast_i32range_st", (sql, keep, gb.stats.kernel)
    g.destroy()
    o.destroy()


def test_kernel_follows_the_candidate_rate_of_the_last_execution(gpu_api, oracle_api, gpu_knobs):
    """pg_fast_i32range_s streams every column whole, pg_fast_i32range_p skips quads without candidates: a plan's first execution follows the
    candidate rate the postings' cardinalities give at plan time (round 6: exact per column, multiplied across columns — AndDocIdSet.java:110
    orders by the same numbers), later ones the rate the kernels counted (>= 15 %: config 3 lets 25 % through, the selective query 3 %);
    PG_NO_WAVE_SPECIALISED pins pg_fast_i32range_p.  Results are the oracle's either way."""
    if not knobs_off:
        pytest.skip("kernel-selection knobs set")
    host = synth.generate_segment(3_000_017, segment_index=4, columns=COLUMNS)
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    selective = ("SELECT g1, SUM(m), MAX(m) FROM gpuBench WHERE c_inv1 = 0 AND c_inv2 = 0 AND r_int BETWEEN 250000 AND 749999 "
                 "GROUP BY g1 ORDER BY g1 LIMIT 1000")
    for sql, later in ((synth.QUERY_CFG3, "pg_fast_i32range_s"), (selective, PIPE)):
        qc = parse_sql(sql)
        ob = o.execute(sql)
        kernels = []
        for _ in range(3):
            gb = g.execute(qc)
            assert gb.rows() == ob.rows()
            assert gb.stats.num_entries_scanned_in_filter == ob.stats.num_entries_scanned_in_filter
            kernels.append(gb.stats.kernel.decode())
        assert kernels == [later, later, later], (sql, kernels)
    g.destroy()
    o.destroy()
    gpu_knobs(PG_NO_WAVE_SPECIALISED="1")
    g = NativeSegment(gpu_api, host)
    assert [g.execute(synth.QUERY_CFG3).stats.kernel.decode() for _ in range(3)] == [PIPE] * 3
    g.destroy()


GENERAL_SPEC = [
    ("SELECT g1, SUM(m), MAX(m) FROM gpuBench GROUP BY g1 LIMIT 1000", "pg_spec_none"),
    ("SELECT g1, g2, COUNT(*), MIN(m) FROM gpuBench GROUP BY g1, g2 LIMIT 10000", "pg_spec_none"),
    ("SELECT g1, g2, SUM(m) FROM gpuBench WHERE r_int BETWEEN 250000 AND 749999 GROUP BY g1, g2 LIMIT 10000", "pg_spec_scan"),
    ("SELECT g1, SUM(m), COUNT(*) FROM gpuBench WHERE r_int < 7 GROUP BY g1 LIMIT 1000", "pg_spec_scan"),
    ("SELECT g1, SUM(m) FROM gpuBench WHERE c_inv1 IN (0, 1, 2, 3) AND c_inv2 IN (0, 1) GROUP BY g1 LIMIT 1000", "pg_spec_index"),
    ("SELECT g2, g1, MAX(m), COUNT(*) FROM gpuBench WHERE c_inv1 NOT IN (3, 4) GROUP BY g2, g1 LIMIT 10000", "pg_spec_index"),
    ("SELECT g1, MIN(m), SUM(m) FROM gpuBench WHERE c_inv1 IN (0, 1, 2, 3) GROUP BY g1 LIMIT 1000", "pg_spec_index"),
]


@pytest.mark.parametrize("n", [2049, 700_001, 9_030_011])
def test_wave_specialised_general_shapes(gpu_api, oracle_api, gpu_knobs, n):
    """The pipeline's general shapes (no filter, a lone range scan, index leaves only) through the loader / consumer kernels — forced: they
    are not faster there (every candidate matches: the consumers' LDS atomics are the long path) and never chosen; results and statistics are
    the oracle's."""
    gpu_knobs(PG_WAVE_SPECIALISED="1")
    host = synth.generate_segment(n, segment_index=6, columns=COLUMNS)
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    for sql, kernel in GENERAL_SPEC:
        qc = parse_sql(sql)
        qc.flags |= capi.QUERY_FLAG_EXACT_FILTER_STATS
        gb, ob = g.execute(qc), o.execute(sql)
        assert gb.rows() == ob.rows(), sql
        for f in ("num_docs_scanned", "num_entries_scanned_in_filter", "num_entries_scanned_post_filter", "num_total_docs"):
            assert getattr(gb.stats, f) == getattr(ob.stats, f), (f, sql)
        if knobs_off and n >= 700_001:
            assert gb.stats.kernel.decode() == kernel, sql
    g.destroy()
    o.destroy()
