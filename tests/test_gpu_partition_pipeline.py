"""GPU parity of the partition pipeline (pinot_amd/csrc/pg_kernels_part.hip: group-by over key spaces beyond one LDS table) vs the
CPU oracle: every tuple shape the planner packs (one dword with HyperLogLog (index, rank) / dictIds / a raw INT minus the column's
minimum; several planes with whole 32- and 64-bit values; the docId plane of numGroupsLimit trimming), bucket counts on both sides of
64, skewed keys (one bucket takes everything), segment sizes around the round / line / chunk boundaries, filtered and unfiltered.

Reference semantics: DictionaryBasedGroupKeyGenerator.java:416-446 (map-based holders), DefaultGroupByExecutor.java:191-220,
DistinctCountHLLAggregationFunction.java:152-222 — bit-exact counts, group keys, MIN / MAX, integer SUMs, HyperLogLog registers.
"""
import os

import numpy as np
import pytest

from pinot_amd import capi, synth
from pinot_amd.executor import NativeSegment
from pinot_amd.query import parse_sql
from pinot_amd.segment import build_segment

pytestmark = pytest.mark.gpu


def both(gpu_api, oracle_api, host):
    return NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)


def same(g, o):
    gr, orr = g.rows(), o.rows()
    assert sorted(gr.keys()) == sorted(orr.keys())
    for k in orr:
        assert gr[k] == orr[k], (k, gr[k], orr[k])
    assert g.stats.num_docs_scanned == o.stats.num_docs_scanned
    assert g.stats.num_entries_scanned_in_filter == o.stats.num_entries_scanned_in_filter
    assert g.stats.num_groups_limit_reached == o.stats.num_groups_limit_reached


def run(g, o, sql, limit=None, kernel="pg_part_group_by"):
    qg, qo = parse_sql(sql), parse_sql(sql)
    if limit:
        qg.num_groups_limit = qo.num_groups_limit = limit
    gb, ob = g.execute(qg), o.execute(qo)
    same(gb, ob)
    if kernel and gb.stats.num_docs_scanned > 0:
        assert gb.stats.kernel.decode() == kernel, gb.stats.kernel
    return gb


# ---- BASELINE config 5 (flat): 12 800 groups x 256 HyperLogLog registers: one dword per doc (9-bit local key... + 13-bit payload) -----
@pytest.mark.parametrize("n", [1, 31, 33, 2047, 2049, 4097, 16_385, 150_001, 1_000_003])
def test_config5_sizes(gpu_api, oracle_api, n):
    host = synth.generate_segment(n, segment_index=2, columns=synth.CFG5_COLUMNS, native=(n > 200_000))
    g, o = both(gpu_api, oracle_api, host)
    run(g, o, synth.QUERY_CFG5, limit=100_000)
    run(g, o, "SELECT h1, h2, h3, h4, COUNT(*), DISTINCTCOUNTHLL(u) FROM gpuBench WHERE h2 < 5 AND u > 1000 GROUP BY h1, h2, h3, h4 LIMIT 20000",
        limit=100_000)
    g.destroy()
    o.destroy()


CFG5_SHAPES = [
    # key only (1 M keys, 64 buckets)
    ("SELECT u, COUNT(*) FROM gpuBench GROUP BY u LIMIT 2000000", 2_000_000),
    # key + one small dictId
    ("SELECT u, COUNT(*), SUM(h1) FROM gpuBench WHERE h2 IN (1, 2) GROUP BY u LIMIT 2000000", 2_000_000),
    # 16 M keys behind a selective filter
    ("SELECT u, h1, COUNT(*) FROM gpuBench WHERE h2 = 3 AND h3 > 4 GROUP BY u, h1 LIMIT 20000000", 20_000_000),
    # numGroupsLimit bites: the docId plane (MIN(docId) per group)
    ("SELECT u, COUNT(*) FROM gpuBench GROUP BY u LIMIT 2000000", 1000),
    ("SELECT h4, u, MAX(h2), MIN(h3), AVG(h1) FROM gpuBench WHERE u < 300000 GROUP BY h4, u LIMIT 100", 50),
    # two HyperLogLogs + a SUM over a third column
    ("SELECT h1, h2, h3, h4, DISTINCTCOUNTHLL(u), COUNT(*) FROM gpuBench WHERE h1 < 8 GROUP BY h1, h2, h3, h4 LIMIT 20000", 100_000),
]


@pytest.fixture(scope="module")
def cfg5_seg(gpu_api, oracle_api):
    host = synth.generate_segment(300_007, segment_index=4, columns=synth.CFG5_COLUMNS, native=False)
    g, o = both(gpu_api, oracle_api, host)
    yield g, o
    g.destroy()
    o.destroy()


@pytest.mark.parametrize("q,limit", CFG5_SHAPES)
def test_config5_shapes(cfg5_seg, q, limit):
    g, o = cfg5_seg
    run(g, o, q, limit, kernel=None)   # 16 M keys x 8 bytes need > 256 buckets: the planner's other routes answer those


def test_config5_shapes_use_the_pipeline(cfg5_seg):
    g, _ = cfg5_seg
    for q, _limit in (CFG5_SHAPES[0], CFG5_SHAPES[5]):
        assert g.execute(q).stats.kernel.decode() == "pg_part_group_by", q


# ---- value columns of every kind: raw INT (range-packed), raw LONG / DOUBLE (two planes), raw FLOAT, dictionary LONG / DOUBLE ---------
@pytest.fixture(scope="module")
def wide_seg(gpu_api, oracle_api):
    rng = np.random.default_rng(5)
    n = 200_003
    data = {
        "k": rng.integers(0, 30_000, n).astype(np.int32),                 # 15-bit dictionary column: 30 000 keys
        "k2": rng.integers(0, 9, n).astype(np.int32),
        "ri": rng.integers(-5000, 60_000, n).astype(np.int32),             # raw INT, 17-bit range, negative minimum
        "rbig": rng.integers(-2**31, 2**31 - 1, n).astype(np.int32),       # raw INT using all 32 bits
        "lm": rng.integers(-10**12, 10**12, n).astype(np.int64),           # raw LONG
        "dm": (rng.integers(-10**6, 10**6, n) * 0.25).astype(np.float64),  # raw DOUBLE
        "fm": (rng.integers(-1000, 1000, n) * 0.5).astype(np.float32),     # raw FLOAT
        "ld": rng.integers(0, 300, n).astype(np.int64) * 10**10,            # dictionary LONG
        "dd": (rng.integers(0, 500, n) * 0.125).astype(np.float64),        # dictionary DOUBLE
        "r": rng.integers(0, 1000, n).astype(np.int32),
    }
    host = build_segment("wide", data, {"k": "INT", "k2": "INT", "ri": "INT", "rbig": "INT", "lm": "LONG", "dm": "DOUBLE", "fm": "FLOAT",
                                         "ld": "LONG", "dd": "DOUBLE", "r": "INT"},
                         no_dictionary_columns=["ri", "rbig", "lm", "dm", "fm", "r"])
    g, o = both(gpu_api, oracle_api, host)
    yield g, o
    g.destroy()
    o.destroy()


WIDE_SHAPES = [
    "SELECT k, COUNT(*), SUM(ri), MIN(ri), MAX(ri) FROM wide GROUP BY k LIMIT 100000",                       # one dword: key + 17 bits
    "SELECT k, k2, SUM(ri) FROM wide WHERE r < 700 GROUP BY k, k2 LIMIT 1000000",                             # 270 000 keys, filtered
    "SELECT k, SUM(rbig), MAX(rbig), COUNT(*) FROM wide GROUP BY k LIMIT 100000",                             # two planes: key | 32-bit value
    "SELECT k, SUM(lm), MIN(lm) FROM wide GROUP BY k LIMIT 100000",                                           # three planes: 64-bit value
    "SELECT k, SUM(dm), MAX(dm), MIN(fm), SUM(fm) FROM wide WHERE r BETWEEN 100 AND 900 GROUP BY k LIMIT 100000",   # DOUBLE + FLOAT: four planes
    "SELECT k, SUM(ld), MAX(dd), AVG(ri) FROM wide GROUP BY k LIMIT 100000",                                  # dictionary LONG / DOUBLE: dictIds
    "SELECT k, k2, MINMAXRANGE(ri), AVG(lm) FROM wide WHERE lm > 0 GROUP BY k, k2 LIMIT 1000000",
]


@pytest.mark.parametrize("q", WIDE_SHAPES)
def test_value_kinds(wide_seg, q):
    g, o = wide_seg
    run(g, o, q, limit=1_000_000, kernel=None)


def test_value_kinds_use_the_pipeline(wide_seg):
    g, _ = wide_seg
    for q in WIDE_SHAPES[:4]:
        assert g.execute(q).stats.kernel.decode() == "pg_part_group_by", q


def test_trimmed_wide(wide_seg):
    g, o = wide_seg
    run(g, o, "SELECT k, SUM(ri), COUNT(*) FROM wide GROUP BY k LIMIT 100000", limit=777, kernel=None)
    run(g, o, "SELECT k, SUM(lm) FROM wide WHERE r < 500 GROUP BY k LIMIT 100000", limit=5000, kernel=None)


# ---- skew: every doc in one group / one bucket, a few hot keys among many cold ones, runs longer than a chunk ---------------------------
@pytest.mark.parametrize("pattern", ["one_key", "one_bucket", "hot_and_cold", "sorted_keys"])
def test_skewed_keys(gpu_api, oracle_api, pattern):
    rng = np.random.default_rng(9)
    n = 120_011
    card = 40_000
    if pattern == "one_key":
        k = np.full(n, 31_337, dtype=np.int32)
    elif pattern == "one_bucket":
        k = rng.integers(8192, 8192 + 100, n).astype(np.int32)
    elif pattern == "hot_and_cold":
        k = np.where(rng.random(n) < 0.9, 7, rng.integers(0, card, n)).astype(np.int32)
    else:
        k = np.sort(rng.integers(0, card, n)).astype(np.int32)
    k[:card] = np.arange(card, dtype=np.int32)   # the dictionary holds every key
    v = rng.integers(0, 1 << 20, n).astype(np.int32)
    u = rng.integers(0, 50_000, n).astype(np.int32)
    host = build_segment("skew", {"k": k, "v": v, "u": u}, {"k": "INT", "v": "INT", "u": "INT"}, no_dictionary_columns=["v"])
    g, o = both(gpu_api, oracle_api, host)
    run(g, o, "SELECT k, COUNT(*), SUM(v), MAX(v) FROM skew GROUP BY k LIMIT 100000", limit=100_000)
    run(g, o, "SELECT k, COUNT(*), DISTINCTCOUNTHLL(u) FROM skew GROUP BY k LIMIT 100000", limit=100_000, kernel=None)
    run(g, o, "SELECT k, SUM(v) FROM skew WHERE v < 300000 GROUP BY k LIMIT 100000", limit=100_000)
    g.destroy()
    o.destroy()


def test_many_buckets(gpu_api, oracle_api):
    """800 000 keys x 3 accumulators: 196 buckets (the scatter keeps per-bucket state for up to 256)."""
    rng = np.random.default_rng(3)
    n = 400_000
    a = rng.integers(0, 800, n).astype(np.int32)
    b = rng.integers(0, 1000, n).astype(np.int32)
    v = rng.integers(0, 100, n).astype(np.int32)
    a[:800] = np.arange(800)
    b[:1000] = np.arange(1000)
    host = build_segment("mb", {"a": a, "b": b, "v": v}, {"a": "INT", "b": "INT", "v": "INT"})
    g, o = both(gpu_api, oracle_api, host)
    run(g, o, "SELECT a, b, COUNT(*), SUM(v), MAX(v) FROM mb GROUP BY a, b LIMIT 3000000", limit=3_000_000)
    run(g, o, "SELECT a, b, COUNT(*), SUM(v), MAX(v) FROM mb WHERE v < 10 GROUP BY a, b LIMIT 3000000", limit=3_000_000)
    g.destroy()
    o.destroy()


def test_round2_passes_still_agree(gpu_api, oracle_api, gpu_knobs):
    """PG_NO_P2 keeps the round-2 radix passes reachable (shapes outside the pipeline take them): same results."""
    host = synth.generate_segment(90_001, segment_index=7, columns=synth.CFG5_COLUMNS, native=False)
    gpu_knobs(PG_NO_P2="1")
    g, o = both(gpu_api, oracle_api, host)
    gb = run(g, o, synth.QUERY_CFG5, limit=100_000, kernel=None)
    assert gb.stats.kernel.decode() == "pg_radix_group_by"
    g.destroy()
    o.destroy()


# ---- oct-layout phase A (round 5: pg_p2_scatter_o*, plans without a filter pass): every kernel variant, segment sizes around the sub-tile
#      (512 docs), tile (2 048) and round (4 x 1 024) boundaries, against the oracle and against the quad-layout kernels (PG_NO_P2_OCT) ----
OCT_SHAPES = [
    # <= 8-bit group columns (1..4 of them), raw INT source (the "40 k / 160 k groups" rows of the variants table)
    ("SELECT g1, g2, c_inv1, COUNT(*), SUM(m) FROM gpuBench GROUP BY g1, g2, c_inv1 LIMIT 100000", 100_000),
    ("SELECT g1, g2, c_inv1, c_inv2, COUNT(*), SUM(m), MAX(m) FROM gpuBench GROUP BY g1, g2, c_inv1, c_inv2 LIMIT 200000", 200_000),
    # no source: COUNT over the key
    ("SELECT g1, g2, c_inv1, COUNT(*) FROM gpuBench GROUP BY g1, g2, c_inv1 LIMIT 100000", 100_000),
    # dictId source (<= 8 bits, and 20 bits)
    ("SELECT g1, g2, c_inv1, COUNT(*), SUM(c_inv2), MAX(c_inv2) FROM gpuBench GROUP BY g1, g2, c_inv1 LIMIT 100000", 100_000),
    ("SELECT g1, g2, c_inv1, MIN(u), MAX(u) FROM gpuBench GROUP BY g1, g2, c_inv1 LIMIT 100000", 100_000),
    # first group column wider than 8 bits: key only, + small columns, + sources
    ("SELECT u, COUNT(*) FROM gpuBench GROUP BY u LIMIT 2000000", 2_000_000),
    ("SELECT u, c_inv2, COUNT(*) FROM gpuBench GROUP BY u, c_inv2 LIMIT 5000000", 5_000_000),
    ("SELECT u, COUNT(*), SUM(g1) FROM gpuBench GROUP BY u LIMIT 2000000", 2_000_000),
]
OCT_COLUMNS = ["c_inv1", "c_inv2", "g1", "g2", "m", "u"]


@pytest.mark.parametrize("n", [1, 7, 9, 511, 513, 1025, 2047, 2049, 4095, 4097, 8193, 100_003, 1_000_003])
def test_oct_scatter_sizes(gpu_api, oracle_api, n):
    host = synth.generate_segment(n, segment_index=3, columns=OCT_COLUMNS, native=(n > 200_000))
    g, o = both(gpu_api, oracle_api, host)
    for q, limit in OCT_SHAPES[:2] + OCT_SHAPES[5:6]:
        run(g, o, q, limit)
    g.destroy()
    o.destroy()


def test_oct_scatter_shapes_and_the_quad_kernels_agree(gpu_api, oracle_api, gpu_knobs):
    host = synth.generate_segment(777_001, segment_index=5, columns=OCT_COLUMNS, native=True)
    g, o = both(gpu_api, oracle_api, host)
    oct_rows = [run(g, o, q, limit).rows() for q, limit in OCT_SHAPES]
    g.destroy()
    gpu_knobs(PG_NO_P2_OCT="1")
    g = NativeSegment(gpu_api, host)   # (plans are cached per segment: a new one sees the knob)
    for (q, limit), rows in zip(OCT_SHAPES, oct_rows):   # (the oracle answered above: the quad kernels' rows are compared with the oct kernels')
        qg = parse_sql(q)
        if limit:
            qg.num_groups_limit = limit
        gb = g.execute(qg)
        assert gb.stats.kernel.decode() == "pg_part_group_by" and gb.rows() == rows, q
    g.destroy()
    o.destroy()


# ---- numGroupsLimit by a prefix pass (round 5, pg_exec.hip execute_limit_by_prefix): the first `limit` groups in docId order are decided
#      on a doc prefix, the segment is aggregated without the docId plane.  Reference: DictionaryBasedGroupKeyGenerator.java:416-446 ------
def _limit_cases(rng, n):
    uniform = rng.integers(0, 200_000, n).astype(np.int32)
    slow = (np.arange(n) // 7).astype(np.int32) % 200_000          # groups appear one every 7 docs: the first prefixes hold too few
    late = np.where(np.arange(n) < n // 2, rng.integers(0, 50, n), rng.integers(0, 200_000, n)).astype(np.int32)   # 50 groups, then the flood
    return {"uniform": uniform, "slow": slow, "late": late}


@pytest.mark.parametrize("pattern", ["uniform", "slow", "late"])
@pytest.mark.parametrize("limit", [100, 5000])
def test_limit_by_prefix(gpu_api, oracle_api, gpu_knobs, pattern, limit):
    """The prefix (4 096 docs at first here, grown 8 x while it holds fewer than `limit` groups, abandoned past a quarter of the segment)
    decides the admitted groups; same rows, same numGroupsLimitReached as the oracle and as the one-pass plan (PG_NO_LIMIT_PREFIX)."""
    rng = np.random.default_rng(11)
    n = 1_200_011
    k = _limit_cases(rng, n)[pattern]
    k[-200_000:] = np.maximum(k[-200_000:], np.arange(200_000, dtype=np.int32) * (pattern != "slow"))   # the dictionary holds 200 000 keys
    v = rng.integers(0, 1 << 20, n).astype(np.int32)
    r = rng.integers(0, 1000, n).astype(np.int32)
    host = build_segment("lim", {"k": k, "v": v, "r": r}, {"k": "INT", "v": "INT", "r": "INT"}, no_dictionary_columns=["v", "r"])
    gpu_knobs(PG_LIMIT_PREFIX_MIN_DOCS="4096")
    g, o = both(gpu_api, oracle_api, host)
    qs = ["SELECT k, COUNT(*), SUM(v), MAX(v) FROM lim GROUP BY k LIMIT 1000000",
          "SELECT k, SUM(v) FROM lim WHERE r < 400 GROUP BY k LIMIT 1000000"]
    # "late": 50 groups in the first half of the segment — no prefix up to a quarter of it holds `limit` groups: the one-pass plan answers
    expect = "pg_part_group_by" if pattern == "late" else "pg_part_group_by_prefix"
    rows = [run(g, o, q, limit=limit, kernel=expect).rows() for q in qs]
    g.destroy()
    gpu_knobs(PG_NO_LIMIT_PREFIX="1")
    g = NativeSegment(gpu_api, host)
    for q, want in zip(qs, rows):
        assert run(g, o, q, limit=limit, kernel="pg_part_group_by").rows() == want
    g.destroy()
    # the admission on the host (tables copied, first docIds compared there) instead of the selection on the device
    gpu_knobs(PG_NO_LIMIT_PREFIX=None, PG_NO_DEVICE_TRIM="1")
    g = NativeSegment(gpu_api, host)
    for q, want in zip(qs, rows):
        assert run(g, o, q, limit=limit, kernel=expect).rows() == want
    g.destroy()
    o.destroy()


# ---- DISTINCTCOUNT whose dictId sets exceed one LDS (round 6): tuples (key, dictId) through the scatter, sets ORed in LDS per bucket -------------
DISTINCT_SHAPES = [
    ("SELECT h1, DISTINCTCOUNT(u) FROM gpuBench GROUP BY h1 LIMIT 100", "pg_part_group_by"),                       # 16 groups x 2^20-bit sets: one group per bucket
    ("SELECT h1, DISTINCTCOUNT(u), COUNT(*) FROM gpuBench GROUP BY h1 LIMIT 100", "pg_part_group_by"),
    ("SELECT h1, h2, COUNT(*), DISTINCTCOUNT(u) FROM gpuBench WHERE h3 < 7 GROUP BY h1, h2 LIMIT 1000", "pg_part_group_by"),   # 160 buckets behind a filter pass
    ("SELECT h4, DISTINCTCOUNT(u) FROM gpuBench WHERE u BETWEEN 1000 AND 500000 AND h2 IN (1, 3, 5) GROUP BY h4 LIMIT 100", None),   # 15 % pass the filter: the planner's cost model keeps the HBM sets
    ("SELECT h1, DISTINCTCOUNT(u), SUM(h2) FROM gpuBench GROUP BY h1 LIMIT 100", None),                             # another accumulator beside it: the older route
]


@pytest.mark.parametrize("n", [1, 2049, 70_001, 1_500_003])
def test_distinctcount_sets_through_the_partition_pipeline(gpu_api, oracle_api, n):
    """BaseDistinctAggregateAggregationFunction.java:306-345: a dictId set per group; the GPU's sets (and COUNTs) equal the oracle's at every
    size around the scatter's rounds, and the final values with PG_QUERY_FLAG_FINAL_DISTINCT equal the sets' sizes."""
    host = synth.generate_segment(n, segment_index=6, columns=synth.CFG5_COLUMNS, native=(n > 200_000))
    g, o = both(gpu_api, oracle_api, host)
    for sql, kernel in DISTINCT_SHAPES:
        gb = run(g, o, sql, kernel=kernel if n >= 70_001 and not os.environ.get("PG_NO_P2") else None)
        qf = parse_sql(sql)
        qf.flags |= capi.QUERY_FLAG_FINAL_DISTINCT
        fb = g.execute(qf)
        assert fb.rows().keys() == gb.rows().keys()
        a = [i for i, s in enumerate(qf.aggregations) if s.function == "DISTINCTCOUNT"][0]
        for k, v in gb.rows().items():
            assert fb.rows()[k][a] == len(v[a]), (sql, k)
    g.destroy()
    o.destroy()
