"""GPU parity of the oct-layout kernels (pinot_amd/csrc/pg_kernels_oct.hip) vs the CPU oracle.

  pg_oct_l / pg_oct_lm            DISTINCTCOUNTHLL / DISTINCTCOUNT states in the workgroup's LDS next to <= 4 group columns of <= 8 bits
  pg_oct_pruned_group_by          the pruned-offer passes: floors of the groups' registers, survivors through the partition pipeline

Reference semantics: DistinctCountHLLAggregationFunction.java:152-222 (hll.offer per doc: index / rank of stream-lib's MurmurHash),
BaseDistinctAggregateAggregationFunction.java:306-345 (dictId sets), CountAggregationFunction.java:110-143,
DictionaryBasedGroupKeyGenerator.java:312-354.  Bit-exact: group keys, counts, every HyperLogLog register, every set, ExecutionStatistics.

Covered: every group-column width 1..8 bits (lane windows that start at every byte alignment), source widths 5..20 bits, the three
hash routes (arithmetic INT dictionary incl. negative values, any dictionary through the per-dictId table, raw INT values), log2m 4..12,
filters in front (match words), no GROUP BY, segment sizes around the sub-tile / wave-tile boundaries, and — pruned offers — one to five
passes, a group that keeps an empty register (floor 0 for ever), skewed groups.
"""
import numpy as np
import pytest

from pinot_amd import synth
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


def run(g, o, sql, kernels, limit=100_000):
    qg, qo = parse_sql(sql), parse_sql(sql)
    qg.num_groups_limit = qo.num_groups_limit = limit
    gb, ob = g.execute(qg), o.execute(qo)
    same(gb, ob)
    if kernels and gb.stats.num_docs_scanned > 0:
        assert gb.stats.kernel.decode() in kernels, (sql, gb.stats.kernel)
    return gb


def make_host(n, seed=11):
    rng = np.random.default_rng(seed)
    data = {
        "g1": rng.integers(0, 2, n).astype(np.int32),          # 1 bit
        "g2": rng.integers(0, 3, n).astype(np.int32),          # 2 bits
        "g3": rng.integers(0, 5, n).astype(np.int32),          # 3 bits
        "g4": rng.integers(0, 16, n).astype(np.int32),         # 4 bits
        "g5": rng.integers(0, 20, n).astype(np.int32),         # 5 bits
        "g6": rng.integers(0, 40, n).astype(np.int32),         # 6 bits
        "g7": rng.integers(0, 100, n).astype(np.int32),        # 7 bits
        "g8": rng.integers(0, 200, n).astype(np.int32),        # 8 bits
        "ua": (np.arange(n) % 5000).astype(np.int32) * 3 + 7,            # arithmetic dictionary (every id present), 13 bits
        "un": ((np.arange(n) * 7) % 3001).astype(np.int32) * 2 - 3000,   # arithmetic, negative values, 12 bits
        "ul": rng.integers(-10**9, 10**9, n).astype(np.int32),           # random values: a dictionary with gaps (table route), <= 18 bits
        "us": rng.integers(0, 29, n).astype(np.int32),                   # 5-bit source
        "ur": rng.integers(-2**31, 2**31 - 1, n).astype(np.int32),       # raw INT
        "r": rng.integers(0, 1000, n).astype(np.int32),
    }
    schema = {k: "INT" for k in data}
    return build_segment("oct", data, schema, no_dictionary_columns=["ur", "r"])


LDS_SHAPES = [
    "SELECT g4, COUNT(*), DISTINCTCOUNTHLL(ua) FROM oct GROUP BY g4 LIMIT 1000",
    "SELECT g4, DISTINCTCOUNTHLL(ua) FROM oct GROUP BY g4 LIMIT 1000",
    "SELECT g1, g2, g3, COUNT(*), DISTINCTCOUNTHLL(un) FROM oct GROUP BY g1, g2, g3 LIMIT 1000",
    "SELECT g5, g3, DISTINCTCOUNTHLL(ul), COUNT(*) FROM oct GROUP BY g5, g3 LIMIT 1000",
    "SELECT g6, COUNT(*), DISTINCTCOUNTHLL(ur) FROM oct GROUP BY g6 LIMIT 1000",
    "SELECT g7, COUNT(*), DISTINCTCOUNTHLL(us) FROM oct GROUP BY g7 LIMIT 1000",
    "SELECT g8, COUNT(*), DISTINCTCOUNTHLL(ua) FROM oct GROUP BY g8 LIMIT 1000",
    "SELECT g2, g1, g4, g3, COUNT(*), DISTINCTCOUNTHLL(ul, 6) FROM oct GROUP BY g2, g1, g4, g3 LIMIT 1000",   # 480 groups x 64 registers
    # no GROUP BY
    "SELECT DISTINCTCOUNTHLL(ua), COUNT(*) FROM oct WHERE r >= 0",
    "SELECT DISTINCTCOUNTHLL(ur) FROM oct WHERE g4 < 9",
    # log2m
    "SELECT g1, DISTINCTCOUNTHLL(ua, 12), COUNT(*) FROM oct GROUP BY g1 LIMIT 10",
    "SELECT g4, DISTINCTCOUNTHLL(ul, 4) FROM oct GROUP BY g4 LIMIT 100",
    "SELECT g3, DISTINCTCOUNTHLL(un, 10), COUNT(*) FROM oct WHERE r < 500 GROUP BY g3 LIMIT 100",
    # filters in front: match words
    "SELECT g4, COUNT(*), DISTINCTCOUNTHLL(ua) FROM oct WHERE r BETWEEN 100 AND 600 GROUP BY g4 LIMIT 1000",
    "SELECT g6, COUNT(*), DISTINCTCOUNTHLL(ul) FROM oct WHERE g3 IN (1, 2) OR g1 = 0 GROUP BY g6 LIMIT 1000",
    # DISTINCTCOUNT: dictId sets in LDS
    "SELECT g4, COUNT(*), DISTINCTCOUNT(g7) FROM oct GROUP BY g4 LIMIT 1000",
    "SELECT DISTINCTCOUNT(ua) FROM oct WHERE g2 = 1",
    "SELECT g3, DISTINCTCOUNT(un) FROM oct WHERE r > 300 GROUP BY g3 LIMIT 1000",
]


@pytest.fixture(scope="module")
def oct_seg(gpu_api, oracle_api):
    g, o = both(gpu_api, oracle_api, make_host(200_003))
    yield g, o
    g.destroy()
    o.destroy()


@pytest.mark.parametrize("sql", LDS_SHAPES)
def test_lds_resident_states(oct_seg, sql):
    g, o = oct_seg
    run(g, o, sql, kernels=("pg_oct_l", "pg_oct_lm"))


@pytest.mark.parametrize("n", [1, 7, 8, 9, 511, 512, 513, 2047, 2048, 2049, 16_385, 70_001])
def test_lds_resident_sizes(gpu_api, oracle_api, n):
    g, o = both(gpu_api, oracle_api, make_host(n, seed=n))
    for sql in (LDS_SHAPES[0], LDS_SHAPES[2], LDS_SHAPES[4], LDS_SHAPES[13], LDS_SHAPES[15]):
        run(g, o, sql, kernels=("pg_oct_l", "pg_oct_lm"))
    g.destroy()
    o.destroy()


def test_round3_kernels_agree(gpu_api, oracle_api, gpu_knobs):
    """PG_NO_OCT: the interpreter runs the same plans (the A/B knob of the variants table)."""
    gpu_knobs(PG_NO_OCT="1")
    g, o = both(gpu_api, oracle_api, make_host(50_001, seed=3))
    gb = run(g, o, LDS_SHAPES[0], kernels=None)
    assert gb.stats.kernel.decode() == "pg_generic_query_l"
    g.destroy()
    o.destroy()


def test_byte_registers_in_lds_agree(gpu_api, oracle_api, gpu_knobs):
    """PG_OCT_BYTE_REGS: key spaces small enough for dword registers in LDS (ds_max_u32 offers) run with byte registers (compare-and-swap)
    instead — the layout larger key spaces use; the A/B knob of the variants table."""
    gpu_knobs(PG_OCT_BYTE_REGS="1")
    g, o = both(gpu_api, oracle_api, make_host(100_003, seed=5))
    for sql in (LDS_SHAPES[0], LDS_SHAPES[2], LDS_SHAPES[8], LDS_SHAPES[10], LDS_SHAPES[13]):
        run(g, o, sql, kernels=("pg_oct_l", "pg_oct_lm"))
    g.destroy()
    o.destroy()


# ---- pruned offers ---------------------------------------------------------------------------------------------------------------------
PRUNED_SHAPES = [
    synth.QUERY_CFG5,
    "SELECT h1, h2, h3, h4, COUNT(*), DISTINCTCOUNTHLL(u) FROM gpuBench WHERE h2 < 5 AND u > 1000 GROUP BY h1, h2, h3, h4 LIMIT 20000",
    "SELECT h1, h2, h3, h4, DISTINCTCOUNTHLL(u) FROM gpuBench GROUP BY h1, h2, h3, h4 LIMIT 20000",
    "SELECT h4, h3, h2, h1, COUNT(*), DISTINCTCOUNTHLL(u, 6) FROM gpuBench WHERE h1 <> 3 GROUP BY h4, h3, h2, h1 LIMIT 20000",
]


@pytest.mark.parametrize("passes", ["1", "0.5,1", "0.1,0.4,1", "0.02,0.08,0.3,1", "0.01,0.02,0.05,0.3,1"])
@pytest.mark.parametrize("n", [2049, 300_007])
def test_pruned_offers_config5(gpu_api, oracle_api, gpu_knobs, passes, n):
    gpu_knobs(PG_OCT_MIN_DOCS="0")
    gpu_knobs(PG_OCT_PASSES=passes)
    host = synth.generate_segment(n, segment_index=3, columns=synth.CFG5_COLUMNS, native=(n > 200_000))
    g, o = both(gpu_api, oracle_api, host)
    for sql in PRUNED_SHAPES:
        run(g, o, sql, kernels=("pg_oct_pruned_group_by",))
    g.destroy()
    o.destroy()


@pytest.mark.parametrize("n", [1, 31, 513, 4097, 16_385, 1_000_003])
def test_pruned_offers_sizes(gpu_api, oracle_api, gpu_knobs, n):
    gpu_knobs(PG_OCT_MIN_DOCS="0")
    gpu_knobs(PG_OCT_PASSES="0.05,0.25,1")
    host = synth.generate_segment(n, segment_index=5, columns=synth.CFG5_COLUMNS, native=(n > 200_000))
    g, o = both(gpu_api, oracle_api, host)
    run(g, o, PRUNED_SHAPES[0], kernels=("pg_oct_pruned_group_by",))
    run(g, o, PRUNED_SHAPES[1], kernels=("pg_oct_pruned_group_by",))
    g.destroy()
    o.destroy()


def test_pruned_offers_skew_and_empty_registers(gpu_api, oracle_api, gpu_knobs):
    """Groups whose values are few keep registers at zero (floor 0: nothing is ever pruned there), one group takes half of the docs, the
    source's dictionary is not arithmetic (table route) in one query and raw INT in the other."""
    gpu_knobs(PG_OCT_MIN_DOCS="0")
    gpu_knobs(PG_OCT_PASSES="0.05,0.2,1")
    rng = np.random.default_rng(9)
    n = 400_003
    k1 = rng.integers(0, 150, n).astype(np.int32)
    k1[rng.random(n) < 0.5] = 17                                    # one heavy group
    k2 = rng.integers(0, 100, n).astype(np.int32)                    # 150 x 100 = 15 000 keys x 256 registers: beyond LDS; counters + floors fit
    v = rng.integers(0, 10**6, n).astype(np.int32)
    v[k2 < 40] = v[k2 < 40] % 5                                      # a third of the groups sees 5 distinct values (floor 0 for ever)
    data = {"k1": k1, "k2": k2, "v": v, "vr": v.copy(), "r": rng.integers(0, 100, n).astype(np.int32)}
    host = build_segment("skew", data, {k: "INT" for k in data}, no_dictionary_columns=["vr", "r"])
    g, o = both(gpu_api, oracle_api, host)
    run(g, o, "SELECT k1, k2, COUNT(*), DISTINCTCOUNTHLL(v) FROM skew GROUP BY k1, k2 LIMIT 100000", kernels=("pg_oct_pruned_group_by",))
    run(g, o, "SELECT k1, k2, COUNT(*), DISTINCTCOUNTHLL(vr) FROM skew WHERE r < 70 GROUP BY k1, k2 LIMIT 100000", kernels=("pg_oct_pruned_group_by",))
    g.destroy()
    o.destroy()


def test_pruned_offers_need_many_distinct_values(gpu_api, oracle_api, gpu_knobs):
    """A source of few distinct values (16 here) never fills the registers: the planner keeps the plain partition pipeline; the knob
    PG_OCT_ANY_CARDINALITY forces the passes (every offer survives every pass) and the result is the same."""
    gpu_knobs(PG_OCT_MIN_DOCS="0")
    host = synth.generate_segment(120_001, segment_index=8, columns=synth.CFG5_COLUMNS, native=False)
    sql = "SELECT h1, h2, h3, h4, COUNT(*), DISTINCTCOUNTHLL(h3) FROM gpuBench GROUP BY h1, h2, h3, h4 LIMIT 20000"
    g, o = both(gpu_api, oracle_api, host)
    run(g, o, sql, kernels=("pg_part_group_by",))
    g.destroy()
    gpu_knobs(PG_OCT_ANY_CARDINALITY="1")
    gpu_knobs(PG_OCT_PASSES="0.1,0.5,1")
    g = NativeSegment(gpu_api, host)
    run(g, o, sql, kernels=("pg_oct_pruned_group_by",))
    g.destroy()
    o.destroy()


def test_pruned_offers_default_threshold(gpu_api, oracle_api, gpu_knobs):
    """Without the knobs a segment with fewer than 16 offers per register (12 800 groups x 256 registers: 52 M docs) keeps the partition
    pipeline — its floors would not rise; with the threshold lowered the same docs take the pruned passes and answer the same.  (The
    default's positive side is the full-size test: tests/test_gpu_full_size.py.)"""
    seg = synth.generate_segment(1_200_000, segment_index=6, columns=synth.CFG5_COLUMNS, native=True)
    g, o = both(gpu_api, oracle_api, seg)
    run(g, o, synth.QUERY_CFG5, kernels=("pg_part_group_by",))
    g.destroy()
    gpu_knobs(PG_OCT_MIN_DOCS="1000000")
    g = NativeSegment(gpu_api, seg)   # (plans are cached per segment: a new one sees the knob)
    run(g, o, synth.QUERY_CFG5, kernels=("pg_oct_pruned_group_by",))
    g.destroy()
    o.destroy()



COUNT_ONLY_SHAPES = [
    "SELECT g4, COUNT(*) FROM oct GROUP BY g4 LIMIT 1000",
    "SELECT g1, g2, g3, g4, COUNT(*) FROM oct GROUP BY g1, g2, g3, g4 LIMIT 1000",            # four columns of <= 4 bits: two-dword buffers
    "SELECT g8, g6, COUNT(*) FROM oct GROUP BY g8, g6 LIMIT 100000",                      # 8 000 groups: no replicas; three-dword buffers
    "SELECT g5, g6, g2, COUNT(*) FROM oct GROUP BY g5, g6, g2 LIMIT 100000",              # three wide columns: three buffers in rotation
    "SELECT g8, g7, COUNT(*) FROM oct GROUP BY g8, g7 LIMIT 100000",                      # 20 000 groups: beyond one LDS table, not this route
    "SELECT g5, g6, COUNT(*) FROM oct WHERE r BETWEEN 100 AND 600 GROUP BY g5, g6 LIMIT 1000",   # behind a filter: the fused kernels, not this route
    "SELECT g3, COUNT(*) FROM oct WHERE g1 = 0 OR g2 = 2 GROUP BY g3 LIMIT 1000",
]


def _count_only_kernels(sql):
    return None if "g8, g7" in sql or "WHERE" in sql else ("pg_oct_c",)


@pytest.mark.parametrize("n", [1, 9, 2049, 200_003])
def test_count_only_group_by_in_the_oct_layout(gpu_api, oracle_api, gpu_knobs, n):
    """COUNT(*) GROUP BY <= 4 columns of <= 8 bits over every doc: the oct-layout decode without a source column (plan_oct, count_only ->
    pg_oct_c, four load buffers); the default threshold keeps small segments on the quad kernels, so the test lowers it — and pg_oct_l (two
    buffers) and the quad kernels give the same rows."""
    gpu_knobs(PG_OCT_COUNT_MIN_DOCS="0")
    g, o = both(gpu_api, oracle_api, make_host(n, seed=n + 5))
    rows = [run(g, o, sql, kernels=_count_only_kernels(sql)).rows() for sql in COUNT_ONLY_SHAPES]
    for knob, names in (("PG_NO_OCT_COUNT_KERNEL", ("pg_oct_l",)), ("PG_NO_OCT_COUNT", None)):
        g.destroy()
        o.destroy()
        gpu_knobs(**{knob: "1"})   # plans are cached per segment: a fresh one
        g, o = both(gpu_api, oracle_api, make_host(n, seed=n + 5))
        for sql, r in zip(COUNT_ONLY_SHAPES, rows):
            gb = run(g, o, sql, kernels=names if _count_only_kernels(sql) else None)
            if names is None:
                assert gb.stats.num_docs_scanned == 0 or not gb.stats.kernel.decode().startswith("pg_oct")
            assert gb.rows() == r
    g.destroy()
    o.destroy()


@pytest.mark.parametrize("n", [9_030_011, 26_000_001])
def test_count_only_rotation_over_many_tiles(gpu_api, oracle_api, n):
    """Enough docs for every wavefront to own several wave tiles (16 wavefronts x 256 workgroups x 2048 docs = 8.4 M docs per round): the
    buffers' rotation, its partial last round with three buffers, the clamped loads past the last tile and the ragged last sub-tile."""
    from pinot_amd import synth
    host = synth.generate_segment(n, segment_index=2, columns=["h1", "h2", "h3", "h4", "g1", "g2"])
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    for sql in ("SELECT h1, h2, h3, h4, COUNT(*) FROM t GROUP BY h1, h2, h3, h4 LIMIT 20000",
                "SELECT h3, COUNT(*) FROM t GROUP BY h3",
                "SELECT g1, COUNT(*) FROM t GROUP BY g1 LIMIT 1000",
                "SELECT g1, g2, COUNT(*) FROM t GROUP BY g1, g2 LIMIT 10000",
                "SELECT g2, h1, h2, COUNT(*) FROM t GROUP BY g2, h1, h2 LIMIT 10000"):
        run(g, o, sql, kernels=("pg_oct_c",))
    g.destroy()
    o.destroy()
