"""numEntriesScannedInFilter of leapfrogged filter shapes, counted on the device (pinot_amd/csrc/pg_filter_stats_tiles.h).

The reference's count is a property of its iterator automaton (AndDocIdIterator.java:37-66, OrDocIdIterator.java:57-119, NotDocIdIterator.java:45-70,
SVScanDocIdIterator.java:76-112).  Three evaluations of it must agree: the oracle's (doc-at-a-time iterators over column values), the library's host
walk over the leaves' match bitmaps (PG_FILTER_STATS_HOST=1) and the tile automaton on the device — at sizes from one doc to 10^7, on shapes with
scans, inverted-index and sorted leaves, ORs and NOTs under an AND, ANDs inside those ORs, and drained ORs / NOTs around them.  CPU part: the host model of the tile algorithm
against a sequential restatement (tests/filter_stats_tiles_test.cpp)."""
import os
import subprocess

import numpy as np
import pytest

from pinot_amd import capi
from pinot_amd.executor import NativeSegment
from pinot_amd.query import parse_sql
from pinot_amd.segment import build_segment

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tile_model_matches_the_sequential_automaton(tmp_path):
    binary = str(tmp_path / "filter_stats_tiles_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "pinot_amd", "csrc"),
                           os.path.join(ROOT, "tests", "filter_stats_tiles_test.cpp"), "-o", binary])
    out = subprocess.run([binary, "600"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("OK 600 rounds")


def _segment(n, seed):
    rng = np.random.default_rng(seed)
    data = {
        "ci": rng.integers(0, 8, n).astype(np.int32),            # inverted index
        "so": np.sort(rng.integers(0, 50, n)).astype(np.int32),  # sorted
        "u": rng.integers(0, 1000, n).astype(np.int32),          # dictionary, scanned
        "g": (rng.integers(0, 40, n) * 3).astype(np.int32),      # dictionary, scanned
        "r": rng.integers(0, 1_000_000, n).astype(np.int32),     # raw, scanned
        "k": rng.integers(-100, 100, n).astype(np.int32),        # raw, scanned
        "b": (np.arange(n) // 5000 % 2).astype(np.int32),        # raw, long runs of equal values (tiles without a match)
        "m": rng.integers(0, 1 << 20, n).astype(np.int32),
    }
    schema = {c: "INT" for c in data}
    return build_segment(f"fs_{n}", data, schema, inverted_index_columns=["ci"], no_dictionary_columns=["r", "k", "b", "m"])


# (filter, counted on the device?) — the host walk keeps a NOT over a compound child, and an OR / NOT inside an OR, under an AND
SHAPES = [
    ("r < 500000 AND k > 0", True),                                              # leapfrog of two scans
    ("r < 100000 AND k > 50 AND u < 300", True),                                 # ... of three
    ("r < 999000 AND k >= -100 AND g < 117", True),                              # nearly everything matches
    ("r < 1000 AND k = 7", True),                                                # nearly nothing
    ("b = 1 AND r < 300000", True),                                              # runs: whole tiles without a match of one child
    ("ci = 3 AND (r < 200000 OR k > 60)", True),                                 # index AND OR(scan, scan)
    ("ci IN (1, 2) AND r < 700000 AND (u < 50 OR k < -90)", True),               # index AND scan (applyAnd) AND OR(scan, scan)
    ("r < 400000 AND (k > 20 OR ci = 5)", True),                                 # scan AND OR(scan, index)
    ("so < 10 AND (r < 300000 OR u > 900)", True),                               # sorted AND OR
    ("so BETWEEN 5 AND 30 AND ci <> 2 AND (k > 90 OR b = 1)", True),             # sorted AND index (merged) AND OR
    ("(r < 300000 AND k > 0) OR u = 17", True),                                  # drained OR over an AND and a scan
    ("NOT (r < 300000 AND k > 0)", True),                                        # drained NOT over an AND
    ("NOT (r < 300000 OR k > 0)", True),                                         # drained NOT over an OR: every scan runs to the end
    ("(r < 100000 AND k > 0) OR (u < 100 AND g > 30) OR ci = 1", True),          # two ANDs drained side by side
    ("(r < 200000 OR k > 80) AND (u < 200 OR b = 0)", True),                     # AND of two ORs
    ("ci = 3 AND NOT (r < 200000)", True),                                       # NOT over a scan under an AND: advance() resets + batches of next()
    ("r < 500000 AND NOT (k > 0)", True),                                        # ... beside a scan
    ("NOT (r < 300000) AND NOT (k > 50) AND u < 500", True),                     # two of them
    ("ci = 3 AND NOT (b = 1)", True),                                            # runs of 5 000 matches the NOT steps over one next() at a time
    ("so < 10 AND NOT (r < 990000)", True),                                      # nearly everything matches the scan
    ("ci <> 1 AND NOT (r < 3000)", True),                                        # nearly nothing does: batches far apart
    ("NOT (ci = 2) AND r < 100000 AND NOT (k < -95)", True),                     # NOT over an index leaf (nothing to count) next to one over a scan
    ("(ci = 3 AND NOT (r < 200000)) OR (k > 90 AND NOT (u < 500))", True),       # two such ANDs drained by an OR
    ("ci = 3 AND (r < 200000 OR NOT (k > 50))", True),                           # a NOT inside an OR: it receives the targets its cursor lies before
    ("r < 600000 AND (NOT (b = 1) OR NOT (k < 0) OR u = 5)", True),              # two of them, one over long runs
    ("ci = 3 AND NOT (r < 200000 OR k > 50)", True),                             # NOT over an OR of leaves under an AND: each scan's own advances and episodes
    ("r < 700000 AND NOT (k > 90 OR b = 1 OR ci = 2)", True),                    # ... scans, long runs and an index leaf in the union
    ("u < 900 AND NOT (so < 3 OR ci = 1)", True),                                # ... no scan in it: nothing to count there
    ("ci <> 0 AND NOT (r < 100000 OR k > 95) AND NOT (u < 30)", True),           # ... beside a NOT over one scan
    ("ci = 3 AND NOT (r < 200000 AND k > 50)", False),                           # NOT over an AND under an AND: the host walk
    ("r < 200000 AND (k > 0 OR (u < 500 AND g < 60))", True),                    # an AND inside an OR under an AND: started at the OR's targets
    ("ci = 3 AND (r < 200000 OR (k > 0 AND u < 500) OR (g < 60 AND b = 1))", True),              # two of them beside a scan
    ("(r < 500000 OR (k > 0 AND NOT (u < 500))) AND ci <> 4", True),                              # ... with a NOT over a scan inside
    ("r < 900000 AND (k > -50 OR (u < 800 AND (g < 90 OR (b = 1 AND so > 20))))", True),          # two levels down
    ("(ci = 1 AND so < 25 AND (r < 300000 OR (k > 0 AND ci <> 3 AND u > 100))) OR g = 30", True),   # merged index children at both levels, drained by an OR
]


@pytest.fixture(scope="module", params=[1, 2047, 2049, 70_001, 1_000_003, 10_000_019])
def pair(request, gpu_api, oracle_api):
    n = request.param
    host = _segment(n, seed=n % 977)
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    yield g, o, n
    g.destroy()
    o.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("where,on_device", SHAPES)
def test_device_count_equals_oracle_and_host_walk(pair, gpu_knobs, where, on_device):
    g, o, n = pair
    sql = f"SELECT ci, COUNT(*), SUM(m) FROM t WHERE {where} GROUP BY ci LIMIT 100"
    ob = o.execute(sql)
    qc = parse_sql(sql)
    qc.flags |= capi.QUERY_FLAG_EXACT_FILTER_STATS
    gb = g.execute(qc)
    assert gb.rows() == ob.rows()
    assert gb.stats.stats_exact == 1
    assert gb.stats.num_docs_scanned == ob.stats.num_docs_scanned
    assert gb.stats.num_entries_scanned_in_filter == ob.stats.num_entries_scanned_in_filter, (where, n)
    if n >= 2047:   # (a segment of one doc folds most of these filters into constants: nothing left to count)
        assert gb.stats.filter_stats_path == (2 if on_device else 1), where
    if n <= 1_000_003:   # the host walk over the same bitmaps (seconds per query beyond that)
        gpu_knobs(PG_FILTER_STATS_HOST=1)
        qh = parse_sql(sql)
        qh.flags |= capi.QUERY_FLAG_EXACT_FILTER_STATS
        hb = g.execute(qh)
        assert hb.stats.filter_stats_path <= 1
        assert hb.stats.num_entries_scanned_in_filter == ob.stats.num_entries_scanned_in_filter, (where, n)


@pytest.mark.gpu
def test_filter_only_entry_point_counts_on_the_device(pair):
    """pg_filter_exec (FilterOperator + DocIdSetOperator) reports the same statistic"""
    g, o, n = pair
    if n > 1_000_003:
        pytest.skip("covered by the query entry point")
    sql = "SELECT COUNT(*) FROM t WHERE ci = 3 AND (r < 200000 OR k > 60)"
    gd, od = g.filter(parse_sql(sql)), o.filter(parse_sql(sql))
    assert gd.cardinality() == od.cardinality()
    assert gd.stats().num_entries_scanned_in_filter == od.stats().num_entries_scanned_in_filter
    if n >= 2047:
        assert gd.stats().filter_stats_path == 2
