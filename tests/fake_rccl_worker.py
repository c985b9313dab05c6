"""Worker of tests/test_gpu_fake_rccl.py: one process, WORLD ranks (threads), every rank's segment and communicator on device 0, the
collectives served by the test double (PG_RCCL_LIBRARY=tests/fake_rccl/libfake_rccl.so).  Runs the body of
test_all_reduce_two_devices (tests/test_gpu_multi.py) with WORLD ranks: pg_result_all_reduce against GroupByCombineOperator over the
oracle's blocks (GroupByCombineOperator.java:102-165, IndexedTable.java:90-120), the refusals on every rank, the survival of the
communicator, both ways of creating it.  Prints one JSON line; any assertion kills the process (non-zero exit)."""
import ctypes as C
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pinot_amd import capi, synth   # noqa: E402
from pinot_amd.executor import Comm, GroupByCombineOperator, NativeSegment   # noqa: E402
from pinot_amd.segment import build_segment   # noqa: E402
from tests.oracle_binding import load_oracle   # noqa: E402
from tests.test_gpu_multi import QUERIES   # noqa: E402

CFG5_QUERIES = [
    synth.QUERY_CFG5,
    "SELECT h1, DISTINCTCOUNT(u), DISTINCTCOUNTHLL(u), COUNT(*) FROM gpuBench GROUP BY h1",
    "SELECT h1, h2, MIN(u), MAX(u), SUM(u) FROM gpuBench WHERE h3 < 5 GROUP BY h1, h2 LIMIT 1000",
]


def all_reduce_in_threads(results, comms, timeout=120):
    outcome = [None] * len(results)

    def work(i):
        try:
            results[i].all_reduce(comms[i])
        except capi.NativeError as e:
            outcome[i] = e.status
    ts = [threading.Thread(target=work, args=(i,), daemon=True) for i in range(len(results))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=timeout)
    assert not any(t.is_alive() for t in ts), "a rank is stuck inside pg_result_all_reduce"
    return outcome


def execute_in_threads(segs, q):
    """One querying thread per rank, as the worker threads of a server (BaseCombineOperator.java:97-142)."""
    out = [None] * len(segs)
    errors = []

    def work(i):
        try:
            out[i] = segs[i].execute_native(q, keep_device_table=True)
        except Exception as e:   # noqa: BLE001
            errors.append(e)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(segs))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    return out


def check_merges(api, ora_api, comms, world, columns, queries, docs):
    hosts = [synth.generate_segment(docs + 17 * i, segment_index=i, columns=columns) for i in range(world)]
    gpu = [NativeSegment(api, h, device=0) for h in hosts]
    ora = [NativeSegment(ora_api, h) for h in hosts]
    n = 0
    for q in queries:
        results = execute_in_threads(gpu, q)
        assert all_reduce_in_threads(results, comms) == [None] * world, q
        oblocks = [o.execute(q) for o in ora]
        expect = GroupByCombineOperator(oblocks).merge()
        for r in results:
            b = r.block()
            assert b.rows() == expect, q   # every rank holds the merged table
            assert b.stats.num_docs_scanned == sum(x.stats.num_docs_scanned for x in oblocks)
            assert b.stats.num_entries_scanned_in_filter == sum(x.stats.num_entries_scanned_in_filter for x in oblocks)
            assert b.stats.num_total_docs == sum(h.total_docs for h in hosts)
            r.free()
        n += 1
    for s in gpu + ora:
        s.destroy()
    return n


def check_refusals(api, comms, world):
    """Ranks that disagree on the dictionary behind a DISTINCTCOUNT's dictId sets, on the kind of a SUM accumulator, or whose merged SUM could leave int64: EVERY rank gets
    PG_ERR_UNSUPPORTED — the deviant is the LAST rank, so with world 8 seven ranks agree among themselves and must still refuse."""
    rng = np.random.default_rng(3)
    n = 40_000

    def seg_of(g_values, m_values, name):
        data = {"g": g_values.astype(np.int32), "m": m_values}
        schema = {"g": "INT", "m": "LONG" if m_values.dtype == np.int64 else "DOUBLE"}
        return NativeSegment(api, build_segment(name, data, schema, no_dictionary_columns=["m"]), device=0)
    g_a = rng.integers(0, 50, n)
    small = rng.integers(-1000, 1000, n).astype(np.int64)
    as_double = small.astype(np.float64)
    cases = {
        "dictionary": lambda last: seg_of(g_a + 1000, small, "a") if last else seg_of(g_a, small, "a"),
        "double sum on one rank": lambda last: seg_of(g_a, np.where(np.arange(n) == 7, np.nan, as_double), "b") if last else seg_of(g_a, as_double, "b"),
        "overflow bound": lambda last: seg_of(g_a, (small + (1 << 62) // n * 3).astype(np.int64), "c") if last else seg_of(g_a, small, "c"),
    }
    q_sum = "SELECT g, SUM(m), COUNT(*) FROM t GROUP BY g LIMIT 1000"
    # (different group-by dictionaries alone no longer refuse: check_value_keyed_merges; dictId SETS over different dictionaries still do)
    queries = {"dictionary": "SELECT g, DISTINCTCOUNT(g), COUNT(*) FROM t GROUP BY g LIMIT 1000"}
    for what, make in cases.items():
        q = queries.get(what, q_sum)
        segs = [make(i == world - 1) for i in range(world)]
        results = [s.execute_native(q, keep_device_table=True) for s in segs]
        got = all_reduce_in_threads(results, comms)
        assert got == [capi.PG_ERR_UNSUPPORTED] * world, (what, got)
        for r in results:
            r.free()
        for s in segs:
            s.destroy()
    return len(cases)


def check_value_keyed_merges(api, ora_api, comms, world):
    """Every rank's segment has dictionaries of its OWN (as every real Pinot segment does): partially overlapping, disjoint and of different
    cardinalities, INT / LONG / DOUBLE / STRING group-by columns.  pg_result_all_reduce re-keys the tables into the union of the dictionaries
    and merges by value: the rows equal GroupByCombineOperator over the oracle's blocks (GroupByCombineOperator.java:135-144,
    IndexedTable.java:90-120) on every rank."""
    rng = np.random.default_rng(11)
    n = 30_000
    shapes = {
        "overlapping": lambda r: (30 * r, 50 + 7 * r),          # [30 r, 30 r + 50 + 7 r): neighbours share some values, cardinalities differ
        "disjoint": lambda r: (1000 * r, 40),
        "nested": lambda r: (0, 20 + 15 * r),                    # rank 0's dictionary is a prefix of every other
    }
    queries = [
        "SELECT d, COUNT(*), SUM(m), MIN(v), MAX(v) FROM t GROUP BY d LIMIT 100000",
        "SELECT s, d, COUNT(*), SUM(m) FROM t WHERE v < 700 GROUP BY s, d LIMIT 100000",
        "SELECT l, AVG(m), MINMAXRANGE(v) FROM t GROUP BY l LIMIT 100000",
        "SELECT f, s, MAX(m), COUNT(*) FROM t GROUP BY f, s LIMIT 100000",
    ]
    merged = 0
    for what, shape in shapes.items():
        hosts = []
        for r in range(world):
            lo, card = shape(r)
            ids = rng.integers(lo, lo + card, n)
            data = {"d": (ids * 3 - 17).astype(np.int32), "l": ids.astype(np.int64) * 10**10 - 5, "f": (ids % 23).astype(np.float64) / 4.0 - 1.5,
                    "s": np.array([f"city_{x % 37:03d}_{'x' * (x % 5)}" for x in ids], dtype=object),
                    "m": rng.integers(-10**6, 10**6, n).astype(np.int64), "v": rng.integers(0, 1000, n).astype(np.int32)}
            hosts.append(build_segment(f"own_dicts_{what}_{r}", data, {"d": "INT", "l": "LONG", "f": "DOUBLE", "s": "STRING", "m": "LONG", "v": "INT"},
                                       no_dictionary_columns=["m"]))
        gpu = [NativeSegment(api, h, device=0) for h in hosts]
        ora = [NativeSegment(ora_api, h) for h in hosts]
        for q in queries:
            results = execute_in_threads(gpu, q)
            assert all_reduce_in_threads(results, comms) == [None] * world, (what, q)
            oblocks = [o.execute(q) for o in ora]
            expect = GroupByCombineOperator(oblocks).merge()
            for r in results:
                b = r.block()
                assert b.rows() == expect, (what, q)
                assert b.stats.num_docs_scanned == sum(x.stats.num_docs_scanned for x in oblocks)
                assert b.stats.num_total_docs == n * world
                r.free()
            merged += 1
        # two segments per rank, folded on the rank's device first (pg_result_merge re-keys them), the re-keyed heads across the ranks after
        # that: what a server with several segments per GPU does (GpuGroupByCombineOperator: fold per device, then the collective)
        if what == "overlapping":
            q = queries[1]
            heads = []
            for r in range(world):
                a, b_ = gpu[r].execute_native(q, keep_device_table=True), gpu[(r + 1) % world].execute_native(q, keep_device_table=True)
                a.merge(b_)
                b_.free()
                heads.append(a)
            assert all_reduce_in_threads(heads, comms) == [None] * world, "re-keyed heads"
            oblocks = [ora[r].execute(q) for r in range(world)] + [ora[(r + 1) % world].execute(q) for r in range(world)]
            expect = GroupByCombineOperator(oblocks).merge()
            for h_ in heads:
                assert h_.block().rows() == expect
                h_.free()
            merged += 1
        for s_ in gpu + ora:
            s_.destroy()
    return merged


def check_concurrent_sets(api, ora_api, world, n_sets=4):
    """A pool of communicator sets (pg_comm_init_all x K): K cross-GPU merges proceed AT ONCE, each on a set of its own — the threading
    contract of BaseCombineOperator.java:97-142 for merged queries (GpuGroupByCombineOperator takes a set from the pool per merge).  Every
    merge's rows equal GroupByCombineOperator over the oracle's blocks; the double counts no lonely rank and no mismatched collective."""
    sets = [Comm.init_all(api, [0] * world) for _ in range(n_sets)]
    hosts = [synth.generate_segment(40_009 + 13 * i, segment_index=i, columns=synth.CFG3_COLUMNS) for i in range(world)]
    gpu = [NativeSegment(api, h, device=0) for h in hosts]
    ora = [NativeSegment(ora_api, h) for h in hosts]
    queries = [QUERIES[k % len(QUERIES)] for k in range(n_sets)]
    expect = [GroupByCombineOperator([o.execute(q) for o in ora]).merge() for q in queries]
    errors = []

    def one_merge(k):
        try:
            for _ in range(3):
                results = execute_in_threads(gpu, queries[k])
                assert all_reduce_in_threads(results, sets[k]) == [None] * world, k
                for r in results:
                    assert r.block().rows() == expect[k], k
                    r.free()
        except Exception as e:   # noqa: BLE001
            errors.append((k, repr(e)))
    ts = [threading.Thread(target=one_merge, args=(k,)) for k in range(n_sets)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in ts), "a merge is stuck"
    assert not errors, errors
    for s_ in gpu + ora:
        s_.destroy()
    for cs in sets:
        for c in cs:
            c.destroy()
    return n_sets * 3


def main():
    world = int(sys.argv[1])
    fake_path = os.environ["PG_RCCL_LIBRARY"]
    import torch  # noqa: F401  (initialises the ROCm runtime the same way bench.py does)
    api = capi.gpu_api()
    api.call("init", 0)
    ora_api = load_oracle()
    comms = Comm.init_all(api, [0] * world)
    assert all(c.world_size() == world for c in comms)
    merged = check_merges(api, ora_api, comms, world, synth.CFG3_COLUMNS, QUERIES, 60_013)
    merged += check_merges(api, ora_api, comms, world, synth.CFG5_COLUMNS, CFG5_QUERIES, 150_011)
    refused = check_refusals(api, comms, world)
    value_keyed = check_value_keyed_merges(api, ora_api, comms, world)
    concurrent = check_concurrent_sets(api, ora_api, world)
    merged += check_merges(api, ora_api, comms, world, synth.CFG3_COLUMNS, [synth.QUERY_CFG3], 30_011)   # the communicator survives the refusals
    for c in comms:
        c.destroy()
    # the other way in: one pg_comm_init_rank per rank (collective), as one process per GPU would
    uid = Comm.unique_id(api)
    by_rank = [None] * world

    def init(r):
        by_rank[r] = Comm.init_rank(api, 0, world, r, uid)
    ts = [threading.Thread(target=init, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=60)
    assert all(c is not None for c in by_rank)
    merged += check_merges(api, ora_api, by_rank, world, synth.CFG3_COLUMNS, [synth.QUERY_NORTH_STAR], 30_011)
    for c in by_rank:
        c.destroy()
    fake = C.CDLL(fake_path)
    for f in ("fake_rccl_lonely_ranks", "fake_rccl_mismatched_collectives", "fake_rccl_collectives"):
        getattr(fake, f).restype = C.c_int64
    print(json.dumps({"world": world, "merged_queries": merged, "refusal_cases": refused, "value_keyed_merges": value_keyed, "concurrent_merges": concurrent, "lonely_ranks": fake.fake_rccl_lonely_ranks(),
                      "mismatched_collectives": fake.fake_rccl_mismatched_collectives(), "collectives": fake.fake_rccl_collectives()}))


if __name__ == "__main__":
    main()
