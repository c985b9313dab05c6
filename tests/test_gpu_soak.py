"""Long-running-server behaviour of libpinot_gpu.so: thousands of queries and repeated segment load / destroy cycles must not
leak HBM or host memory; swapping the upsert queryableDocIds snapshot while other threads query the same segment must give
every query the answer of one of the snapshots (a running query keeps the plan, and the bitmap, it started with)."""
import threading

import numpy as np
import psutil
import pytest

from pinot_amd import synth
from pinot_amd.executor import NativeSegment
from tests.fuzz_queries import Gen, clone, fuzz_segment


def _free_hbm():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def _hbm_is_ours():
    """Free HBM is a property of the device: under pytest-xdist the other workers' segments come and go on the same GPU (a 26 M-doc segment next
    door read as a 7 GB 'leak'), so the HBM-growth assertions only hold when this process has the device to itself — as in the driver's run."""
    import os
    return not os.environ.get("PYTEST_XDIST_WORKER")


@pytest.mark.gpu
def test_no_memory_growth_over_thousands_of_queries(gpu_api):
    from pinot_amd import capi
    host, data, _ = fuzz_segment(60_000, seed=9)
    g = NativeSegment(gpu_api, host)
    gen = Gen(data, seed=4242)
    queries = [gen.query() for _ in range(150)]

    def run_all():
        for q in queries:
            try:
                g.execute(clone(q))
            except capi.NativeError as e:
                assert e.status == capi.PG_ERR_UNSUPPORTED
    for _ in range(2):
        run_all()                      # warm: per-thread workspaces reach their high-water mark, plans are cached
    free0, rss0 = _free_hbm(), psutil.Process().memory_info().rss
    for _ in range(20):                # 3 000 queries
        run_all()
    free1, rss1 = _free_hbm(), psutil.Process().memory_info().rss
    assert not _hbm_is_ours() or free0 - free1 < 32 << 20, f"HBM grew by {(free0 - free1) >> 20} MB over 3000 queries"
    assert rss1 - rss0 < 128 << 20, f"host RSS grew by {(rss1 - rss0) >> 20} MB over 3000 queries"
    g.destroy()


@pytest.mark.gpu
def test_segment_load_destroy_cycles_return_their_memory(gpu_api):
    host = synth.generate_segment(400_000, segment_index=2)
    first = NativeSegment(gpu_api, host)
    first.execute(synth.QUERY_CFG3)
    first.destroy()
    free0, rss0 = _free_hbm(), psutil.Process().memory_info().rss
    for _ in range(40):
        seg = NativeSegment(gpu_api, host)
        seg.execute(synth.QUERY_CFG3)
        seg.execute(synth.QUERY_CFG5)
        seg.destroy()
    free1, rss1 = _free_hbm(), psutil.Process().memory_info().rss
    assert not _hbm_is_ours() or free0 - free1 < 32 << 20, f"HBM not returned: {(free0 - free1) >> 20} MB after 40 load/destroy cycles"
    assert rss1 - rss0 < 128 << 20, f"host RSS grew by {(rss1 - rss0) >> 20} MB after 40 load/destroy cycles"


@pytest.mark.gpu
def test_snapshot_swaps_under_concurrent_queries(gpu_api, oracle_api):
    n = 300_000
    host = synth.generate_segment(n, segment_index=4, columns=synth.CFG3_COLUMNS)
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    rng = np.random.default_rng(0)
    snapshots = [np.flatnonzero(rng.random(n) < p) for p in (0.9, 0.5, 0.2)]
    q = synth.QUERY_CFG3
    expected = []
    for ids in snapshots:
        o.set_queryable_doc_ids(ids)
        expected.append(o.execute(q).rows())
    o.destroy()
    g.set_queryable_doc_ids(snapshots[0])
    stop = threading.Event()
    errors, seen = [], set()

    def worker():
        try:
            while not stop.is_set():
                rows = g.execute(q).rows()
                which = [i for i, e in enumerate(expected) if e == rows]
                if not which:
                    errors.append("a query saw a mixture of snapshots")
                    return
                seen.add(which[0])
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker) for _ in range(6)]
    for t in threads:
        t.start()
    for i in range(60):
        g.set_queryable_doc_ids(snapshots[i % 3])
    stop.set()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
    assert len(seen) >= 2
    g.destroy()


@pytest.mark.gpu
def test_null_handling_queries_from_several_threads(gpu_api, oracle_api):
    """The null-aware path runs several GPU queries per call and joins them on the host (pg_nullaware.cpp): eight threads over one segment,
    every answer the oracle's; no memory growth over the repeats."""
    from pinot_amd import capi
    host, data, _ = fuzz_segment(60_000, seed=13)
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    gen = Gen(data, seed=777)
    cases = []
    for _ in range(60):
        q = gen.query()
        oq = clone(q)
        oq.flags |= capi.QUERY_FLAG_NULL_HANDLING
        try:
            cases.append((q, o.execute(oq).rows()))
        except capi.NativeError:
            pass
    o.destroy()
    errors = []

    def worker(k):
        try:
            for rep in range(2):
                for i, (q, want) in enumerate(cases):
                    if (i + k) % 2:
                        continue
                    gq = clone(q)
                    gq.flags |= capi.QUERY_FLAG_NULL_HANDLING
                    try:
                        got = g.execute(gq).rows()
                    except capi.NativeError as e:
                        assert e.status == capi.PG_ERR_UNSUPPORTED
                        continue
                    assert got == want, i
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e)[:300])
    for k in range(2):
        worker(k)     # warm up: plans cached, per-thread workspaces at their high-water mark
    import time

    def threaded_phase():
        threads = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        time.sleep(0.5)    # (the threads' contexts — streams, work areas — are released by their thread-local destructors)
        return _free_hbm()
    free1 = threaded_phase()
    assert not errors, errors[:3]
    free2 = threaded_phase()
    assert not errors, errors[:3]
    assert not _hbm_is_ours() or free1 - free2 < 256 << 20, f"HBM grew by {(free1 - free2) >> 20} MB over a second round of eight threads"
    g.destroy()
