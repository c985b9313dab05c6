"""One process, N segments, one worker thread per segment, host-side combine — the reference's deployment shape
(BaseCombineOperator.java:81-142: one task per segment; GroupByCombineOperator.java:102-165: upsert by key VALUES).  CPU leg:
the oracle stands in for the per-segment executor (there is no GPU here); the merged table is checked against a brute-force
numpy evaluation over the concatenated docs.  The GPU leg of the same shape (pg_segment_create_on_device + pg_result_merge /
pg_result_all_reduce) is tests/test_gpu_multi.py."""
import threading

import numpy as np

from pinot_amd import synth
from pinot_amd.executor import GroupByCombineOperator, NativeSegment


def test_threaded_one_process_n_segments(oracle_api):
    n_seg = 6
    hosts = [synth.generate_segment(40_009 + 13 * i, segment_index=i, columns=synth.CFG3_COLUMNS) for i in range(n_seg)]
    segs = [NativeSegment(oracle_api, h) for h in hosts]
    q = synth.QUERY_NORTH_STAR
    blocks = [None] * n_seg

    def work(i):
        blocks[i] = segs[i].execute(q)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(n_seg)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    merged = GroupByCombineOperator(blocks).merge()
    # brute force over the concatenation
    cols = {c: np.concatenate([synth.values_numpy(synth.GPU_BENCH[c], synth.SEED_BASE ^ i, h.total_docs)
                               for i, h in enumerate(hosts)]) for c in ("c_inv1", "c_inv2", "r_int", "g1", "g2", "m")}
    sel = np.isin(cols["c_inv1"], [0, 1, 2, 3]) & np.isin(cols["c_inv2"], [0, 1]) & (cols["r_int"] >= 250000) & (cols["r_int"] <= 749999)
    expect = {}
    g1, g2, m = cols["g1"][sel], cols["g2"][sel], cols["m"][sel].astype(np.int64)
    key = g1.astype(np.int64) * 1000 + g2
    order = np.argsort(key, kind="stable")
    uk, start = np.unique(key[order], return_index=True)
    sums = np.add.reduceat(m[order], start)
    for k, s in zip(uk.tolist(), sums.tolist()):
        expect[(k // 1000, k % 1000)] = [float(s)]
    assert merged == expect
    for s in segs:
        s.destroy()
