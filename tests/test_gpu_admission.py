"""Per-device admission (pg_exec.hip AdmissionGuard, PG_MAX_INFLIGHT): more callers than admitted queries — the surplus waits first come,
first served and every caller still gets its own, correct result (the threading contract of BaseCombineOperator.java:97-142: many worker
threads, one query each, on one segment)."""
import threading

import pytest

from pinot_amd import synth
from pinot_amd.executor import NativeSegment

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("max_inflight", ["2", "16", "0"])
def test_more_callers_than_admitted_queries(gpu_api, oracle_api, gpu_knobs, max_inflight):
    gpu_knobs(PG_MAX_INFLIGHT=max_inflight)
    host = synth.generate_segment(300_007, columns=synth.CFG3_COLUMNS)
    g, o = NativeSegment(gpu_api, host), NativeSegment(oracle_api, host)
    queries = [synth.QUERY_CFG2, synth.QUERY_CFG3, synth.QUERY_NORTH_STAR,
               "SELECT g1, g2, COUNT(*), MIN(r_int) FROM gpuBench WHERE c_inv1 < 5 GROUP BY g1, g2 LIMIT 10000"]
    expect = [o.execute(q).rows() if "GROUP" in q else o.execute(q).aggregation_result() for q in queries]
    errors = []

    def work(k):
        try:
            for i in range(12):
                j = (i + k) % len(queries)
                b = g.execute(queries[j])
                got = b.rows() if "GROUP" in queries[j] else b.aggregation_result()
                assert got == expect[j], queries[j]
        except Exception as e:   # noqa: BLE001
            errors.append(e)
    ts = [threading.Thread(target=work, args=(k,)) for k in range(40)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ts), "a caller is stuck in the admission queue"
    assert not errors, errors[:1]
    g.destroy()
    o.destroy()
