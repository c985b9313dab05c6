"""Wall-clock latency of small queries through the C-ABI (dev tool): p50 / p90 of pg_query_exec + pg_result_free, next to the library's own
host total and the kernels' device time.  A/B knobs: PG_NO_DIRECT_RESULT, PG_NO_SPIN_WAIT."""
import argparse
import ctypes as C
import os
import statistics
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from pinot_amd import capi, synth
from pinot_amd.executor import NativeSegment
from pinot_amd.query import CQuery, parse_sql
from pinot_amd.segment import HostSegment

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=100_000_000)
ap.add_argument("--calls", type=int, default=300)
args = ap.parse_args()
api = capi.gpu_api()
api.call("init", 0)
seg = NativeSegment(api, HostSegment("lat", args.docs))
for name in synth.CFG3_COLUMNS:
    one = synth.generate_segment(args.docs, columns=[name])
    seg.add_column(one.columns[name], keep_host_buffers=False)
QUERIES = {
    "cfg2": synth.QUERY_CFG2,
    "cfg3": synth.QUERY_CFG3,
    "postings only count": "SELECT COUNT(*) FROM t WHERE c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1)",
    "sum(m) group g1": "SELECT g1, SUM(m), MAX(m) FROM t GROUP BY g1",
}
for name, sql in QUERIES.items():
    for profile in (False, True):
        qc = parse_sql(sql)
        if profile:
            qc.flags |= capi.QUERY_FLAG_PROFILE
        cq = CQuery(qc)
        wall, host, dev = [], [], []
        for i in range(args.calls + 20):
            h = C.c_void_p()
            t0 = time.perf_counter()
            api.call("query_exec", seg.handle, cq.ptr(), C.byref(h))
            t1 = time.perf_counter()
            st = capi.PgExecStats()
            api.call("result_stats", h, C.byref(st))
            api.call("result_free", h)
            if i >= 20:
                wall.append((t1 - t0) * 1e6)
                host.append(st.host_ms_total * 1e3)
                dev.append(st.device_ms_total * 1e3)
        wall.sort()
        q = lambda f: wall[int(f * (len(wall) - 1))]
        tail = f"  device {statistics.median(dev):7.1f} us" if profile else ""
        print(f"{name:22s} {'profiled' if profile else 'plain   '}  pg_query_exec p50 {q(0.5):7.1f} us  p90 {q(0.9):7.1f}  p99 {q(0.99):7.1f}   "
              f"library host total p50 {statistics.median(host):7.1f} us{tail}")
