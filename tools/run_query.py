"""Runs ONE query of the gpuBench set repeatedly on a synthetic segment (dev tool for rocprofv3 --pmc passes)."""
import argparse
import ctypes as C
import os
import statistics
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from pinot_amd import capi, synth
from pinot_amd.executor import NativeSegment
from pinot_amd.query import CQuery, parse_sql
from pinot_amd.segment import HostSegment

QUERIES = {
    "cfg2": synth.QUERY_CFG2,
    "postings": "SELECT COUNT(*) FROM t WHERE c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1)",
    "cfg3filter": "SELECT COUNT(*) FROM t WHERE c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1) AND r_int BETWEEN 250000 AND 749999",
    "groupall": "SELECT g1, SUM(m), MAX(m) FROM t GROUP BY g1",
    "sumall": "SELECT SUM(m) FROM t",
    "cfg3": synth.QUERY_CFG3,
    "northstar": synth.QUERY_NORTH_STAR,
    "g1eq": "SELECT COUNT(*) FROM t WHERE g1 = 7",
    "cfg5": synth.QUERY_CFG5,
    "cfg5count": "SELECT h1, h2, h3, h4, COUNT(*) FROM gpuBench GROUP BY h1, h2, h3, h4 LIMIT 20000",
    "hllonly": "SELECT DISTINCTCOUNTHLL(u) FROM gpuBench",
    "hllg1": "SELECT g1, DISTINCTCOUNTHLL(u) FROM gpuBench GROUP BY g1",
}
ap = argparse.ArgumentParser()
ap.add_argument("query")
ap.add_argument("--docs", type=int, default=200_000_000)
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
api = capi.gpu_api()
api.call("init", 0)
seg = NativeSegment(api, HostSegment("prof", args.docs))
for name in (synth.CFG5_COLUMNS + ["g1"] if args.query in ("cfg5", "cfg5count", "hllonly", "hllg1") else synth.CFG3_COLUMNS):
    one = synth.generate_segment(args.docs, columns=[name])
    seg.add_column(one.columns[name], keep_host_buffers=False)
qc = parse_sql(QUERIES[args.query])
qc.flags |= capi.QUERY_FLAG_PROFILE
cq = CQuery(qc)
ms = []
for i in range(args.reps):
    h = C.c_void_p()
    api.call("query_exec", seg.handle, cq.ptr(), C.byref(h))
    st = capi.PgExecStats()
    api.call("result_stats", h, C.byref(st))
    api.call("result_free", h)
    ms.append(st.device_ms_aggregate)
print(args.query, "median ms", statistics.median(ms), "matched", st.num_docs_scanned)
