"""Dev tool: host timeline (PG_TRACE_HOST) and latency of config 5 answered from its star-tree."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pinot_amd import capi, startree, synth
from pinot_amd.executor import NativeSegment
from pinot_amd.query import parse_sql
api = capi.gpu_api(); api.call("init", 0)
parent = synth.generate_segment(400_000, segment_index=0, columns=list(synth.CFG5_COLUMNS), native=False)
startree.add_star_tree(parent, ["h1", "h2", "h3", "h4"], [("COUNT", "*"), ("DISTINCTCOUNTHLL", "u")], max_leaf_records=10000)
seg = NativeSegment(api, parent)
for flags in (capi.QUERY_FLAG_FINAL_DISTINCT, 0):
    q = parse_sql(synth.QUERY_CFG5); q.flags |= flags
    lat, lib = [], []
    for i in range(60):
        t = time.perf_counter(); b = seg.execute(q)
        if i >= 10: lat.append((time.perf_counter() - t) * 1e3); lib.append(b.stats.host_ms_total)
    print(f"flags {flags:#x}: wall p50 {statistics.median(lat):.3f} ms, library {statistics.median(lib):.3f} ms, kernel {b.stats.kernel.decode()}", file=sys.stderr)
