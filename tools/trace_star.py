"""Host-side timeline (PG_TRACE_HOST=1) of the star-tree route of BASELINE config 5."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from pinot_amd import capi, startree, synth
from pinot_amd.executor import NativeSegment
from pinot_amd.query import parse_sql
api = capi.gpu_api(); api.call("init", 0)
parent = synth.generate_segment(2_000_000, segment_index=0, columns=list(synth.CFG5_COLUMNS), native=False)
startree.add_star_tree(parent, ["h1", "h2", "h3", "h4"], [("COUNT", "*"), ("DISTINCTCOUNTHLL", "u")], max_leaf_records=10000)
seg = NativeSegment(api, parent)
q = parse_sql(synth.QUERY_CFG5)
for i in range(6):
    t = time.perf_counter(); b = seg.execute(q); print("python total ms", (time.perf_counter() - t) * 1e3, file=sys.stderr)
