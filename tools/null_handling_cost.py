"""Dev tool: what query-level null handling costs on the GPU path — the config-3 shape over a segment whose aggregation argument `m` (and, second
row, the group column g1) hold nulls: wall time per call without the flag, with the flag and no nulls, with the flag and nulls."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pinot_amd import capi, formats, synth
from pinot_amd.executor import NativeSegment
from pinot_amd.query import parse_sql
docs = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
api = capi.gpu_api(); api.call("init", 0)
host = synth.generate_segment(docs, segment_index=0, columns=list(synth.CFG3_COLUMNS), native=False)
SQL = ["SELECT g1, SUM(m), MAX(m), COUNT(*) FROM t WHERE c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1) AND r_int BETWEEN 250000 AND 749999 GROUP BY g1 LIMIT 1000",
       "SELECT g1, SUM(m), COUNT(*) FROM t GROUP BY g1 LIMIT 1000"]
def timed(seg, sql, flag):
    q = parse_sql(sql)
    if flag: q.flags |= capi.QUERY_FLAG_NULL_HANDLING
    t = []
    for i in range(12):
        t0 = time.perf_counter(); seg.execute(q); t.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(t[2:])
seg = NativeSegment(api, host)
base = [timed(seg, s, False) for s in SQL]
flag_no_nulls = [timed(seg, s, True) for s in SQL]
seg.destroy()
rng = np.random.default_rng(1)
host.columns["m"].null_vector = np.frombuffer(formats.serialize_roaring(np.flatnonzero(rng.random(docs) < 0.1)), dtype=np.uint8)
seg = NativeSegment(api, host)
null_arg = [timed(seg, s, True) for s in SQL]
seg.destroy()
host.columns["g1"].null_vector = np.frombuffer(formats.serialize_roaring(np.flatnonzero(rng.random(docs) < 0.05)), dtype=np.uint8)
seg = NativeSegment(api, host)
null_arg_key = [timed(seg, s, True) for s in SQL]
for i, s in enumerate(SQL):
    print(f"{docs} docs | {s[:60]}... | no flag {base[i]:.3f} ms | flag, no nulls {flag_no_nulls[i]:.3f} | nulls in m (1 sub-query) {null_arg[i]:.3f} | nulls in m and g1 (2 partitions x 2) {null_arg_key[i]:.3f}", file=sys.stderr)
