#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
timeout 300 python -m pytest tests/test_gpu_partition_pipeline.py -x -q -m gpu -k "limit" 2>&1 | tail -3
PG_TRACE_HOST=1 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -30
import ctypes as C, os, statistics, sys, time
sys.path.insert(0, os.getcwd())
import torch
from pinot_amd import capi, synth
from pinot_amd.executor import NativeSegment
from pinot_amd.query import CQuery, parse_sql
api = capi.gpu_api(); api.call("init", 0)
host = synth.generate_segment(100_000_000, columns=["u", "m"])
seg = NativeSegment(api, host)
for sql in ("SELECT u, COUNT(*) FROM t GROUP BY u LIMIT 10", "SELECT u, COUNT(*), SUM(m), MAX(m) FROM t GROUP BY u LIMIT 10"):
    cq = CQuery(parse_sql(sql))
    wall = []
    for i in range(6):
        h = C.c_void_p(); t0 = time.perf_counter()
        api.call("query_exec", seg.handle, cq.ptr(), C.byref(h)); wall.append((time.perf_counter() - t0) * 1e3)
        st = capi.PgExecStats(); api.call("result_stats", h, C.byref(st)); api.call("result_free", h)
    print(f"{sql[7:40]:34s} kernel {st.kernel.decode()} wall p50 {statistics.median(wall[2:]):7.3f} ms", file=sys.stderr)
PY
