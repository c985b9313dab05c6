#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests/test_gpu_partition_pipeline.py tests/test_gpu_group_trim.py tests/test_raw_group_by.py tests/test_fuzz.py tests/test_gpu_parity.py tests/test_null_and_valid_docs.py -x -q -m gpu 2>&1 | tail -6
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_q_limit_latency.txt
import ctypes as C, os, statistics, sys, time
sys.path.insert(0, os.getcwd())
import torch
from pinot_amd import capi, synth
from pinot_amd.executor import NativeSegment
from pinot_amd.query import CQuery, parse_sql
api = capi.gpu_api(); api.call("init", 0)
host = synth.generate_segment(100_000_000, columns=["u", "m"])
for mode, env in (("one pass, docId plane", {"PG_NO_LIMIT_PREFIX": "1"}), ("prefix, admission on the host", {"PG_NO_DEVICE_TRIM": "1"}), ("prefix, admission on the device", {})):
    for k in ("PG_NO_LIMIT_PREFIX", "PG_NO_DEVICE_TRIM"):
        os.environ.pop(k, None)
    os.environ.update(env)
    api.call("options_reload")
    seg = NativeSegment(api, host)
    for sql in ("SELECT u, COUNT(*) FROM t GROUP BY u LIMIT 10", "SELECT u, COUNT(*), SUM(m), MAX(m) FROM t GROUP BY u LIMIT 10"):
        cq = CQuery(parse_sql(sql))   # default numGroupsLimit 100 000 over 10^6 groups
        wall = []
        for i in range(12):
            h = C.c_void_p(); t0 = time.perf_counter()
            api.call("query_exec", seg.handle, cq.ptr(), C.byref(h)); wall.append((time.perf_counter() - t0) * 1e3)
            n = C.c_int32(); api.call("result_num_groups", h, C.byref(n)); api.call("result_free", h)
        print(f"{mode:34s} {sql[7:40]:34s} groups {n.value:7d}  wall p50 {statistics.median(wall[2:]):7.3f} ms")
    seg.destroy()
PY
