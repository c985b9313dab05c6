"""Segment-level group trim (GROUP BY u ORDER BY ... LIMIT 10 over 10^6 groups): wall-clock p50 of pg_query_exec and the bytes the result moves
over PCIe, untrimmed / trimmed at assembly (PG_NO_DEVICE_TRIM=1) / trimmed on the device.  usage: python tools/trim_latency.py [docs]"""
import ctypes as C
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from pinot_amd import capi, synth  # noqa: E402
from pinot_amd.executor import NativeSegment  # noqa: E402
from pinot_amd.query import CQuery, parse_sql  # noqa: E402

docs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
api = capi.gpu_api()
api.call("init", 0)
host = synth.generate_segment(docs, columns=["u", "m"])
QUERIES = [
    ("count(*) desc", "SELECT u, COUNT(*) FROM t GROUP BY u ORDER BY COUNT(*) DESC LIMIT 10", 2),
    ("sum(m) desc, u", "SELECT u, COUNT(*), SUM(m) FROM t GROUP BY u ORDER BY SUM(m) DESC, u LIMIT 10", 2),
    ("u desc", "SELECT u, COUNT(*), SUM(m), MAX(m) FROM t GROUP BY u ORDER BY u DESC LIMIT 10", 3),
]
G = 1_000_000
for mode in ("untrimmed", "assembly", "device"):
    if mode == "assembly":
        os.environ["PG_NO_DEVICE_TRIM"] = "1"
    else:
        os.environ.pop("PG_NO_DEVICE_TRIM", None)
    api.call("options_reload")
    seg = NativeSegment(api, host)
    for name, sql, n_ops in QUERIES:
        qc = parse_sql(sql)
        qc.num_groups_limit = 2_000_000
        qc.min_segment_group_trim_size = -1 if mode == "untrimmed" else 5000
        cq = CQuery(qc)
        wall, groups = [], 0
        for i in range(12):
            h = C.c_void_p()
            t0 = time.perf_counter()
            api.call("query_exec", seg.handle, cq.ptr(), C.byref(h))
            wall.append((time.perf_counter() - t0) * 1e3)
            n = C.c_int32()
            api.call("result_num_groups", h, C.byref(n))
            groups = n.value
            api.call("result_free", h)
        cap = 5000 + 4096
        moved = G * n_ops * 8 if mode != "device" else (n_ops + 1) * cap * 8
        print(f"{mode:10s} {name:16s} groups returned {groups:8d}  p50 {statistics.median(wall[2:]):8.3f} ms  result bytes over PCIe {moved / 1e6:8.2f} MB")
    seg.destroy()
