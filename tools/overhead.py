import sys, os, time, statistics
sys.path.insert(0, "/root/repo")
import torch
from pinot_amd import capi, synth, distributed as pd
from pinot_amd.executor import NativeSegment
from pinot_amd.query import parse_sql
from pinot_amd.segment import HostSegment
api = capi.gpu_api(); api.call("init", 0)
docs = 200_000_000
seg = NativeSegment(api, HostSegment("x", docs))
for name in ["c_inv1", "c_inv2", "r_int", "g1", "m"]:
    one = synth.generate_segment(docs, columns=[name]); seg.add_column(one.columns[name], keep_host_buffers=False)
qc = parse_sql(synth.QUERY_CFG3); qc.flags |= capi.QUERY_FLAG_PROFILE
for prof in (True, False):
    if not prof: qc.flags &= ~capi.QUERY_FLAG_PROFILE
    t_exec, t_dense, host_total, dev = [], [], [], []
    for i in range(30):
        t0 = time.perf_counter(); b = seg.execute(qc); t1 = time.perf_counter(); d = pd.dense_from_block(b, [100]); t2 = time.perf_counter()
        if i >= 5:
            t_exec.append((t1 - t0) * 1e3); t_dense.append((t2 - t1) * 1e3); host_total.append(b.stats.host_ms_total); dev.append(b.stats.device_ms_aggregate + b.stats.device_ms_reduce)
    print("profile" if prof else "noprofile", "execute()", statistics.median(t_exec), "dense", statistics.median(t_dense), "native host_ms_total", statistics.median(host_total), "device", statistics.median(dev))
