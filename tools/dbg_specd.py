import sys; sys.path.insert(0,'.')
import torch
from pinot_amd import capi, synth
from pinot_amd.executor import NativeSegment
api = capi.gpu_api(); api.call("init", 0)
h = synth.generate_segment(700_001, segment_index=3, columns=["c_inv1","c_inv2","g1","g2","r_int_d","m_d","m_s","r_int_s"])
g = NativeSegment(api, h)
for q in (synth.QUERY_NORTH_STAR_DICT, synth.QUERY_CFG3_DICT, "SELECT g1, g2, COUNT(*), SUM(m_s) FROM gpuBench WHERE c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1) AND r_int_s BETWEEN 750000 AND 2249999 GROUP BY g1, g2 LIMIT 10000"):
    r = g.execute(q); print(r.stats.kernel.decode(), len(r.rows()))
