"""Times a set of queries on one synthetic segment with the library's HIP-event timers (dev tool, not a test)."""
import argparse
import ctypes as C
import statistics
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from pinot_amd import capi, synth
from pinot_amd.executor import NativeSegment
from pinot_amd.query import CQuery, parse_sql
from pinot_amd.segment import HostSegment

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=200_000_000)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--only", default="", help="substring of the query names to run (a leading '=' asks for the exact name)")
ap.add_argument("--set", choices=["cfg3", "cfg5", "general", "wide", "upsert", "postings"], default="cfg3")
args = ap.parse_args()
api = capi.gpu_api()
api.call("init", 0)
seg = NativeSegment(api, HostSegment("prof", args.docs))
for name in (synth.CFG5_COLUMNS if args.set == "cfg5" else synth.CFG3_COLUMNS):
    one = synth.generate_segment(args.docs, columns=[name])
    seg.add_column(one.columns[name], keep_host_buffers=False)

if args.set == "wide":
    # columns the synthetic table lacks: a LONG raw metric and an 11-bit dictionary column (built with numpy: small sizes only)
    import numpy as np
    from pinot_amd.segment import build_column
    n = args.docs
    mvals = synth.values_numpy(synth.GPU_BENCH["m"], synth.SEED_BASE, n).astype(np.int64)
    seg.add_column(build_column("m64", mvals * 1000 + 7, "LONG", dictionary=False), keep_host_buffers=False)
    seg.add_column(build_column("d64", (mvals * 0.5).astype(np.float64), "DOUBLE", dictionary=False), keep_host_buffers=False)
    w = (synth.values_numpy(synth.GPU_BENCH["u"], synth.SEED_BASE, n) % 2000).astype(np.int32)
    seg.add_column(build_column("w1", w, "INT"), keep_host_buffers=False)
    del mvals, w

if args.set == "postings":
    # inverted indexes whose containers are NOT bitmaps: a 1000-value column (about 65 docs per dictId and 2^16-doc chunk: array
    # containers) and a column of long runs (run containers); built with numpy / Python: keep --docs around 3e7
    import numpy as np
    from pinot_amd.segment import build_column
    n = args.docs
    rng = np.random.default_rng(1)
    seg.add_column(build_column("s1", rng.integers(0, 1000, n).astype(np.int32), "INT", inverted=True), keep_host_buffers=False)
    seg.add_column(build_column("srun", ((np.arange(n) // 5000) % 50).astype(np.int32), "INT", inverted=True), keep_host_buffers=False)

QUERIES_POSTINGS = {   # bytes per row: the forward-index bytes a scan-only plan would read (postings themselves are tiny here)
    "array postings: s1 = 5 count": ("SELECT COUNT(*) FROM t WHERE s1 = 5", 0.0),
    "array postings: s1 IN (3 ids) count": ("SELECT COUNT(*) FROM t WHERE s1 IN (5, 77, 901)", 0.0),
    "array postings: 40 ids + range + group": ("SELECT g1, SUM(m), MAX(m) FROM t WHERE s1 < 40 AND r_int BETWEEN 250000 AND 749999 GROUP BY g1", 0.0),
    "array postings NOT IN + group": ("SELECT g1, COUNT(*) FROM t WHERE s1 NOT IN (5, 77, 901) AND c_inv2 = 1 GROUP BY g1", 0.0),
    "run postings: srun = 7 sum": ("SELECT SUM(m), COUNT(*) FROM t WHERE srun = 7", 0.0),
    "run postings: srun IN (5 ids) + range + group": ("SELECT g1, SUM(m) FROM t WHERE srun IN (1, 2, 3, 4, 5) AND r_int < 500000 GROUP BY g1", 0.0),
    "run + array postings": ("SELECT g1, SUM(m) FROM t WHERE srun IN (1, 2, 3, 4, 5) AND s1 < 100 GROUP BY g1", 0.0),
    "dense postings (reference point)": ("SELECT g1, SUM(m), MAX(m) FROM t WHERE c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1) AND r_int BETWEEN 250000 AND 749999 GROUP BY g1", 9.625),
}

QUERIES = {
    "cfg2 count(range scan)": (synth.QUERY_CFG2, 4.0),
    "postings only count": ("SELECT COUNT(*) FROM t WHERE c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1)", 0.75),
    "cfg3 filter only count": ("SELECT COUNT(*) FROM t WHERE c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1) AND r_int BETWEEN 250000 AND 749999", 4.75),
    "no filter sum(m) group g1": ("SELECT g1, SUM(m), MAX(m) FROM t GROUP BY g1", 4.875),
    "no filter sum(m)": ("SELECT SUM(m) FROM t", 4.0),
    "cfg3": (synth.QUERY_CFG3, 9.625),
    "northstar": (synth.QUERY_NORTH_STAR, 10.375),
    "g1 scan eq": ("SELECT COUNT(*) FROM t WHERE g1 = 7", 0.875),
}
QUERIES5 = {
    "cfg5": (synth.QUERY_CFG5, 4.375),
    "cfg5 count only": ("SELECT h1, h2, h3, h4, COUNT(*) FROM t GROUP BY h1, h2, h3, h4 LIMIT 20000", 1.875),
    "cfg5 hll(u) no group": ("SELECT DISTINCTCOUNTHLL(u) FROM t", 2.5),
    "cfg5 hll(u) group h1": ("SELECT h1, DISTINCTCOUNTHLL(u) FROM t GROUP BY h1", 3.0),
    "cfg5 distinctcount(u) group h1": ("SELECT h1, DISTINCTCOUNT(u) FROM t GROUP BY h1", 3.0),
    "hll(u) where h2<5": ("SELECT DISTINCTCOUNTHLL(u) FROM t WHERE h2 < 5", 3.0),
    "distinctcount(u) where h2<5": ("SELECT DISTINCTCOUNT(u) FROM t WHERE h2 < 5", 3.0),
    "hll(u) group h1,h2 (160)": ("SELECT h1, h2, DISTINCTCOUNTHLL(u), COUNT(*) FROM t GROUP BY h1, h2", 3.5),
    "count group u (1M groups)": ("SELECT u, COUNT(*) FROM t GROUP BY u LIMIT 2000000", 2.5),
    "count group u,h1 (16M groups)": ("SELECT u, h1, COUNT(*) FROM t WHERE h2 = 3 GROUP BY u, h1 LIMIT 20000000", 3.5),
    "hash: group u,h1,h2 where u<20000": ("SELECT u, h1, h2, COUNT(*), SUM(h3) FROM t WHERE u < 20000 GROUP BY u, h1, h2 LIMIT 10000000", 4.0),
    "hash: group u,h1,h2 where h3=1,h4=2": ("SELECT u, h1, h2, COUNT(*) FROM t WHERE h3 = 1 AND h4 = 2 GROUP BY u, h1, h2 LIMIT 10000000", 4.0),
    "probe radix sum(h3)": ("SELECT h1, h2, h3, h4, COUNT(*), SUM(h3) FROM t GROUP BY h1, h2, h3, h4 LIMIT 20000", 1.875),
    "probe radix hll(h3)": ("SELECT h1, h2, h3, h4, COUNT(*), DISTINCTCOUNTHLL(h3) FROM t GROUP BY h1, h2, h3, h4 LIMIT 20000", 1.875),
    "cfg5 count group h1..h3": ("SELECT h1, h2, h3, COUNT(*) FROM t GROUP BY h1, h2, h3 LIMIT 20000", 1.5),
}
QUERIES_GENERAL = {   # shapes outside the specialised kernels: several scans, OR of scans, tables beyond LDS
    "2 scans + group g1": ("SELECT g1, SUM(m) FROM t WHERE r_int BETWEEN 250000 AND 749999 AND m < 524288 GROUP BY g1", 8.875),
    "postings + 2 scans + group": ("SELECT g1, SUM(m), MAX(m) FROM t WHERE c_inv1 IN (0,1,2,3) AND r_int BETWEEN 250000 AND 749999 AND m < 524288 GROUP BY g1", 9.375),
    "or of 2 scans count": ("SELECT COUNT(*) FROM t WHERE r_int < 100000 OR m > 900000", 8.0),
    "dict scan + raw scan": ("SELECT COUNT(*) FROM t WHERE g1 < 50 AND r_int < 500000", 4.875),
    "group g1,g2,c_inv1 (40k groups)": ("SELECT g1, g2, c_inv1, COUNT(*), SUM(m) FROM t GROUP BY g1, g2, c_inv1 LIMIT 100000", 6.0),
    "filtered 40k groups": ("SELECT g1, g2, c_inv1, SUM(m) FROM t WHERE r_int < 125000 GROUP BY g1, g2, c_inv1 LIMIT 100000", 10.0),
    "group g1,g2,c_inv1,c_inv2 (160k)": ("SELECT g1, g2, c_inv1, c_inv2, COUNT(*), SUM(m), MAX(m) FROM t GROUP BY g1, g2, c_inv1, c_inv2 LIMIT 200000", 6.25),
    "cfg3 filter, 160k groups": ("SELECT g1, g2, c_inv1, c_inv2, SUM(m) FROM t WHERE c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1) AND r_int BETWEEN 250000 AND 749999 GROUP BY g1, g2, c_inv1, c_inv2 LIMIT 200000", 11.0),
    "avg/min/max/count no group": ("SELECT COUNT(*), AVG(m), MIN(r_int), MAX(m) FROM t WHERE c_inv2 = 1", 8.125),
}
QUERIES_WIDE = {   # LDS-table aggregations over 64-bit sources / an 11-bit group column
    "sum(m64) group g1": ("SELECT g1, SUM(m64), MAX(m64) FROM t GROUP BY g1", 8.875),
    "cfg3 filter, sum(m64) group g1": ("SELECT g1, SUM(m64) FROM t WHERE c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1) AND r_int BETWEEN 250000 AND 749999 GROUP BY g1", 13.625),
    "sum(d64) avg(m) group g1": ("SELECT g1, SUM(d64), AVG(m) FROM t WHERE r_int < 500000 GROUP BY g1", 16.875),
    "sum(m) group w1 (2000 groups)": ("SELECT w1, SUM(m), COUNT(*) FROM t GROUP BY w1 LIMIT 5000", 5.375),
    "filtered sum(m64) group w1": ("SELECT w1, SUM(m64) FROM t WHERE r_int BETWEEN 250000 AND 749999 GROUP BY w1 LIMIT 5000", 13.375),
    "sum(m64) no group": ("SELECT SUM(m64), MIN(m64), COUNT(*) FROM t WHERE c_inv2 = 1", 8.125),
}
if args.set == "upsert":   # the same shapes behind an upsert queryableDocIds snapshot (90 % of the docs valid): +1 bit per doc
    import numpy as np
    rng = np.random.default_rng(0)
    bits = rng.random(args.docs) < 0.9
    seg.set_queryable_doc_ids(np.flatnonzero(bits))
    del bits
    QUERIES = {k: (sql, bpr + 0.125) for k, (sql, bpr) in QUERIES.items() if k in ("cfg2 count(range scan)", "cfg3 filter only count", "cfg3", "northstar", "no filter sum(m) group g1")}
if args.set == "cfg5":
    QUERIES = QUERIES5
if args.set == "wide":
    QUERIES = QUERIES_WIDE
if args.set == "general":
    QUERIES = QUERIES_GENERAL
if args.set == "postings":
    QUERIES = QUERIES_POSTINGS
for name, (sql, bpr) in QUERIES.items():
    if args.only and (args.only[1:] != name if args.only.startswith("=") else args.only not in name):
        continue
    qc = parse_sql(sql)
    qc.flags |= capi.QUERY_FLAG_PROFILE
    cq = CQuery(qc)
    ms = []
    for i in range(args.reps + 2):
        h = C.c_void_p()
        api.call("query_exec", seg.handle, cq.ptr(), C.byref(h))
        st = capi.PgExecStats()
        api.call("result_stats", h, C.byref(st))
        api.call("result_free", h)
        if i >= 2:
            ms.append(st.device_ms_aggregate)
    m = statistics.median(ms)
    if m <= 0:
        print(f"{name:32s} no kernel ran (answered on the host)")
        continue
    if bpr <= 0:   # index-driven: bytes per row of the segment say little; report the time per doc of the segment and per match
        print(f"{name:46s} {st.kernel.decode():24s} {m * 1e3:8.1f} us  {m * 1e6 / max(st.num_docs_scanned, 1):7.3f} ns per matching doc  matched={st.num_docs_scanned} ({100.0 * st.num_docs_scanned / args.docs:.2f} %)")
        continue
    print(f"{name:32s} {st.kernel.decode():24s} {m:8.3f} ms  {bpr * args.docs / m / 1e6:8.1f} GB/s  ({bpr * args.docs / m / 1e6 / 80:5.1f}% of 8 TB/s)  matched={st.num_docs_scanned}")
