"""Times a set of queries on one synthetic segment with the library's HIP-event timers (dev tool, not a test)."""
import argparse
import ctypes as C
import statistics
import time
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
ap.add_argument("--set", choices=["cfg3", "cfg5", "general", "wide", "upsert", "postings", "mv", "strings", "dict"], default="cfg3")
args = ap.parse_args()
api = capi.gpu_api()
api.call("init", 0)
seg = NativeSegment(api, HostSegment("prof", args.docs))
DICT_COLUMNS = synth.CFG3_COLUMNS + ["r_int_d", "m_d", "r_int_s", "m_s"]   # --set dict: config 3's columns + their dictionary-encoded twins
for name in (synth.CFG5_COLUMNS if args.set == "cfg5" else (DICT_COLUMNS if args.set == "dict" else synth.CFG3_COLUMNS)):
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
    # a metric behind a 20 000-value dictionary that is NOT arithmetic (15-bit dictIds): its values are gathered — from LDS where the dictionary fits
    sp = np.sort(np.random.default_rng(3).choice(4_000_000, 20000, replace=False)).astype(np.int32)
    seg.add_column(build_column("ms20k", sp[mvals % 20000], "INT"), keep_host_buffers=False)
    del mvals, w, sp

MV_BYTES = {}
if args.set == "mv":
    # multi-value dictionary columns (FixedBitMVForwardIndexReader layout) next to the synthetic table's columns, built with numpy:
    # mv1 — 1000 values, 1-5 entries per doc; mv2 — 20 values, 1-3 entries per doc; keep --docs around 5e7 (17 s of packing per column)
    import numpy as np
    from pinot_amd import formats
    from pinot_amd.segment import HostColumn
    n = args.docs
    rng = np.random.default_rng(2)
    for cname, card, hi in (("mv1", 1000, 6), ("mv2", 20, 4)):
        lengths = rng.integers(1, hi, n).astype(np.int64)
        total = int(lengths.sum())
        ids = rng.integers(0, card, total).astype(np.int32)
        bits = formats.num_bits_per_value(card - 1)
        col = HostColumn(cname, "INT", capi.FWD_DICT_FIXED_BIT_MV, True, card, bits, False, 4, formats.write_fixed_bit_mv(ids, lengths, bits),
                         formats.write_numeric_dictionary(np.arange(card, dtype=np.int32), "INT"), None, None)
        col.total_number_of_entries = total
        seg.add_column(col, keep_host_buffers=False)
        MV_BYTES[cname] = (total * bits / 8 + total / 8) / n   # entries + the row-start bitmap, per doc
        del lengths, ids, col

if args.set == "strings":
    # a raw (no-dictionary) STRING column, var-byte chunks V4: 5000 distinct 10-byte values; Python builds the values: keep --docs around 2e7
    import numpy as np
    from pinot_amd.segment import build_column
    n = args.docs
    rng = np.random.default_rng(3)
    vals = [f"city_{i:05d}" for i in rng.integers(0, 5000, n)]
    seg.add_column(build_column("s", vals, "STRING", dictionary=False), keep_host_buffers=False)
    del vals

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
    # (routed to the wide pipeline's register-held accumulators, PG NG = 0, these measured 78.5 / 68.9 / 80.7 / 77.2 % over 10^9 docs against 82.3 / 57.2 /
    # 83.0 / 81.0 % on the narrow kernels: not routed)
    "no group: sum min max count(m)": ("SELECT SUM(m), MIN(m), MAX(m), COUNT(*) FROM t", 4.0),
    "no group: range scan, sum(m)": ("SELECT SUM(m), COUNT(*) FROM t WHERE r_int BETWEEN 250000 AND 749999", 8.0),
    "no group: cfg3 filter, sum(m)": ("SELECT SUM(m), MAX(m) FROM t WHERE c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1) AND r_int BETWEEN 250000 AND 749999", 8.75),
    "range scan, sum(m) group g1": ("SELECT g1, SUM(m), MAX(m) FROM t WHERE r_int BETWEEN 250000 AND 749999 GROUP BY g1", 8.875),
    "index only, sum(m) group g1": ("SELECT g1, SUM(m), MAX(m) FROM t WHERE c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1) GROUP BY g1", 5.625),
    "cfg3": (synth.QUERY_CFG3, 9.625),
    "northstar": (synth.QUERY_NORTH_STAR, 10.375),
    # the headline shape at other index selectivities (candidates of the range scan as a fraction of the docs): pg_fast_i32range_p skips the
    # quads without candidates, pg_fast_i32range_s streams everything
    "sel 3%": ("SELECT g1, SUM(m), MAX(m) FROM t WHERE c_inv1 = 0 AND c_inv2 = 0 AND r_int BETWEEN 250000 AND 749999 GROUP BY g1", 9.625),
    "sel 6%": ("SELECT g1, SUM(m), MAX(m) FROM t WHERE c_inv1 IN (0,1) AND c_inv2 = 0 AND r_int BETWEEN 250000 AND 749999 GROUP BY g1", 9.625),
    "sel 12%": ("SELECT g1, SUM(m), MAX(m) FROM t WHERE c_inv1 IN (0,1,2,3) AND c_inv2 = 0 AND r_int BETWEEN 250000 AND 749999 GROUP BY g1", 9.625),
    "sel 50%": ("SELECT g1, SUM(m), MAX(m) FROM t WHERE c_inv1 IN (0,1,2,3) AND c_inv2 NOT IN (3) AND r_int BETWEEN 250000 AND 749999 GROUP BY g1", 9.625),
    "sel 75% all match": ("SELECT g1, SUM(m), MAX(m) FROM t WHERE c_inv1 NOT IN (7) AND c_inv2 NOT IN (3) AND r_int BETWEEN 0 AND 2000000 GROUP BY g1", 9.625),
    "g1 scan eq": ("SELECT COUNT(*) FROM t WHERE g1 = 7", 0.875),
}
QUERIES5 = {
    "cfg5": (synth.QUERY_CFG5, 4.375),
    "cfg5 count only": ("SELECT h1, h2, h3, h4, COUNT(*) FROM t GROUP BY h1, h2, h3, h4 LIMIT 20000", 1.875),
    "count only group h1 (16)": ("SELECT h1, COUNT(*) FROM t GROUP BY h1", 0.5),
    "count only group h1,h2 (160)": ("SELECT h1, h2, COUNT(*) FROM t GROUP BY h1, h2", 1.0),
    "count only group h1,h2 where h3<3": ("SELECT h1, h2, COUNT(*) FROM t WHERE h3 < 3 GROUP BY h1, h2", 1.5),
    "cfg5 hll(u) no group": ("SELECT DISTINCTCOUNTHLL(u) FROM t", 2.5),
    "cfg5 hll(u) group h1": ("SELECT h1, DISTINCTCOUNTHLL(u) FROM t GROUP BY h1", 3.0),
    "cfg5 distinctcount(u) group h1": ("SELECT h1, DISTINCTCOUNT(u) FROM t GROUP BY h1", 3.0),
    "hll(u) where h2<5": ("SELECT DISTINCTCOUNTHLL(u) FROM t WHERE h2 < 5", 3.0),
    "distinctcount(u) where h2<5": ("SELECT DISTINCTCOUNT(u) FROM t WHERE h2 < 5", 3.0),
    "hll(u) group h1,h2 (160)": ("SELECT h1, h2, DISTINCTCOUNTHLL(u), COUNT(*) FROM t GROUP BY h1, h2", 3.5),
    "count group u (1M groups)": ("SELECT u, COUNT(*) FROM t GROUP BY u LIMIT 2000000", 2.5),
    "count group u (1M groups), numGroupsLimit 2M": ("SELECT u, COUNT(*) FROM t GROUP BY u LIMIT 2000000 /*limit=2000000*/", 2.5),
    "count group u,h1 (16M groups)": ("SELECT u, h1, COUNT(*) FROM t WHERE h2 = 3 GROUP BY u, h1 LIMIT 20000000", 3.5),
    "hash: group u,h1,h2 where u<20000": ("SELECT u, h1, h2, COUNT(*), SUM(h3) FROM t WHERE u < 20000 GROUP BY u, h1, h2 LIMIT 10000000", 4.0),
    "hash: group u,h1,h2 where h3=1,h4=2": ("SELECT u, h1, h2, COUNT(*) FROM t WHERE h3 = 1 AND h4 = 2 GROUP BY u, h1, h2 LIMIT 10000000", 4.0),
    "probe radix sum(h3)": ("SELECT h1, h2, h3, h4, COUNT(*), SUM(h3) FROM t GROUP BY h1, h2, h3, h4 LIMIT 20000", 1.875),
    "probe radix hll(h3)": ("SELECT h1, h2, h3, h4, COUNT(*), DISTINCTCOUNTHLL(h3) FROM t GROUP BY h1, h2, h3, h4 LIMIT 20000", 1.875),
    "cfg5 final values": (synth.QUERY_CFG5 + " /*final*/", 4.375),
    "hll(u) group h1,h2 final": ("SELECT h1, h2, DISTINCTCOUNTHLL(u), COUNT(*) FROM t GROUP BY h1, h2 /*final*/", 3.5),
    "cfg5 count group h1..h3": ("SELECT h1, h2, h3, COUNT(*) FROM t GROUP BY h1, h2, h3 LIMIT 20000", 1.5),
}
QUERIES_GENERAL = {   # shapes outside the specialised kernels: several scans, OR of scans, tables beyond LDS
    "2 scans + group g1": ("SELECT g1, SUM(m) FROM t WHERE r_int BETWEEN 250000 AND 749999 AND m < 524288 GROUP BY g1", 8.875),
    "postings + 2 scans + group": ("SELECT g1, SUM(m), MAX(m) FROM t WHERE c_inv1 IN (0,1,2,3) AND r_int BETWEEN 250000 AND 749999 AND m < 524288 GROUP BY g1", 9.375),
    "or of 2 scans count": ("SELECT COUNT(*) FROM t WHERE r_int < 100000 OR m > 900000", 8.0),
    "dict scan + raw scan": ("SELECT COUNT(*) FROM t WHERE g1 < 50 AND r_int < 500000", 4.875),
    "group g1,g2,c_inv1 (40k groups)": ("SELECT g1, g2, c_inv1, COUNT(*), SUM(m) FROM t GROUP BY g1, g2, c_inv1 LIMIT 100000", 6.0),
    "40k groups, sum of a dictionary column": ("SELECT g1, g2, c_inv1, COUNT(*), SUM(c_inv2), MAX(c_inv2) FROM t GROUP BY g1, g2, c_inv1 LIMIT 100000", 2.25),
    "filtered 40k groups": ("SELECT g1, g2, c_inv1, SUM(m) FROM t WHERE r_int < 125000 GROUP BY g1, g2, c_inv1 LIMIT 100000", 10.0),
    "group g1,g2,c_inv1,c_inv2 (160k)": ("SELECT g1, g2, c_inv1, c_inv2, COUNT(*), SUM(m), MAX(m) FROM t GROUP BY g1, g2, c_inv1, c_inv2 LIMIT 200000", 6.25),
    "160k groups, numGroupsLimit 200k": ("SELECT g1, g2, c_inv1, c_inv2, COUNT(*), SUM(m), MAX(m) FROM t GROUP BY g1, g2, c_inv1, c_inv2 LIMIT 200000 /*limit=200000*/", 6.25),
    "cfg3 filter, 160k groups": ("SELECT g1, g2, c_inv1, c_inv2, SUM(m) FROM t WHERE c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1) AND r_int BETWEEN 250000 AND 749999 GROUP BY g1, g2, c_inv1, c_inv2 LIMIT 200000", 11.0),
    "avg/min/max/count no group": ("SELECT COUNT(*), AVG(m), MIN(r_int), MAX(m) FROM t WHERE c_inv2 = 1", 8.125),
}
QUERIES_WIDE = {   # LDS-table aggregations over 64-bit sources / an 11-bit group column
    "sum(m64) group g1": ("SELECT g1, SUM(m64), MAX(m64) FROM t GROUP BY g1", 8.875),
    "cfg3 filter, sum(m64) group g1": ("SELECT g1, SUM(m64) FROM t WHERE c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1) AND r_int BETWEEN 250000 AND 749999 GROUP BY g1", 13.625),
    "sum(d64) avg(m) group g1": ("SELECT g1, SUM(d64), AVG(m) FROM t WHERE r_int < 500000 GROUP BY g1", 16.875),
    "sum(d64) max(d64) group g1": ("SELECT g1, SUM(d64), MAX(d64) FROM t GROUP BY g1", 8.875),
    "filtered sum(d64) group g1": ("SELECT g1, SUM(d64), COUNT(*) FROM t WHERE r_int BETWEEN 250000 AND 749999 GROUP BY g1", 12.875),
    "sum(m) group w1 (2000 groups)": ("SELECT w1, SUM(m), COUNT(*) FROM t GROUP BY w1 LIMIT 5000", 5.375),
    "filtered sum(m64) group w1": ("SELECT w1, SUM(m64) FROM t WHERE r_int BETWEEN 250000 AND 749999 GROUP BY w1 LIMIT 5000", 13.375),
    "sum(m64) no group": ("SELECT SUM(m64), MIN(m64), COUNT(*) FROM t WHERE c_inv2 = 1", 8.125),
    "no group: sum min max count(ms20k)": ("SELECT SUM(ms20k), MIN(ms20k), MAX(ms20k), COUNT(*) FROM t", 1.875),
}
QUERIES_DICT = {   # config 3 in Pinot's default encoding: 20-bit dictId streams for the scan column and the value column (2.5 B/row each)
    "cfg3 raw (reference point)": (synth.QUERY_CFG3, 9.625),
    "cfg3 dict": (synth.QUERY_CFG3_DICT, 6.625),
    "northstar dict": (synth.QUERY_NORTH_STAR_DICT, 7.375),
    "cfg3 sparse dictionaries (gather)": (synth.QUERY_CFG3_SPARSE, 6.625),
    "northstar sparse, count sum": ("SELECT g1, g2, COUNT(*), SUM(m_s) FROM t WHERE c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1) AND r_int_s BETWEEN 750000 AND 2249999 GROUP BY g1, g2 LIMIT 10000", 7.375),
    "raw scan, dict value": ("SELECT g1, SUM(m_d), MAX(m_d) FROM t WHERE c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1) AND r_int BETWEEN 250000 AND 749999 GROUP BY g1", 8.125),
    "dict scan, raw value": ("SELECT g1, SUM(m), MAX(m) FROM t WHERE c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1) AND r_int_d BETWEEN 250000 AND 749999 GROUP BY g1", 8.125),
    "no filter sum(m_d) group g1": ("SELECT g1, SUM(m_d), MAX(m_d) FROM t GROUP BY g1", 3.375),
    "no filter sum(m_s) group g1": ("SELECT g1, SUM(m_s), MAX(m_s) FROM t GROUP BY g1", 3.375),
    "index only, sum(m_d) group g1": ("SELECT g1, SUM(m_d), MAX(m_d) FROM t WHERE c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1) GROUP BY g1", 4.125),
    "dict range scan, sum(m_d) group g1": ("SELECT g1, SUM(m_d), MAX(m_d) FROM t WHERE r_int_d BETWEEN 250000 AND 749999 GROUP BY g1", 5.875),
    "dict sel 3%": ("SELECT g1, SUM(m_d), MAX(m_d) FROM t WHERE c_inv1 = 0 AND c_inv2 = 0 AND r_int_d BETWEEN 250000 AND 749999 GROUP BY g1", 6.625),
    "dict sel 50%": ("SELECT g1, SUM(m_d), MAX(m_d) FROM t WHERE c_inv1 IN (0,1,2,3) AND c_inv2 NOT IN (3) AND r_int_d BETWEEN 250000 AND 749999 GROUP BY g1", 6.625),
    "dict sel 75% all match": ("SELECT g1, SUM(m_d), MAX(m_d) FROM t WHERE c_inv1 NOT IN (7) AND c_inv2 NOT IN (3) AND r_int_d BETWEEN 0 AND 2000000 GROUP BY g1", 6.625),
    "sparse sel 75% all match": ("SELECT g1, SUM(m_s), MAX(m_s) FROM t WHERE c_inv1 NOT IN (7) AND c_inv2 NOT IN (3) AND r_int_s BETWEEN 0 AND 9000000 GROUP BY g1", 6.625),
    "dict filter only count": ("SELECT COUNT(*) FROM t WHERE c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1) AND r_int_d BETWEEN 250000 AND 749999", 3.25),
    "no group: dict filter, sum(m_d)": ("SELECT SUM(m_d), MAX(m_d) FROM t WHERE c_inv1 IN (0,1,2,3) AND c_inv2 IN (0,1) AND r_int_d BETWEEN 250000 AND 749999", 5.75),
    "cfg2 over r_int_d: count(dict range scan)": ("SELECT COUNT(*) FROM t WHERE r_int_d BETWEEN 250000 AND 749999", 2.5),
    "no group: sum min max count(m_d)": ("SELECT SUM(m_d), MIN(m_d), MAX(m_d), COUNT(*) FROM t", 2.5),
    "no group: sum min max count(m_s)": ("SELECT SUM(m_s), MIN(m_s), MAX(m_s), COUNT(*) FROM t", 2.5),
}
if args.set == "dict":
    QUERIES = QUERIES_DICT
if args.set == "mv":
    b1, b2 = MV_BYTES["mv1"], MV_BYTES["mv2"]
    QUERIES = {   # bytes per row: every entry of the multi-value columns read + the single-value columns
        "mv scan: mv1 = 7 count": ("SELECT COUNT(*) FROM t WHERE mv1 = 7", b1),
        "mv scan: mv1 between + sum group g1": ("SELECT g1, SUM(m), COUNT(*) FROM t WHERE mv1 BETWEEN 100 AND 199 GROUP BY g1", b1 + 4.875),
        "mv scan NOT IN + sv scan": ("SELECT COUNT(*), SUM(m) FROM t WHERE mv2 NOT IN (1, 2, 3) AND r_int < 500000", b2 + 8.0),
        "group by mv2": ("SELECT mv2, COUNT(*), SUM(m) FROM t GROUP BY mv2 LIMIT 100", b2 + 4.0),
        "group by mv1 (1000)": ("SELECT mv1, COUNT(*), MAX(m) FROM t GROUP BY mv1 LIMIT 2000", b1 + 4.0),
        "group by mv2, g1": ("SELECT mv2, g1, COUNT(*), SUM(m) FROM t GROUP BY mv2, g1 LIMIT 10000", b2 + 4.875),
        "group by mv1, mv2 (cartesian)": ("SELECT mv1, mv2, COUNT(*) FROM t WHERE r_int < 250000 GROUP BY mv1, mv2 LIMIT 100000", b1 + b2 + 4.0),
        "summv/countmv/maxmv group g1": ("SELECT g1, SUMMV(mv1), COUNTMV(mv1), MAXMV(mv1) FROM t GROUP BY g1", b1 + 0.875),
        "avgmv/minmaxrangemv no group": ("SELECT AVGMV(mv1), MINMAXRANGEMV(mv1), COUNT(*) FROM t WHERE c_inv2 = 1", b1 + 0.125),
        "distinctcountmv(mv1) group g1": ("SELECT g1, DISTINCTCOUNTMV(mv1) FROM t GROUP BY g1", b1 + 0.875),
        "distinctcounthllmv(mv1)": ("SELECT DISTINCTCOUNTHLLMV(mv1) FROM t WHERE r_int < 500000", b1 + 4.0),
    }
if args.set == "strings":
    QUERIES = {   # 14 bytes per row: a 10-byte value and its 4-byte offset
        "group by raw string": ("SELECT s, COUNT(*), SUM(m) FROM t GROUP BY s LIMIT 10000", 18.0),
        "group by raw string, g1": ("SELECT s, g1, COUNT(*) FROM t WHERE r_int < 500000 GROUP BY s, g1 LIMIT 1000000", 18.875),
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
    import re as _re
    lim = _re.search(r" /\*limit=(\d+)\*/", sql)   # numGroupsLimit of the row (default 100 000: rows beyond it carry the docId plane of the trimming)
    qc = parse_sql(_re.sub(r" /\*limit=\d+\*/", "", sql).replace(" /*final*/", ""))
    if lim:
        qc.num_groups_limit = int(lim.group(1))
    qc.flags |= capi.QUERY_FLAG_PROFILE
    if "/*final*/" in sql:
        qc.flags |= capi.QUERY_FLAG_FINAL_DISTINCT
    cq = CQuery(qc)
    ms, wall = [], []
    for i in range(args.reps + 2):
        h = C.c_void_p()
        t0 = time.perf_counter()
        api.call("query_exec", seg.handle, cq.ptr(), C.byref(h))
        wall.append((time.perf_counter() - t0) * 1e3)
        st = capi.PgExecStats()
        api.call("result_stats", h, C.byref(st))
        api.call("result_free", h)
        if i >= 2:
            ms.append(st.device_ms_aggregate)
    m = statistics.median(ms)
    if args.set == "strings" or "/*final*/" in sql:   # what the first use costs (virtual dictionary build) and the call as the host sees it
        print(f"{name:32s} wall: first call {wall[0]:9.3f} ms, second {wall[1]:8.3f} ms, median of the rest {statistics.median(wall[2:]):8.3f} ms")
    if m <= 0:
        print(f"{name:32s} no kernel ran (answered on the host)")
        continue
    if bpr <= 0:   # index-driven: bytes per row of the segment say little; report the time per doc of the segment and per match
        print(f"{name:46s} {st.kernel.decode():24s} {m * 1e3:8.1f} us  {m * 1e6 / max(st.num_docs_scanned, 1):7.3f} ns per matching doc  matched={st.num_docs_scanned} ({100.0 * st.num_docs_scanned / args.docs:.2f} %)")
        continue
    print(f"{name:32s} {st.kernel.decode():24s} {m:8.3f} ms  {bpr * args.docs / m / 1e6:8.1f} GB/s  ({bpr * args.docs / m / 1e6 / 80:5.1f}% of 8 TB/s)  matched={st.num_docs_scanned}")
