"""Cost of the exact numEntriesScannedInFilter of leapfrogged filter shapes (OR / NOT over scans inside an AND): wall time of
pg_query_exec with the exact count (PG_QUERY_FLAG_EXACT_FILTER_STATS), with the approximate one (PG_QUERY_FLAG_APPROX_FILTER_STATS) and
by default, at two segment sizes (dev tool; its output is kept under profiles/)."""
import argparse
import statistics
import sys
import os
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from pinot_amd import capi, synth
from pinot_amd.executor import NativeSegment
from pinot_amd.query import parse_sql
from pinot_amd.segment import HostSegment

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, nargs="+", default=[100_000_000, 1_000_000_000])
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--no-host-walk", action="store_true", help="skip the runs that force the host walk (seconds per query at 10^9 docs)")
args = ap.parse_args()
api = capi.gpu_api()
api.call("init", 0)
QUERIES = {
    "or of 2 scans (drained OR: closed form)": "SELECT COUNT(*) FROM t WHERE r_int < 100000 OR m > 900000",
    "postings AND (scan OR scan)": "SELECT g1, SUM(m) FROM t WHERE c_inv1 IN (0,1,2,3) AND (r_int < 100000 OR m > 900000) GROUP BY g1",
    "postings AND NOT scan": "SELECT g1, SUM(m) FROM t WHERE c_inv2 = 1 AND NOT r_int < 900000 GROUP BY g1",
    "scan AND (postings OR scan)": "SELECT COUNT(*) FROM t WHERE m < 524288 AND (c_inv1 = 3 OR r_int > 990000)",
    "scan AND scan (leapfrog)": "SELECT g1, SUM(m) FROM t WHERE r_int < 500000 AND m > 400000 GROUP BY g1",
}
for docs in args.docs:
    seg = NativeSegment(api, HostSegment("fs", docs))
    for name in synth.CFG3_COLUMNS:
        one = synth.generate_segment(docs, columns=[name])
        seg.add_column(one.columns[name], keep_host_buffers=False)
    print(f"# {docs} docs")
    for name, sql in QUERIES.items():
        if args.no_host_walk and "NOT" in sql:
            continue
        row = []
        for label, flag in (("default", 0), ("approx", capi.QUERY_FLAG_APPROX_FILTER_STATS), ("exact", capi.QUERY_FLAG_EXACT_FILTER_STATS),
                            ("exact, host walk", capi.QUERY_FLAG_EXACT_FILTER_STATS)):
            q = parse_sql(sql)
            q.flags |= flag
            ts = []
            host = label.endswith("host walk")
            if host and (args.no_host_walk or "NOT" in sql and args.no_host_walk):
                continue   # PG_FILTER_STATS_HOST: the automaton walked on the host also where the device counts it
            if host:
                os.environ["PG_FILTER_STATS_HOST"] = "1"
                api.call("options_reload")
            for i in range((1 if host and docs > 200_000_000 else args.reps) + 1):
                t = time.perf_counter()
                b = seg.execute(q)
                if i:
                    ts.append((time.perf_counter() - t) * 1e3)
            if host:
                os.environ.pop("PG_FILTER_STATS_HOST")
                api.call("options_reload")
            row.append(f"{label} {statistics.median(ts):9.3f} ms (stats_exact={b.stats.stats_exact}, path={b.stats.filter_stats_path}, entries={b.stats.num_entries_scanned_in_filter})")
        print(f"{name:42s} " + " | ".join(row))
    seg.destroy()
