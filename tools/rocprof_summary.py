"""Summarises a rocprofv3 results database (rocpd sqlite, the default output format of rocprofv3 in ROCm 7.2) into the
per-kernel table `--stats` would print: calls, total/avg/min/max duration, % of GPU kernel time, plus VGPR/LDS of the
dispatches.  Usage: python tools/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/rNN_xxx_kernel_stats.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
# One row per (kernel, launch shape): the same kernel launched for two queries — pg_fast_i32range_p runs BASELINE config 3 (100 groups) and the
# north-star variant (5 000 groups) in one bench run — differs in its dynamic LDS size / grid, and averaging the two under one name hid the
# headline kernel's own duration.  `--by-name` gives the old one-row-per-name table.
by_name = "--by-name" in sys.argv
group = "name" if by_name else "name, lds_size, grid_x, workgroup_x"
rows = list(cur.execute(
    "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), max(vgpr_count), "
    f"max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) from kernels group by {group} order by 3 desc"))
total = sum(r[2] for r in rows) or 1
print(f"{'kernel':44s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s} "
      f"{'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>7s} {'grid':>8s} {'wg':>5s}")
for r in rows:
    print(f"{r[0][:44]:44s} {r[1]:6d} {r[2] / 1e6:10.3f} {r[3] / 1e3:10.2f} {r[4] / 1e3:10.2f} {r[5] / 1e3:10.2f} "
          f"{100.0 * r[2] / total:6.2f} {r[6]:5d} {r[7]:5d} {r[8]:5d} {r[9]:7d} {r[10]:8d} {r[11]:5d}")
try:
    pm = list(cur.execute("select kernel_name, counter_name, count(*), avg(value) from "
                          "(select dispatch_id, kernel_name, counter_name, sum(value) as value from counters_collection "
                          " group by dispatch_id, kernel_name, counter_name) group by kernel_name, counter_name order by 1, 2"))
    if pm:
        print("\ncounters (per-dispatch average, summed over instances):")
        for r in pm:
            if r[0].startswith("__amd"):
                continue
            print(f"  {r[0][:36]:36s} {r[1]:28s} n={r[2]:5d} avg={r[3]:.6g}")
except Exception as e:  # noqa
    print("no counters:", e)
