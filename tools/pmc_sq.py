"""Dev tool: SQ counters per kernel for one prof_variants.py row (two rocprofv3 --pmc passes, 8 SQ slots each)."""
import os, sqlite3, subprocess, sys, tempfile, shutil
PASSES = [
    ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"],
    ["SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_WAVES", "SQ_BUSY_CYCLES"],
]
if os.environ.get("PMC_MEMORY"):   # + the memory side: FETCH_SIZE / WRITE_SIZE (KB as rocprofv3 reports them; x 2 on gfx950 per the guide), L2 hits / misses
    PASSES += [["FETCH_SIZE"], ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"], ["SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"]]
setname, only, docs = sys.argv[1], sys.argv[2], sys.argv[3]
per_dispatch = sys.argv[4] if len(sys.argv) > 4 else ""   # kernel name: its counters dispatch by dispatch (last 8 dispatches)
per = {}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for counters in PASSES:
    d = tempfile.mkdtemp(prefix="pg_sq_", dir="/tmp")
    cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "-d", d, "-o", "x", "--", sys.executable, os.path.join(root, "tools/prof_variants.py"),
           "--set", setname, "--only", only, "--docs", docs, "--reps", "3"]
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=400)
    dbs = [os.path.join(rr, f) for rr, _, fs in os.walk(d) for f in fs if f.endswith("_results.db")]
    if not dbs:
        print(r.stdout.decode()[-2000:]); continue
    db = sqlite3.connect(dbs[0])
    for k, c, v, n in db.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection where kernel_name like 'pg_%' group by kernel_name, counter_name"):
        out.setdefault(k, {})[c] = v / max(n, 1)
    if per_dispatch:
        for did, c, v in db.execute("select dispatch_id, counter_name, sum(value) from counters_collection where kernel_name = ? group by dispatch_id, counter_name order by dispatch_id", (per_dispatch,)):
            per.setdefault(c, []).append(v)
    shutil.rmtree(d, ignore_errors=True)
for k, cs in sorted(out.items()):
    print(k)
    for c, v in cs.items():
        print(f"    {c:24s} {v:16.0f}")
if per_dispatch:
    print("per dispatch (last 8):", per_dispatch)
    for c, vs in per.items():
        print(f"    {c:24s} " + " ".join(f"{v:13.0f}" for v in vs[-8:]))
