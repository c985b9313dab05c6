#!/bin/bash
# round 4, session f: byte registers in the pruned passes' aggregation (25 buckets for config 5), per-dispatch counters of the stream scatter
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r04_f}
timeout 900 python -m pytest tests/test_gpu_oct.py tests/test_gpu_bench_contract.py -q -p no:cacheprovider -x > $O/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -8 $O/${TAG}_tests.log | cut -c1-300
timeout 200 python tools/prof_variants.py --set cfg5 --only "=cfg5" --docs 200000000 --reps 8 2>&1 | grep -v "^/opt"
PG_OCT_DWORD_REGS=1 timeout 200 python tools/prof_variants.py --set cfg5 --only "=cfg5" --docs 200000000 --reps 8 2>&1 | grep -v "^/opt"
cd /tmp; export TMPDIR=/tmp
for q in "=cfg5"; do
  n=$(echo "$q" | tr -c 'a-z0-9' '_')
  rm -rf /tmp/prof_$n; timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_$n -o x -- python $R/tools/prof_variants.py --set cfg5 --only "$q" --docs 200000000 --reps 6 > $O/${TAG}_prof_$n.log 2>&1
  db=$(find /tmp/prof_$n -name "*_results.db" | head -1); python $R/tools/rocprof_summary.py $db > $O/${TAG}_kernels_$n.txt 2>&1; echo "== $q"; head -16 $O/${TAG}_kernels_$n.txt
  python - "$db" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
tail = rows[-45:]
t0 = tail[0][1]
for n, s, e in tail:
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f} us  {n[:40]}")
PY
done
cd $R
timeout 500 python tools/pmc_sq.py cfg5 "=cfg5" 200000000 pg_p2_scatter_stream > $O/${TAG}_sq_cfg5.txt 2>&1; grep -A12 "per dispatch" $O/${TAG}_sq_cfg5.txt
timeout 300 python bench.py --query cfg5 --no-variants --no-traffic --no-cpu-baseline --steps 10 --warmup 3 > $O/${TAG}_bench_cfg5_1b.json 2> $O/${TAG}_bench_cfg5_1b.err; python -c "
import json,sys
d=json.loads(open('$O/${TAG}_bench_cfg5_1b.json').read().strip().splitlines()[-1]); print('cfg5 1B: ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])"
