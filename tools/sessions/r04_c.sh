#!/bin/bash
# round 4, session c: oct kernels after the store / CAS rework
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r04_c}
timeout 600 python -m pytest tests/test_gpu_oct.py tests/test_gpu_bench_contract.py -q -p no:cacheprovider -x > $O/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -25 $O/${TAG}_tests.log | cut -c1-300
timeout 200 python tools/prof_variants.py --set cfg5 --docs 200000000 --reps 8 > $O/${TAG}_cfg5_oct.txt 2>&1; grep -v "^/opt" $O/${TAG}_cfg5_oct.txt
cd /tmp; export TMPDIR=/tmp
for q in "=cfg5"; do
  n=$(echo "$q" | tr -c 'a-z0-9' '_')
  rm -rf /tmp/prof_$n; timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_$n -o x -- python $R/tools/prof_variants.py --set cfg5 --only "$q" --docs 200000000 --reps 6 > $O/${TAG}_prof_$n.log 2>&1
  db=$(find /tmp/prof_$n -name "*_results.db" | head -1); python $R/tools/rocprof_summary.py $db > $O/${TAG}_kernels_$n.txt 2>&1; echo "== $q"; head -30 $O/${TAG}_kernels_$n.txt
  python - "$db" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
# the last query's passes: kernel by kernel, with the gaps between them
tail = rows[-45:]
t0 = tail[0][1]
for n, s, e in tail:
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f} us  {n[:40]}")
PY
done
cd $R
timeout 300 python bench.py --query cfg5 --no-variants --no-traffic --no-cpu-baseline --steps 10 --warmup 3 > $O/${TAG}_bench_cfg5_1b.json 2> $O/${TAG}_bench_cfg5_1b.err; tail -c 1500 $O/${TAG}_bench_cfg5_1b.json
