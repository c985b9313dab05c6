#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
nproc > $OUT/r05_j_nproc.txt; timeout 600 python bench.py --docs 200000000 --no-cpu-baseline --no-traffic --steps 5 --warmup 2 > $OUT/r05_j_bench.json 2> $OUT/r05_j_bench.log
tail -5 $OUT/r05_j_bench.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_j_bench.json').read().strip().splitlines()[-1])
c=d.get("concurrency",{})
for name,w in c.get("workloads",{}).items():
    print(name, "16/1 =", w.get("qps_16_over_1"))
    for p in w["points"]: print("   ", p)
    print("    bg:", w.get("with_1e9_row_scan_in_background"))
PY
