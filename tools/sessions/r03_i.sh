#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r03_i}
PG_TRACE_HOST=1 timeout 300 python bench.py --query cfg5 --docs 20000000 --steps 8 --warmup 3 --no-traffic --no-cpu-baseline > $O/${TAG}_bench_cfg5.json 2> $O/${TAG}_bench_cfg5.err; echo "bench rc=$?"
grep "pg_generic_query_l" $O/${TAG}_bench_cfg5.err | tail -24
python - <<PY
import json
d=json.loads(open("$O/${TAG}_bench_cfg5.json").read().strip().splitlines()[-1])
print(json.dumps(d["star_tree_route"], indent=0))
PY
