#!/bin/bash
# round 3, GPU call F: whole GPU suite + default bench (with the cfg2 / cfg5 blocks)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r03_f}
timeout 900 python -m pytest tests -x -q -m gpu > $O/${TAG}_tests.log 2>&1; echo "gpu suite rc=$?"; tail -6 $O/${TAG}_tests.log
( time timeout 600 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err ) 2> $O/${TAG}_bench.time; echo "bench rc=$?"; cat $O/${TAG}_bench.time | grep real
tail -3 $O/${TAG}_bench.err
python - <<PY
import json
d=json.loads(open("$O/${TAG}_bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["roofline"]["frac"], d["roofline"]["kernel_ms"])
for k in ("north_star_variant","cfg2","cfg5_flat","cfg5_star_tree"):
    b=d.get(k,{}); print(k, {x:b.get(x) for x in ("kernel","kernel_ms","roofline_frac","traffic_over_algorithmic","gpu_equals_oracle_at_full_size","gpu_equals_oracle_on_sample","full_size_invariant","p50_query_latency_ms","device_ms","wall_s")})
PY
