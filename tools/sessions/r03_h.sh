#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r03_h}
timeout 600 python -m pytest tests/test_gpu_final_distinct.py tests/test_gpu_multi.py tests/test_gpu_partition_pipeline.py -q -p no:cacheprovider > $O/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -8 $O/${TAG}_tests.log | cut -c1-220
timeout 300 python bench.py --query cfg5 --docs 200000000 --no-traffic --no-cpu-baseline > $O/${TAG}_bench_cfg5.json 2> $O/${TAG}_bench_cfg5.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/${TAG}_bench_cfg5.json").read().strip().splitlines()[-1])
print(d["roofline"]["kernel_ms"], d["roofline"]["frac"]); print(json.dumps(d["star_tree_route"], indent=0))
PY
