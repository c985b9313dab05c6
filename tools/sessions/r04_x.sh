#!/bin/bash
# round 4, session x: result assembly sizes only the components a result kind has: star-tree route latency; parity of everything that reads results
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r04_x}
timeout 900 python -m pytest tests/test_gpu_startree.py tests/test_gpu_final_distinct.py tests/test_gpu_parity.py tests/test_gpu_multi.py -q -p no:cacheprovider -x --timeout 600 -n 4 > $O/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/${TAG}_tests.log | cut -c1-300
PG_TRACE_HOST=1 timeout 300 python bench.py --query cfg5 --no-traffic --no-cpu-baseline --no-variants --docs 20000000 --steps 20 --warmup 5 2> $O/${TAG}_trace.err | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d.get('cfg5_star_tree', d); print({k:c.get(k) for k in ('p50_query_latency_ms','library_ms','device_ms')}); print(c.get('intermediate_registers'))"
grep "pg_generic_query_l" $O/${TAG}_trace.err | tail -6
