#!/bin/bash
# round 4, session j: small-query latency A/B (result written straight into the pinned block, polling wait); host timeline of the raw STRING group-by
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r04_j}
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_oct.py tests/test_gpu_final_distinct.py -q -p no:cacheprovider -x --timeout 300 -n 4 > $O/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/${TAG}_tests.log | cut -c1-300
{
echo "== default"; timeout 200 python tools/latency.py 2>&1 | grep -v "^/opt"
echo "== PG_NO_DIRECT_RESULT=1"; PG_NO_DIRECT_RESULT=1 timeout 200 python tools/latency.py 2>&1 | grep -v "^/opt"
echo "== PG_NO_SPIN_WAIT=1"; PG_NO_SPIN_WAIT=1 timeout 200 python tools/latency.py 2>&1 | grep -v "^/opt"
echo "== PG_NO_DIRECT_RESULT=1 PG_NO_SPIN_WAIT=1 (round 3)"; PG_NO_DIRECT_RESULT=1 PG_NO_SPIN_WAIT=1 timeout 200 python tools/latency.py 2>&1 | grep -v "^/opt"
} | tee $O/${TAG}_small_query_latency.txt
PG_TRACE_HOST=1 timeout 200 python tools/prof_variants.py --set strings --docs 20000000 --reps 2 2>&1 | grep -v "^/opt" | tail -12 | tee $O/${TAG}_strings_host_timeline.txt
