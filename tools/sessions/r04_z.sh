#!/bin/bash
# round 4, session z: COUNT(*) over dense postings / one <= 8-bit dictionary scan as streams: parity + the cfg3 variants rows they touch, A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r04_z}
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -q -p no:cacheprovider -x --timeout 300 -n 4 > $O/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/${TAG}_tests.log | cut -c1-300
{
timeout 200 python tools/prof_variants.py --set cfg3 --only "count" --docs 200000000 --reps 9 2>&1 | grep -v "^/opt"
timeout 200 python tools/prof_variants.py --set cfg3 --only "g1 scan" --docs 200000000 --reps 9 2>&1 | grep -v "^/opt"
echo "== PG_NO_DENSE_COUNT=1 (round 3)"; PG_NO_DENSE_COUNT=1 timeout 200 python tools/prof_variants.py --set cfg3 --only "postings only" --docs 200000000 --reps 9 2>&1 | grep -v "^/opt"
PG_NO_DENSE_COUNT=1 timeout 200 python tools/prof_variants.py --set cfg3 --only "g1 scan" --docs 200000000 --reps 9 2>&1 | grep -v "^/opt"
} | tee $O/${TAG}_count_streams.txt
