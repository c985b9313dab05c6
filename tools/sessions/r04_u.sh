#!/bin/bash
# round 4, session u: multi-value scan leaves by 32-byte windows: parity, then the mv variants with and without
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r04_u}
timeout 600 python -m pytest tests/test_gpu_mv.py tests/test_mv_reference_goldens.py -q -p no:cacheprovider -x --timeout 300 > $O/${TAG}_tests.log 2>&1; echo "mv tests rc=$?"; tail -6 $O/${TAG}_tests.log | cut -c1-300
{
timeout 400 python tools/prof_variants.py --set mv --docs 50000000 --reps 5 2>&1 | grep -v "^/opt"
echo "== PG_MV_NO_WINDOWS=1 (round 3 walk)"; PG_MV_NO_WINDOWS=1 timeout 400 python tools/prof_variants.py --set mv --only "mv scan" --docs 50000000 --reps 5 2>&1 | grep -v "^/opt"
} | tee $O/${TAG}_variants_mv_50m.txt
