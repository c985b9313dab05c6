#!/bin/bash
# round 4, session k: the wide pipeline (pg_pipe_w_*): parity, then the wide variants table at 1e8 docs with and without it
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r04_k}
timeout 600 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -x --timeout 300 -k "wide" > $O/${TAG}_tests.log 2>&1; echo "wide tests rc=$?"; tail -12 $O/${TAG}_tests.log | cut -c1-400
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline_kernels.py -q -p no:cacheprovider -x --timeout 300 -n 4 > $O/${TAG}_tests2.log 2>&1; echo "parity tests rc=$?"; tail -4 $O/${TAG}_tests2.log | cut -c1-300
{
echo "== wide pipeline"; timeout 300 python tools/prof_variants.py --set wide --docs 100000000 --reps 7 2>&1 | grep -v "^/opt"
echo "== PG_NO_PIPE_WIDE=1 (round 3)"; PG_NO_PIPE_WIDE=1 timeout 300 python tools/prof_variants.py --set wide --docs 100000000 --reps 7 2>&1 | grep -v "^/opt"
} | tee $O/${TAG}_variants_wide_100m.txt
