#!/bin/bash
# round 4, session m: wide pipeline with four quarter-tile buffers: parity + wide variants
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r04_m}
timeout 600 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -x --timeout 300 -k "wide" > $O/${TAG}_tests.log 2>&1; echo "wide tests rc=$?"; tail -4 $O/${TAG}_tests.log | cut -c1-400
timeout 300 python tools/prof_variants.py --set wide --docs 100000000 --reps 7 2>&1 | grep -v "^/opt" | tee $O/${TAG}_variants_wide_100m.txt
timeout 400 python tools/pmc_sq.py wide "=sum(d64) max(d64) group g1" 100000000 > $O/${TAG}_sq_w_none.txt 2>&1; grep -A18 "^pg_pipe_wd_none" $O/${TAG}_sq_w_none.txt | head -12
echo "== PG_NO_PIPE_WIDE_DOUBLE=1" | tee -a $O/${TAG}_variants_wide_100m.txt; PG_NO_PIPE_WIDE_DOUBLE=1 timeout 300 python tools/prof_variants.py --set wide --only "d64" --docs 100000000 --reps 7 2>&1 | grep -v "^/opt" | tee -a $O/${TAG}_variants_wide_100m.txt
