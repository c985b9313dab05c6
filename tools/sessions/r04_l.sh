#!/bin/bash
# round 4, session l: wide pipeline parity after the test fix; SQ counters of pg_pipe_w_none (unfiltered LONG sum) next to pg_pipe_none (INT sum)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r04_l}
timeout 600 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -x --timeout 300 -k "wide" > $O/${TAG}_tests.log 2>&1; echo "wide tests rc=$?"; tail -4 $O/${TAG}_tests.log | cut -c1-400
timeout 400 python tools/pmc_sq.py wide "=sum(m64) group g1" 100000000 > $O/${TAG}_sq_w_none.txt 2>&1; grep -A18 "^pg_pipe_w_none" $O/${TAG}_sq_w_none.txt
timeout 400 python tools/pmc_sq.py cfg3 "=no filter sum(m) group g1" 100000000 > $O/${TAG}_sq_pipe_none.txt 2>&1; grep -A18 "^pg_pipe_none" $O/${TAG}_sq_pipe_none.txt
timeout 100 python tools/prof_variants.py --set cfg3 --only "=no filter sum(m) group g1" --docs 100000000 --reps 5 2>&1 | tail -1
