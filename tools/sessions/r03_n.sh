#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r03_n}
timeout 900 python -m pytest tests/test_gpu_headline_kernels.py -q -p no:cacheprovider -x > $O/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/${TAG}_tests.log | cut -c1-250
timeout 800 python tools/pmc_sq.py cfg5 "cfg5" 200000000 > $O/${TAG}_sq_cfg5.txt 2>&1; grep -v "^/opt" $O/${TAG}_sq_cfg5.txt | head -150
