#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r03_u}
timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_bench_contract.py -q -p no:cacheprovider > $O/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -8 $O/${TAG}_tests.log | cut -c1-250
