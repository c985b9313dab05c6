#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r03_o}
timeout 900 python -m pytest tests/test_gpu_partition_pipeline.py tests/test_gpu_headline_kernels.py -q -p no:cacheprovider -x > $O/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/${TAG}_tests.log | cut -c1-250
timeout 300 python tools/prof_variants.py --set cfg5 --docs 200000000 --reps 6 > $O/${TAG}_cfg5.txt 2>&1; grep -v "^/opt" $O/${TAG}_cfg5.txt
timeout 200 python tools/prof_variants.py --set general --only groups --docs 200000000 --reps 6 > $O/${TAG}_general.txt 2>&1; grep -v "^/opt" $O/${TAG}_general.txt
