#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r03_m}
timeout 900 python -m pytest tests/test_gpu_headline_kernels.py -q -p no:cacheprovider > $O/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -12 $O/${TAG}_tests.log | cut -c1-250
for set in general upsert; do
  timeout 200 python tools/prof_variants.py --set $set --docs 200000000 --reps 8 > $O/${TAG}_${set}.txt 2>&1; grep -v "^/opt" $O/${TAG}_${set}.txt
done
timeout 200 python tools/prof_variants.py --set wide --docs 100000000 --reps 8 > $O/${TAG}_wide.txt 2>&1; grep -v "^/opt" $O/${TAG}_wide.txt
