#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r03_r}
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/${TAG}_tests.log | cut -c1-300
timeout 600 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"; tail -c 3000 $O/${TAG}_bench.json; tail -5 $O/${TAG}_bench.err
