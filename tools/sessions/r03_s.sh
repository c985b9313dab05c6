#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r03_s}
T0=$(date +%s); timeout 600 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$? wall=$(( $(date +%s) - T0 ))s"; grep "bench\]" $O/${TAG}_bench.err
