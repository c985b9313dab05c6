#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r03_t}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-variants > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
echo "rc=$?"; find $O/${TAG}_prof -name "*kernel_stats*" | head; f=$(find $O/${TAG}_prof -name "*kernel_stats.csv" | head -1); head -12 "$f"
