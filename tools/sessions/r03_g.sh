#!/bin/bash
# round 3, GPU call G: whole GPU suite (no -x) + filter-stats cost
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r03_g}
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/${TAG}_tests.log 2>&1; echo "gpu suite rc=$?"; tail -15 $O/${TAG}_tests.log | cut -c1-200
timeout 600 python tools/time_filter_stats.py > $O/${TAG}_filter_stats.txt 2>&1; cat $O/${TAG}_filter_stats.txt | grep -v "^/opt"
