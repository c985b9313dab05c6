#!/bin/bash
# the whole -m gpu suite as the driver runs it (plus xdist), and smoke()
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r04_full}
timeout 1500 python -m pytest tests/ -q -m gpu -p no:cacheprovider --timeout 900 -n 4 > $O/${TAG}_tests.log 2>&1; echo "gpu suite rc=$?"; tail -15 $O/${TAG}_tests.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
