#!/bin/bash
# round 4, session a: oct kernels (parity first), full GPU suite, cfg5 variants with / without the oct kernels, SQ counters of the pruned pass
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r04_a}
timeout 600 python -m pytest tests/test_gpu_oct.py -q -p no:cacheprovider -x > $O/${TAG}_oct_tests.log 2>&1; echo "oct tests rc=$?"; tail -40 $O/${TAG}_oct_tests.log | cut -c1-300
timeout 200 python tools/prof_variants.py --set cfg5 --docs 200000000 --reps 8 > $O/${TAG}_cfg5_oct.txt 2>&1; grep -v "^/opt" $O/${TAG}_cfg5_oct.txt
PG_NO_OCT=1 timeout 200 python tools/prof_variants.py --set cfg5 --docs 200000000 --reps 8 > $O/${TAG}_cfg5_nooct.txt 2>&1; grep -v "^/opt" $O/${TAG}_cfg5_nooct.txt
timeout 900 python -m pytest tests -q -p no:cacheprovider -m gpu > $O/${TAG}_all_tests.log 2>&1; echo "all tests rc=$?"; tail -15 $O/${TAG}_all_tests.log | cut -c1-300
