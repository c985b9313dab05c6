#!/bin/bash
# round 4, session h: raw multi-value columns on the GPU (after the registration deadlock fix); the pass schedule by offers per register at 2e8 and 1e9 docs
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r04_h}
timeout 300 python -m pytest tests/test_gpu_mv.py tests/test_mv_reference_goldens.py -q -p no:cacheprovider -x --timeout 120 > $O/${TAG}_tests.log 2>&1; echo "mv tests rc=$?"; tail -8 $O/${TAG}_tests.log | cut -c1-300
timeout 240 python -m pytest tests/test_gpu_oct.py -q -p no:cacheprovider -x --timeout 120 > $O/${TAG}_oct.log 2>&1; echo "oct tests rc=$?"; tail -4 $O/${TAG}_oct.log | cut -c1-300
PG_TRACE_OCT=1 timeout 200 python tools/prof_variants.py --set cfg5 --only "=cfg5" --docs 200000000 --reps 3 2>&1 | grep -v "^/opt" | tail -14 | tee $O/${TAG}_cfg5_200m.txt
for P in "" "0.02,0.08,0.3,1" "0.015,0.06,0.2,0.5,1"; do
  echo "== PG_OCT_PASSES='$P' at 1e9"
  PG_OCT_PASSES=$P timeout 300 python tools/prof_variants.py --set cfg5 --only "=cfg5" --docs 1000000000 --reps 3 2>&1 | grep -v "^/opt" | tail -4
done | tee $O/${TAG}_cfg5_1b.txt
