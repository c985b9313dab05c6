#!/bin/bash
# round 4, session g: raw multi-value columns on the GPU; what each pruned pass leaves in the stream; full-size config 5 against the sharded oracle
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r04_g}
timeout 900 python -m pytest tests/test_gpu_mv.py tests/test_mv_reference_goldens.py tests/test_gpu_oct.py -q -p no:cacheprovider -x > $O/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -12 $O/${TAG}_tests.log | cut -c1-300
PG_TRACE_OCT=1 timeout 200 python tools/prof_variants.py --set cfg5 --only "=cfg5" --docs 200000000 --reps 1 2>&1 | grep -v "^/opt" | tail -14
timeout 900 python -m pytest tests/test_gpu_full_size.py -q -p no:cacheprovider -x -k "config5" > $O/${TAG}_full.log 2>&1; echo "full-size cfg5 rc=$?"; tail -5 $O/${TAG}_full.log | cut -c1-300
