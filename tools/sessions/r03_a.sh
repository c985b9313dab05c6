#!/bin/bash
# round 3, GPU call A: parity of the partition pipeline + old/new timings of the high-cardinality group-bys
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O
timeout 420 python -m pytest tests/test_gpu_partition_pipeline.py -x -q > $O/r03_a_tests_part.log 2>&1; echo "part tests rc=$?"
tail -5 $O/r03_a_tests_part.log
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "radix or partitioned or config5 or hashed or num_groups or distinct" > $O/r03_a_tests_parity.log 2>&1; echo "parity subset rc=$?"
tail -3 $O/r03_a_tests_parity.log
for set in cfg5 general; do
  PG_NO_P2=1 timeout 200 python tools/prof_variants.py --set $set --docs 200000000 --reps 6 > $O/r03_a_${set}_old.txt 2>&1
  timeout 200 python tools/prof_variants.py --set $set --docs 200000000 --reps 6 > $O/r03_a_${set}_new.txt 2>&1
  echo "== $set old"; cat $O/r03_a_${set}_old.txt | grep -v "^#"
  echo "== $set new"; cat $O/r03_a_${set}_new.txt | grep -v "^#"
done
