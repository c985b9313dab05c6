#!/bin/bash
# One-shot measurement of a round on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 600 -- 'bash tools/profile_round.sh r02_v1'
# Writes gpurun_out/<tag>_*: the bench JSON line, the same bench under rocprofv3 --kernel-trace with its per-kernel summary, and
# the per-shape table of tools/prof_variants.py.  Copy what should be judged into profiles/.  Every step has its own timeout and
# reads nothing from stdin (a `head` without a file once cost ten GPU-minutes).
set -u
TAG=${1:-round}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd "$R" || exit 1
timeout 240 python bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.log" < /dev/null
tail -c 600 "$OUT/${TAG}_bench.json"; echo
( cd /tmp && export TMPDIR=/tmp && timeout 150 rocprofv3 --kernel-trace -d "$OUT/${TAG}_prof" -o x -- \
    python "$R/bench.py" --no-cpu-baseline --no-variants --steps 20 --warmup 5 > "$OUT/${TAG}_bench_under_rocprof.json" 2> /dev/null < /dev/null )
timeout 60 python tools/rocprof_summary.py "$OUT/${TAG}_prof/x_results.db" > "$OUT/${TAG}_kernel_stats.txt" 2>&1 < /dev/null
head -6 "$OUT/${TAG}_kernel_stats.txt" | cut -c1-150
# the dictionary-encoded config 3 and the flat config 5 as bench lines of their own under the kernel trace
for q in cfg3_dict cfg5; do
  ( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace -d "$OUT/${TAG}_prof_$q" -o x -- \
      python "$R/bench.py" --query $q --no-cpu-baseline --no-variants --no-traffic --steps 20 --warmup 5 > "$OUT/${TAG}_${q}_bench_under_rocprof.json" 2> /dev/null < /dev/null )
  timeout 60 python tools/rocprof_summary.py "$OUT/${TAG}_prof_$q/x_results.db" > "$OUT/${TAG}_${q}_kernel_stats.txt" 2>&1 < /dev/null
  head -5 "$OUT/${TAG}_${q}_kernel_stats.txt" | cut -c1-150
done
for set in cfg3 dict general cfg5 upsert wide; do
  docs=200000000; [ $set = wide ] && docs=100000000      # the wide set builds two 64-bit columns with numpy
  echo "# --set $set --docs $docs" >> "$OUT/${TAG}_variants_200m.txt"
  timeout 150 python tools/prof_variants.py --set $set --docs $docs 2>&1 < /dev/null | grep -v amdgpu.ids >> "$OUT/${TAG}_variants_200m.txt"
done
tail -40 "$OUT/${TAG}_variants_200m.txt" | cut -c1-170
