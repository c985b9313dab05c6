#!/bin/bash
# round 4, session w: the default bench line (as the driver runs it) and the same command's kernel stats under rocprofv3
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O; TAG=${1:-r04_w}
( time timeout 900 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err ) 2>&1 | tail -3; echo "bench rc=$?"; grep "^{" $O/${TAG}_bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'traffic/alg', d['roofline']['traffic'] / d['roofline']['algorithmic_bytes_per_launch'] if d['roofline']['traffic'] else None)
c=d['cfg5_flat']; print('cfg5_flat kernel_ms', c['kernel_ms'], 'ms_per_step', c['ms_per_step'], 'frac', c['roofline_frac'], 'traffic/alg', c.get('traffic_over_algorithmic'))
c=d['cfg2']; print('cfg2 kernel_ms', c['kernel_ms'], 'p50', c['p50_query_latency_ms'])
c=d['cfg5_star_tree']; print('star-tree p50', c['p50_query_latency_ms'], 'device', c['device_ms'])"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bench; timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_bench -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $O/${TAG}_bench_under_rocprof.json 2> $O/${TAG}_bench_under_rocprof.err
echo "rocprof rc=$?"; db=$(find /tmp/prof_bench -name "*_results.db" | head -1); python $R/tools/rocprof_summary.py $db > $O/${TAG}_kernel_stats.txt 2>&1; head -30 $O/${TAG}_kernel_stats.txt | cut -c1-170
