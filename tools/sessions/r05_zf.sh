#!/bin/bash
# same-box A/B of the headline bench line: pg_fast_i32range_s (default) against pg_fast_i32range_p (PG_NO_WAVE_SPECIALISED=1), alternating
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
mkdir -p gpurun_out
for round in 1 2; do
  for v in "" "PG_NO_WAVE_SPECIALISED=1"; do
    env $v timeout 300 python bench.py --no-cpu-baseline --no-traffic --no-concurrency --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; n=d.get('north_star_variant',{})
print('${v:-default}'.ljust(28), r['kernel'], 'kernel_ms %.4f frac %.4f  ms/step %.4f | north-star kernel_ms %.4f frac %.4f' % (r['kernel_ms'], r['frac'], d['ms_per_step'], n.get('kernel_ms',0), n.get('roofline_frac',0)))"
  done
done | tee gpurun_out/r05_zf_headline_ab.txt
