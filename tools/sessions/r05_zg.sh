#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
mkdir -p gpurun_out
timeout 400 python bench.py --no-cpu-baseline --no-concurrency --no-variants --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(r['kernel'], 'kernel_ms %.4f frac %.4f traffic %.4g (%.4f x algorithmic)' % (r['kernel_ms'], r['frac'], r['traffic'] or 0, (r['traffic'] or 0)/r['algorithmic_bytes_per_launch']))" | tee gpurun_out/r05_zg_traffic.txt
timeout 300 python -m pytest tests/test_gpu_headline_kernels.py -x -q -m gpu -k "specialised or follows" 2>&1 | tail -1
