#!/bin/bash
# A/B of kernel-shape variants (tools/build_variants.sh) on the headline bench: prints kernel ms / roofline fraction per variant.
# usage (GPU box): tools/ab_variants.sh "base scan4 w8" ["1 2"]   (second argument: PG_WGS_PER_CU values to try per variant)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for v in $1; do
  for w in ${2:-1}; do
    out=$(PG_GPU_LIB=$R/tools/variants/libpinot_gpu_$v.so PG_WGS_PER_CU=$w timeout 120 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null < /dev/null | tail -1)
    python - "$v" "$w" "$out" <<'PY'
import json, sys
v, w, out = sys.argv[1:4]
try:
    d = json.loads(out)
    r, n = d["roofline"], d.get("north_star_variant", {})
    print(f"{v:10s} wgs/cu={w}  cfg3 kernel {r['kernel_ms']:.4f} ms frac {r['frac']:.4f} step {d['ms_per_step']:.4f} ms | north-star kernel {n.get('kernel_ms', 0):.4f} ms frac {n.get('roofline_frac', 0):.4f}")
except Exception as e:
    print(v, w, "failed:", e, out[-300:])
PY
  done
done
