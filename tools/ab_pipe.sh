#!/bin/bash
# A/B of the pipelined headline kernel (pg_fast_i32range_p): wavefronts per workgroup (variants pw2 / pw4 / pw8 built with
# tools/build_variants.sh name "PIPE_WAVES=n") x workgroups per CU, against pg_fast_i32range_d (PG_NO_PIPE=1).
# usage (GPU box): tools/ab_pipe.sh "pw8:1 pw8:2 pw4:1 pw4:2 pw4:4 pw2:4 pw2:8 nopipe"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for spec in $1; do
  v=${spec%%:*}; w=${spec##*:}
  if [ "$v" = nopipe ]; then envs="PG_NO_PIPE=1"; else envs="PG_GPU_LIB=$R/tools/variants/libpinot_gpu_$v.so PG_PIPE_WGS_PER_CU=$w"; fi
  out=$(env $envs timeout 120 python bench.py --no-cpu-baseline --no-traffic --no-full-check --steps 20 --warmup 5 2>/dev/null < /dev/null | tail -1)
  python - "$spec" "$out" <<'PY'
import json, sys
v, out = sys.argv[1:3]
try:
    d = json.loads(out)
    r, n = d["roofline"], d.get("north_star_variant", {})
    print(f"{v:10s} {r['kernel']}  cfg3 kernel {r['kernel_ms']:.4f} ms frac {r['frac']:.4f} step {d['ms_per_step']:.4f} ms | north-star kernel {n.get('kernel_ms', 0):.4f} ms frac {n.get('roofline_frac', 0):.4f} parity {d.get('parity_checked')}")
except Exception as e:
    print(v, "failed:", e, out[-300:])
PY
done
