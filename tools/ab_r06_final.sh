cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 120 python bench.py --no-cpu-baseline --no-traffic --no-full-check --no-variants --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(' '.join(sys.argv[1:]), '|', r['kernel'], 'kernel_ms', round(r['kernel_ms'],4), 'frac', round(r['frac'],4))
" "$@"; }
OLD=PG_GPU_LIB=$PWD/tools/variants/libpinot_gpu_old.so
for i in 1 2 3; do run $OLD; run X=new; done
