#!/bin/bash
# Same-box A/B of environment knobs over tools/prof_variants.py: usage tools/ab_env.sh <set> <docs> <grep-pattern> "<ENV=1 or ->" ... (each spec run twice, alternating)
set=$1; docs=$2; pat=$3; shift 3
for i in 1 2; do
  for spec in "$@"; do
    echo "=== $spec"
    if [ "$spec" = "-" ]; then timeout 300 python tools/prof_variants.py --set $set --docs $docs 2>&1 | grep -E "$pat"
    else env $spec timeout 300 python tools/prof_variants.py --set $set --docs $docs 2>&1 | grep -E "$pat"; fi
  done
done
