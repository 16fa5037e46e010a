#!/bin/bash
# same-box A/B of an environment switch:  bash tools/ab_env.sh VAR [rounds] [bench args]   (VAR=0 vs VAR=1, alternating)
V=$1; R=${2:-3}; shift 2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for r in $(seq 1 $R); do for x in 0 1; do
  echo "$V=$x $(env $V=$x python bench.py --only-timed --steps 40 --warmup 5 "$@" 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')"
done; done
