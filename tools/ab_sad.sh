#!/bin/bash
# A/B of lp_l1_sad.hip build variants on one box:  bash tools/ab_sad.sh "<flags A>" "<flags B>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for flags in "$@"; do
  KGE_HIPCC_EXTRA="$flags" python -c "
import os
from torchkge_amd.csrc import build
os.utime(os.path.join(build.HERE, 'lp_l1_sad.hip'))
build.build()" > /dev/null 2>&1
  echo "== flags: $flags"
  bash tools/kprof_eval.sh --steps 10 --warmup 3 --workload transe_l1_fb15k237 --weights xavier 2>&1 | grep -E "lp_l1_sad|recheck"
done
