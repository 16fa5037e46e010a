#!/bin/bash
# A/B of the split count kernel's DMA schedules on one box:  bash tools/run_sched.sh "0 1 2 3 4"
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
for sc in ${1:-0 1 2 3 4}; do
  echo "== KGE_SPLIT_SCHED=$sc (rep $rep)"
  KGE_SPLIT_SCHED=$sc python tools/split_time.py 2>&1 | grep -E "^count|split timing" | sort | uniq -c | sort -rn | head -4
done
done
for sc in ${1:-0 1 2 3 4}; do
  [ "$sc" = "0" ] && continue
  echo "== tests KGE_SPLIT_SCHED=$sc"
  KGE_SPLIT_SCHED=$sc timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "split_prefilter and not projection" 2>&1 | tail -2
done
