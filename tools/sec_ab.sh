#!/bin/bash
# A/B of the backward's id ordering (KGE_BWD_PERM=sort|torch|count) on the secondary bench numbers
for m in sort torch count; do
  echo "== KGE_BWD_PERM=$m"
  KGE_BWD_PERM=$m python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-full-parity --settle-ms 0 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
s = d['secondary']
print('train_step eager ms', s['train_step']['ms'], ' hipgraph ms', s['train_step_hipgraph']['ms'], ' build-time training s', d['workload_detail']['train_s'])
"
done
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "backward" 2>&1 | tail -2
