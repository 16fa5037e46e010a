#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_13
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -k "query_pipeline or evaluator_vs_reference or split_prefilter or full_size" > $O/tests.log 2>&1
echo "tests rc=$?" > $O/status.txt; tail -2 $O/tests.log
for q in 16 8 32; do
  KGE_QPIPE_QPW=$q timeout 600 bash tools/kprof.sh --steps 20 --warmup 5 --only-timed --weights xavier 2>&1 | grep -E "query_pipeline|^\{" | cut -c1-120
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['parity_full_split']
print('default ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'parity', p['ranks_differing'], p['outside_tie_interval'])"
