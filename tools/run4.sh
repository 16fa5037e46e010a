#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_4
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 > $O/tests.log 2>&1
echo "tests rc=$?" > $O/status.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>&1
echo "bench rc=$?" >> $O/status.txt
timeout 1500 bash tools/gloo_dryrun.sh > $O/gloo_dryrun.log 2>&1
timeout 900 bash tools/nccl_ws1.sh > $O/nccl_ws1.log 2>&1
for w in transh_fb15k237 transd_fb15k237 complex_wn18rr distmult_fb15k; do
  timeout 600 python bench.py --steps 5 --warmup 2 --workload $w --no-cpu-baseline --no-full-parity --weights xavier 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['secondary']
print('$w', 'ms/step', d['ms_per_step'], 'K1', s['scoring_function'], 'train', s['train_step']['ms'])" >> $O/k1.log 2>&1
done
tail -3 $O/tests.log; cat $O/status.txt; cat $O/k1.log; cat $O/gloo_dryrun.log; tail -c 400 $O/bench_default.log
