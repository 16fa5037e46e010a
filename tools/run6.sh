#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_6
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 -k "not full_test_split" > $O/tests.log 2>&1
echo "tests rc=$?" > $O/status.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>&1
echo "bench rc=$?" >> $O/status.txt
timeout 600 bash tools/kprof.sh --steps 20 --warmup 5 --only-timed --weights xavier > $O/kprof_eval_zipf_xavier.log 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 --workload distmult_fb15k --no-cpu-baseline --no-full-parity > $O/bench_distmult.log 2>&1
tail -3 $O/tests.log; cat $O/status.txt; grep -v "^W2026" $O/kprof_eval_zipf_xavier.log | cut -c1-140 | head -16
for f in bench_default bench_distmult; do tail -1 $O/$f.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$f', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'])"; done
