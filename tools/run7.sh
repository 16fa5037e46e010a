#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_7
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 -k "not full_test_split and not two_ranks and not forced" > $O/tests.log 2>&1
echo "tests rc=$?" > $O/status.txt
timeout 600 bash tools/kprof.sh --steps 20 --warmup 5 --only-timed --weights xavier > $O/kprof_eval_zipf_xavier.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_default.log 2>&1
tail -3 $O/tests.log; cat $O/status.txt; grep -v "^W2026" $O/kprof_eval_zipf_xavier.log | cut -c1-140 | head -12
tail -1 $O/bench_default.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('default', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['parity_full_split']['ranks_differing'], d['parity_full_split']['outside_tie_interval'])"
