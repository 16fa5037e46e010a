#!/bin/bash
# round-2 GPU session 3: plan path + K1 DPP; evaluate-only kernel tables; K1 timings
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_3
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 -k "not full_test_split and not two_ranks" > $O/tests.log 2>&1
echo "tests rc=$?" > $O/status.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>&1
echo "bench rc=$?" >> $O/status.txt
timeout 600 bash tools/kprof.sh --steps 20 --warmup 5 --only-timed > $O/kprof_eval_default.log 2>&1
timeout 600 bash tools/kprof.sh --steps 20 --warmup 5 --only-timed --kg uniform --weights xavier > $O/kprof_eval_uniform_xavier.log 2>&1
for w in transh_fb15k237 transd_fb15k237 complex_wn18rr distmult_fb15k transe_l1_fb15k237; do
  timeout 600 python bench.py --steps 5 --warmup 2 --workload $w --no-cpu-baseline --no-full-parity --weights xavier > $O/bench_$w.log 2>&1
done
tail -3 $O/tests.log; cat $O/status.txt; tail -c 300 $O/bench_default.log
