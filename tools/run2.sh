#!/bin/bash
# round-2 GPU session 2: failing tests + new tests, eval-only kernel profile, bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_2
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 -s -k "not full_test_split" > $O/tests.log 2>&1
echo "tests rc=$?" > $O/status.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>&1
echo "bench rc=$?" >> $O/status.txt
timeout 600 bash tools/kprof.sh --steps 10 --warmup 3 --no-full-parity --no-secondary --weights xavier > $O/kprof_zipf_xavier.log 2>&1
timeout 600 bash tools/kprof.sh --steps 10 --warmup 3 --no-full-parity --no-secondary --weights xavier --kg uniform > $O/kprof_uniform_xavier.log 2>&1
timeout 600 bash tools/kprof.sh --steps 5 --warmup 2 --no-full-parity --no-secondary --weights xavier --workload distmult_fb15k > $O/kprof_distmult.log 2>&1
tail -3 $O/tests.log; cat $O/status.txt; tail -c 300 $O/bench_default.log
