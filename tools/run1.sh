#!/bin/bash
# round-2 GPU session 1: tests, bench (default + r01-style), per-kernel times
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_1
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 -s > $O/tests.log 2>&1
echo "tests rc=$?" > $O/status.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>&1
echo "bench rc=$?" >> $O/status.txt
timeout 600 python bench.py --steps 20 --warmup 5 --kg uniform --weights xavier --no-full-parity --no-cpu-baseline > $O/bench_uniform_xavier.log 2>&1
echo "bench_ux rc=$?" >> $O/status.txt
timeout 600 python bench.py --steps 20 --warmup 5 --kg zipf --weights xavier --no-full-parity --no-cpu-baseline --no-secondary > $O/bench_zipf_xavier.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --kg uniform --weights trained --no-full-parity --no-cpu-baseline --no-secondary > $O/bench_uniform_trained.log 2>&1
timeout 600 bash tools/kprof.sh --steps 10 --warmup 3 --no-full-parity --no-secondary > $O/kprof_default.log 2>&1
tail -3 $O/tests.log; cat $O/status.txt; tail -c 600 $O/bench_default.log
