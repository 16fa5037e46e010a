#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_10
mkdir -p $O
cd $R
timeout 1200 python bench.py --workload complex_wikidata5m --no-secondary --no-cpu-baseline --no-full-parity --batch 8192 --kg uniform --steps 3 --warmup 1 > $O/bench_cfg5.log 2> $O/bench_cfg5.err
echo "rc=$?"; tail -c 1500 $O/bench_cfg5.log; tail -5 $O/bench_cfg5.err
