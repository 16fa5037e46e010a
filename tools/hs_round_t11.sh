#!/bin/bash
# r05: final build -- region tests, the default bench line as the driver runs it
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/profiles_r05
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "region_recheck or dot_query_side or query_pipeline" 2>&1 | tail -3 > gpurun_out/t11_tests.txt
( time python bench.py ) > gpurun_out/t11_bench.txt 2> gpurun_out/t11_bench.err
tail -1 gpurun_out/t11_bench.txt > gpurun_out/profiles_r05/bench_transe_fb15k237.json
cat gpurun_out/t11_tests.txt; tail -4 gpurun_out/t11_bench.err; tail -1 gpurun_out/t11_bench.txt | cut -c1-600
