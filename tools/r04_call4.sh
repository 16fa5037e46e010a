#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "one_product or level_policy" > $O/tests4.log 2>&1; echo "pytest rc=$?" | tee -a $O/tests4.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/tests4.log | head -40 | cut -c1-300
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-traffic --split-level auto > $O/bench_auto.out 2> $O/bench_auto.err; echo "bench rc=$?"
tail -5 $O/bench_auto.err | cut -c1-600
tail -1 $O/bench_auto.out | cut -c1-400
