#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "level_policy or one_product" 2>&1 | tail -2
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-traffic --no-full-parity"
run() { timeout 400 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['ms_per_step'], 'cold', j['cold_ms_per_step'], 'level', j['split_prefilter']['level_of_the_timed_evaluations'], 'rescored', j['split_prefilter']['rescored_pairs_per_query'], 'kernel_ms', j['roofline']['kernel_ms'])"; }
for wl in transe_fb15k237 complex_wn18rr distmult_fb15k transh_fb15k237; do for dd in 1 0 1 0; do echo -n "$wl DEDUPE_LEVEL1=$dd: "; KGE_DEDUPE_LEVEL1=$dd run $B --workload $wl; done; done 2>&1 | tee $O/dedupe_level1_ab.txt
echo -n "uniform xavier: "; run $B --kg uniform --weights xavier | tee -a $O/dedupe_level1_ab.txt
echo -n "zipf xavier: "; run $B --weights xavier | tee -a $O/dedupe_level1_ab.txt
