#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "one_product or level_policy or evaluator_vs_reference or dedupes" 2>&1 | tail -3
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-traffic --no-full-parity"
run() { timeout 400 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['ms_per_step'], 'cold', j['cold_ms_per_step'], 'level', j['split_prefilter']['level_of_the_timed_evaluations'], 'rescored', round(j['split_prefilter']['rescored_pairs_per_query'],2), 'kernel_ms', j['roofline']['kernel_ms'], 'f32same', (j.get('f32_mfma_only') or {}).get('ranks_identical_to_headline_run'))"; }
for rp in 1 0 1 0; do echo -n "transe RP=$rp: "; KGE_SPLIT_RP=$rp run $B; done 2>&1 | tee $O/resident_panel_ab.txt
for rp in 1 0; do echo -n "transe d=200 with columns (KGE_DEDUPE..) RP=$rp: "; KGE_SPLIT_RP=$rp run $B --workload transh_fb15k237; done 2>&1 | tee -a $O/resident_panel_ab.txt
