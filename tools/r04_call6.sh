#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsplit.py -m gpu -q -x -k "topk or inference or projection_modes or level_policy or cfg5" > $O/tests6.log 2>&1; echo "pytest rc=$?" | tee -a $O/tests6.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  |cfg5 subsample" $O/tests6.log | head -20 | cut -c1-400
timeout 300 python tools/topk_time.py 2>/dev/null | grep "^{" | tee $O/topk_inference.jsonl | cut -c1-600
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-traffic --no-full-parity"
for wl in transh_fb15k237 transd_fb15k237; do for lv in auto 0; do timeout 400 $B --workload $wl --split-level $lv 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$wl', '$lv', j['ms_per_step'], j['split_prefilter']['level_of_the_timed_evaluations'], j['split_prefilter']['rescored_pairs_per_query'], j['roofline']['kernel_ms'])"; done; done 2>&1 | tee $O/level_ab_proj.txt
bash tools/kprof_eval.sh --steps 20 --warmup 5 2>&1 | tee $O/kprof_eval_level1.txt
