#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "one_product or level_policy" > $O/tests5.log 2>&1; echo "pytest rc=$?" | tee -a $O/tests5.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $O/tests5.log | head -20 | cut -c1-300
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-traffic"
for lv in auto 0 auto 0; do timeout 400 $B --split-level $lv > $O/bench_level.out 2> $O/bench_level.err; tail -1 $O/bench_level.out > $O/bench_level_$lv.json; python - <<PY
import json
try:
    j = json.load(open('$O/bench_level_$lv.json'))
    r = j['roofline']
    print('level arg $lv:', j['ms_per_step'], 'ms/step', j['split_prefilter']['level_of_the_timed_evaluations'], j['split_prefilter']['rescored_pairs_per_query'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'exec', r.get('executed_frac'), 'parity', (j.get('parity_full_split') or {}).get('within_reference_tie_interval_2e-5'), 'hits', j['filtered_hits_at_10'], 'f32', j.get('f32_mfma_only'))
except Exception as e:
    print('FAILED', e); print(open('$O/bench_level.err').read()[-1500:])
PY
done 2>&1 | tee $O/level_ab.txt
