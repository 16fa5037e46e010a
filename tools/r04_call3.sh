#!/bin/bash
# round-4 GPU call 3: the one-product level -- new tests, then same-box A/B of the evaluate with the level policy on / off
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "one_product or level_policy or split_prefilter or split_count or query_pipeline or evaluator_vs_reference" > $O/tests3.log 2>&1; echo "pytest rc=$?" | tee -a $O/tests3.log
tail -15 $O/tests3.log | cut -c1-250
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-traffic"
for lv in auto 0 auto 0; do timeout 400 $B --split-level $lv 2>/dev/null | tail -1 > $O/bench_level_$lv.json; python - <<PY
import json
j = json.load(open('$O/bench_level_$lv.json'))
r = j['roofline']
print('level arg $lv:', j['ms_per_step'], 'ms/step', j['split_prefilter'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'exec', r.get('executed_frac'), 'parity', (j.get('parity_full_split') or {}).get('within_reference_tie_interval_2e-5'), 'hits', j['filtered_hits_at_10'])
PY
done 2>&1 | tee $O/level_ab.txt
for wl in distmult_fb15k complex_wn18rr; do for lv in auto 0; do timeout 400 $B --workload $wl --split-level $lv --no-full-parity 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$wl', '$lv', j['ms_per_step'], j['split_prefilter']['level_of_the_timed_evaluations'], j['split_prefilter']['rescored_pairs_per_query'], j['roofline']['kernel_ms'])"; done; done 2>&1 | tee -a $O/level_ab.txt
