#!/bin/bash
# r05: fused DOT query side in the evaluator -- parity tests, full-split comparisons, timed evaluations A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "dot_query_side" 2>&1 | tail -15 > gpurun_out/t3_tests.txt
timeout 900 python -m pytest tests/test_gpu_fullsplit.py tests/test_gpu_parity.py -x -q -m gpu -k "complex or distmult or ComplEx or DistMult or bilinear" 2>&1 | tail -8 >> gpurun_out/t3_tests.txt
{
for w in complex_wn18rr distmult_fb15k; do
  for f in 1 0 1 0; do
    echo -n "$w KGE_DOT_FUSED=$f: "; KGE_DOT_FUSED=$f python bench.py --workload $w --only-timed --steps 40 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('split_prefilter',{}).get('level_of_the_timed_evaluations'), d.get('filtered_hits_at_10'))"
  done
done
} > gpurun_out/t3_ab.txt 2>&1
bash tools/eval_timeline.sh gpurun_out/t3_timeline_complex.txt --workload complex_wn18rr > /dev/null 2>&1
bash tools/eval_timeline.sh gpurun_out/t3_timeline_distmult.txt --workload distmult_fb15k > /dev/null 2>&1
cat gpurun_out/t3_tests.txt gpurun_out/t3_ab.txt; cut -c1-130 gpurun_out/t3_timeline_complex.txt gpurun_out/t3_timeline_distmult.txt
