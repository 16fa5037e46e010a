#!/bin/bash
# r05: DOT candidate side in two launches, small-batch query pipeline, batch coalescing -- tests, A/B, timelines
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "dot_query_side" 2>&1 | tail -15 > gpurun_out/t4_tests.txt
timeout 900 python -m pytest tests/test_gpu_fullsplit.py tests/test_gpu_parity.py -x -q -m gpu -k "complex or distmult or ComplEx or DistMult or bilinear" 2>&1 | tail -8 >> gpurun_out/t4_tests.txt
one() { python bench.py --workload $1 --only-timed --steps 40 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('split_prefilter',{}).get('level_of_the_timed_evaluations'), d.get('filtered_hits_at_10'))"; }
{
for f in 1 0 1 0; do echo -n "complex_wn18rr KGE_DOT_FUSED=$f: "; KGE_DOT_FUSED=$f one complex_wn18rr; done
for c in 32768 65536 131072 32768 65536 131072; do echo -n "distmult_fb15k KGE_COALESCE_BATCH=$c: "; KGE_COALESCE_BATCH=$c one distmult_fb15k; done
} > gpurun_out/t4_ab.txt 2>&1
bash tools/eval_timeline.sh gpurun_out/t4_timeline_complex.txt --workload complex_wn18rr > /dev/null 2>&1
KGE_COALESCE_BATCH=131072 bash tools/eval_timeline.sh gpurun_out/t4_timeline_distmult_one_batch.txt --workload distmult_fb15k > /dev/null 2>&1
TL_BACK=1 bash tools/eval_timeline.sh gpurun_out/t4_timeline_distmult_first_batch.txt --workload distmult_fb15k > /dev/null 2>&1
cat gpurun_out/t4_tests.txt gpurun_out/t4_ab.txt; cut -c1-130 gpurun_out/t4_timeline_complex.txt gpurun_out/t4_timeline_distmult_one_batch.txt gpurun_out/t4_timeline_distmult_first_batch.txt
