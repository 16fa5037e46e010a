#!/bin/bash
# r05: the uncertain-pair list in regions + region recheck -- tests, A/B, timeline
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "region_recheck or dot_query_side or query_pipeline or fused or hi_stream or level1 or one_product" 2>&1 | tail -15 > gpurun_out/t7_tests.txt
timeout 900 python -m pytest tests/test_gpu_fullsplit.py -x -q -m gpu 2>&1 | tail -5 >> gpurun_out/t7_tests.txt
{
for w in transe_fb15k237 complex_wn18rr distmult_fb15k; do
  bash tools/ab_env.sh KGE_REGION_RECHECK 2 --workload $w | sed "s/^/$w /"
done
for nw in 2 4; do echo "transe KGE_RECHECK_REGION_WAVES=$nw $(KGE_RECHECK_REGION_WAVES=$nw python bench.py --only-timed --steps 40 --warmup 5 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')"; done
} > gpurun_out/t7_ab.txt 2>&1
bash tools/eval_timeline.sh gpurun_out/t7_timeline_transe.txt > /dev/null 2>&1
bash tools/eval_timeline.sh gpurun_out/t7_timeline_distmult.txt --workload distmult_fb15k > /dev/null 2>&1
cat gpurun_out/t7_tests.txt gpurun_out/t7_ab.txt; cut -c1-130 gpurun_out/t7_timeline_transe.txt gpurun_out/t7_timeline_distmult.txt
