#!/bin/bash
# r05: region recheck with two candidate chunks in flight
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "region_recheck or dot_query_side or query_pipeline" 2>&1 | tail -5 > gpurun_out/t9_tests.txt
{
for nw in 2 4 2 4; do echo "transe KGE_RECHECK_REGION_WAVES=$nw $(KGE_RECHECK_REGION_WAVES=$nw python bench.py --only-timed --steps 40 --warmup 5 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')"; done
for b in 36864 65536; do echo "complex KGE_REGION_MAX_BYTES=$b $(KGE_REGION_MAX_BYTES=$b python bench.py --only-timed --steps 40 --warmup 5 --workload complex_wn18rr 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')"; done
for b in 36864 65536; do echo "distmult KGE_REGION_MAX_BYTES=$b $(KGE_REGION_MAX_BYTES=$b python bench.py --only-timed --steps 40 --warmup 5 --workload distmult_fb15k 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')"; done
} > gpurun_out/t9_ab.txt 2>&1
bash tools/eval_timeline.sh gpurun_out/t9_timeline_transe.txt > /dev/null 2>&1
cat gpurun_out/t9_tests.txt gpurun_out/t9_ab.txt; cut -c1-130 gpurun_out/t9_timeline_transe.txt
