#!/bin/bash
# r05: regions for TransH / TransD; queries per wavefront of the TransE pipeline
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsplit.py -x -q -m gpu -k "transh or transd or TransH or TransD or projection or proj" 2>&1 | tail -5 > gpurun_out/t10_tests.txt
{
for w in transh_fb15k237 transd_fb15k237; do bash tools/ab_env.sh KGE_REGION_RECHECK 2 --workload $w | sed "s/^/$w /"; done
for q in 16 8 16 8; do echo "transe KGE_QPIPE_QPW=$q $(KGE_QPIPE_QPW=$q python bench.py --only-timed --steps 40 --warmup 5 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')"; done
} > gpurun_out/t10_ab.txt 2>&1
bash tools/eval_timeline.sh gpurun_out/t10_timeline_transh.txt --workload transh_fb15k237 > /dev/null 2>&1
cat gpurun_out/t10_tests.txt gpurun_out/t10_ab.txt; cut -c1-130 gpurun_out/t10_timeline_transh.txt
