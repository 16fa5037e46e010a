#!/bin/bash
# r05: TransE query pipeline with the residual sums on all lanes -- parity tests, timeline, timed evaluate
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "query_pipeline or fused or level1 or hi_stream or one_product" 2>&1 | tail -5 > gpurun_out/t1_tests.txt
bash tools/eval_timeline.sh gpurun_out/t1_timeline_transe.txt > /dev/null 2>&1
for i in 1 2 3; do python bench.py --only-timed --steps 40 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done > gpurun_out/t1_ms.txt
cat gpurun_out/t1_tests.txt gpurun_out/t1_timeline_transe.txt gpurun_out/t1_ms.txt
