#!/bin/bash
# r05: after the list changes -- parity subset + timed evaluations of every workload
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "region_recheck or dot_query_side or query_pipeline or hi_stream or level1 or one_product or projection" 2>&1 | tail -5 > gpurun_out/t8_tests.txt
timeout 900 python -m pytest tests/test_gpu_fullsplit.py -x -q -m gpu -k "not reference" 2>&1 | tail -5 >> gpurun_out/t8_tests.txt
{
for w in transe_fb15k237 transh_fb15k237 transd_fb15k237 complex_wn18rr distmult_fb15k; do
  for i in 1 2; do echo "$w $(python bench.py --only-timed --steps 40 --warmup 5 --workload $w 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')"; done
done
} > gpurun_out/t8_ms.txt 2>&1
cat gpurun_out/t8_tests.txt gpurun_out/t8_ms.txt
