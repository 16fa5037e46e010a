#!/bin/bash
# r05: [r | r] with the plan, cheaper capture key -- tests + timed evaluations
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsplit.py tests/test_reference_style.py -x -q -m gpu -k "not reference and not large_entity" 2>&1 | tail -3 > gpurun_out/t14_tests.txt
{
for w in transe_fb15k237 transh_fb15k237 transd_fb15k237 complex_wn18rr; do
  for i in 1 2; do echo "$w $(python bench.py --only-timed --steps 40 --warmup 5 --workload $w 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')"; done
done
} > gpurun_out/t14_ms.txt 2>&1
cat gpurun_out/t14_tests.txt gpurun_out/t14_ms.txt
