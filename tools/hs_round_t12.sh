#!/bin/bash
# r05: larger per-wave sub-lists -- tests, timed evaluations, the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/profiles_r05
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsplit.py -x -q -m gpu -k "not reference and not large_entity" 2>&1 | tail -3 > gpurun_out/t12_tests.txt
{
for w in transe_fb15k237 transh_fb15k237 complex_wn18rr distmult_fb15k; do
  for i in 1 2; do echo "$w $(python bench.py --only-timed --steps 40 --warmup 5 --workload $w 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')"; done
done
} > gpurun_out/t12_ms.txt 2>&1
python bench.py 2>/dev/null | tail -1 > gpurun_out/t12_bench.json
cat gpurun_out/t12_tests.txt gpurun_out/t12_ms.txt; python3 -c "
import json; d=json.loads(open('gpurun_out/t12_bench.json').read()); r=d['roofline']; print(d['ms_per_step'], d['value'], r['kernel_ms'], r['frac'], r.get('mfma_busy_frac'), r.get('package_power'))"
