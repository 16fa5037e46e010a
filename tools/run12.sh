#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_12
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 -k "not full_test_split" > $O/tests.log 2>&1
echo "tests rc=$?" > $O/status.txt
tail -3 $O/tests.log
for w in transh_fb15k237 transd_fb15k237 transe_fb15k237; do
  timeout 600 python bench.py --steps 10 --warmup 3 --workload $w --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['parity_full_split']
print('$w', 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], d['roofline']['frac'], 'parity', p['ranks_differing'], p['outside_tie_interval'], 'f32', d['f32_mfma_only']['ms_per_step'])"
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
