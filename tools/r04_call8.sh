#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests8.log 2>&1; echo "pytest rc=$?" | tee -a $O/tests8.log
tail -6 $O/tests8.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python bench.py --steps 20 --warmup 3 > $O/bench_default.out 2> $O/bench_default.err; echo "bench rc=$?"; tail -1 $O/bench_default.out | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; print(j['ms_per_step'], j['value'], j['split_prefilter'], r['frac'], r.get('executed_frac'), r.get('mfma_busy_frac'), r.get('traffic'), (r.get('pmc') or {}).get('write_bytes'), j['first_evaluate_ms'], (j['parity_full_split'] or {}).get('outside_tie_interval'))"
