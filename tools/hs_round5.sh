#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/hs5
mkdir -p $OUT
export PYTHONUNBUFFERED=1
bash tools/eval_timeline.sh $OUT/timeline_transe.txt
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/hs5/bench.json').read().strip().splitlines()[-1])
    print({k: d[k] for k in ('value', 'ms_per_step', 'first_evaluate_ms', 'first_evaluate_parts') if k in d}, {k: d['roofline'].get(k) for k in ('frac', 'kernel_ms')})
    print(d.get('fresh_evaluator_per_validation'))
except Exception as e:
    print('bench parse failed', e)
PY
tail -3 $OUT/bench.err
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1
tail -15 $OUT/pytest_gpu.log
