#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/hs4
mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_product or level_policy or table_prep or evaluator or query_pipeline" > $OUT/pytest_subset.log 2>&1
tail -3 $OUT/pytest_subset.log
bash tools/eval_timeline.sh $OUT/timeline_transe.txt
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/hs4/bench.json').read().strip().splitlines()[-1])
    print({k: d[k] for k in ('value', 'ms_per_step') if k in d}, {k: d['roofline'].get(k) for k in ('frac', 'kernel_ms', 'kernel')})
    print({k: v for k, v in d['cpu_baseline'].items() if k in ('value', 'kind', 'cores', 'reference_ranks_differing_from_port', 'ranks_differing')})
    print(d['cpu_baseline']['sample'][:300])
except Exception as e:
    print('bench parse failed', e)
PY
tail -3 $OUT/bench.err
