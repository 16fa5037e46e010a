#!/bin/bash
# r05: compact epilogue + fused table prep -- subset tests, probes, timeline, host profile, bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/hs3
mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_product or level_policy or projection_modes or table_prep or evaluator" > $OUT/pytest_subset.log 2>&1
tail -3 $OUT/pytest_subset.log
{
for pr in 0 1 12 16; do
  LEVEL=1 TAIL=1 FRAG=1 K=200 KGE_HS_PROBE=$pr timeout 120 python tools/split_time.py 2>&1 | grep count | sed "s/^/probe=$pr /"
done
LEVEL=1 TAIL=1 FRAG=1 K=400 timeout 120 python tools/split_time.py 2>&1 | grep count
LEVEL=1 TAIL=1 FRAG=1 K=200 B=40932 timeout 120 python tools/split_time.py 2>&1 | grep count
} > $OUT/probes.txt 2>&1
cat $OUT/probes.txt
bash tools/eval_timeline.sh $OUT/timeline_transe.txt
timeout 300 python tools/host_profile_eval.py > $OUT/host_profile.txt 2>&1
head -40 $OUT/host_profile.txt | cut -c1-150
timeout 300 python bench.py --no-cpu-baseline --no-secondary > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/hs3/bench.json').read().strip().splitlines()[-1])
    print({k: d[k] for k in ('value', 'ms_per_step') if k in d}, {k: d['roofline'].get(k) for k in ('frac', 'kernel_ms', 'kernel')})
    print(d.get('parity_full_split'))
except Exception as e:
    print('bench parse failed', e)
PY
tail -3 $OUT/bench.err
