#!/bin/bash
# r05: the free-running one-product count kernel (lp_hi_stream.hip) -- correctness subset, then timing against the r04 kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/hs1
mkdir -p $OUT
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_product or projection_modes or level_policy" > $OUT/pytest_subset.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_subset.log
tail -5 $OUT/pytest_subset.log
{
for K in 200 400; do
  for frag in 0 1; do
    LEVEL=1 TAIL=1 FRAG=$frag K=$K timeout 120 python tools/split_time.py 2>&1 | grep -v Warning
  done
  LEVEL=1 TAIL=1 FRAG=1 K=$K KGE_HS_WAVES=8 timeout 120 python tools/split_time.py 2>&1 | grep count
  LEVEL=1 TAIL=1 FRAG=1 K=$K KGE_HS_QG=16 timeout 120 python tools/split_time.py 2>&1 | grep count
done
LEVEL=1 TAIL=1 FRAG=1 K=200 B=40932 timeout 120 python tools/split_time.py 2>&1 | grep count
LEVEL=1 TAIL=1 FRAG=0 K=200 B=40932 timeout 120 python tools/split_time.py 2>&1 | grep count
} > $OUT/split_time.txt 2>&1
cat $OUT/split_time.txt
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/hs1/bench.json').read().strip().splitlines()[-1])
    print({k: d[k] for k in ('value', 'ms_per_step') if k in d}, d.get('roofline'))
    print({k: v for k, v in d.items() if 'rank' in k or 'level' in k or 'parity' in k})
except Exception as e:
    print('bench parse failed', e)
PY
tail -3 $OUT/bench.err
