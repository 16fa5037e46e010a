#!/bin/bash
# r05: probes of the free-running kernel + timeline of one evaluate
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/hs2
mkdir -p $OUT
export PYTHONUNBUFFERED=1
{
for pr in 0 1 2 4 8 12 13 16 17; do
  LEVEL=1 TAIL=1 FRAG=1 K=200 KGE_HS_PROBE=$pr timeout 120 python tools/split_time.py 2>&1 | grep count | sed "s/^/probe=$pr /"
done
} > $OUT/probes.txt 2>&1
cat $OUT/probes.txt
bash tools/eval_timeline.sh $OUT/timeline_transe.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_product or level_policy or evaluator" > $OUT/pytest_subset.log 2>&1
tail -3 $OUT/pytest_subset.log
