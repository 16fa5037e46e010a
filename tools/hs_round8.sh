#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/hs8
mkdir -p $OUT
export PYTHONUNBUFFERED=1
python tools/dbg_transh.py 2>&1 | grep "differing"
python tools/dbg_transh.py transd_fb15k237 2>&1 | grep "differing"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_collectives.py -x -q -m gpu -k "transh or transd or projection or level_policy or one_product or dedupes or evaluator_vs or fresh" > $OUT/pytest_subset.log 2>&1
tail -5 $OUT/pytest_subset.log
bash tools/eval_timeline.sh $OUT/timeline_transh.txt --workload transh_fb15k237
bash tools/eval_timeline.sh $OUT/timeline_transd.txt --workload transd_fb15k237 > /dev/null
tail -3 $OUT/timeline_transd.txt
bash tools/eval_timeline.sh $OUT/timeline_transe.txt > /dev/null; tail -2 $OUT/timeline_transe.txt
