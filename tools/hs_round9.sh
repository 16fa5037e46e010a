#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/hs9
mkdir -p $OUT
export PYTHONUNBUFFERED=1
python tools/dbg_transh.py 2>&1 | grep "differing"
python tools/dbg_transh.py transd_fb15k237 2>&1 | grep "differing"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_collectives.py -x -q -m gpu -k "transh or transd or projection or level_policy or one_product or dedupes or evaluator_vs or fresh or two_launches" > $OUT/pytest_subset.log 2>&1
tail -5 $OUT/pytest_subset.log
for w in transh_fb15k237 transd_fb15k237 complex_wn18rr distmult_fb15k transe_fb15k237; do
  bash tools/eval_timeline.sh $OUT/timeline_$w.txt --workload $w > /dev/null 2>&1
  echo "== $w"; grep -v "^kernel\|^#" $OUT/timeline_$w.txt | awk '{printf "%s %s %s | ", substr($0,1,40), $(NF-3), $(NF-2)} END {print ""}' | cut -c1-1500; tail -1 $OUT/timeline_$w.txt
done
