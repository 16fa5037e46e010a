#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_14
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -k "query_pipeline or evaluator_vs_reference or split_prefilter" > $O/tests.log 2>&1
echo "tests rc=$?" > $O/status.txt; tail -2 $O/tests.log
for q in 16 32; do
  KGE_QPIPE_QPW=$q timeout 600 bash tools/kprof.sh --steps 20 --warmup 5 --only-timed --weights xavier 2>&1 | grep -E "query_pipeline|^\{" | cut -c1-120
done
