#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_5
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_collectives.py -q > $O/tests.log 2>&1
echo "tests rc=$?" > $O/status.txt
for b in default 1024 4096 8192; do
  if [ $b = default ]; then unset KGE_K1_BLOCKS; else export KGE_K1_BLOCKS=$b; fi
  timeout 300 python tools/k1_time.py >> $O/k1.log 2>&1
done
unset KGE_K1_BLOCKS
tail -3 $O/tests.log; cat $O/k1.log
timeout 2400 bash tools/profile_round.sh r02 > $O/profile_round.log 2>&1
tail -5 $O/profile_round.log
