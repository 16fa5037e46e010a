#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_11
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_collectives.py -q -x > $O/tests.log 2>&1
echo "tests rc=$?" > $O/status.txt
tail -5 $O/tests.log; grep -n "MISMATCH\|Error\|rror:" $O/tests.log | head -20
timeout 900 bash tools/nccl_ws1.sh > $O/nccl_ws1.log 2>&1; cut -c1-260 $O/nccl_ws1.log | grep -o 'ms_per_step": [0-9.]*\|^[a-z-]*$' | paste - - 
