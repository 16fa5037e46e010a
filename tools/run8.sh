#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_8
mkdir -p $O
cd $R
for B in 4096 8192 16384 32768 65536 131072 262144; do timeout 300 python tools/k1_time.py $B 2>/dev/null | grep -E "^transe     d=200|^complex|^transh" >> $O/k1_B.log; done
cat $O/k1_B.log
