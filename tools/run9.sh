#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2_9
mkdir -p $O
cd $R
timeout 300 python tools/k1_time.py 32768 2>/dev/null > $O/k1.log
KGE_K1_BLOCKS=4096 timeout 300 python tools/k1_time.py 32768 2>/dev/null | head -3 >> $O/k1.log
KGE_K1_BLOCKS=8192 timeout 300 python tools/k1_time.py 32768 2>/dev/null | head -3 >> $O/k1.log
cat $O/k1.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-parity 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['secondary'])"
