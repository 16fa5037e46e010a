#!/bin/bash
# Power / clock while the split count kernel runs back to back:  bash tools/power_probe.sh [KGE_SPLIT_DBG]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
# (optional first argument: value for KGE_SPLIT_DBG, the timing probes of the kernel)
export KGE_SPLIT_DBG=${1:-0}
python - <<'PY' &
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
sys.argv = ['x']
import runpy
# reuse the harness of split_time.py up to the prepared problem
src = open('tools/split_time.py').read().split("for name, fn in")[0]
exec(compile(src, 'split_time_head', 'exec'))
torch.cuda.synchronize()
t0 = time.time()
n = 0
while time.time() - t0 < 6.0:
    for _ in range(200):
        count()
    torch.cuda.synchronize()
    n += 200
print('launches', n, 'ms/launch', (time.time() - t0) / n * 1e3)
PY
PID=$!
sleep 3.5
for i in 1 2 3; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | head -6
  sleep 0.7
done
wait $PID
