#!/bin/bash
# r05: how evaluate() waits for the ranks -- event polling vs hipStreamSynchronize, and the runtime's own active-wait knob
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
{
bash tools/ab_env.sh KGE_SPIN_WAIT 3
for v in 0 100 1000; do echo "ROC_ACTIVE_WAIT_TIMEOUT=$v $(ROC_ACTIVE_WAIT_TIMEOUT=$v python bench.py --only-timed --steps 40 --warmup 5 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')"; done
bash tools/ab_env.sh KGE_SPIN_WAIT 2 --workload complex_wn18rr
} > gpurun_out/t6_ab.txt 2>&1
KGE_SPIN_WAIT=1 bash tools/eval_timeline.sh gpurun_out/t6_timeline_transe_spin.txt > /dev/null 2>&1
cat gpurun_out/t6_ab.txt; cut -c1-130 gpurun_out/t6_timeline_transe_spin.txt
