#!/bin/bash
# r05: where the host time between two replays goes; filter correction beside the recheck instead of beside the sweep
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 300 python tools/host_profile_eval.py > gpurun_out/t5_host_profile.txt 2>&1
{
for w in transe_fb15k237 complex_wn18rr transh_fb15k237 distmult_fb15k; do
  bash tools/ab_env.sh KGE_FILTER_BESIDE_RECHECK 2 --workload $w | sed "s/^/$w /"
done
} > gpurun_out/t5_ab.txt 2>&1
KGE_FILTER_BESIDE_RECHECK=1 bash tools/eval_timeline.sh gpurun_out/t5_timeline_transe_fbr.txt > /dev/null 2>&1
head -60 gpurun_out/t5_host_profile.txt | cut -c1-150; cat gpurun_out/t5_ab.txt; cut -c1-130 gpurun_out/t5_timeline_transe_fbr.txt
