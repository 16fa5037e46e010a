#!/bin/bash
# r05, last call: the whole GPU suite on the final build + the TransH / TransD timelines after the last change
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/profiles_r05
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/final_tests.txt
bash tools/eval_timeline.sh gpurun_out/profiles_r05/timeline_transh_fb15k237.txt --workload transh_fb15k237 > /dev/null 2>&1
bash tools/eval_timeline.sh gpurun_out/profiles_r05/timeline_transd_fb15k237.txt --workload transd_fb15k237 > /dev/null 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> gpurun_out/final_tests.txt
cat gpurun_out/final_tests.txt; cut -c1-120 gpurun_out/profiles_r05/timeline_transh_fb15k237.txt
