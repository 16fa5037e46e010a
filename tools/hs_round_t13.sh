#!/bin/bash
# r05: same-box A/B of the per-wave list size (old: 384 entries, flush at a third; new: 768, flush at half)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
{
for w in transh_fb15k237 transe_fb15k237 distmult_fb15k; do
  bash tools/ab.sh gpurun_ab_old.so gpurun_ab_new.so 2 -- python bench.py --only-timed --steps 40 --warmup 5 --workload $w | sed "s/^/$w /"
done
} > gpurun_out/t13_ab.txt 2>&1
cat gpurun_out/t13_ab.txt
