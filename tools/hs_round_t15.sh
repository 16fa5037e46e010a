#!/bin/bash
# r05: one RCCL rank with the collectives forced on -- the full JSON line (level, per-phase times)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export HSA_ENABLE_IPC_MODE_LEGACY=0 KGE_FORCE_COLLECTIVES=1
for args in "--steps 3 --warmup 1" "--steps 20 --warmup 5"; do
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 $args --no-cpu-baseline --no-secondary --no-weak 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$args', d['ms_per_step'], d.get('split_prefilter'), d.get('collective_time',{}).get('ms_per_evaluate'), d.get('config',{}).get('parallelism'))
print({k:d[k] for k in d if k in ('first_evaluate_ms','cold_ms_per_step','clock_settle')})"
done
