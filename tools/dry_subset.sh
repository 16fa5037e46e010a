#!/bin/bash
# the N > 1 bench line on a ONE-GPU box (what the builder can run): two gloo ranks on GPU 0 for the logic, then one RCCL
# rank with the collectives forced on for the nccl backend.  Output: profiles/<round>/n2_dryrun_gloo.txt
cd ${GRAFT_REPO_ROOT:-.}
source <(sed -n '/^run()/,/^}/p' tools/gloo_dryrun.sh)
echo "== 2 gloo ranks on one GPU: default (strong scaling of the cfg2 job, row-sharded tables, counts all-reduce; score all-to-all + weak mode beside it)"; run
echo "== 2 gloo ranks: --exchange scores --weights xavier --no-weak"; run --exchange scores --weights xavier --no-weak
echo "== 2 gloo ranks: KGE_EAGER_COLLECTIVES=1 --exchange counts"; KGE_EAGER_COLLECTIVES=1 run --exchange counts --weights xavier --no-weak
export HSA_ENABLE_IPC_MODE_LEGACY=0 KGE_FORCE_COLLECTIVES=1
nrun() { timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | tail -1 | grep -o '"value": [0-9.e+]*\|"ms_per_step": [0-9.]*\|"scaling": "[^"]*"\|"parallelism": "[^"]*"\|"collective_time": {[^}]*}\|"other_exchange": {[^}]*}[^}]*}\|"weak_mode": {[^}]*}[^}]*}' | paste -s -d' '; }
echo "== 1 RCCL rank, collectives forced: default"; nrun
echo "== 1 RCCL rank: --exchange counts --graph-collectives --no-weak"; nrun --exchange counts --graph-collectives --no-weak
