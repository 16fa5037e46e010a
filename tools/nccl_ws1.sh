#!/bin/bash
# RCCL smoke on a single-GPU box: one rank, sharded code paths and collectives forced on.
export HSA_ENABLE_IPC_MODE_LEGACY=0 KGE_FORCE_COLLECTIVES=1
run() { timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary "$@" 2>&1 | tail -1 | cut -c1-2400; }
echo "default (strong, counts all-reduce; score all-to-all beside it)"; run
echo strong-entities-counts; run --exchange counts --no-weak
echo strong-queries; run --shard queries
echo weak; run --scaling weak --exchange counts
echo complex; run --workload complex_wn18rr --no-weak
echo transh-counts; run --workload transh_fb15k237 --exchange counts --no-weak
echo counts-graph-collectives; run --exchange counts --graph-collectives --no-weak
echo counts-eager-collectives-env; KGE_EAGER_COLLECTIVES=1 run --exchange counts --no-weak
