#!/bin/bash
# RCCL smoke on a single-GPU box: one rank, sharded code paths and collectives forced on.
export HSA_ENABLE_IPC_MODE_LEGACY=0 KGE_FORCE_COLLECTIVES=1
run() { timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary "$@" 2>&1 | tail -1 | cut -c1-1200; }
echo weak; run
echo strong-entities-counts; run --scaling strong --shard entities --exchange counts
echo strong-entities-scores; run --scaling strong --shard entities --exchange scores
echo strong-queries; run --scaling strong --shard queries
echo complex-weak; run --workload complex_wn18rr
echo transh-strong; run --workload transh_fb15k237 --scaling strong
echo weak-graph-collectives; run --graph-collectives
echo transh-strong-graph-collectives; run --workload transh_fb15k237 --scaling strong --graph-collectives
