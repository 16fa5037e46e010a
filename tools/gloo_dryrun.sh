#!/bin/bash
# logic dry run of the N > 1 bench paths on a one-GPU box: 2 ranks on GPU 0, gloo backend
run() { timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29578 bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --no-cpu-baseline --no-secondary "$@" 2>/dev/null | tail -1 | grep -o '^{"metric\|"value": [0-9.e+]*\|ms_per_step[^,]*\|filtered_mrr[^,]*\|"parallelism": "[^"]*"\|hip_graph[^,]*' | paste -s -d' '; }
echo weak; run
echo strong-entities-counts; run --scaling strong --shard entities --exchange counts
echo strong-entities-scores; run --scaling strong --shard entities --exchange scores
echo strong-queries; run --scaling strong --shard queries
echo weak-nograph; run --no-graph
echo single; python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary | tail -1 | grep -o 'filtered_mrr[^,]*'
