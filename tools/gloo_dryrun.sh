#!/bin/bash
# logic dry run of the N > 1 bench paths on a one-GPU box: 2 ranks on GPU 0, gloo backend
run() {
  timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29578 bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --no-cpu-baseline --no-secondary "$@" > /tmp/dry.out 2> /tmp/dry.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "FAILED rc=$rc"; tail -15 /tmp/dry.err; fi
  tail -1 /tmp/dry.out | grep -o '^{"metric\|"value": [0-9.e+]*\|ms_per_step[^,]*\|filtered_mrr[^,]*\|"parallelism": "[^"]*"\|hip_graph[^,]*\|"layout": "[^"]*"\|bytes_this_rank[^,}]*\|"other_exchange": {[^}]*}' | paste -s -d' '
}
echo weak-sharded-tables; run
echo weak-sharded-tables-xavier; run --weights xavier
echo weak-replicated-tables; run --tables replicated --weights xavier
echo strong-entities-counts; run --scaling strong --shard entities --exchange counts --weights xavier
echo strong-entities-scores; run --scaling strong --shard entities --exchange scores --weights xavier --batch 2048
echo strong-queries; run --scaling strong --shard queries --weights xavier
echo weak-nograph; run --no-graph --weights xavier
echo complex-weak; run --workload complex_wn18rr --weights xavier
echo transh-strong; run --workload transh_fb15k237 --scaling strong --weights xavier
echo single; python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-full-parity --weights xavier | tail -1 | grep -o 'filtered_mrr[^,]*'
