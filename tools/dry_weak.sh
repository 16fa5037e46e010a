#!/bin/bash
# the default `bench.py --gpus 2` mode (strong scaling of the cfg2 job, row-sharded tables, score all-gather, trained weights) with two gloo ranks on GPU 0:
# the whole JSON line, field by field
mkdir -p gpurun_out/c2
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29578 bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --no-cpu-baseline --no-secondary "$@" > gpurun_out/c2/dry_weak.out 2> gpurun_out/c2/dry_weak.err
tail -1 gpurun_out/c2/dry_weak.out | python -c "
import sys, json
d = json.loads(sys.stdin.read())
for k, v in d.items():
    print(k, ':', json.dumps(v)[:600])
"
tail -5 gpurun_out/c2/dry_weak.err
