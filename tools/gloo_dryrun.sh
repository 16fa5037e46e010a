#!/bin/bash
# logic dry run of the N > 1 bench paths on a one-GPU box: 2 ranks on GPU 0, gloo backend
run() {
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29578 bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --no-cpu-baseline --no-secondary "$@" > /tmp/dry.out 2> /tmp/dry.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "FAILED rc=$rc"; tail -15 /tmp/dry.err; fi
  tail -1 /tmp/dry.out | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); c=d.get('cfg4_mode') or {}
    print('   cfg4_mode:', {k:c.get(k) for k in ('ms_per_step','value','split_level','filtered_hits_at_10')}, '| split level of the headline run:', (d.get('roofline') or {}).get('split_level'))
    print('   strong_scaling_model (cfg2):', (d.get('strong_scaling_model') or {}).get('modelled_speedup_vs_1gpu'), '(cfg4):', ((c.get('strong_scaling_model') or {}).get('modelled_speedup_vs_1gpu')))
except Exception as e: print('   (no json)', e)"
  tail -1 /tmp/dry.out | grep -o '^{"metric\|"value": [0-9.e+]*\|ms_per_step[^,]*\|"scaling": "[^"]*"\|filtered_mrr[^,]*\|"workload": "[^"]*"\|"parallelism": "[^"]*"\|hip_graph[^,]*\|"layout": "[^"]*"\|bytes_this_rank[^,}]*\|"collective_time": {[^}]*}\|"other_exchange": {[^}]*}[^}]*}\|"weak_mode": {[^}]*}[^}]*}' | paste -s -d' '
}
echo "default (strong scaling of the cfg2 job, row-sharded tables, counts all-reduce; score all-to-all + weak mode beside it)"; run
echo default-xavier; run --weights xavier
echo strong-entities-counts; run --exchange counts --weights xavier --no-weak
echo strong-replicated-tables; run --tables replicated --weights xavier --exchange counts
echo strong-entities-scores-small-batch; run --exchange scores --weights xavier --batch 2048 --no-weak
echo strong-queries; run --shard queries --weights xavier
echo weak-sharded-tables; run --scaling weak --exchange counts --weights xavier
echo counts-nograph; run --exchange counts --no-graph --weights xavier --no-weak
echo counts-eager-collectives-env; KGE_EAGER_COLLECTIVES=1 run --exchange counts --weights xavier --no-weak
echo complex; run --workload complex_wn18rr --weights xavier --no-weak
echo transh; run --workload transh_fb15k237 --weights xavier --no-weak
echo single; python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-full-parity --no-traffic --weights xavier | tail -1 | grep -o 'filtered_mrr[^,]*'
