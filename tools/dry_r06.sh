#!/bin/bash
# (r06) the N > 1 paths on the one-GPU box the builder has: PLAIN `python bench.py --gpus 2` invocations (bench.py launches its
# own ranks since r06), two gloo ranks sharing GPU 0; then one RCCL rank with the collectives forced; then BASELINE cfg5
# (ComplEx d = 512, 4.59 M entities, trained-like) in its SHARDED form on the new chunked-panel kernel.
cd ${GRAFT_REPO_ROOT:-.}
export KGE_BENCH_TABLE_CACHE=/tmp/kge_cache
show() { python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read())
except Exception as e:
    print('   (no json)', e); sys.exit(0)
sp=d.get('split_prefilter') or {}
print('   n_gpus', d.get('n_gpus'), '| ms_per_step', d.get('ms_per_step'), '| value %.3e' % d.get('value'), '| split level', sp.get('level_of_the_timed_evaluations'), '| re-scored pairs per query', sp.get('rescored_pairs_per_query'))
print('   parallelism:', (d.get('config') or {}).get('parallelism'), '| filtered Hits@10', d.get('filtered_hits_at_10'))
print('   collective_time:', json.dumps(d.get('collective_time'))[:260])
q=d.get('query_partition'); print('   query_partition:', json.dumps(q)[:400])
o=d.get('other_exchange') or {}; print('   other_exchange:', {k:o.get(k) for k in ('exchange','ms_per_step','ranks_identical_to_headline_run')})
w=d.get('weak_mode') or {}; print('   weak_mode:', {k:w.get(k) for k in ('n_ent','ms_per_step','value')})
c=d.get('cfg4_mode') or {}; print('   cfg4_mode:', {k:c.get(k) for k in ('ms_per_step','value','split_level','filtered_hits_at_10')})
m=d.get('strong_scaling_model') or {}; print('   strong_scaling_model: entity shards', m.get('modelled_speedup_vs_1gpu'), '| query partition', (m.get('query_partition') or {}).get('modelled_speedup_vs_1gpu'))
print('   collectives_fallback:', (d.get('collectives_fallback') or '')[:80])
"; }
echo "== python bench.py --gpus 2 --backend gloo   (default: strong scaling of the cfg2 job over entity shards, counts all-reduce; score all-to-all, query partition, weak mode and cfg4 beside it)"
timeout 900 python bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/tmp/dry.err | tail -1 | show
echo "== python bench.py --gpus 2 --backend gloo --shard queries"
timeout 600 python bench.py --gpus 2 --backend gloo --shard queries --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/tmp/dry.err | tail -1 | show
echo "== KGE_EAGER_COLLECTIVES=1 python bench.py --gpus 2 --backend gloo --no-weak"
KGE_EAGER_COLLECTIVES=1 timeout 600 python bench.py --gpus 2 --backend gloo --no-weak --steps 3 --warmup 1 --no-cpu-baseline --no-secondary 2>/tmp/dry.err | tail -1 | show
echo "== one RCCL rank, collectives forced (KGE_FORCE_COLLECTIVES=1, torch.distributed.run --nproc-per-node 1)"
HSA_ENABLE_IPC_MODE_LEGACY=0 KGE_FORCE_COLLECTIVES=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-weak --no-cpu-baseline --no-secondary 2>/tmp/dry.err | tail -1 | show
echo "== BASELINE cfg5 sharded: two gloo ranks on one GPU, each HOLDING half of the 4.59 M entity rows, trained-like tables"
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 tools/sharded_shapes.py --workload complex_wikidata5m --weights trained --evals 3 --score-facts 64 --batch 8192 --out gpurun_out/r06/sharded_complex_wikidata5m.json 2>/tmp/dry.err | tail -1 | cut -c1-1800
