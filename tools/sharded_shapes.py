#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""BASELINE cfg4 / cfg5 in their SHARDED form on a one-GPU box: P gloo ranks share GPU 0, each HOLDS 1/P of the
entity-table rows (distributed.shard_model_), and the four rank vectors of the sharded evaluation -- counts exchange
over the whole test split, score all-to-all over the first --score-facts facts (gloo moves the score bytes through
host memory: the full split's tiles would take minutes at Wikidata5M size) -- must equal the UNSHARDED single-GPU
ranks position by position.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
        tools/sharded_shapes.py --workload distmult_fb15k --out gpurun_out/sharded_distmult_fb15k.json

Exit code 1 on any mismatch.  (What a real 8-GPU node adds is RCCL instead of gloo and 8 devices instead of one.)
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='distmult_fb15k')
    ap.add_argument('--score-facts', type=int, default=0, help='facts of the score-exchange run (0: the whole split)')
    ap.add_argument('--batch', type=int, default=32768)
    ap.add_argument('--out', default=None)
    ap.add_argument('--weights', default=None, help="default: 'unit' at Wikidata5M size, else 'xavier'; 'trained' (r06) = the "
                    "bench's trained-like tables -- rank 0 trains (or reads KGE_BENCH_TABLE_CACHE), the others read its cache")
    ap.add_argument('--evals', type=int, default=2, help='evaluations of the counts exchange before the timed one (>= 3 lets the '
                                                          'level policy reach the one-product level and its captured segments)')
    args = ap.parse_args()
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    dist.init_process_group('gloo')
    import bench
    import torchkge_amd as tk
    from torchkge_amd import distributed as kd
    from torchkge_amd import evaluation as ev_mod

    kind, shape, d, p = bench.WORKLOADS[args.workload]
    weights = args.weights or ('unit' if shape == 'wikidata5m' else 'xavier')
    if weights == 'trained':        # one rank trains, the others read what it cached (same tables on every rank)
        os.environ.setdefault('KGE_BENCH_TABLE_CACHE', '/tmp/kge_cache')
        if rank == 0:
            model, tables, kg, kg_test, info = bench.build_workload(args.workload, dev, weights=weights)
        dist.barrier()
        if rank != 0:
            model, tables, kg, kg_test, info = bench.build_workload(args.workload, dev, weights=weights)
    else:
        model, tables, kg, kg_test, info = bench.build_workload(args.workload, dev, weights=weights)
    for x in ('head_idx', 'tail_idx', 'relations'):
        setattr(kg_test, x, getattr(kg_test, x).to(dev))
    n_test, n_ent = kg_test.n_facts, info['n_ent']
    names = ('rank_true_heads', 'rank_true_tails', 'filt_rank_true_heads', 'filt_rank_true_tails')

    def sync():
        torch.cuda.synchronize(dev)
        dist.barrier()

    # the unsharded single-GPU ranks (rank 0 evaluates, everybody gets them)
    want = torch.zeros(4, n_test, dtype=torch.int64)
    t_single = None
    if rank == 0:
        ev = tk.LinkPredictionEvaluator(model, kg_test, graph=False)
        ev.evaluate(args.batch, verbose=False)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        ev.evaluate(args.batch, verbose=False)
        torch.cuda.synchronize(dev)
        t_single = (time.perf_counter() - t0) * 1e3
        want = torch.stack([getattr(ev, nm) for nm in names]).clone()
        del ev
    dist.broadcast(want, src=0)
    full_bytes = model.entity_table_bytes()
    lo, hi = kd.shard_model_(model)
    torch.cuda.empty_cache()
    res = {'workload': args.workload, 'kind': kind, 'd': d, 'n_ent': n_ent, 'n_test': n_test, 'world': world,
           'backend': 'gloo (ranks share one MI355X)', 'weights': weights,
           'entity_table_bytes_full': full_bytes, 'entity_table_bytes_this_rank': model.entity_table_bytes(),
           'rows_this_rank': [lo, hi], 'single_gpu_ms_per_evaluate': None if t_single is None else round(t_single, 3)}
    ok = True

    def run(exchange, kgx, n_facts, graph, evals=1):
        evx = tk.LinkPredictionEvaluator(model, kgx, shard='entities', exchange=exchange, graph=graph)
        for _ in range(max(1, evals)):
            evx.evaluate(args.batch, verbose=False)
        sync()
        t0 = time.perf_counter()
        evx.evaluate(args.batch, verbose=False)
        sync()
        ms = (time.perf_counter() - t0) * 1e3
        got = torch.stack([getattr(evx, nm) for nm in names])
        diff = int((got != want[:, :n_facts]).sum())
        return {'exchange': exchange, 'facts': n_facts, 'ranks_compared': int(got.numel()),
                'ranks_differing_from_unsharded': diff, 'ms_per_evaluate_two_gloo_ranks_one_gpu': round(ms, 2),
                'hip_graph_segments': bool(graph), 'split_level': int(getattr(model, '_split_level', 0)),
                'rescored_pairs_per_query': evx.last_rescored_per_query, 'evaluations_before_the_timed_one': max(1, evals)}

    res['counts'] = run('counts', kg_test, n_test, graph=None, evals=args.evals)
    ok = ok and res['counts']['ranks_differing_from_unsharded'] == 0
    ns = n_test if args.score_facts <= 0 else min(n_test, args.score_facts)
    sub = kg_test
    if ns < n_test:
        sub = tk.KnowledgeGraph(kg={'heads': kg_test.head_idx[:ns].cpu(), 'tails': kg_test.tail_idx[:ns].cpu(),
                                    'relations': kg_test.relations[:ns].cpu()}, ent2ix=kg.ent2ix, rel2ix=kg.rel2ix,
                                _filter_src=kg._lazy)
        for x in ('head_idx', 'tail_idx', 'relations'):
            setattr(sub, x, getattr(sub, x).to(dev))
    res['scores'] = run('scores', sub, ns, graph=False)
    per = kd.shard_size(n_ent, world)
    m_max = max(1, min(-(-2 * min(ns, args.batch) // world), ev_mod.SCORE_TILE_BYTES // (4 * per * world)))
    res['scores'].update({'exchange_is': 'all-to-all of (P*m, N/P) fp32 row tiles, ranked in place from the rank-major tiles',
                          'rows_per_tile': m_max * world, 'tile_bytes': m_max * world * per * 4,
                          'tiles_per_batch': -(-2 * min(ns, args.batch) // (m_max * world))})
    ok = ok and res['scores']['ranks_differing_from_unsharded'] == 0
    res['ok'] = bool(ok)
    flag = torch.tensor([0 if ok else 1])
    dist.all_reduce(flag)
    if rank == 0:
        line = json.dumps(res)
        print(line, flush=True)
        if args.out:
            os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
            with open(args.out, 'w') as f:
                f.write(line + '\n')
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 0 else 1)


if __name__ == '__main__':
    main()
