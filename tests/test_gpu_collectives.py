# -*- coding: utf-8 -*-
"""The multi-rank code paths on ONE GPU: a world of one rank with the collectives forced on
(KGE_FORCE_COLLECTIVES=1, RCCL backend) must reproduce the single-GPU metrics -- entity shards
exchanging counts (hipGraph segments with the all-reduces between them) and query shards."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra, forced, port):
    env = dict(os.environ)
    env.pop('KGE_FORCE_COLLECTIVES', None)
    if forced:
        env.update({'KGE_FORCE_COLLECTIVES': '1', 'RANK': '0', 'LOCAL_RANK': '0', 'WORLD_SIZE': '1',
                    'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'HSA_ENABLE_IPC_MODE_LEGACY': '0'})
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
           '--workload', 'complex_wn18rr', '--no-cpu-baseline', '--no-secondary', '--no-full-parity', '--no-traffic', '--no-weak',
           '--weights', 'xavier'] + extra     # (trained weights differ run to run: atomics in the backward)
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith('{')]
    assert lines and out.stdout.strip().splitlines()[-1] == lines[-1]      # the JSON line is the last line of stdout
    return json.loads(lines[-1])


def test_forced_collectives_reproduce_single_gpu_metrics():
    ref = _bench([], False, 0)
    assert ref['config']['parallelism'] == 'single'
    for k, extra in enumerate((['--scaling', 'strong', '--shard', 'entities', '--exchange', 'counts'],
                               ['--scaling', 'strong', '--shard', 'queries'])):
        got = _bench(extra, True, 29731 + k)
        assert got['config']['parallelism'] != 'single' and got['config']['hip_graph'] is True
        assert got['filtered_mrr'] == ref['filtered_mrr']
        assert got['filtered_hits_at_10'] == ref['filtered_hits_at_10']
        if k == 0:      # row-sharded entity tables (a world of one holds the only shard) + both exchanges measured
            assert got['entity_tables']['layout'].startswith('row-sharded')
            assert got['entity_tables']['bytes_this_rank'] == got['entity_tables']['bytes_full']
            assert got['other_exchange']['exchange'] == 'scores' and got['other_exchange']['ranks_identical_to_headline_run']


def test_c_abi_collectives_world_of_one():
    """include/kge_hip_coll.h through ctypes on a communicator of ONE rank (what a one-GPU box can run): the score
    all-gather + re-layout reproduces the local tile (also with a padded last shard), the count / float all-reduces
    are the identity; the gathered scores rank exactly like the local ones (kge_filtered_rank_from_scores)."""
    import torch
    from torchkge_amd import _hip, _hip_coll
    _hip.load_library()
    comm = _hip_coll.Comm(1, 0, _hip_coll.unique_id())
    try:
        g = torch.Generator(device='cuda').manual_seed(3)
        B, N = 37, 1001
        local = torch.randn(B, N, device='cuda', generator=g)
        full = comm.allgather_scores(local, N)
        assert torch.equal(full, local)
        padded = torch.zeros(B, N + 7, device='cuda')
        padded[:, :N] = local
        assert torch.equal(comm.allgather_scores(padded, N), local)      # n_per > N: the caller's zero padding is dropped
        counts = torch.randint(0, 1000, (3, B), device='cuda', dtype=torch.int32)
        ref = counts.clone()
        assert torch.equal(comm.allreduce_counts(counts), ref)
        x = torch.randn(B, device='cuda', generator=g)
        xr = x.clone()
        assert torch.equal(comm.allreduce_sum(x), xr)
        true = torch.randint(0, N, (B,), device='cuda', generator=g)
        assert torch.equal(_hip.get_rank(full, true), _hip.get_rank(local, true))
        # the score all-to-all on one rank = the own block's device copy; the int64 rank all-reduce = identity
        tiles = comm.alltoall_scores(local)
        assert tiles.shape == (1, B, N) and torch.equal(tiles[0], local)
        untouched = torch.full((1, B, N), 7.0, device='cuda')
        comm.alltoall_scores(local, untouched, recv_own=False)      # own block left in `local`: nothing is written
        assert bool((untouched == 7.0).all())
        rk = torch.randint(0, 10 ** 12, (4, B), device='cuda', dtype=torch.int64)
        rk0 = rk.clone()
        assert torch.equal(comm.allreduce_ranks(rk), rk0)
        torch.cuda.synchronize()
    finally:
        comm.close()


WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
rank, world, port, kind, out_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = port
torch.cuda.set_device(0)
dist.init_process_group('gloo', rank=rank, world_size=world)      # 2 ranks share the one GPU: gloo moves the bytes
import torchkge_amd as tk
from torchkge_amd import distributed as kd
from oracle import kge_oracle as orc
from tests.test_gpu_parity import build_model
n_ent, n_rel, d = 3001, 11, 64
p = 1 if kind == 'transe_l1' else 2
k = 'transe' if kind == 'transe_l1' else kind
tables = orc.init_tables(k, n_ent, n_rel, d, seed=3, d_rel=(48 if k == 'transd' else None))
m = build_model(k, p, tables, n_ent, n_rel)
h, t, r = orc.synthetic_triples_zipf(n_ent, n_rel, 20000, 9, hubs=((900, 'head'), (300, 'tail')))
kg = tk.KnowledgeGraph(kg={'heads': h, 'tails': t, 'relations': r}, ent2ix={i: i for i in range(n_ent)},
                       rel2ix={i: i for i in range(n_rel)})
_, kg_test = kg.split_kg(sizes=(19000, 1000))
ref = tk.LinkPredictionEvaluator(m, kg_test, graph=False)
ref.evaluate(b_size=256, verbose=False)                            # the unsharded single-GPU ranks
want = [ref.rank_true_heads, ref.rank_true_tails, ref.filt_rank_true_heads, ref.filt_rank_true_tails]
want_inf = {}
if kind in ('transe', 'transh', 'complex'):
    dh_, dt_, _ = orc.build_filter_dicts(h, t, r)
    for missing, dic in (('tails', dt_), ('heads', None)):
        a = tk.EntityInference(m, h[-300:], r[-300:], top_k=7, missing=missing, dictionary=dic)
        a.evaluate(b_size=128, verbose=False)
        want_inf[missing] = (a.predictions, a.scores)
full = m.entity_table_bytes()
lo, hi = kd.shard_model_(m)
assert m.entity_table_bytes() <= full // world + 4 * 2 * d * 2 * world
ok = True
# (this worker runs the SHIPPED coalescing default: b_size 256 -> one internal batch; co = 0 takes b_size literally
#  and drives the multi-batch sharded flow: 4 batches, short last one)
for exchange, graph, qx, co in (('counts', False, 'evaluate', None), ('counts', True, 'evaluate', None),
                                ('counts', False, 'batch', None), ('counts', True, 'batch', None),
                                ('scores', False, 'evaluate', None), ('counts', True, 'evaluate', 0),
                                ('counts', False, 'batch', 0),
                                # the score ALL-TO-ALL (r04): as hipGraph segments, multi-batch, and cut into several row
                                # tiles per batch ('tiles': 64 rows x P ranks per all-to-all, the last tile short)
                                ('scores', True, 'evaluate', None), ('scores', False, 'batch', 0),
                                ('scores', True, 'evaluate', 'tiles')):
    import torchkge_amd.evaluation as ev_mod
    ev_mod.SCORE_TILE_BYTES = (4 * kd.shard_size(n_ent, world) * world * 64) if co == 'tiles' else (256 << 20)
    co = None if co == 'tiles' else co
    ev = tk.LinkPredictionEvaluator(m, kg_test, shard='entities', exchange=exchange, graph=graph, query_exchange=qx,
                                    coalesce=co)
    for _ in range(2):
        ev.evaluate(b_size=256, verbose=False)
    got = [ev.rank_true_heads, ev.rank_true_tails, ev.filt_rank_true_heads, ev.filt_rank_true_tails]
    for a, b in zip(want, got):
        if not torch.equal(a, b):
            ok = False
            print('MISMATCH', rank, kind, exchange, graph, qx, int((a != b).sum()), flush=True)
# (r05) the ONE-PRODUCT level on entity shards: forced (split_level = 1) and through the policy ('auto': the re-scored pair
# count rides the counts all-reduce, so every rank switches together) -- eager and as hipGraph segments
if kind != 'transe_l1':
    for lvl, graph in ((1, False), (1, True), ('auto', None)):
        m.split_level = lvl
        ev = tk.LinkPredictionEvaluator(m, kg_test, shard='entities', exchange='counts', graph=graph)
        lv_seen = []
        for _ in range(4):
            ev.evaluate(b_size=256, verbose=False)
            lv_seen.append(int(m._split_level))
            got = [ev.rank_true_heads, ev.rank_true_tails, ev.filt_rank_true_heads, ev.filt_rank_true_tails]
            for a, b in zip(want, got):
                if not torch.equal(a, b):
                    ok = False
                    print('MISMATCH (level %%s)' %% lvl, rank, kind, graph, int((a != b).sum()), flush=True)
        if lvl == 1 and lv_seen != [1, 1, 1, 1]:
            ok = False
            print('LEVEL NOT TAKEN', rank, kind, lv_seen, flush=True)
        if ev.last_rescored_per_query is None or ev.last_rescored_per_query <= 0:
            ok = False
            print('NO RE-SCORED COUNT ON SHARDS', rank, kind, lvl, ev.last_rescored_per_query, flush=True)
    m.split_level = 'auto'
# top-k inference on the row-sharded model (per-shard tiles -> partial lists -> all-gather -> merge) == unsharded
if kind in ('transe', 'transh', 'complex'):
    dh_, dt_, _ = orc.build_filter_dicts(h, t, r)
    qe, qr = h[-300:], r[-300:]
    for missing, dic in (('tails', dt_), ('heads', None)):
        a = tk.EntityInference(m, qe, qr, top_k=7, missing=missing, dictionary=dic, tile=512)
        a.evaluate(b_size=128, verbose=False)
        if not (torch.equal(a.predictions, want_inf[missing][0]) and torch.equal(a.scores, want_inf[missing][1])):
            ok = False
            print('INFERENCE MISMATCH', rank, kind, missing, flush=True)
dist.barrier()
dist.destroy_process_group()
open(out_path, 'w').write('ok' if ok else 'bad')
sys.exit(0 if ok else 1)
'''


@pytest.mark.parametrize('kind,world', [('transe', 2), ('transe_l1', 2), ('transh', 2), ('transd', 2), ('distmult', 2),
                                        ('complex', 2), ('transe', 3), ('complex', 3)])
def test_row_sharded_entity_tables_two_ranks_on_one_gpu(kind, world, tmp_path):
    """Two ranks (gloo) sharing the one GPU: each keeps HALF of every entity-indexed table
    (distributed.shard_model_), query rows are built by the owner rank (kge_lp_prep_sharded) and
    summed over the ranks, each rank counts its own candidates -- the four rank vectors equal the
    unsharded single-GPU ones position by position, for both exchanges and the hipGraph segments.
    (world = 3: uneven shards, the replica all-gather with padded blocks.)"""
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % {'root': ROOT})
    port = str(29900 + (os.getpid() % 50) * 13 + ['transe', 'transe_l1', 'transh', 'transd', 'distmult', 'complex'].index(kind)
               + 6 * (world - 2))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('KGE_FORCE_COLLECTIVES', None)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), port, kind, str(tmp_path / ('r%d' % r))],
                              env=env, cwd=ROOT) for r in range(world)]
    codes = [p.wait(timeout=600) for p in procs]
    assert codes == [0] * world


@pytest.mark.parametrize('world,N,B', [(1, 777, 9), (2, 1001, 33), (3, 1000, 64), (8, 14541, 50), (8, 4097, 7)])
def test_rank_from_rank_major_tiles_equals_rank_from_the_score_matrix(world, N, B):
    """kge_filtered_rank_from_tiles on P VIRTUAL shards of one device: the (2B, N) score matrix cut into the rank-major
    tiles the score all-to-all delivers ((P, m, per), short last shard, garbage in the padding columns / rows) ranks
    exactly like kge_filtered_rank_from_scores on the matrix itself -- raw and filtered, exact ties, -inf scores,
    filter lists with the true entity present / absent / empty, ranks written through off / pos."""
    import torch
    from torchkge_amd import _hip
    _hip.load_library()
    dev = torch.device('cuda')
    g = torch.Generator(device='cuda').manual_seed(world * 1000 + N)
    n2 = 2 * B
    scores = torch.randn(n2, N, device=dev, generator=g)
    scores = (scores * 4).round() / 4                      # plenty of exact ties
    scores[torch.rand(n2, N, device=dev, generator=g) < 0.01] = float('-inf')
    true = torch.randint(0, N, (n2,), device=dev, generator=g)
    # filter segments: lengths 0 .. 40, the true entity planted in two thirds of the non-empty ones
    lens = torch.randint(0, 41, (n2,), device=dev, generator=g)
    seg_hi = torch.cumsum(lens, 0)
    seg_lo = seg_hi - lens
    targets = torch.randint(0, N, (int(seg_hi[-1]) + 1,), device=dev, generator=g).int()
    for i in range(n2):
        if int(lens[i]) > 0 and i % 3 != 0:
            targets[int(seg_lo[i]) + (i % int(lens[i]))] = int(true[i])
    rk, frk = _hip.filtered_rank_from_scores(scores, true, seg_lo, seg_hi, targets)
    want = torch.zeros(4, B + 5, dtype=torch.int64, device=dev)
    want[1, 3:3 + B], want[3, 3:3 + B], want[0, 3:3 + B], want[2, 3:3 + B] = rk[:B], frk[:B], rk[B:], frk[B:]
    per = -(-N // world)
    m = -(-n2 // world)
    got = torch.zeros(4, B + 5, dtype=torch.int64, device=dev)
    padded = torch.full((world * m, world * per), float('nan'), device=dev)
    padded[:n2, :N] = scores
    for j in range(world):          # "rank j": the tiles it would receive for its m rows
        tiles = padded[j * m:(j + 1) * m].view(m, world, per).permute(1, 0, 2).contiguous()
        my0 = j * m
        rows = max(0, min(m, n2 - my0))
        if rows > 0:
            if j % 2 == 0:
                _hip.filtered_rank_from_tiles(tiles, N, true[my0:], seg_lo[my0:], seg_hi[my0:], targets, rows, my0, B, got, 3)
            else:       # the own block read in place (the exchange did not copy it: garbage in its slot of the tiles)
                own = tiles[j].clone()
                tiles[j] = float('nan')
                _hip.filtered_rank_from_tiles(tiles, N, true[my0:], seg_lo[my0:], seg_hi[my0:], targets, rows, my0, B, got, 3,
                                              None, own, j)
    assert torch.equal(got, want)
    # through pos (facts processed in another order)
    pos = torch.randperm(B + 5, device=dev, generator=g)
    got2 = torch.zeros(4, B + 5, dtype=torch.int64, device=dev)
    tiles = padded[:m].view(m, world, per).permute(1, 0, 2).contiguous()
    rows = min(m, n2)
    _hip.filtered_rank_from_tiles(tiles, N, true, seg_lo, seg_hi, targets, rows, 0, B, got2, 3, pos)
    for q in range(rows):
        f = int(pos[3 + (q if q < B else q - B)])
        assert int(got2[1 if q < B else 0, f]) == int(rk[q]) and int(got2[3 if q < B else 2, f]) == int(frk[q])
