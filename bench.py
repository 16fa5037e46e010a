#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""Benchmark of the link-prediction hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one full ``LinkPredictionEvaluator.evaluate`` over the test split of
the workload (default: BASELINE.json configs[1] = TransE dim=200 L2 on an
FB15k-237-shaped synthetic KG: 14,541 entities, 237 relations, 20,466 test
triples => 20,466 x 2 sides x 14,541 = 5.95e8 scored triples per step), with
the model tables, the test triples and the filter index already resident in HBM.
Metric = link-prediction triples scored / second (whole job, all ranks).

N > 1 (launched by torch.distributed.run, one rank per GPU over RCCL):
  --scaling strong --shard entities  (default) the ONE dataset-sized job BASELINE.json names with the entity
                   tables row-sharded: every rank scores ITS N/P candidates for all test triples and the ranks meet
                   in one all-reduce of the (3, 2B) partial rank counts (--exchange counts, the headline: the
                   evaluator's own default) -- or in the exchange north_star names, the partial score tiles, as an
                   all-to-all in which every rank receives (and ranks) only the rows of its 2B/P queries (--exchange
                   scores; timed beside the headline: it needs the exact fp32 score GEMM and 595 MB per link at N = 2).
  --scaling weak --shard entities  the entity table grows to N dataset-sized shards (also reported beside a
                   strong run: weak_mode).
  --scaling weak --shard queries   independent replicas: every rank evaluates its own
                   dataset-sized test split against its replica of the tables (no collective).
  --scaling strong --shard queries   the test facts split across ranks, tables replicated.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra
objects: "roofline" (dominant kernel timed live with HIP events on the launch
stream) and "cpu_baseline" (the oracle = reference CPU algorithm, timed on the
host cores of this box on a bounded sample; N=1 only), plus "parity_full_split":
every rank of the WHOLE test split against the reference algorithm run on ATen
GPU ops (oracle.lp_evaluate(device=cuda)).

Workload realism: the synthetic KG is Zipf-skewed with planted hub
keys (filter lists of thousands of entities, --kg uniform gives the r01 graph), and
the model is "trained-like" by default (--weights trained: a few hundred steps of the
engine's own training step -- Bernoulli negatives, margin loss, Adam -- on the full
synthetic graph before the timed region, so true ranks are small and near-ties
dense; --weights xavier = the constructor's initialisation).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (kind, dataset shape, dim, p)
    'transe_fb15k237': ('transe', 'fb15k237', 200, 2),
    'complex_wn18rr': ('complex', 'wn18rr', 200, 2),
    'distmult_fb15k': ('distmult', 'fb15k', 400, 2),
    'transe_nations': ('transe', 'nations', 50, 2),
    'transh_fb15k237': ('transh', 'fb15k237', 200, 2),
    'transd_fb15k237': ('transd', 'fb15k237', 200, 2),
    'transe_l1_fb15k237': ('transe', 'fb15k237', 200, 1),
    'complex_wikidata5m': ('complex', 'wikidata5m', 512, 2),    # cfg5: 18.8 GB of tables; use --no-secondary
}
PEAK_FP32_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 MFMA = fp32 vector peak
PEAK_F16_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense f16/bf16 MFMA (~2.5 PF, no sparsity)
PEAK_HBM_GBS = 8000.0
PEAK_SAD_TINST = 39.3        # v_sad_u16 (half-rate VALU op): 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz lane-instructions / s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--settle-ms', type=float, default=250.0,
                    help='untimed evaluate() replays before the warm-up steps so that the clocks reach their sustained '
                         '(power-capped) state; 0 = none')
    ap.add_argument('--workload', default='transe_fb15k237', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=32768, help='evaluate() b_size')
    ap.add_argument('--scaling', default='strong', choices=['weak', 'strong'],
                    help='N>1: strong (default) = the ONE dataset-sized job BASELINE.json names, split across the ranks; '
                         'weak = the entity table grows to N dataset-sized shards (also reported beside a strong run: weak_mode)')
    ap.add_argument('--shard', default='entities', choices=['entities', 'queries'],
                    help='N>1: what is partitioned across ranks (weak+queries = independent replicas)')
    ap.add_argument('--exchange', default=None, choices=['counts', 'scores'],
                    help="entity shards: what the ranks exchange.  Default: 'counts' = one all-reduce of 3 x 2B int32 per batch "
                         "behind the f16 prefilter (the evaluator's own default).  'scores' = the partial (2B, N/P) score tiles "
                         "north_star names, exchanged as an RCCL all-to-all of row blocks (every rank receives and ranks only "
                         "its 2B/P queries; bit-identical ranks).  The other exchange is timed beside the headline at the same "
                         "b_size (other_exchange)")
    ap.add_argument('--no-weak', action='store_true', help='N>1 strong run: skip the secondary weak-scaling measurement')
    ap.add_argument('--tables', default='sharded', choices=['sharded', 'replicated'],
                    help='entity shards: each rank HOLDS only its rows of the entity tables (default) or a full replica')
    ap.add_argument('--materialize', action='store_true', help='fused=False: write the (B,N) scores')
    ap.add_argument('--l2-mode', default='auto', choices=['auto', 'expand', 'direct'])
    ap.add_argument('--no-split', action='store_true',
                    help='rank counts on the fp32 MFMA kernel only (no f16-split prefilter)')
    ap.add_argument('--split-level', default='auto', choices=['auto', '0', '1'],
                    help="level of the f16-split prefilter: 'auto' (the evaluator's policy: one product per k16 unit when the "
                         "previous evaluation re-scored few pairs per query, else three), or forced")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the scoring_function / sampler / train-step timings')
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of one hipGraph per evaluate()')
    ap.add_argument('--graph-collectives', action='store_true',
                    help='entity shards: capture the RCCL collectives inside the one hipGraph of evaluate() (opt-in)')
    ap.add_argument('--overlap', action='store_true', help='two-stream overlap of the short kernels (default: single stream)')
    ap.add_argument('--no-both', action='store_true', help='rank the two sides of a batch one after the other '
                                                          '(default: one 2B-query problem per batch)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help="torch.distributed backend; 'gloo' only to dry-run the N>1 logic on one GPU")
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='(unused; kept for compatibility)')
    ap.add_argument('--kg', default='zipf', choices=['zipf', 'uniform'],
                    help='synthetic KG: Zipf-skewed entities/relations + planted hub keys (default), or uniform draws')
    ap.add_argument('--weights', default='trained', choices=['trained', 'xavier', 'unit'],
                    help="model weights: 'trained' = a few hundred engine training steps before the timed region; "
                         "'unit' = uniform(-0.5, 0.5) tables (scores of unit scale like a fitted model's; what the "
                         "Wikidata5M-size workload uses instead of 'trained': dense Adam state would triple 18.8 GB)")
    ap.add_argument('--parity-sample', type=int, default=512,
                    help='workloads too large for the full-split comparison (cfg5): this many test facts through the '
                         'reference algorithm on ATen GPU ops at b_size 2 (SURVEY 8d says <= 64; r06: 512, which on the Zipf graph '
                         'includes the hub keys -- filter lists of 10^5 entities)')
    ap.add_argument('--train-steps', type=int, default=None)
    ap.add_argument('--only-timed', action='store_true',
                    help='profiling aid: nothing but the warm-up and the timed loop (no roofline / f32 / cpu / parity legs)')
    ap.add_argument('--no-traffic', action='store_true',
                    help='skip the two rocprofv3 --pmc child passes that measure the dominant kernel\'s HBM-side bytes')
    ap.add_argument('--no-full-parity', action='store_true',
                    help='skip the full-test-split comparison against the GPU-resident reference algorithm')
    ap.add_argument('--launch-check', action='store_true',
                    help='only prove the N-rank launch: rendezvous on 127.0.0.1, one all-reduce over the ranks, rank 0 prints '
                         'a JSON line -- no GPU is touched (CPU test of the self-launch path, --backend gloo)')
    return ap.parse_args()


def make_model(kind, p, tables, n_ent, n_rel):
    import torchkge_amd as tk
    d = tables[0].shape[1]
    if kind == 'transe':
        m, names = tk.TransEModel(d, n_ent, n_rel, 'L%d' % p), ['ent_emb', 'rel_emb']
    elif kind == 'distmult':
        m, names = tk.DistMultModel(d, n_ent, n_rel), ['ent_emb', 'rel_emb']
    elif kind == 'complex':
        m, names = tk.ComplExModel(d, n_ent, n_rel), ['re_ent_emb', 'im_ent_emb', 're_rel_emb', 'im_rel_emb']
    elif kind == 'transh':
        m, names = tk.TransHModel(d, n_ent, n_rel), ['ent_emb', 'rel_emb', 'norm_vect']
    elif kind == 'transd':
        m, names = tk.TransDModel(d, tables[1].shape[1], n_ent, n_rel), ['ent_emb', 'rel_emb', 'ent_proj_vect',
                                                                          'rel_proj_vect']
    else:
        raise ValueError(kind)
    m.load_state_dict({n + '.weight': t for n, t in zip(names, tables)})
    return m


def make_triples(orc, shape, n_ent, n_rel, seed, kg_kind='zipf'):
    """All facts (train + valid + test) of a dataset-shaped synthetic KG, the test split last."""
    _, _, n_train, n_valid, n_test = orc.DATASET_SHAPES[shape]
    n = n_train + n_valid + n_test
    if kg_kind == 'zipf':
        return orc.synthetic_triples_zipf(n_ent, n_rel, n, seed)
    return orc.synthetic_triples(n_ent, n_rel, n, seed)


TRAIN_DEFAULTS = {'steps': 500, 'batch': 32768, 'lr': 1e-2, 'margin': 0.5}
# cfg5 (ComplEx d = 512 on 4.59 M entities, 20.6 M facts): the same training step at a batch of 2 M facts -- a dense Adam step
# costs the same whatever the batch (18.8 GB of tables + gradients + two moments = 75 GB of the 288), so the ~30 epochs a
# fitted score distribution takes are ~300 steps instead of 19,000
TRAIN_DEFAULTS_BIG = {'steps': 300, 'batch': 1 << 21, 'lr': 1e-2, 'margin': 0.5}


def train_like(model, kg, steps=None, batch=None, lr=None, margin=None, seed=0):
    """"Trained-like" weights: `steps` steps of the engine's own training path on the FULL graph
    `kg` (corrupt_batch -> Model.forward -> MarginLoss -> backward -> Adam), entity tables
    re-normalised every epoch and at the end as utils/training.py:186-188 does.  The synthetic graph has
    no held-out structure, so the test facts are trained on too: what is wanted is the score
    DISTRIBUTION of a fitted model (small true ranks, clustered embeddings, dense near-ties)."""
    import torchkge_amd as tk
    cfg = dict(TRAIN_DEFAULTS_BIG if model.n_ent > 2000000 else TRAIN_DEFAULTS)
    for k, v in (('steps', steps), ('batch', batch), ('lr', lr), ('margin', margin)):
        if v is not None:
            cfg[k] = v
    dev = next(model.parameters()).device
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)
    h, t, r = kg.head_idx.to(dev), kg.tail_idx.to(dev), kg.relations.to(dev)
    n = h.shape[0]
    B = min(cfg['batch'], n)
    samp = tk.BernoulliNegativeSampler(kg)
    crit = tk.MarginLoss(cfg['margin'])
    opt = torch.optim.Adam(model.parameters(), lr=cfg['lr'])
    per_epoch = max(1, n // B)
    for s in range(cfg['steps']):
        idx = torch.randint(0, n, (B,), device=dev)
        hh, tt, rr = h[idx], t[idx], r[idx]
        nh, nt = samp.corrupt_batch(hh, tt, rr)
        pos, neg = model(hh, tt, rr, nh, nt)
        loss = crit(pos, neg)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        if (s + 1) % per_epoch == 0:
            model.normalize_parameters()
    model.normalize_parameters()
    for prm in model.parameters():
        prm.grad = None
    del opt
    torch.cuda.synchronize(dev)
    torch.cuda.empty_cache()        # (cfg5: 56 GB of gradients and Adam moments go back before the evaluation allocates)
    return model


def build_workload(name, device, weights='trained', kg_kind='zipf', n_ent_mult=1, train_cfg=None):
    """(model on `device`, CPU tables for the oracle, full KnowledgeGraph, test split, info dict)."""
    import torchkge_amd as tk
    from oracle import kge_oracle as orc
    kind, shape, d, p = WORKLOADS[name]
    n_ent1, n_rel, n_train, n_valid, n_test = orc.DATASET_SHAPES[shape]
    n_ent = n_ent1 * n_ent_mult
    if n_ent > 2000000 and kind == 'complex':
        # Wikidata5M scale: let the constructor draw the 2 x 9.4 GB tables once (same distribution)
        torch.manual_seed(0)
        model = tk.ComplExModel(d, n_ent, n_rel).to(device)
        if weights != 'xavier':
            unit_scale_(model)
        tables = None
    else:
        tables = orc.init_tables(kind, n_ent, n_rel, d, seed=0)
        model = make_model(kind, p, tables, n_ent, n_rel).to(device)
    cfg_seed = 1000 + sorted(WORKLOADS).index(name)
    heads, tails, rels = make_triples(orc, shape, n_ent, n_rel, cfg_seed, kg_kind)
    ident_e = {i: i for i in range(n_ent)}
    ident_r = {i: i for i in range(n_rel)}
    kg = tk.KnowledgeGraph(kg={'heads': heads, 'tails': tails, 'relations': rels}, ent2ix=ident_e, rel2ix=ident_r)
    _, _, kg_test = kg.split_kg(sizes=(n_train, n_valid, n_test))
    info = {'kind': kind, 'shape': shape, 'd': d, 'p': p, 'n_ent': n_ent, 'n_rel': n_rel, 'n_test': n_test,
            'kg_kind': kg_kind, 'weights': weights}
    if weights == 'trained':
        t0 = time.perf_counter()
        # KGE_BENCH_TABLE_CACHE=<dir>: the trained tables of a workload are written there once and re-read by later runs of
        # the same call (the rocprofv3 child passes of a Wikidata5M-sized workload would otherwise each train for a minute)
        cache = os.environ.get('KGE_BENCH_TABLE_CACHE')
        cpath = None
        if cache:
            tc = dict(TRAIN_DEFAULTS_BIG if n_ent > 2000000 else TRAIN_DEFAULTS, **(train_cfg or {}))
            cpath = os.path.join(cache, '%s_%s_x%d_%s.pt' % (name, kg_kind, n_ent_mult, '_'.join('%s%g' % kv for kv in sorted(tc.items()))))
        if cpath and os.path.exists(cpath):
            sd = torch.load(cpath, map_location=device)
            with torch.no_grad():
                for k_, prm in model.state_dict().items():
                    prm.copy_(sd[k_])
            del sd
            info['train_cache'] = 'loaded'
        else:
            train_like(model, kg, **(train_cfg or {}))
            if cpath:
                os.makedirs(cache, exist_ok=True)
                torch.save(model.state_dict(), cpath)
                info['train_cache'] = 'written'
        info['train_s'] = round(time.perf_counter() - t0, 2)
        info['train'] = dict(TRAIN_DEFAULTS_BIG if n_ent > 2000000 else TRAIN_DEFAULTS, **(train_cfg or {}))
        if tables is not None:
            tables = [x.detach().cpu().clone() for x in model._tables()]
    return model, tables, kg, kg_test, info


def unit_scale_(model, seed=0):
    """Tables ~ uniform(-0.5, 0.5): ComplEx d = 512 scores then have unit scale (std of a score = sqrt(4 d) sigma^3
    = 1.1), the scale of a fitted model's -- Xavier at N = 4.6 M gives entries ~1e-3 and scores ~1e-8, where an
    absolute score tolerance says nothing.  The split prefilter's band is relative: timing does not depend on the scale."""
    g = torch.Generator(device=next(model.parameters()).device).manual_seed(seed)
    for prm in model.parameters():
        prm.data.uniform_(-0.5, 0.5, generator=g)
    return model


def build_cfg5_sample(device, n_facts=5000000, n_test=64, seed=1005, hub_tests=32):
    """cfg5's MODEL (ComplEx d = 512 on 4,594,485 entities / 822 relations, unit-scale tables) with a 5 M-fact Zipf graph
    over those entities (planted hub keys with thousands of known heads / tails): what tests/test_gpu_fullsplit.py
    compares with the reference algorithm at b_size = 2.  Half of the `n_test` test facts are taken FROM the hub keys, so
    the filtered ranks of the sample walk filter lists of thousands of entities at N = 4.6 M (r03's 400 k-fact graph
    left them almost empty)."""
    import torchkge_amd as tk
    from oracle import kge_oracle as orc
    n_ent, n_rel = orc.DATASET_SHAPES['wikidata5m'][:2]
    torch.manual_seed(0)
    model = unit_scale_(tk.ComplExModel(512, n_ent, n_rel).to(device))
    hubs = ((6000, 'head'), (3000, 'head'), (1500, 'tail'), (800, 'tail'))
    heads, tails, rels = orc.synthetic_triples_zipf(n_ent, n_rel, n_facts, seed, hubs=hubs)
    ident_e, ident_r = {i: i for i in range(n_ent)}, {i: i for i in range(n_rel)}
    kg = tk.KnowledgeGraph(kg={'heads': heads, 'tails': tails, 'relations': rels}, ent2ix=ident_e, rel2ix=ident_r)
    # test facts: the last n_test - hub_tests facts of the (shuffled) graph + hub_tests facts of the most frequent
    # (h, r) / (t, r) keys, alternating sides
    sel = list(range(n_facts - (n_test - hub_tests), n_facts))
    if hub_tests > 0:
        for side_key, cnt in ((heads * n_rel + rels, hub_tests - hub_tests // 2), (tails * n_rel + rels, hub_tests // 2)):
            uk, inv, c = torch.unique(side_key, return_inverse=True, return_counts=True)
            top = torch.argsort(c, descending=True)[:4]
            per = -(-cnt // 4)
            for k in top.tolist():
                sel += torch.nonzero(inv == k).view(-1)[:per].tolist()
        sel = sel[:n_test]
    sel = torch.tensor(sel, dtype=torch.long)
    kg_test = tk.KnowledgeGraph(kg={'heads': heads[sel].clone(), 'tails': tails[sel].clone(),
                                    'relations': rels[sel].clone()}, ent2ix=ident_e, rel2ix=ident_r,
                                _filter_src=kg._lazy)
    info = {'kind': 'complex', 'shape': 'wikidata5m', 'd': 512, 'p': 2, 'n_ent': n_ent, 'n_rel': n_rel,
            'n_test': int(sel.shape[0]), 'kg_kind': 'zipf', 'weights': 'unit', 'graph_facts': n_facts}
    return model, kg, kg_test, info


def _parity_summary(ref, ties, got, n_test, what, secs):
    inside = (got >= ties[..., 0]) & (got <= ties[..., 1])
    from oracle import kge_oracle as orc
    mo = orc.lp_metrics(*ref, 10)
    mg = orc.lp_metrics(*got, 10)
    # ranks on opposite sides of k = 10 (each such near-tie flip moves Hits@10 by 0.5 / n_test: 8.5e-6 at cfg4)
    flips10 = int(((ref[2:] <= 10) != (got[2:] <= 10)).sum())
    return {'filtered_ranks_across_the_hits10_boundary': flips10, 'oracle': what,
            'ranks_compared': int(ref.numel()), 'ranks_differing': int((ref != got).sum()),
            'max_abs_rank_diff': int((ref - got).abs().max()) if ref.numel() else 0,
            'within_reference_tie_interval_2e-5': bool(inside.all()), 'outside_tie_interval': int((~inside).sum()),
            'filt_mrr_ref_hip': [mo['mrr'][1], mg['mrr'][1]], 'filt_hits10_ref_hip': [mo['hit_at_k'][1], mg['hit_at_k'][1]],
            'mrr_ref_hip': [mo['mrr'][0], mg['mrr'][0]],
            'abs_diff_filt_mrr': abs(mo['mrr'][1] - mg['mrr'][1]),
            'abs_diff_filt_hits10': abs(mo['hit_at_k'][1] - mg['hit_at_k'][1]),
            'median_filt_rank_ref': float(torch.cat([ref[2], ref[3]]).float().median()), 'oracle_seconds': round(secs, 1)}


def sample_parity(model, info, kg, kg_test, ev_ranks, device, n=64, b=2, tol=2e-5):
    """cfg5-size workloads ("subsample to <= 64 test triples and b <= 2"): the first `n` test facts through
    the reference algorithm on ATen GPU ops (oracle.lp_evaluate, its (b, N, d) temporaries are 18.8 GB each at b = 2),
    the model's tables read in place, the filter sets of the FULL graph for the looked-up keys by numpy scans
    (oracle.filter_dicts_for_facts -- independent of the engine's filter index)."""
    from oracle import kge_oracle as orc
    n = min(n, info['n_test'])
    th, tt, tr = kg_test.head_idx[:n].cpu(), kg_test.tail_idx[:n].cpu(), kg_test.relations[:n].cpu()
    t0 = time.perf_counter()
    dh, dt = orc.filter_dicts_for_facts(kg.head_idx, kg.tail_idx, kg.relations, th, tt, tr)
    tables = [x.data for x in model._tables()]
    with torch.no_grad():
        rh, rt, frh, frt, ties = orc.lp_evaluate(info['kind'], tables, th, tt, tr, dh, dt, b, info['p'], tie_tol=tol,
                                                 device=device)
    torch.cuda.empty_cache()
    secs = time.perf_counter() - t0
    ref = torch.stack([rh, rt, frh, frt])
    got = torch.stack([x.cpu()[:n] for x in ev_ranks])
    out = _parity_summary(ref, ties, got, n, 'oracle.lp_evaluate = reference algorithm on ATen GPU ops, b_size=%d, the '
                          'first %d test facts (SURVEY 8d subsample for this size)' % (b, n), secs)
    out['filter_list_entries_of_the_sample'] = int(sum(len(v) for v in dh.values()) + sum(len(v) for v in dt.values()))
    return out


def full_split_parity(info, tables, kg, kg_test, ev_ranks, device, b=256, tol=2e-5):
    """Every rank of the whole test split against the reference algorithm on ATen GPU ops
    (oracle.lp_evaluate(device=...), evaluation.py:263-308).  ev_ranks: the engine's four rank
    vectors (heads, tails, filtered heads, filtered tails).  Ranks are an integer function of
    fp32 scores: a near-tie may move a rank by one between two correct fp32 implementations, so
    the per-rank criterion is containment in the tie interval the reference's own scores allow
    within `tol`; the metrics must agree to 1e-5."""
    from oracle import kge_oracle as orc
    th, tt, tr = kg_test.head_idx.cpu(), kg_test.tail_idx.cpu(), kg_test.relations.cpu()
    t0 = time.perf_counter()
    rh, rt, frh, frt, ties = orc.lp_evaluate(info['kind'], tables, th, tt, tr, kg.dict_of_heads, kg.dict_of_tails,
                                             b, info['p'], tie_tol=tol, device=device)
    secs = time.perf_counter() - t0
    ref = torch.stack([rh, rt, frh, frt])
    got = torch.stack([x.cpu() for x in ev_ranks])
    return _parity_summary(ref, ties, got, info['n_test'],
                           'oracle.lp_evaluate = reference algorithm on ATen GPU ops, b_size=%d' % b, secs)


def strong_scaling_model(n_test, n_ent, d, measured_ms_1gpu=None):
    """Per-phase MODEL of one entity-sharded evaluate() of this job on P GPUs (exchange='counts', r05 path: fused query side
    fed from the query-entity replicas, one-product level on every shard).  No multi-GPU box was available to any round, so
    this is arithmetic on the single-GPU phase times measured in profiles/r05 (cfg2: timeline_transe_fb15k237.txt), scaled
    by the work each phase does -- printed beside every measurement so that a measured curve can be held against it:
      fixed      : host gap + graph launch + rank copy (0.075 ms), the query pipeline over all 2B queries (0.040 -- every rank
                   builds every query row), finalize; does not shrink with P
      1/P        : candidate-table preparation, the count sweep, its exact recheck, the filter correction's scoring
      collectives: the replica all-gather of the U ~ 0.83 N distinct query entities' rows once per evaluate ((P-1)/P * U * d * 4 B
                   per rank at 120 GB/s effective per direction) + two latency-bound all-reduces (true scores ride the
                   replicas: none; counts + flags: one of 24 B x 2B) at ~25 us each on xGMI."""
    pairs = 2.0 * n_test * n_ent
    scale = pairs / (2.0 * 20466 * 14541) * (d / 200.0)         # the 1/P phases scale with pairs x width
    qscale = (n_test / 20466.0) * (d / 200.0)
    fixed = 0.075 + 0.040 * qscale + 0.003
    per_p = (0.010 * (n_ent / 14541.0) * (d / 200.0) + 0.290 * scale + 0.070 * scale + 0.035 * qscale)
    out = {}
    for P in (1, 2, 4, 8):
        coll = 0.0
        if P > 1:
            u_bytes = 0.83 * n_ent * d * 4.0
            coll = (P - 1) / P * u_bytes / 120e9 * 1e3 + 2 * 0.025
        out[str(P)] = round(fixed + per_p / P + coll, 4)
    # the OTHER partition of the same job (shard='queries'): tables replicated, the 2 x n_test queries split across the ranks,
    # ONE all-gather of the (4, n) int64 ranks at the end.  Every rank still prepares the whole candidate table (not / P); the
    # query pipeline, sweep, recheck and filter correction shrink with the queries.
    tprep = 0.010 * (n_ent / 14541.0) * (d / 200.0)
    per_p_q = 0.040 * qscale + 0.290 * scale + 0.070 * scale + 0.035 * qscale
    out_q = {}
    for P in (1, 2, 4, 8):
        coll = 0.0 if P == 1 else (P - 1) / P * 4 * n_test * 8 / 76.8e9 * 1e3 + 0.025
        out_q[str(P)] = round(0.078 + tprep + per_p_q / P + coll, 4)
    res = {'modelled_ms_per_evaluate': out,
           'modelled_speedup_vs_1gpu': {k: round(out['1'] / v, 2) for k, v in out.items()},
           'query_partition': {'modelled_ms_per_evaluate': out_q,
                               'modelled_speedup_vs_1gpu': {k: round(out_q['1'] / v, 2) for k, v in out_q.items()},
                               'what': "shard='queries': replicated tables, 2B/P queries per GPU, one all-gather of the ranks; "
                                       'the better partition while the tables fit one GPU (cfg2: 11.6 MB) -- entity shards are '
                                       'what north_star names and what cfg5 needs'},
           'phases_ms_at_P1': {'fixed': round(fixed, 4), 'shrinks_as_1_over_P': round(per_p, 4)},
           'note': 'a MODEL (single-GPU phase times of profiles/r05 scaled by work; collectives from link arithmetic), not a '
                   'measurement: no multi-GPU node was available to any round'}
    if measured_ms_1gpu is not None:
        res['measured_ms_1gpu_this_run'] = round(measured_ms_1gpu, 4)
    return res


def _load_reference():
    """The REAL reference package, staged by oracle/Makefile as the git-ignored oracle/_ref/torchkge (it travels to the GPU
    box with the snapshot).  None when absent.  Only bench.py's cpu_baseline leg uses it."""
    ref_root = os.path.join(ROOT, 'oracle', '_ref')
    if not os.path.isdir(os.path.join(ref_root, 'torchkge')):
        return None
    if ref_root not in sys.path:
        sys.path.insert(0, ref_root)
    try:
        import torchkge            # noqa: F401  (the reference; this repo's package is torchkge_amd)
        return torchkge
    except Exception:
        return None


def reference_evaluator(ref, kind, p, tables, n_ent, n_rel, heads, tails, rels, dict_of_heads, dict_of_tails):
    """torchkge's own model + KnowledgeGraph + LinkPredictionEvaluator (evaluation.py:207-308) on CPU tensors holding
    `tables`, for the given facts; the filter dictionaries are the full graph's."""
    from collections import defaultdict
    d = tables[0].shape[1]
    if kind == 'transe':
        m, names = ref.models.TransEModel(d, n_ent, n_rel, dissimilarity_type='L%d' % p), ['ent_emb', 'rel_emb']
    elif kind == 'distmult':
        m, names = ref.models.DistMultModel(d, n_ent, n_rel), ['ent_emb', 'rel_emb']
    elif kind == 'complex':
        m, names = ref.models.ComplExModel(d, n_ent, n_rel), ['re_ent_emb', 'im_ent_emb', 're_rel_emb', 'im_rel_emb']
    elif kind == 'transh':
        m, names = ref.models.TransHModel(d, n_ent, n_rel), ['ent_emb', 'rel_emb', 'norm_vect']
    elif kind == 'transd':
        m, names = ref.models.TransDModel(d, tables[1].shape[1], n_ent, n_rel), ['ent_emb', 'rel_emb', 'ent_proj_vect',
                                                                                 'rel_proj_vect']
    else:
        raise ValueError(kind)
    m.load_state_dict({n + '.weight': t.clone() for n, t in zip(names, tables)}, strict=False)
    kg_ref = ref.data_structures.KnowledgeGraph(
        kg={'heads': heads.clone(), 'tails': tails.clone(), 'relations': rels.clone()},
        ent2ix={i: i for i in range(n_ent)}, rel2ix={i: i for i in range(n_rel)},
        dict_of_heads=defaultdict(set, dict_of_heads), dict_of_tails=defaultdict(set, dict_of_tails),
        dict_of_rels=defaultdict(set))
    return ref.evaluation.LinkPredictionEvaluator(m, kg_ref)


def _flush_c_stdio():
    """RCCL writes a banner through C stdio; on a piped stdout it would otherwise surface after the JSON line."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def _sample_power(run, device, seconds=1.6):
    """Package power / shader clock while the dominant kernel runs back to back (rocm-smi sampled from a side thread):
    says whether the kernel sits at the package power cap -- then the clock, not the instruction schedule, sets its rate."""
    import re
    import shutil
    import subprocess
    import threading
    smi = shutil.which('rocm-smi') or '/opt/rocm/bin/rocm-smi'
    if not os.path.exists(smi):
        return None
    out = {}

    def sampler():
        time.sleep(0.6)
        try:
            txt = subprocess.run([smi, '--showpower', '--showclocks', '--showmaxpower'], stdout=subprocess.PIPE,
                                 stderr=subprocess.DEVNULL, text=True, timeout=20).stdout
            m = re.search(r'GPU\[0\].*?Package Power \(W\):\s*([0-9.]+)', txt)
            c = re.search(r'GPU\[0\].*?Max Graphics Package Power \(W\):\s*([0-9.]+)', txt)
            k = re.search(r'GPU\[0\].*?sclk clock level:.*?\((\d+)Mhz\)', txt)
            if m:
                out['package_W'] = float(m.group(1))
            if c:
                out['cap_W'] = float(c.group(1))
            if k:
                out['sclk_MHz_reported'] = int(k.group(1))
        except Exception:
            pass
    th = threading.Thread(target=sampler)
    th.start()
    ts = time.perf_counter()
    n = 0
    while th.is_alive() or time.perf_counter() - ts < seconds:
        for _ in range(50):
            run()
        torch.cuda.synchronize(device)
        n += 50
        if time.perf_counter() - ts > 30:
            break
    th.join()
    if not out:
        return None
    out['sustained_kernel_ms'] = round((time.perf_counter() - ts) / n * 1e3, 4)
    out['how'] = 'rocm-smi sampled while the kernel runs back to back for %.1f s (host-timed, incl. launch gaps)' % (time.perf_counter() - ts)
    return out


def _measure_traffic(args, kernel_sym, extra_out=None):
    """HBM-side bytes per launch of the dominant kernel, measured in THIS run: bench.py re-runs itself (timed loop only,
    eager launches, the SAME weights and split level as the headline: the uncertain-pair list a model leaves is part of
    the kernel's writes) under
    ``rocprofv3 --pmc <counter> --kernel-trace`` once per counter -- FETCH_SIZE (3 TCC slots) and WRITE_SIZE (2) do
    not fit one pass -- and averages the counter over the launches of `kernel_sym`.  FETCH_SIZE / WRITE_SIZE are in
    KB; gfx950 reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> 2 x FETCH_SIZE.
    Infinity-Cache hits are counted by these memory-side counters, i.e. this is L2-miss traffic, not DRAM traffic.
    A third pass (extra_out: dict) collects the matrix-pipe counters of the same kernel: SQ_INSTS_MFMA (executed MFMA
    instructions per evaluate, summed over the kernel's instantiations), SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(rp):
        return None
    vals = {}
    passes = [('FETCH_SIZE',), ('WRITE_SIZE',)]
    if extra_out is not None:
        passes.append(('SQ_INSTS_MFMA', 'SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE'))
    for ctrs in passes:
        d = tempfile.mkdtemp(prefix='kge_pmc_', dir='/tmp')
        cmd = [rp, '--pmc'] + list(ctrs) + ['--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 't', '--', sys.executable,
               os.path.abspath(__file__), '--only-timed', '--no-graph', '--no-traffic', '--weights', args.weights,
               '--steps', '3', '--warmup', '0',
               '--settle-ms', '0', '--workload', args.workload, '--batch', str(args.batch), '--kg', args.kg, '--l2-mode', args.l2_mode,
               '--split-level', str(getattr(args, 'level_used', args.split_level))]
        cmd += (['--no-split'] if args.no_split else []) + (['--materialize'] if args.materialize else []) \
            + (['--no-both'] if args.no_both else [])
        env = dict(os.environ, TMPDIR='/tmp')
        for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'KGE_FORCE_COLLECTIVES'):
            env.pop(k, None)
        try:
            if args.train_steps is not None:
                cmd += ['--train-steps', str(args.train_steps)]
            subprocess.run(cmd, cwd="/tmp", env=env, timeout=(1200 if args.workload == 'complex_wikidata5m' else 300),
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            got = {}        # per (counter, kernel NAME): the count kernel may run as two instantiations per launch
            for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):      # (single / grouped columns)
                for r in csv.DictReader(open(f)):
                    if r.get('Counter_Name') in ctrs and kernel_sym in r.get('Kernel_Name', ''):
                        got.setdefault((r['Counter_Name'], r['Kernel_Name']), []).append(float(r['Counter_Value']))
            for ctr in ctrs:
                per_kernel = [sum(v) / len(v) for (c, _), v in got.items() if c == ctr]
                if per_kernel:
                    vals[ctr] = sum(per_kernel)
        except Exception:
            pass
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if extra_out is not None and 'SQ_INSTS_MFMA' in vals:
        extra_out['SQ_INSTS_MFMA'] = vals['SQ_INSTS_MFMA']
        if vals.get('GRBM_GUI_ACTIVE') and 'SQ_VALU_MFMA_BUSY_CYCLES' in vals:
            extra_out['SQ_VALU_MFMA_BUSY_CYCLES'] = vals['SQ_VALU_MFMA_BUSY_CYCLES']
            extra_out['GRBM_GUI_ACTIVE'] = vals['GRBM_GUI_ACTIVE']
            # busy cycles are summed over the chip's 1024 SIMDs, GUI_ACTIVE over its 8 XCDs (how r03's review derived it)
            extra_out['mfma_busy_frac'] = round(vals['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / (vals['GRBM_GUI_ACTIVE'] / 8.0), 4)
    if 'FETCH_SIZE' not in vals or 'WRITE_SIZE' not in vals:
        return None
    if extra_out is not None:
        extra_out['write_bytes'] = int(vals['WRITE_SIZE'] * 1024)
    return int((2 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024)


def _self_launch(n):
    """Re-run this command line under ``python -m torch.distributed.run --nnodes=1 --nproc-per-node n`` on 127.0.0.1
    with a free port; stdout / stderr pass through (rank 0 prints the JSON line).  Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC (RCCL / tensor sharing across processes)
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // n)))
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.only_timed:
        args.no_cpu_baseline = args.no_secondary = args.no_full_parity = args.no_traffic = True
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            # plain `python bench.py --gpus N`: launch the N ranks ourselves (what the driver's torch.distributed.run line
            # does), one per GPU over RCCL -- or, with --backend gloo, N ranks sharing GPU 0 (logic dry run)
            raise SystemExit(_self_launch(args.gpus))
    if args.launch_check:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if world > 1:
            dist.init_process_group('gloo')
            tt = torch.tensor([float(rank + 1)], dtype=torch.float64)
            dist.all_reduce(tt)
            dist.barrier()
            got = float(tt.item())
            dist.destroy_process_group()
        else:
            got = 1.0
        if rank == 0:
            print(json.dumps({'launch_check': True, 'n_gpus': args.gpus, 'world_size': world, 'backend': 'gloo',
                              'sum_of_rank_plus_one': got, 'expected': world * (world + 1) / 2.0}), flush=True)
        return
    dev_index = local_rank % max(torch.cuda.device_count(), 1)   # (>1 rank per GPU only in --backend gloo dry runs)
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    # what the process's FIRST GPU operation costs, whoever issues it (ROCm device wake-up: context, queues, first copy)
    _t0 = time.perf_counter()
    torch.zeros(1, device=device).add_(1.0)
    torch.cuda.synchronize(device)
    rocm_wakeup_ms = (time.perf_counter() - _t0) * 1e3
    forced = os.environ.get('KGE_FORCE_COLLECTIVES') == '1' and 'RANK' in os.environ   # debug: collectives at N=1
    if world > 1 or forced:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)      # RCCL over xGMI
        else:
            dist.init_process_group('gloo')                        # logic dry-run (several ranks on one GPU)

    import torchkge_amd as tk
    from torchkge_amd import _hip
    from oracle import kge_oracle as orc     # synthetic-KG generator + the cpu_baseline leg only

    kind, shape, d, p = WORKLOADS[args.workload]
    n_ent1, n_rel, n_train, n_valid, n_test = orc.DATASET_SHAPES[shape]
    multi = world > 1 or forced
    if args.exchange is None:
        # N > 1 headline = the exchange the evaluator itself defaults to: the all-reduce of the (3, 2B) partial rank counts
        # behind the f16 prefilter.  The score exchange north_star names (all-to-all of the partial score tiles) is measured
        # beside it at the same b_size (other_exchange): it needs the EXACT fp32 score GEMM (>= 1.5 ms of fp32 MFMA peak for
        # this job on one GPU) and moves (P-1)/P^2 * 2B*N*4 bytes per rank -- 595 MB over ONE xGMI link at N = 2 (7.8 ms
        # modelled against 0.63 ms for the whole single-GPU evaluation) -- so it cannot strong-scale a job this small.
        args.exchange = 'counts'
    # weak scaling over entity shards (default for N > 1): the entity table grows to N
    # dataset-sized shards, each GPU scores ITS shard for every test triple and the ranks
    # exchange partial results over RCCL -- per-GPU work is fixed as N grows.
    ent_weak = multi and args.scaling == 'weak' and args.shard == 'entities'
    weights = args.weights
    # (cfg5 trains like the others since r06: 18.8 GB of tables + gradients + Adam moments = 75 GB of the 288 GB;
    #  --weights unit keeps r03-r05's untrained unit-scale tables)
    train_cfg = {'steps': args.train_steps} if args.train_steps is not None else None
    # N > 1: only rank 0 trains (atomics make training run-to-run different); its tables are broadcast below
    model, tables, kg, kg_test, info = build_workload(args.workload, device, kg_kind=args.kg,
                                                      weights=(weights if rank == 0 or not multi or weights != 'trained'
                                                               else 'xavier'),
                                                      n_ent_mult=(world if ent_weak else 1), train_cfg=train_cfg)
    info['weights'] = weights
    n_ent = info['n_ent']
    if multi:
        for prm in model.parameters():      # identical tables on every rank before they are sharded
            if args.backend == 'nccl':
                dist.broadcast(prm.data, src=0)
            else:
                buf = prm.data.cpu()
                dist.broadcast(buf, src=0)
                prm.data.copy_(buf)
    heads, tails, rels = kg.head_idx, kg.tail_idx, kg.relations
    ident_e, ident_r = kg.ent2ix, kg.rel2ix
    if kind in ('transe', 'transh', 'transd'):
        model.l2_mode = args.l2_mode
    model.split_filter = not args.no_split
    model.split_level = 'auto' if args.split_level == 'auto' else int(args.split_level)
    replicas = multi and args.scaling == 'weak' and args.shard != 'entities'
    if replicas:
        # every rank evaluates its own test split of the same size (facts of the same graph)
        g = torch.Generator().manual_seed(7 + rank)
        sel = torch.randperm(kg.n_facts, generator=g)[:n_test]
        kg_test = tk.KnowledgeGraph(kg={'heads': heads[sel], 'tails': tails[sel], 'relations': rels[sel]},
                                    ent2ix=ident_e, rel2ix=ident_r, _filter_src=kg._lazy)
    # test triples resident in HBM before the timed region
    kg_test.head_idx = kg_test.head_idx.to(device)
    kg_test.tail_idx = kg_test.tail_idx.to(device)
    kg_test.relations = kg_test.relations.to(device)

    flt_stats = None
    if rank == 0 and n_ent <= 2000000:
        flt_stats = orc.filter_list_stats(heads, tails, rels, kg_test.head_idx.cpu(), kg_test.tail_idx.cpu(),
                                          kg_test.relations.cpu(), n_ent, n_rel)
    shard = None
    if multi and not replicas:
        shard = 'entities' if args.shard == 'entities' else 'queries'
    table_bytes_full = model.entity_table_bytes()
    model_rep = None
    if shard == 'entities' and args.tables == 'sharded':
        # ROW-SHARDED entity tables: this rank keeps rows [lo, hi) of every entity-indexed table;
        # relation tables stay replicated; query rows are built by the owner rank and summed over the ranks
        from torchkge_amd import distributed as kd
        if table_bytes_full <= (1 << 30) and not args.materialize:
            import copy as _copy0
            model_rep = _copy0.deepcopy(model)      # (a replica: the query partition of the same job, measured beside the headline)
        kd.shard_model_(model)
        torch.cuda.empty_cache()
    # (N > 1: graph=None = 'auto' -- first call eager, second captures, and a capture that fails on real multi-GPU
    # hardware falls back to eager launches instead of aborting the line; every rank still issues the same collectives)
    graph_arg = (not args.no_graph) if not multi else (None if not args.no_graph else False)
    ev = tk.LinkPredictionEvaluator(model, kg_test, fused=not args.materialize, shard=shard,
                                    exchange=args.exchange, graph=graph_arg, overlap=args.overlap,
                                    both_sides=not args.no_both, graph_collectives=True if args.graph_collectives else None)

    def sync():
        torch.cuda.synchronize(device)
        if multi:
            dist.barrier()
            torch.cuda.synchronize(device)

    # Clock / power settle (untimed, before the W warm-up steps): the count kernel runs at the package power cap
    # (1400 W, tools/power_probe.sh), and the first milliseconds after an idle gap run at a different DVFS point
    # than the sustained state (same-box: 0.68 ms per launch cold, 0.54 sustained).  The timed region below is
    # still exactly W warm-up + K timed steps.
    # What a reference-style script pays that calls evaluate() ONCE per epoch: the first call of a fresh evaluator
    # (device-side filter index by sort / unique, FilterPlans of every batch, the MFMA accumulation self-test, eager
    # launches, the rank vectors' copy to the host) ...
    sync()
    first_parts = {'rocm_wakeup_ms_first_gpu_op_of_the_process': round(rocm_wakeup_ms, 2)}
    t0 = time.perf_counter()
    if not multi:
        # the library's own first-use work, in parts: its first launch (code object load), the device-side filter index of the
        # full graph, then the evaluation itself (FilterPlans, MFMA self-test, capture or eager launches, ranks to the host)
        from torchkge_amd import _hip as _hipb
        _hipb.row_sqnorm(torch.ones(4, 8, device=device))
        sync()
        first_parts['first_library_launch_ms'] = round((time.perf_counter() - t0) * 1e3, 2)
        t1 = time.perf_counter()
        if hasattr(kg_test, 'filter_index'):
            kg_test.filter_index('heads', device)
            kg_test.filter_index('tails', device)
        sync()
        first_parts['filter_index_ms'] = round((time.perf_counter() - t1) * 1e3, 2)
    t1 = time.perf_counter()
    ev.evaluate(args.batch, verbose=False)      # (first call: eager, builds index + plans)
    sync()
    first_parts['evaluate_ms'] = round((time.perf_counter() - t1) * 1e3, 2)
    first_ms = (time.perf_counter() - t0) * 1e3
    ev.evaluate(args.batch, verbose=False)      # (second call: captures the hipGraph)
    sync()
    # ... and the per-step time of a few evaluations after an idle gap, clocks NOT settled (what r01 / early r02 reported)
    time.sleep(0.5)
    sync()
    t0 = time.perf_counter()
    for _ in range(5):
        ev.evaluate(args.batch, verbose=False)
    sync()
    cold_ms = (time.perf_counter() - t0) / 5 * 1e3
    settle_steps = 0
    if args.settle_ms > 0:
        sync()
        t0 = time.perf_counter()
        ev.evaluate(args.batch, verbose=False)
        sync()
        t_one = time.perf_counter() - t0
        if multi:
            tt = torch.tensor([t_one], device=device if args.backend == 'nccl' else 'cpu', dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_one = float(tt.item())
        settle_steps = int(min(400, args.settle_ms * 1e-3 / max(t_one, 1e-6)))
        for _ in range(settle_steps):
            ev.evaluate(args.batch, verbose=False)
        sync()
    for _ in range(args.warmup):
        ev.evaluate(args.batch, verbose=False)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ev.evaluate(args.batch, verbose=False)
    sync()
    elapsed = time.perf_counter() - t0
    # (read HERE: the legs that follow -- the other exchange, the weak mode -- run evaluations of their own on this model)
    level_timed, rescored_timed = int(getattr(model, '_split_level', 0)), getattr(ev, 'last_rescored_per_query', None)
    if multi:
        tt = torch.tensor([elapsed], device=device if args.backend == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    def timed_steps(evx, bsz, n):
        sync()
        t1 = time.perf_counter()
        for _ in range(n):
            evx.evaluate(bsz, verbose=False)
        sync()
        el = time.perf_counter() - t1
        if multi:
            tt_ = torch.tensor([el], device=device if args.backend == 'nccl' else 'cpu', dtype=torch.float64)
            dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
            el = float(tt_.item())
        return el

    def modelled_xgmi(exchange, n_ent_all):
        """Fabric time of one evaluate()'s rank exchange FROM BYTES: MI355X has 7 xGMI links per GPU, one per peer on an
        8-GPU node, 153.6 GB/s each bidirectional = 76.8 GB/s per direction (the task's '7 x ~153 GB/s'); a direct
        all-to-all / all-gather drives every link at once, so time = bytes one link carries in one direction / 76.8e9."""
        n2 = 2 * n_test
        per = -(-n_ent_all // world)
        if exchange == 'scores':        # all-to-all of row tiles: one (2B/P, N/P) block to every peer
            link = -(-n2 // world) * per * 4
            what = 'all-to-all of (2B/P, N/P) fp32 score blocks + one int64 all-reduce of the (4, n) ranks'
            sent = link * (world - 1) + 2 * 4 * n_test * 8 * (world - 1) // max(world, 1)
        else:                           # counts: (3, 2B) int32 ring all-reduce, 2 (P-1)/P of the buffer per link
            link = 2 * (world - 1) * 3 * n2 * 4 // max(world, 1)
            what = 'all-reduce of the (3, 2B) int32 partial counts'
            sent = link
        if world == 1:
            link = sent = 0
        return {'what': what, 'bytes_sent_per_rank': int(sent), 'bytes_per_link_and_direction': int(link),
                'link_GBps_per_direction': 76.8, 'ms': round(link / 76.8e9 * 1e3, 4),
                'note': 'bandwidth term only: every collective also pays ~10-30 us of launch / protocol latency'}

    def collective_ms(evx, bsz, n=3, exchange=None, n_ent_all=None):
        """Device time of the data-path collectives per evaluate() (events around every RCCL call on the launch stream),
        with the fabric time modelled from the bytes beside it."""
        evx.collective_timing(True)
        for _ in range(n):
            evx.evaluate(bsz, verbose=False)
        c = evx.collective_timing(False)
        out = {'collectives_per_evaluate': c['collectives'] // n, 'ms_per_evaluate': round(c['ms'] / n, 4)}
        if exchange is not None:
            out['modelled_xgmi'] = modelled_xgmi(exchange, n_ent_all if n_ent_all is not None else n_ent)
        return out

    headline_coll = None
    if multi and shard == 'entities' and not args.materialize and device.type == 'cuda':
        headline_coll = collective_ms(ev, args.batch, exchange=args.exchange)

    # entity shards: the OTHER exchange measured beside the headline one, at the SAME b_size -- the all-gather of the
    # partial score tiles (B, N/P) -> (B, N) that north_star names, vs the all-reduce of rank counts (bit-identical ranks)
    other_x = None
    if multi and shard == 'entities' and not args.materialize:
        ox = 'scores' if args.exchange == 'counts' else 'counts'
        ob = args.batch
        ev_o = tk.LinkPredictionEvaluator(model, kg_test, shard=shard, exchange=ox, graph=graph_arg,
                                          both_sides=not args.no_both,
                                          graph_collectives=True if args.graph_collectives else None)
        main_ranks = [ev.rank_true_heads.clone(), ev.rank_true_tails.clone(), ev.filt_rank_true_heads.clone(),
                      ev.filt_rank_true_tails.clone()]
        for _ in range(3):      # eager call, capture, one replay
            ev_o.evaluate(ob, verbose=False)
        n_o = max(2, args.steps // 2)
        el_o = timed_steps(ev_o, ob, n_o)
        same_o = all(torch.equal(a, b) for a, b in zip(main_ranks, [ev_o.rank_true_heads, ev_o.rank_true_tails,
                                                                    ev_o.filt_rank_true_heads, ev_o.filt_rank_true_tails]))
        other_x = {'exchange': ox, 'b_size': ob, 'steps': n_o, 'ms_per_step': round(el_o / n_o * 1e3, 4),
                   'value': round(n_test * 2 * n_ent * n_o / el_o, 1), 'ranks_identical_to_headline_run': bool(same_o),
                   'collective': 'RCCL all-to-all of the partial score tiles: every rank receives the (2B/P, N) score rows '
                                 'of the queries it ranks as P rank-major (2B/P, N/P) tiles' if ox == 'scores'
                                 else 'RCCL all-reduce of the (3, 2B) rank counts (+ one all-gather of the query-entity rows per evaluate)',
                   'collective_time': collective_ms(ev_o, ob, exchange=ox) if device.type == 'cuda' else None}
        del ev_o

    # ... the same job under the OTHER partition: test facts split across the ranks, tables replicated, no data-path collective
    # (one all-gather of the ranks at the end) -- both partitions on the line, each with its scaling model
    query_part = None
    if model_rep is not None and multi:
        ev_q = tk.LinkPredictionEvaluator(model_rep, kg_test, shard='queries', graph=graph_arg, both_sides=not args.no_both)
        for _ in range(4):      # eager, level switch + capture, replays
            ev_q.evaluate(args.batch, verbose=False)
        n_q = max(2, args.steps // 2)
        el_q = timed_steps(ev_q, args.batch, n_q)
        same_q = all(torch.equal(a, b) for a, b in zip(
            [ev.rank_true_heads, ev.rank_true_tails, ev.filt_rank_true_heads, ev.filt_rank_true_tails],
            [ev_q.rank_true_heads, ev_q.rank_true_tails, ev_q.filt_rank_true_heads, ev_q.filt_rank_true_tails]))
        query_part = {'partition': "shard='queries' (replicated tables, 2B/P queries per rank, one all-gather of the ranks)",
                      'steps': n_q, 'ms_per_step': round(el_q / n_q * 1e3, 4), 'value': round(n_test * 2 * n_ent * n_q / el_q, 1),
                      'split_level': int(getattr(model_rep, '_split_level', 0)),
                      'ranks_identical_to_headline_run': bool(same_q)}
        del ev_q
    # ... and, beside a strong-scaling run, the WEAK mode: the entity table grown to N dataset-sized shards (Xavier
    # weights), every rank scoring its own shard for all test facts, counts exchange -- per-GPU work fixed as N grows
    weak_mode = None
    if multi and shard == 'entities' and args.scaling == 'strong' and not args.no_weak and not args.materialize \
            and args.tables == 'sharded':
        from torchkge_amd import distributed as kd
        m_w, _, kg_w, kg_test_w, info_w = build_workload(args.workload, device, kg_kind=args.kg, weights='xavier',
                                                         n_ent_mult=world)
        if kind in ('transe', 'transh', 'transd'):
            m_w.l2_mode = args.l2_mode
        kg_test_w.head_idx, kg_test_w.tail_idx, kg_test_w.relations = (kg_test_w.head_idx.to(device), kg_test_w.tail_idx.to(device),
                                                                       kg_test_w.relations.to(device))
        kd.shard_model_(m_w)
        ev_w = tk.LinkPredictionEvaluator(m_w, kg_test_w, shard='entities', exchange='counts', graph=graph_arg,
                                          graph_collectives=True if args.graph_collectives else None)
        for _ in range(3):
            ev_w.evaluate(args.batch, verbose=False)
        n_w = max(2, args.steps // 2)
        el_w = timed_steps(ev_w, args.batch, n_w)
        weak_mode = {'scaling': 'weak', 'n_ent': info_w['n_ent'], 'entity_shards': world, 'exchange': 'counts',
                     'weights': 'xavier', 'steps': n_w, 'ms_per_step': round(el_w / n_w * 1e3, 4),
                     'value': round(n_test * 2 * info_w['n_ent'] * n_w / el_w, 1),
                     'scored_triples_per_step': n_test * 2 * info_w['n_ent'],
                     'collective_time': collective_ms(ev_w, args.batch, exchange='counts', n_ent_all=info_w['n_ent'])
                     if device.type == 'cuda' else None}
        del ev_w, m_w, kg_w, kg_test_w
        torch.cuda.empty_cache()

    # ... and BASELINE cfg4 -- the configuration named for 8 GPUs: DistMult d = 400 on the FB15k shape, entity table sharded --
    # next to the default cfg2 job (strong scaling, counts exchange, trained-like weights from rank 0)
    cfg4_mode = None
    if multi and shard == 'entities' and args.workload == 'transe_fb15k237' and not args.no_weak and not args.materialize \
            and args.tables == 'sharded':
        from torchkge_amd import distributed as kd
        m_4, _, kg_4, kg_test_4, info_4 = build_workload('distmult_fb15k', device, kg_kind=args.kg,
                                                         weights=('trained' if rank == 0 else 'xavier'),
                                                         train_cfg={'steps': 300})
        for prm in m_4.parameters():
            if args.backend == 'nccl':
                dist.broadcast(prm.data, src=0)
            else:
                buf = prm.data.cpu()
                dist.broadcast(buf, src=0)
                prm.data.copy_(buf.to(device))
        kg_test_4.head_idx, kg_test_4.tail_idx, kg_test_4.relations = (kg_test_4.head_idx.to(device), kg_test_4.tail_idx.to(device),
                                                                       kg_test_4.relations.to(device))
        kd.shard_model_(m_4)
        ev_4 = tk.LinkPredictionEvaluator(m_4, kg_test_4, shard='entities', exchange='counts', graph=graph_arg,
                                          graph_collectives=True if args.graph_collectives else None)
        for _ in range(4):      # eager, (level switch +) capture, replays
            ev_4.evaluate(args.batch, verbose=False)
        n_4 = max(2, args.steps // 2)
        el_4 = timed_steps(ev_4, args.batch, n_4)
        cfg4_mode = {'workload': 'distmult dim=400 on the fb15k-shaped synthetic KG (N=%d, R=%d, test=%d): BASELINE cfg4'
                                 % (info_4['n_ent'], info_4['n_rel'], info_4['n_test']),
                     'scaling': 'strong', 'entity_shards': world, 'exchange': 'counts', 'weights': 'trained (rank 0, broadcast)',
                     'steps': n_4, 'ms_per_step': round(el_4 / n_4 * 1e3, 4),
                     'value': round(info_4['n_test'] * 2 * info_4['n_ent'] * n_4 / el_4, 1),
                     'split_level': int(getattr(m_4, '_split_level', 0)),
                     'filtered_hits_at_10': ev_4.hit_at_k(10)[1],
                     'strong_scaling_model': strong_scaling_model(info_4['n_test'], info_4['n_ent'], 400)}
        del ev_4, m_4, kg_4, kg_test_4
        torch.cuda.empty_cache()

    # the same evaluation with the rank counts on the fp32 MFMA kernel only (reported beside the headline)
    f32_only_ms = None
    if getattr(model, 'split_filter', False) and rank == 0 and not multi and not args.only_timed:
        model.split_filter = False
        for _ in range(2):
            ev.evaluate(args.batch, verbose=False)
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            ev.evaluate(args.batch, verbose=False)
        sync()
        f32_only_ms = (time.perf_counter() - t1) / args.steps * 1e3
        f32_ranks = [ev.rank_true_heads.clone(), ev.rank_true_tails.clone(), ev.filt_rank_true_heads.clone(),
                     ev.filt_rank_true_tails.clone()]
        model.split_filter = True
        ev.evaluate(args.batch, verbose=False)
        same = all(torch.equal(a, b) for a, b in zip(f32_ranks, [ev.rank_true_heads, ev.rank_true_tails,
                                                                 ev.filt_rank_true_heads, ev.filt_rank_true_tails]))
        if not same:
            raise SystemExit('bench: f16-split ranks differ from the fp32 ranks')

    # scored triples of the whole job per step
    total_units = n_test * 2 * n_ent * (world if replicas else 1)
    value = total_units * args.steps / elapsed
    hit10, mrr = ev.hit_at_k(10), ev.mrr()
    n_ent_full = n_ent
    if shard == 'entities':            # per-rank candidate range: what one launch of the dominant kernel covers
        from torchkge_amd import distributed as kd
        lo_r, hi_r = kd.shard_range(n_ent, world, rank)
    else:
        lo_r, hi_r = 0, n_ent
    n_ent = hi_r - lo_r

    # ---- roofline of the dominant kernel (the all-candidates count kernel) ----
    roof = None
    if rank == 0 and not args.only_timed:
        B = min(args.batch, n_test)
        h, t, r = kg_test.head_idx[:B], kg_test.tail_idx[:B], kg_test.relations[:B]
        if getattr(ev, '_perm', None) is not None and B == n_test:    # the order evaluate() processes the facts in
            h, t, r = h[ev._perm], t[ev._perm], r[ev._perm]
        guard_on = hasattr(model, 'lp_guard_begin') and model.lp_guard_begin(device) is not None
        with model.lp_session():
            prob = None
            both = (shard is None and not args.materialize and ev.both_sides and not args.overlap
                    and hasattr(model, 'lp_problem_both'))
            cols = None
            if both:    # as evaluate() builds it: both sides of the batch as one problem of 2B queries -- over the
                # batch's COLUMNS (distinct query rows) where the evaluator's plan carries a ColumnPlan
                pl = (getattr(ev, '_plans', None) or {}).get((0, int(h.shape[0])))
                cols = getattr(pl, 'cols', None)
                if model._use_level1() and (not (tk.evaluation.DEDUPE_LEVEL1 and getattr(model, 'lp_dedupe_level1', True))
                                            or model._level1_stream()):
                    cols = None     # (as evaluate() does on the one-product level: Model.lp_dedupe_level1 / the free-running kernel)
                prob = model.lp_problem(h, t, r, 'both', cols=cols) if cols is not None else model.lp_problem_both(h, t, r)
            if prob is not None:
                true = torch.cat([t, h])
                if prob.pre is not None:
                    prob.pre['true_idx'] = true
                s_true = prob.pair_scores(true)
                prob.split_true = (s_true, true)    # (as the evaluator: the sweep does not list the pairs whose score IS the threshold)
                B = 2 * B
            else:
                # (row-sharded tables: only rank 0 runs this timing leg, so the query exchange is skipped --
                # rows of entities other ranks own stay zero, which does not change the kernel's work)
                xk = {'exchange': (lambda ts: None)} if getattr(model, '_row_shard', None) is not None else {}
                prob = model.lp_problem(h, t, r, 'tail', ent_lo=lo_r, ent_hi=hi_r, **xk)
                s_true = prob.pair_scores(t)
            raw = torch.zeros(B, dtype=torch.int32, device=device)
            scores_buf = torch.empty(B, n_ent, device=device) if args.materialize else None
            split = prob.split is not None and not args.materialize
            if split:       # dominant kernel of the fused path: the f16-split MFMA count kernel
                prep = prob.split_prepare()
                run = lambda: prob.split_count(prep, s_true, raw)
            elif args.materialize:
                run = lambda: prob.scores(scores_buf)
            else:
                run = lambda: prob.count_ge(s_true, raw)
            for _ in range(3):
                run()
            if args.settle_ms > 0:      # the same sustained state as the timed region
                torch.cuda.synchronize(device)
                ts = time.perf_counter()
                while (time.perf_counter() - ts) * 1e3 < args.settle_ms:
                    for _ in range(20):
                        run()
                    torch.cuda.synchronize(device)
            reps = 20
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(device)
            e0.record()                     # events on torch's current stream = the launch stream
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize(device)
            kern_s = e0.elapsed_time(e1) / 1e3 / reps
            power = _sample_power(run, device) if args.settle_ms > 0 else None
        if guard_on:
            model.lp_guard_end()
        K = d * (2 if kind == 'complex' else 1)
        mode = prob.desc.mode
        pairs = B * n_ent
        extra = {}
        # ALGORITHMIC flops per (query, candidate) pair -- 2K for the contraction kernels (K3 / K2 as a
        # GEMM), 3d for the broadcast-subtract kernels (sub, mul|abs, add) -- over the measured launch time, against the
        # dense peak of the UNIT the kernel runs on.  What the kernel additionally executes (the f16 split's three
        # products over K padded to 16) is reported as executed_*; it is not the roofline fraction.
        if split:
            level = int(prob.split.get('level', 0))      # 1: the one-product level (planar hi operands, one MFMA per k16 unit)
            k16 = ((K + 2 + 15) // 16 * 16) if level == 1 else ((K + 1 + 15) // 16 * 16)
            alg_flops, exec_flops = 2 * K, (1 if level == 1 else 3) * 2 * k16
            stream = bool(prob.split.get('es_frag'))    # r05: the free-running one-product kernel (lp_hi_stream.hip)
            kname = ('lp_hi_stream_kernel (one-product level, r05: ONE v_mfma_f32_32x32x16_f16 product per k16 unit on f16 hi '
                     'operands, fp32 accumulate; resident 128-query panel in LDS (r06; projection epilogues: 96), candidate fragments straight from the '
                     'fragment-major table into registers, no block-wide barriers in the tile loop; band from the measured f16 '
                     'residuals)') if (level == 1 and stream) else \
                ('lp_split_count_kernel, LV = 1 (f16 hi operands, ONE v_mfma_f32_32x32x16_f16 product per k16 unit, fp32 '
                 'accumulate; band from the measured f16 residuals)') if level == 1 else \
                'lp_split_count_kernel (f16 hi/lo split, v_mfma_f32_32x32x16_f16, fp32 accumulate)'
            ksym = 'lp_hi_stream_kernel' if (level == 1 and stream) else 'lp_split_count_kernel'
            chunked = level == 1 and stream and (K + 2 + 15) // 16 > 32
            if chunked:     # long rows (r06): the query panel streamed through LDS in chunks, lp_hi_chunk.hip
                ksym = 'lp_hi_chunk_kernel'
                kname = ('lp_hi_chunk_kernel (one-product level, r06: ONE v_mfma_f32_32x32x16_f16 product per k16 unit on f16 hi '
                         'operands, fp32 accumulate; candidate fragments straight from the fragment-major table into a register '
                         'ring, the 128-query panel streamed through a two-slot LDS ring in chunks of 13 units, one block-wide '
                         'barrier per chunk; band from the measured f16 residuals)')
            peak, bound = PEAK_F16_TFLOPS, 'mfma'
            # what the matrix cores EXECUTE: the sweep runs over the batch's COLUMNS (distinct query rows, padded to the
            # 192-column panel) x the candidates padded to the 256-row tile, three f16 products per k16 unit -- not over
            # (query, candidate) pairs (r03 multiplied by pairs and over-stated executed_frac: 0.51 where PMC says 0.39)
            mfma_cols = (cols.n_single_p + cols.n_multi_p) if cols is not None else (B + 191) // 192 * 192
            mfma_pairs = mfma_cols * ((n_ent + 255) // 256 * 256)
            if chunked:
                mfma_pairs = ((B + 127) // 128 * 128) * ((n_ent + 511) // 512 * 512)
            if level == 1 and stream:       # units are exact there (no padding of K to 64): ceil((K + 2) / 16) per pair
                exec_flops = 2 * 16 * ((K + 2 + 15) // 16)
            extra = {'peak_is': 'dense f16 MFMA (the unit the kernel runs on)',
                     'executed_flops_per_column_pair': exec_flops,
                     'mfma_column_pairs_per_launch': int(mfma_pairs),
                     'executed_TFLOPs': round(exec_flops * mfma_pairs / kern_s / 1e12, 2),
                     'executed_frac': round(exec_flops * mfma_pairs / kern_s / 1e12 / peak, 4),
                     'split_level': level,
                     'executed_from': '%d x 2 x 16 flop per k16 unit x (padded query columns x padded candidates); '
                                      'pmc.SQ_INSTS_MFMA x 32768 flop is the same figure from the counters' % (1 if level == 1 else 3),
                     'frac_of_fp32_mfma_peak': round(alg_flops * pairs / kern_s / 1e12 / PEAK_FP32_TFLOPS, 4),
                     'note': 'ranks are bit-identical to the fp32 path (pairs inside the proven error band are re-scored '
                             'exactly by kge_lp_split_recheck); frac = 2K algorithmic flop per pair against the f16 MFMA '
                             'peak; the kernel executes 3 f16 products per element (hi*hi, hi*lo, lo*hi) -- or ONE (hi*hi) on '
                             'the one-product level a fitted model is evaluated on, split_level = 1 --, see executed_*; '
                             'SURVEY 8(d) names the fp32 MFMA peak as this row\'s bound: frac_of_fp32_mfma_peak'}
        elif mode in (_hip.LP_DOT, _hip.LP_L2_EXPAND, _hip.LP_L2_PROJH, _hip.LP_L2_PROJD):
            alg_flops = 2 * K               # one fp32 MFMA FMA per (pair, k)
            kname = 'lp_gemm_kernel (fp32 MFMA 32x32x2%s)' % (', per-pair projection gather' if mode >= _hip.LP_L2_PROJH else '')
            ksym, peak, bound = 'lp_gemm_kernel', PEAK_FP32_TFLOPS, 'mfma'
        elif getattr(prob, 'sad', None) is not None and not args.materialize:
            alg_flops = 3 * K               # the fp32 work of the same pairs: sub, abs, add per (pair, k)
            kname = 'lp_l1_sad_count_kernel (+ thresholds + exact recheck): v_sad_u16 on 16-bit fixed-point operands'
            ksym, peak, bound = 'lp_l1_sad_count_kernel', PEAK_FP32_TFLOPS, 'valu'
            sad_cols = (cols.n_single_p + cols.n_multi_p) if cols is not None else B
            sad_insts = sad_cols * n_ent * (K // 2 + K % 2)        # one v_sad_u16 per two elements of a (column, candidate) pair
            sad_extra = {'sad_issue': {'v_sad_u16_per_launch': int(sad_insts), 'T_inst_per_s': round(sad_insts / kern_s / 1e12, 2),
                                       'issue_peak_T_inst_per_s': PEAK_SAD_TINST,
                                       'frac_of_issue_peak': round(sad_insts / kern_s / 1e12 / PEAK_SAD_TINST, 4),
                                       'note': 'the unit this kernel runs on: v_sad_u16 issues at half rate (256 CUs x 4 SIMDs x 16 '
                                               'lanes x 2.4 GHz = 39.3 T lane-instructions/s); swept over the distinct query rows'}}
            extra = {'note': 'ranks are bit-identical to the fp32 VALU path (pairs inside the proven error band are re-scored '
                             'exactly by kge_lp_sad_recheck); achieved = 3K fp32-equivalent flop per pair over the time of '
                             'the whole count (thresholds + SAD kernel + recheck), peak = fp32 VALU; the SAD kernel itself '
                             'issues K/2 half-rate integer ops per pair and, since r03, sweeps the distinct query rows only -- '
                             'frac_fp32_equivalent > 1 means: faster than ANY fp32 VALU kernel could do the same algorithmic work'}
            extra.update(sad_extra)
        else:
            alg_flops = 3 * K               # sub, mul|abs, add on the VALU
            kname, ksym, peak, bound = 'lp_direct_kernel (fp32 VALU)', 'lp_direct_kernel', PEAK_FP32_TFLOPS, 'valu'
        achieved = alg_flops * pairs / kern_s / 1e12
        # HBM-side bytes of the dominant kernel per launch: two rocprofv3 --pmc passes OF THIS RUN (FETCH_SIZE and
        # WRITE_SIZE cannot share a pass; corrections per MI355X_MICROARCH.md), else the checked-in figure, labelled
        traffic, traffic_src = None, None
        pmc = {}
        args.level_used = int(prob.split.get('level', 0)) if split else args.split_level   # the counters' run uses the same level
        if not args.no_traffic and not multi:
            traffic = _measure_traffic(args, ksym, pmc if bound == 'mfma' else None)
            traffic_src = None if traffic is None else 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes spawned by this run ' \
                                                       '(2 x FETCH_SIZE + WRITE_SIZE, KB -> B; MI355X_MICROARCH.md gfx950 note)'
        if traffic is None and not args.no_traffic:      # (measurement failed: the checked-in figure, labelled as such)
            tfile = os.path.join(ROOT, 'profiles', 'traffic.json')
            if os.path.exists(tfile):
                try:
                    traffic = (json.load(open(tfile)).get(args.workload + ('' if split else ':no-split')) or {}).get('bytes_per_launch')
                    traffic_src = None if traffic is None else 'static: profiles/traffic.json (rocprofv3 --pmc of an earlier run)'
                except Exception:
                    traffic = None
        # algorithmic operand bytes of one launch: every operand row read once (fp32, or 4 B / element split cells)
        kcols = ((K + 1 + 15) // 16 * 16) if split else K        # (split cells: f16 hi + lo = 4 B per element, + the norm column)
        alg_bytes = 4 * (B + n_ent) * kcols
        if split and int(prob.split.get('level', 0)) == 1:       # planar hi operands: 2 B per element, K + 2 columns padded to 64
            alg_bytes = 2 * (B + n_ent) * ((K + 2 + 63) // 64 * 64)
        roof = {'bound': bound, 'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s',
                'frac': round(achieved / peak, 4), 'traffic': traffic, 'traffic_source': traffic_src,
                'algorithmic_operand_bytes_per_launch': alg_bytes,
                'kernel': kname, 'kernel_ms': round(kern_s * 1e3, 4), 'timing': 'HIP events on the launch stream, %d launches' % reps,
                'pairs_per_launch': pairs, 'algorithmic_flops_per_pair': alg_flops}
        roof.update(extra)
        if 'sad_issue' in roof:     # TransE-L1: the roofline fraction is taken on the unit the kernel runs on
            roof['frac_fp32_equivalent'] = roof['frac']
            roof['frac'] = roof['sad_issue']['frac_of_issue_peak']
            roof['peak_is'] = 'v_sad_u16 issue rate (39.3 T lane-instructions/s); frac_fp32_equivalent = 3K flop per pair against the fp32 VALU peak'
        if pmc:
            pm = {'source': 'rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass spawned by this run '
                            '(eager launches, same weights and split level), summed over the kernel\'s instantiations per evaluate'}
            pm.update(pmc)
            if 'SQ_INSTS_MFMA' in pmc and split:
                pm['executed_TFLOPs_from_SQ_INSTS_MFMA'] = round(pmc['SQ_INSTS_MFMA'] * 32768 / kern_s / 1e12, 2)
                pm['executed_frac_from_SQ_INSTS_MFMA'] = round(pmc['SQ_INSTS_MFMA'] * 32768 / kern_s / 1e12 / peak, 4)
            roof['pmc'] = pm
            if 'mfma_busy_frac' in pmc:
                roof['mfma_busy_frac'] = pmc['mfma_busy_frac']
        if cols is not None:
            roof['query_columns'] = {'queries': int(cols.n_queries), 'distinct_rows': int(cols.n_distinct_keys),
                                     'columns': int(cols.n_columns), 'single': int(cols.n_single), 'grouped': int(cols.n_multi),
                                     'sets_per_grouped_column': int(cols.sets),
                                     'note': 'queries that share their key share the query row: the matrix-core sweep runs once '
                                             'per column (two launches: single-query columns, grouped columns), the threshold '
                                             'compare once per query; pairs_per_launch counts (query, candidate) pairs'}
        if power is not None:
            roof['package_power'] = power

    # ---- secondary numbers of the same hot path: scoring_function (K1) and corrupt_batch (K5) ----
    sec = None
    if rank == 0 and not args.no_secondary and getattr(model, '_row_shard', None) is None:
        Bt = 32768                                     # training batch of docs/tutorials/transe.rst:25
        h2, t2, r2 = orc.synthetic_triples(n_ent_full, n_rel, Bt, seed=3, device=device)

        def ev_time(fn, reps=20):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(device)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize(device)
            return e0.elapsed_time(e1) / 1e3 / reps
        with torch.no_grad():
            t_sf_call = ev_time(lambda: model.scoring_function(h2, t2, r2))
            # the kernel itself: 20 launches replayed as one hipGraph -- back-to-back eager calls are HOST bound below
            # B ~ 64k (the Python / ctypes path of one call takes ~20 us, the kernel ~11 us; measured flat 20-21 us
            # from B = 4096 to 32768 before this)
            gsf = torch.cuda.CUDAGraph()
            sside = torch.cuda.Stream(device)
            sside.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(sside):
                model.scoring_function(h2, t2, r2)
            torch.cuda.current_stream(device).wait_stream(sside)
            with torch.cuda.graph(gsf):
                for _ in range(20):
                    model.scoring_function(h2, t2, r2)
            t_sf = ev_time(gsf.replay, reps=5) / 20
        bytes_per_triple = {'complex': 24 * d + 28, 'transh': 16 * d + 28, 'transd': 20 * d + 28}.get(kind, 12 * d + 28)
        samp = tk.BernoulliNegativeSampler(kg)
        t_cb = ev_time(lambda: samp.corrupt_batch(h2, t2, r2), reps=10)
        # K1 against HBM proper (the dataset-sized tables live in the 256 MiB Infinity Cache): the same kernel
        # on a table that exceeds it -- 2,000,000 entities (1.6 GB per fp32 table at d = 200) and 1,048,576 triples with
        # independent random ids, so one launch touches ~2 x B distinct rows (>= 1.6 GB) and nothing survives in the
        # caches from launch to launch
        hbm = None
        try:
            n_big, b_big = 2000000, 1 << 20
            torch.manual_seed(11)
            ctor = {'transe': lambda: tk.TransEModel(d, n_big, n_rel, 'L%d' % p), 'transh': lambda: tk.TransHModel(d, n_big, n_rel),
                    'transd': lambda: tk.TransDModel(d, d, n_big, n_rel), 'distmult': lambda: tk.DistMultModel(d, n_big, n_rel),
                    'complex': lambda: tk.ComplExModel(d, n_big, n_rel)}[kind]
            big = ctor().to(device)
            hb, tb, rb = orc.synthetic_triples(n_big, n_rel, b_big, seed=4, device=device)
            with torch.no_grad():
                t_big = ev_time(lambda: big.scoring_function(hb, tb, rb), reps=10)
            hbm = {'n_ent': n_big, 'batch': b_big, 'entity_table_bytes': big.entity_table_bytes(), 'ms': round(t_big * 1e3, 4),
                   'hbm_GBps': round(b_big * bytes_per_triple / t_big / 1e9, 1),
                   'frac_of_8TBps': round(b_big * bytes_per_triple / t_big / 1e9 / PEAK_HBM_GBS, 4),
                   'note': 'entity tables exceed the 256 MiB Infinity Cache and a launch touches > 1.6 GB of distinct rows: '
                           'this is the HBM figure (relation rows, 1/3 of the algorithmic bytes for TransE, still come from L2)'}
            del big, hb, tb, rb
            torch.cuda.empty_cache()
        except Exception as exc:
            hbm = {'error': str(exc)[:200]}
        sec = {'scoring_function': {'triples_per_s': round(Bt / t_sf, 1), 'batch': Bt, 'ms': round(t_sf * 1e3, 4),
                                    'timing': 'device time per launch (20 launches replayed as one hipGraph)',
                                    'ms_per_eager_call_host_bound': round(t_sf_call * 1e3, 4),
                                    'algorithmic_bytes_per_triple': bytes_per_triple,
                                    'cache_GBps': round(Bt * bytes_per_triple / t_sf / 1e9, 1),
                                    'cache_note': 'dataset-sized tables (%.1f MB) sit in the L2 / 256 MiB Infinity Cache: this is '
                                                  'cache bandwidth, not HBM' % (table_bytes_full / 1e6),
                                    'hbm': hbm},
               'corrupt_batch': {'samples_per_s': round(Bt / t_cb, 1), 'batch': Bt, 'ms': round(t_cb * 1e3, 4),
                                 'note': 'includes the reference-compatible mask.sum().item() host sync'}}

    # ---- the reference idiom: a NEW LinkPredictionEvaluator(model, kg) per validation (evaluation.py:252-262) ----
    ev_ranks = [ev.rank_true_heads, ev.rank_true_tails, ev.filt_rank_true_heads, ev.filt_rank_true_tails]
    fresh_loop = None
    if rank == 0 and world == 1 and not args.only_timed and not args.materialize:
        try:
            import copy as _copy
            m2 = _copy.deepcopy(model)          # a model this process has not evaluated yet (same tables)
            times, same = [], True
            for it_ in range(8):
                ev_f = tk.LinkPredictionEvaluator(m2, kg_test)      # default options: graph 'auto'
                sync()
                t1 = time.perf_counter()
                ev_f.evaluate(args.batch, verbose=False)
                sync()
                times.append(round((time.perf_counter() - t1) * 1e3, 3))
                same = same and all(torch.equal(a, b) for a, b in zip(
                    ev_ranks, [ev_f.rank_true_heads, ev_f.rank_true_tails, ev_f.filt_rank_true_heads, ev_f.filt_rank_true_tails]))
            fresh_loop = {'ms_per_iteration': times, 'ranks_identical_to_the_timed_evaluator': same,
                          'what': 'a fresh LinkPredictionEvaluator(model, kg_test) built in every iteration on a model copy '
                                  'this process had not evaluated: plans, level and hipGraph live per (model, kg) at module '
                                  'level (evaluation._EvalState), so iteration 1 is the eager warm-up, 2 captures (level '
                                  'switch + second stream), 3.. replay'}
            del m2
        except Exception as exc:
            fresh_loop = {'error': str(exc)[:200]}

    # ---- reference CPU path (oracle) on a bounded sample, rank 0, N = 1 only ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and tables is not None:
        th, tt_, tr = kg_test.head_idx.cpu(), kg_test.tail_idx.cpu(), kg_test.relations.cpu()
        dh, dtl = kg.dict_of_heads, kg.dict_of_tails

        ref_pkg = _load_reference()
        # TransH / TransD: the reference's evaluate_projections loops over all N entities in Python and caches an (R, N, d)
        # tensor (2.76 GB at cfg2's shape) -- timed as part of its evaluate(), as a user would pay it
        use_ref = ref_pkg is not None

        def cpu_run(off, bs, ties=False):
            if use_ref and not ties:
                evr = reference_evaluator(ref_pkg, kind, p, tables, n_ent_full, info['n_rel'], th[off:off + bs],
                                          tt_[off:off + bs], tr[off:off + bs], dh, dtl)
                c0 = time.perf_counter()
                with torch.no_grad():
                    evr.evaluate(b_size=bs, verbose=False)
                return time.perf_counter() - c0, (evr.rank_true_heads, evr.rank_true_tails, evr.filt_rank_true_heads,
                                                  evr.filt_rank_true_tails)
            c0 = time.perf_counter()
            o = orc.lp_evaluate(kind, tables, th[off:off + bs], tt_[off:off + bs], tr[off:off + bs], dh, dtl, bs, p,
                                tie_tol=2e-5 if ties else None)
            return time.perf_counter() - c0, o
        # thread count: all host cores is not always the fastest for these memory-bound (b, N, d) temporaries
        thread_sweep, best_thr = {}, None
        for nthr in sorted({os.cpu_count(), min(os.cpu_count(), 64), min(os.cpu_count(), 16)}, reverse=True):
            torch.set_num_threads(nthr)
            cpu_run(0, 32)                              # warms the allocator / threads (first touch is page-fault bound)
            dt_, _ = cpu_run(0, 32)
            thread_sweep[nthr] = round(32 * 2 * n_ent_full / dt_, 1)
            if best_thr is None or thread_sweep[nthr] > thread_sweep[best_thr]:
                best_thr = nthr
        torch.set_num_threads(best_thr)
        sweep, best, off = {}, None, 0
        cpu_r, gpu_r, ties_all = [], [], []
        ref_vs_port_diff = 0
        for bs in (32, 64, 128, 256):           # the reference is strongly non-monotonic in b
            if off + bs > n_test:
                break
            if bs > 32:
                cpu_run(off, bs)                        # untimed first touch of this size's temporaries
            if use_ref:
                # the timed run is the REFERENCE's own evaluate(); its ranks must equal the port's (which also supplies
                # the tie intervals the GPU ranks are judged by)
                dt_, (rrh, rrt, rfrh, rfrt) = cpu_run(off, bs)
                _, (rh, rt, frh, frt, ties) = cpu_run(off, bs, ties=True)
                ref_vs_port_diff += int((torch.stack([rrh, rrt, rfrh, rfrt]) != torch.stack([rh, rt, frh, frt])).sum())
            else:
                dt_, (rh, rt, frh, frt, ties) = cpu_run(off, bs, ties=True)
            sweep[bs] = round(bs * 2 * n_ent_full / dt_, 1)
            if best is None or sweep[bs] > sweep[best]:
                best = bs
            cpu_r.append(torch.stack([rh, rt, frh, frt]))
            gpu_r.append(torch.stack([x[off:off + bs] for x in ev_ranks]))
            ties_all.append(ties)
            off += bs
        cpu_r, gpu_r, ties_all = torch.cat(cpu_r, 1), torch.cat(gpu_r, 1), torch.cat(ties_all, 1)
        n_diff = int((gpu_r != cpu_r).sum())
        in_tie = bool(((gpu_r >= ties_all[..., 0]) & (gpu_r <= ties_all[..., 1])).all())
        mo = orc.lp_metrics(*cpu_r, 10)
        mg = orc.lp_metrics(*gpu_r, 10)
        cpu = {'value': sweep[best], 'unit': 'triples_scored/s',
               'cores': torch.get_num_threads(), 'host_cores': os.cpu_count(), 'kind': 'reference' if use_ref else 'port',
               'sample': ('one batch per b_size in {32,64,128,256} (%d of %d test triples; each size run twice, the second timed), '
                          % (off, n_test)) +
                         ('torchkge %s itself (staged from /root/reference as oracle/_ref by oracle/Makefile): its models, '
                          'KnowledgeGraph and LinkPredictionEvaluator.evaluate on CPU tensors with this run\'s tables; '
                          % getattr(ref_pkg, '__version__', '?') if use_ref else
                          'oracle.lp_evaluate = the reference algorithm on torch CPU ops (oracle/_ref absent: no autograd graph, '
                          'measured 1.27x faster than the real reference in the build container, BASELINE.md); ') +
                         'value = the best b_size (%d) at the best thread count (%d)' % (best, best_thr),
               'reference_ranks_differing_from_port': ref_vs_port_diff if use_ref else None,
               'b_size_sweep': sweep, 'thread_sweep_b32': thread_sweep,
               'ranks_equal_to_gpu': n_diff == 0, 'ranks_differing': n_diff, 'ranks_compared': int(cpu_r.numel()),
               'gpu_ranks_within_reference_tie_interval_2e-5': in_tie,
               'filt_hits10_cpu_gpu': [mo['hit_at_k'][1], mg['hit_at_k'][1]],
               'filt_mrr_cpu_gpu': [mo['mrr'][1], mg['mrr'][1]]}

    if rank == 0 and world == 1 and not args.no_cpu_baseline and tables is None:
        # cfg5-size: the reference's (b, N, d) temporaries are 9.4 GB each at b = 1 -> a few facts at b = 1 on the host
        # cores, rate extrapolated linearly in n_test; needs ~60 GB of host memory
        try:
            import psutil
            if psutil.virtual_memory().available > 120 * 2 ** 30:
                nf = 2
                cpu_tabs = [x.data.cpu() for x in model._tables()]
                th, tt_, tr = kg_test.head_idx[:nf].cpu(), kg_test.tail_idx[:nf].cpu(), kg_test.relations[:nf].cpu()
                dh, dtl = orc.filter_dicts_for_facts(kg.head_idx, kg.tail_idx, kg.relations, th, tt_, tr)
                torch.set_num_threads(min(os.cpu_count(), 64))
                with torch.no_grad():
                    orc.lp_evaluate(kind, cpu_tabs, th[:1], tt_[:1], tr[:1], dh, dtl, 1, p)       # first touch
                    c0 = time.perf_counter()
                    rh, rt, frh, frt = orc.lp_evaluate(kind, cpu_tabs, th, tt_, tr, dh, dtl, 1, p)
                    dt_ = time.perf_counter() - c0
                cpu_r = torch.stack([rh, rt, frh, frt])
                gpu_r = torch.stack([x[:nf] for x in ev_ranks])
                cpu = {'value': round(nf * 2 * n_ent_full / dt_, 1), 'unit': 'triples_scored/s',
                       'cores': torch.get_num_threads(), 'host_cores': os.cpu_count(), 'kind': 'port',
                       'sample': '%d test facts at b_size 1 (%.1f s; the (1, N, d) temporaries are 9.4 GB each), oracle.lp_evaluate '
                                 '= the reference algorithm on torch CPU ops' % (nf, dt_),
                       'ranks_differing': int((cpu_r != gpu_r).sum()), 'ranks_compared': int(cpu_r.numel())}
                del cpu_tabs
        except Exception as exc:
            cpu = {'error': str(exc)[:200]}

    # ---- every rank of the whole test split vs the reference algorithm on ATen GPU ops ----
    parity = None
    if rank == 0 and world == 1 and not args.no_full_parity and tables is not None:
        parity = full_split_parity(info, tables, kg, kg_test, ev_ranks, device)
        if not parity['within_reference_tie_interval_2e-5'] or parity['abs_diff_filt_mrr'] >= 1e-5 \
                or parity['abs_diff_filt_hits10'] >= 1e-5 + parity['filtered_ranks_across_the_hits10_boundary'] * 0.5 / n_test:
            raise SystemExit('bench: full-split parity against the reference algorithm failed: %s' % json.dumps(parity))

    if rank == 0 and world == 1 and not args.no_full_parity and tables is None and args.parity_sample > 0:
        parity = sample_parity(model, info, kg, kg_test, ev_ranks, device, n=args.parity_sample, b=2)
        if not parity['within_reference_tie_interval_2e-5'] or parity['abs_diff_filt_mrr'] >= 1e-5:
            raise SystemExit('bench: subsample parity against the reference algorithm failed: %s' % json.dumps(parity))

    # ---- training step through the same kernels (last: it changes the tables) ----
    if rank == 0 and sec is not None:
        crit = tk.MarginLoss(0.5)
        opt = torch.optim.SGD(model.parameters(), lr=1e-3)

        def train_step():
            nh, nt = samp.corrupt_batch(h2, t2, r2)            # K5 (+ torch RNG draws)
            pos, neg = model(h2, t2, r2, nh, nt)               # K1 forward x2
            loss = crit(pos, neg)
            opt.zero_grad(set_to_none=True)
            loss.backward()                                     # K1 backward x2 (atomic scatter)
            opt.step()
        t_tr = ev_time(train_step, reps=10)
        sec['train_step'] = {'triples_per_s': round(Bt / t_tr, 1), 'batch': Bt, 'ms': round(t_tr * 1e3, 4),
                             'what': 'corrupt_batch + Model.forward(pos,neg) + MarginLoss + backward + SGD step',
                             'note': 'eager: ~30 launches per step, host bound at this batch size'}
        # the same step as ONE hipGraph (sampler in its sync-free form: no host read of the mask sum): device time
        try:
            samp.sync_free = True
            for prm in model.parameters():
                prm.grad = torch.zeros_like(prm)

            def graph_step():
                nh, nt = samp.corrupt_batch(h2, t2, r2)
                pos, neg = model(h2, t2, r2, nh, nt)
                loss = crit(pos, neg)
                for prm in model.parameters():
                    prm.grad.zero_()
                loss.backward()
                opt.step()
            sside = torch.cuda.Stream(device)
            sside.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(sside):
                for _ in range(3):
                    graph_step()
            torch.cuda.current_stream(device).wait_stream(sside)
            gtr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gtr):
                graph_step()
            t_gr = ev_time(gtr.replay, reps=20)
            sec['train_step_hipgraph'] = {'triples_per_s': round(Bt / t_gr, 1), 'batch': Bt, 'ms': round(t_gr * 1e3, 4),
                                          'what': 'the same step captured once and replayed as one hipGraph '
                                                  '(sync-free sampler), device time'}
        except Exception as exc:      # a capture problem must not cost the bench line
            sec['train_step_hipgraph'] = {'error': str(exc)[:200]}
        finally:
            samp.sync_free = False

    if rank == 0:
        if not multi:
            par = 'single'
        elif replicas:
            par = 'independent-replicas-%d' % world
        elif shard == 'entities':
            par = 'entity-shards-%d (row-sharded tables), RCCL %s' % (
                world, 'all-reduce of the (3, 2B) rank counts' if args.exchange == 'counts'
                else 'all-to-all of the partial (2B, N/P) score tiles: each rank receives and ranks the rows of its 2B/P queries')
        else:
            par = 'query-shards-%d' % world
        used_split = bool(roof and 'executed_frac' in roof)
        dtype = 'f32 (f16 hi/lo-split MFMA prefilter + exact f32 recheck; ranks bit-identical to f32)' \
            if used_split else ('f32 (u16 fixed-point SAD prefilter + exact f32 recheck; ranks bit-identical to f32)'
                                if roof and 'lp_l1_sad' in roof.get('kernel', '') else 'f32')
        line = {
            'metric': 'link-prediction triples scored/sec (filtered LP eval, both sides)',
            'value': round(value, 1), 'unit': 'triples_scored/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 4),
            'clock_settle': {'untimed_steps_before_warmup': settle_steps, 'target_ms': args.settle_ms},
            'first_evaluate_ms': round(first_ms, 2), 'first_evaluate_parts': first_parts, 'cold_ms_per_step': round(cold_ms, 4),
            'first_evaluate_what': 'wall time of the first evaluate() of a fresh evaluator: device-side filter index, FilterPlans, '
                                   'MFMA self-test, eager launches, ranks to the host; cold_ms_per_step: 5 graph replays after a '
                                   '0.5 s idle gap, before the clock-settle phase',
            # (the contract's key; at N = 1 it only says which mode `--gpus N` would run: the default strong-scales this job)
            'higher_is_better': True, 'scaling': args.scaling,
            'vs_baseline': None, 'dtype': dtype, 'data': 'synthetic',
            'config': {'workload': '%s dim=%d L%d on %s-shaped synthetic KG (N=%d, R=%d, test=%d), '
                                   'LinkPredictionEvaluator.evaluate(b_size=%d)' % (
                                       kind, d, p, shape + (' x%d entity shards' % world if ent_weak else ''), n_ent_full, n_rel, n_test, args.batch),
                       'parallelism': par, 'fused_rank': not args.materialize, 'hip_graph': (not args.no_graph) and (not multi or shard == 'queries' or
                                                               (shard == 'entities'
                                                                and not args.materialize and not args.no_both)),
                       'collectives_in_graph': bool(getattr(ev, 'graph_collectives', False)) and multi and shard == 'entities',
                       'scored_triples_per_step': total_units},
            'filtered_hits_at_10': hit10[1], 'filtered_mrr': mrr[1],
            'split_prefilter': {'level_of_the_timed_evaluations': level_timed,
                                'rescored_pairs_per_query': rescored_timed,
                                'policy': 'level 1 (one MFMA product per k16 unit, wider band) when the previous evaluation '
                                          're-scored <= %.0f pairs per query on three products; back to three products above %.0f '
                                          '(cfg2\'s 4 / 30 scaled by N / 14,541: both the sweep saved and the exact chains paid scale '
                                          'with the row width, their ratio with the candidates per query)'
                                          % tk.evaluation.level1_thresholds(n_ent_full)},
            'workload_detail': {'kg': args.kg, 'weights': weights, 'train': info.get('train'), 'train_s': info.get('train_s'),
                                'filter_lists': flt_stats},
            'roofline': roof, 'cpu_baseline': cpu, 'parity_full_split': parity, 'secondary': sec,
            'fresh_evaluator_per_validation': fresh_loop,
            'strong_scaling_model': strong_scaling_model(n_test, n_ent_full, d, elapsed / args.steps * 1e3 if world == 1 else None),
            'entity_tables': None if not multi else {
                'query_rows': (None if getattr(model, '_row_shard', None) is None else
                               ('rows of the %d distinct entities of the test facts summed over the ranks once per evaluate()'
                                % int(ev._qmap['uniq'].shape[0]) if getattr(ev, '_qmap', None) is not None and ev.query_exchange == 'evaluate'
                                else '(2B, K) query rows summed over the ranks per batch')),
                'layout': ('row-sharded: N/P rows per GPU, relation tables replicated' if (shard == 'entities' and args.tables == 'sharded')
                           else 'replicated'),
                'bytes_full': table_bytes_full, 'bytes_this_rank': model.entity_table_bytes()},
            'collective_time': headline_coll, 'other_exchange': other_x, 'query_partition': query_part, 'weak_mode': weak_mode,
            'cfg4_mode': cfg4_mode,
            'collectives_fallback': None if not multi else
            'KGE_EAGER_COLLECTIVES=1 runs every kernel and every RCCL call of a sharded evaluate() as ordinary eager launches (no '
            'graph segments, no captured collectives): the escape hatch if segment replay misbehaves on a multi-GPU node',
            'f32_mfma_only': None if f32_only_ms is None else {
                'ms_per_step': round(f32_only_ms, 4), 'value': round(total_units / f32_only_ms * 1e3, 1),
                'ranks_identical_to_headline_run': True},
        }
        _flush_c_stdio()
        if multi:
            dist.barrier()      # the other ranks have flushed their stdout too: the JSON line comes last
        print(json.dumps(line), flush=True)
    elif multi:
        _flush_c_stdio()
        dist.barrier()
    if multi:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
