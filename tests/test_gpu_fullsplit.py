"""GPU parity at BASELINE sizes over the WHOLE test split (run with -m gpu on an MI355X).

 * every one of the 4 x n_test ranks of LinkPredictionEvaluator.evaluate (evaluation.py:263-308)
   for cfg2 / cfg3 / cfg4 and for TransH / TransD / TransE-L1 at cfg2's shape (d = 200; the reference's
   (R, N, d) projection cache, translation.py:260-284 / :629-652, is 2.76 GB on the device) against the reference algorithm run on ATen GPU ops
   (oracle.lp_evaluate(device='cuda')), on Xavier weights AND on trained-like weights (true ranks
   small, near-ties dense): tie-interval containment, |dMRR|, |dHits@10| < 1e-5, f16-split == fp32;
 * the grouped / flattened filter correction (kge_lp_filter_sub_grouped) on heavy-tailed filter
   lists with hub keys: bit-exact against the materialised path and the C oracle.
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import kge_oracle as orc
from tests.helpers import oracle_clib, fptr, dict_to_csr

pytestmark = pytest.mark.gpu
i64 = ctypes.c_int64


@pytest.fixture(scope='module')
def hip():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    from torchkge_amd import _hip
    _hip.load_library()
    return _hip


def _ranks(ev):
    return [ev.rank_true_heads.clone(), ev.rank_true_tails.clone(), ev.filt_rank_true_heads.clone(),
            ev.filt_rank_true_tails.clone()]


@pytest.mark.parametrize('workload', ['transe_fb15k237', 'complex_wn18rr', 'distmult_fb15k',
                                      'transh_fb15k237', 'transd_fb15k237', 'transe_l1_fb15k237'])
@pytest.mark.parametrize('weights', ['xavier', 'trained'])
def test_full_test_split_vs_gpu_resident_reference(hip, workload, weights):
    import bench
    import torchkge_amd as tk
    dev = torch.device('cuda', 0)
    model, tables, kg, kg_test, info = bench.build_workload(workload, dev, weights=weights, kg_kind='zipf',
                                                            train_cfg={'steps': 300})
    # The path that SHIPS is the steady state of a persistent evaluator: level 0 eagerly (the warm-up), then -- for a
    # fitted model -- the one-product level, captured as a hipGraph and replayed.  Evaluate until replays run and pin the
    # LAST ranks to the reference; every earlier call must have produced the same ranks, position by position.
    ev = tk.LinkPredictionEvaluator(model, kg_test)
    history, levels = [], []
    for _ in range(5):
        levels.append(ev._level)
        ev.evaluate(b_size=32768, verbose=False)
        history.append(_ranks(ev))
    split = history[-1]
    for earlier in history[:-1]:
        for a, b in zip(split, earlier):
            assert torch.equal(a, b)
    assert ev._graph is not None and ev._graph_key is not None, 'the last evaluations were hipGraph replays'
    if weights == 'trained' and info['p'] == 2:
        # (TransE-L1 counts through the SAD prefilter: no levels there)
        assert levels[0] == 0 and levels[-1] == 1 and ev._level == 1, (levels, ev.last_rescored_per_query)
    else:
        assert levels[-1] == 0 or info['p'] == 2, levels
    print('\n%s / %s: levels of the five evaluations %s, re-scored pairs per query on the last %.2f'
          % (workload, weights, levels, ev.last_rescored_per_query or 0.0))
    par = bench.full_split_parity(info, tables, kg, kg_test, split, dev)
    print('\n%s / %s: %d of %d ranks differ from the GPU-resident reference (max |d| = %d), %d outside the '
          'tie interval; filt MRR ref/hip = %.6f / %.6f, filt Hits@10 = %.6f / %.6f, median filt rank %.0f'
          % (workload, weights, par['ranks_differing'], par['ranks_compared'], par['max_abs_rank_diff'],
             par['outside_tie_interval'], par['filt_mrr_ref_hip'][0], par['filt_mrr_ref_hip'][1],
             par['filt_hits10_ref_hip'][0], par['filt_hits10_ref_hip'][1], par['median_filt_rank_ref']))
    assert par['ranks_compared'] == 4 * info['n_test']
    assert par['within_reference_tie_interval_2e-5'], par
    # Hits@10 is a step function of the ranks: ONE near-tie flip across rank 10 (inside the tie interval, checked
    # above) moves it by 0.5 / n_test (8.5e-6 at cfg4) -- allow exactly the flips that were counted, nothing else
    assert par['abs_diff_filt_mrr'] < 1e-5, par
    assert par['abs_diff_filt_hits10'] < 1e-5 + par['filtered_ranks_across_the_hits10_boundary'] * 0.5 / info['n_test'], par
    assert abs(par['mrr_ref_hip'][0] - par['mrr_ref_hip'][1]) < 1e-5
    if weights == 'trained':
        assert par['filt_hits10_ref_hip'][0] > 0.02, 'the trained-like model should rank its facts high'
    # the f16-split prefilter and the all-fp32 counts give the same ranks, position by position
    model.split_filter = False
    ev2 = tk.LinkPredictionEvaluator(model, kg_test)
    ev2.evaluate(b_size=32768, verbose=False)
    for a, b in zip(split, _ranks(ev2)):
        assert torch.equal(a, b)
    # ... and so does the hipGraph replay of the same evaluation
    model.split_filter = True
    ev3 = tk.LinkPredictionEvaluator(model, kg_test, graph=True)
    ev3.evaluate(b_size=32768, verbose=False)
    ev3.evaluate(b_size=32768, verbose=False)
    for a, b in zip(split, _ranks(ev3)):
        assert torch.equal(a, b)


@pytest.mark.parametrize('kind,d,n_ent,n_facts', [('transe', 128, 1000000, 3000000), ('distmult', 128, 1000000, 3000000),
                                                  ('complex', 512, 200000, 1000000)])
def test_large_entity_set_trained_like_vs_gpu_resident_reference(hip, kind, d, n_ent, n_facts):
    """N = 1,000,000 entities, TRAINED-like tables (a few hundred steps of the engine's own training path on a 3 M-fact Zipf
    graph with hub keys): at this N a 2e-5 score window around a random threshold holds tens of candidates, so tie-interval
    containment only says something when the true entities sit in the sparse upper tail -- which a fitted model's do.
    The steady-state path (one-product level, hipGraph replay) against the reference algorithm on ATen GPU ops for 512 of
    the 2,048 test facts (oracle.lp_evaluate with its (b, N, d) temporaries at b = 8), split == fp32 on all of them.
    r06: ComplEx d = 512 (K = 1024, BASELINE cfg5's row width: the one-product level runs on the CHUNKED-panel kernel
    lp_hi_chunk.hip) on 200,000 entities, trained-like."""
    import bench
    import torchkge_amd as tk
    dev = torch.device('cuda', 0)
    n_rel, n_test = 64, 2048
    tables = orc.init_tables(kind, n_ent, n_rel, d, seed=3)
    model = bench.make_model(kind, 2, tables, n_ent, n_rel).to(dev)
    hubs = ((5000, 'head'), (2500, 'tail'), (1200, 'head'), (600, 'tail'))
    heads, tails, rels = orc.synthetic_triples_zipf(n_ent, n_rel, n_facts, 77, hubs=hubs)
    ident_e, ident_r = {i: i for i in range(n_ent)}, {i: i for i in range(n_rel)}
    kg = tk.KnowledgeGraph(kg={'heads': heads, 'tails': tails, 'relations': rels}, ent2ix=ident_e, rel2ix=ident_r)
    # test facts: the last facts of the shuffled graph + facts of the heaviest (h, r) / (t, r) keys (long filter lists)
    sel = list(range(n_facts - n_test + 256, n_facts))
    for side_key in (heads * n_rel + rels, tails * n_rel + rels):
        uk, inv, c = torch.unique(side_key, return_inverse=True, return_counts=True)
        for k in torch.argsort(c, descending=True)[:4].tolist():
            sel += torch.nonzero(inv == k).view(-1)[:32].tolist()
    sel = torch.tensor(sel[:n_test], dtype=torch.long)
    kg_test = tk.KnowledgeGraph(kg={'heads': heads[sel].clone(), 'tails': tails[sel].clone(), 'relations': rels[sel].clone()},
                                ent2ix=ident_e, rel2ix=ident_r, _filter_src=kg._lazy)
    bench.train_like(model, kg, steps=300)
    info = {'kind': kind, 'p': 2, 'n_test': int(sel.shape[0])}
    ev = tk.LinkPredictionEvaluator(model, kg_test)
    history, levels = [], []
    for _ in range(5):
        levels.append(ev._level)
        ev.evaluate(b_size=32768, verbose=False)
        history.append(_ranks(ev))
    split = history[-1]
    for earlier in history[:-1]:
        for a, b in zip(split, earlier):
            assert torch.equal(a, b)
    # the oracle on the 256 hub facts (the tail of the selection) and the 256 facts in front of them
    pick = torch.cat([torch.arange(0, 256), torch.arange(int(sel.shape[0]) - 256, int(sel.shape[0]))])
    sub = tk.KnowledgeGraph(kg={'heads': kg_test.head_idx[pick].clone(), 'tails': kg_test.tail_idx[pick].clone(),
                                'relations': kg_test.relations[pick].clone()}, ent2ix=ident_e, rel2ix=ident_r,
                            _filter_src=kg._lazy)
    par = bench.sample_parity(model, dict(info, n_test=512), kg, sub, [x[pick] for x in split], dev, n=512, b=8)
    print('\nN = 1e6 %s d = %d trained-like: levels %s, re-scored pairs per query %.1f; %d of %d ranks differ from the '
          'GPU-resident reference (max |d| = %d), %d outside the tie interval; median filtered rank %.0f, filt Hits@10 %.4f; '
          '%d filter-list entries in the sample'
          % (kind, d, levels, ev.last_rescored_per_query or 0.0, par['ranks_differing'], par['ranks_compared'],
             par['max_abs_rank_diff'], par['outside_tie_interval'], par['median_filt_rank_ref'],
             par['filt_hits10_ref_hip'][0], par['filter_list_entries_of_the_sample']))
    assert par['ranks_compared'] == 4 * 512
    assert par['within_reference_tie_interval_2e-5'], par
    assert par['abs_diff_filt_mrr'] < 1e-5 and abs(par['mrr_ref_hip'][0] - par['mrr_ref_hip'][1]) < 1e-5, par
    assert par['abs_diff_filt_hits10'] < 1e-5 + par['filtered_ranks_across_the_hits10_boundary'] * 0.5 / 512, par
    assert par['median_filt_rank_ref'] < 0.05 * n_ent, 'the trained-like model should rank its facts high'
    assert par['filter_list_entries_of_the_sample'] > 5000
    if kind == 'complex':
        # whatever the policy chose: the same ranks with the one-product level (long rows: the chunked-panel kernel) forced
        assert hip.hi_stream_ok(2 * d) and (2 * d + 2 + 15) // 16 > 32
        model.split_level = 1
        ev1 = tk.LinkPredictionEvaluator(model, kg_test, share_state=False)
        for _ in range(2):
            ev1.evaluate(b_size=32768, verbose=False)
            assert model._use_level1()
            for a, b in zip(split, _ranks(ev1)):
                assert torch.equal(a, b)
        model.split_level = 'auto'
    model.split_filter = False
    ev2 = tk.LinkPredictionEvaluator(model, kg_test)
    ev2.evaluate(b_size=32768, verbose=False)
    for a, b in zip(split, _ranks(ev2)):
        assert torch.equal(a, b)


def test_cfg5_wikidata5m_shape_subsample_vs_gpu_resident_reference(hip):
    """BASELINE cfg5 (ComplEx d = 512, 4,594,485 entities, 822 relations; K = 1024: the regime where the f16-split band
    is widest) against the REFERENCE ALGORITHM, as SURVEY 8(d) prescribes for this size: 64 test facts at b_size = 2
    through oracle.lp_evaluate on ATen GPU ops (bilinear.py:501-556 with its (b, N, d) temporaries: 18.8 GB each at
    b = 2), filter sets of the full graph restricted to the looked-up keys (oracle.filter_dicts_for_facts).  All 256
    ranks inside the reference's tie interval, split == fp32, metrics equal.  Tables: uniform(-0.5, 0.5) -- scores of
    unit scale like a fitted model's (Xavier at N = 4.6 M gives scores ~1e-8, where an absolute tolerance says nothing)."""
    import bench
    import torchkge_amd as tk
    dev = torch.device('cuda', 0)
    model, kg, kg_test, info = bench.build_cfg5_sample(dev, n_facts=5000000, n_test=64)      # 5 M facts, hub keys: long filter lists
    ev = tk.LinkPredictionEvaluator(model, kg_test)
    ev.evaluate(b_size=32768, verbose=False)
    split = _ranks(ev)
    par = bench.sample_parity(model, info, kg, kg_test, split, dev, n=64, b=2)
    print('\ncfg5 subsample: %d of %d ranks differ from the GPU-resident reference (max |d| = %d), %d outside the tie '
          'interval; filt MRR ref/hip = %.6f / %.6f; median filtered rank %.0f; %d filter-list entries in the sample'
          % (par['ranks_differing'], par['ranks_compared'], par['max_abs_rank_diff'], par['outside_tie_interval'],
             par['filt_mrr_ref_hip'][0], par['filt_mrr_ref_hip'][1], par['median_filt_rank_ref'],
             par['filter_list_entries_of_the_sample']))
    assert par['ranks_compared'] == 256
    assert par['filter_list_entries_of_the_sample'] > 20000        # the filtered path is really exercised at N = 4.6 M
    assert par['within_reference_tie_interval_2e-5'], par
    assert par['abs_diff_filt_mrr'] < 1e-5 and abs(par['mrr_ref_hip'][0] - par['mrr_ref_hip'][1]) < 1e-5, par
    assert par['abs_diff_filt_hits10'] < 1e-5 + par['filtered_ranks_across_the_hits10_boundary'] * 0.5 / 64, par
    model.split_filter = False
    ev2 = tk.LinkPredictionEvaluator(model, kg_test)
    ev2.evaluate(b_size=32768, verbose=False)
    for a, b in zip(split, _ranks(ev2)):
        assert torch.equal(a, b)


@pytest.mark.parametrize('kind,p', [('transe', 2), ('transe', 1), ('complex', 2), ('transh', 2)])
def test_grouped_filter_correction_with_hub_keys_bit_exact(hip, kind, p):
    """Hub keys (lists of 3000 / 1500 entities shared by hundreds of queries), duplicate keys with
    different true entities, missing keys, true entity absent, entity shards: sub / found of
    kge_lp_filter_sub_grouped == kge_lp_filter_sub (per query walk) == the materialised
    filtered_rank_from_scores == the C oracle on the same score matrix."""
    import torchkge_amd as tk
    from torchkge_amd import distributed as kd
    from torchkge_amd.filter_index import FilterIndex
    from tests.test_gpu_parity import build_model
    n_ent, n_rel, d = 5000, 12, 64
    tables = orc.init_tables(kind, n_ent, n_rel, d, seed=2)
    m = build_model(kind, p, tables, n_ent, n_rel)
    h, t, r = orc.synthetic_triples_zipf(n_ent, n_rel, 40000, 17, hubs=((3000, 'tail'), (1500, 'tail'), (800, 'head')))
    B = 1500
    qh, qt, qr = h[-B:].clone(), t[-B:].clone(), r[-B:].clone()
    qh[3], qr[3] = n_ent - 1, n_rel - 1                  # (almost surely) a key the graph does not have
    qt[5] = (qt[5] + 1) % n_ent                          # true entity absent from its list
    dh, dt, _ = orc.build_filter_dicts(h, t, r)
    idx = FilterIndex.from_triples(h.numpy(), r.numpy(), t.numpy(), 'cuda')       # (h, r) -> tails
    H, T, R = qh.cuda(), qt.cuda(), qr.cuda()
    seg_lo, seg_hi = idx.lookup(H, R)
    assert int((seg_hi - seg_lo).max()) >= 3000
    prob = m.lp_problem(H, T, R, 'tail')
    s_true = prob.pair_scores(T)
    scores = prob.scores()
    raw = prob.count_ge(s_true)
    sub_g, found_g = prob.filter_sub(s_true, T, seg_lo, seg_hi, idx.targets, grouped=True)
    sub_w, found_w = prob.filter_sub(s_true, T, seg_lo, seg_hi, idx.targets, grouped=False)
    assert torch.equal(sub_g, sub_w) and torch.equal(found_g, found_w)
    # ... and with the grouping precomputed once (FilterPlan, what the evaluator keeps per batch)
    from torchkge_amd.filter_index import FilterPlan
    plan = FilterPlan(seg_lo, seg_hi, T, idx.targets)
    assert plan.n_long > 0 and plan.n_pairs <= int(idx.targets.shape[0]) and plan.n_pairs < int((seg_hi - seg_lo).sum())
    for _ in range(2):
        sub_p, found_p = prob.filter_sub(s_true, T, seg_lo, seg_hi, idx.targets, plan=plan)
        assert torch.equal(sub_p, sub_w) and torch.equal(found_p, found_w)
    rk, frk = hip.filtered_rank_from_scores(scores, T, seg_lo, seg_hi, idx.targets)
    rk2, frk2 = hip.rank_finalize(raw, sub_g, found_g)
    assert torch.equal(rk, rk2) and torch.equal(frk, frk2)
    # C oracle (integer semantics of get_rank(filter_scores())) on the same score matrix
    lib = oracle_clib()
    has, off, tgt = dict_to_csr(dt, qh, qr)
    sn = np.ascontiguousarray(scores.cpu().numpy())
    ork = np.empty(B, dtype=np.int64); ofrk = np.empty(B, dtype=np.int64)
    lib.orc_filtered_rank(fptr(sn), i64(B), i64(n_ent), fptr(qt.numpy()), fptr(has), fptr(off), fptr(tgt),
                          fptr(ork), fptr(ofrk))
    assert np.array_equal(frk.cpu().numpy(), ofrk) and np.array_equal(rk.cpu().numpy(), ork)
    assert int(found_g[3]) == 0 and int(found_g[5]) == 0 and int(sub_g[5]) >= 0
    # entity shards: partial sub / found add up to the unsharded ones
    if kind != 'transe' or p == 2:
        acc = torch.zeros(2, B, dtype=torch.int32, device='cuda')
        for pidx in range(3):
            lo, hi = kd.shard_range(n_ent, 3, pidx)
            pp = m.lp_problem(H, T, R, 'tail', ent_lo=lo, ent_hi=hi)
            s_, f_ = pp.filter_sub(s_true, T, seg_lo, seg_hi, idx.targets, grouped=True)
            s2_, f2_ = pp.filter_sub(s_true, T, seg_lo, seg_hi, idx.targets, plan=plan)
            assert torch.equal(s_, s2_) and torch.equal(f_, f2_)
            acc[0] += s_; acc[1] += f_
        assert torch.equal(acc[0], sub_g) and torch.equal(acc[1], found_g)


@pytest.mark.parametrize('coalesce_mode', ['literal', 'default'])
def test_evaluator_on_skewed_graph_equals_materialised_path(hip, coalesce_mode):
    """LinkPredictionEvaluator on a Zipf graph with hubs: fused (grouped filter correction, both sides
    as one batch) == side by side == materialised score matrices, rank for rank."""
    import bench
    import torchkge_amd as tk
    from tests.test_gpu_parity import build_model
    n_ent, n_rel, d = 3000, 9, 48
    for kind in ('transe', 'distmult'):
        tables = orc.init_tables(kind, n_ent, n_rel, d, seed=4)
        m = build_model(kind, 2, tables, n_ent, n_rel)
        h, t, r = orc.synthetic_triples_zipf(n_ent, n_rel, 30000, 23, hubs=((2000, 'head'), (1000, 'tail')))
        kg = tk.KnowledgeGraph(kg={'heads': h, 'tails': t, 'relations': r}, ent2ix={i: i for i in range(n_ent)},
                               rel2ix={i: i for i in range(n_rel)})
        _, kg_test = kg.split_kg(sizes=(27000, 3000))
        res = []
        for kw in ({}, {'both_sides': False}, {'fused': False}, {'graph': True}):
            ev = tk.LinkPredictionEvaluator(m, kg_test, **kw)
            ev.evaluate(b_size=1024, verbose=False)
            res.append(_ranks(ev))
        for other in res[1:]:
            for a, b in zip(res[0], other):
                assert torch.equal(a, b)
        dh, dt, _ = orc.build_filter_dicts(h, t, r)
        rh, rt, frh, frt, ties = orc.lp_evaluate(kind, tables, kg_test.head_idx, kg_test.tail_idx, kg_test.relations,
                                                 dh, dt, 256, 2, tie_tol=2e-5, device='cuda')
        ref = torch.stack([rh, rt, frh, frt])
        got = torch.stack(res[0])
        assert ((got >= ties[..., 0]) & (got <= ties[..., 1])).all()
        assert int((ref != got).sum()) <= 0.001 * ref.numel()


def test_device_side_filter_build_fb15k237_and_wikidata5m_scale(hip, tmp_path):
    """the filter CSR built ON the GPU (FilterIndex.from_triples_torch: sort /
    unique_consecutive) == the reference's dict-of-sets build (data_structures.py:386-397, restated
    by oracle.build_filter_dicts) at FB15k-237 size, through the on-disk cache too; and the build at
    Wikidata5M size (20.6 M facts; the reference's per-fact loop takes ~8 min) with its time printed."""
    import time
    from torchkge_amd.filter_index import FilterIndex, KEY2_SPAN
    import torchkge_amd as tk
    n_ent, n_rel, ntr, nv, nte = orc.DATASET_SHAPES['fb15k237']
    h, t, r = orc.synthetic_triples_zipf(n_ent, n_rel, ntr + nv + nte, 1001)
    dh, dt, _ = orc.build_filter_dicts(h, t, r)
    for dic, k1, v in ((dt, h, t), (dh, t, h)):
        a = FilterIndex.from_triples_torch(k1, r, v, 'cuda')
        b = FilterIndex.from_dict(dic, 'cuda')                 # from the reference-style dict
        c = FilterIndex.from_triples(k1.numpy(), r.numpy(), v.numpy(), 'cuda')
        n_t = int(a.offsets[-1])
        assert n_t == sum(len(s) for s in dic.values())
        assert torch.equal(a.keys, b.keys) and torch.equal(a.offsets, b.offsets)
        assert torch.equal(a.keys, c.keys) and torch.equal(a.offsets, c.offsets) and torch.equal(a.targets, c.targets)
        # a dict's sets are unordered: compare segment contents as sorted lists
        off = a.offsets.cpu().numpy()
        ta, tb = a.targets.cpu().numpy()[:n_t], b.targets.cpu().numpy()[:n_t]
        seg = np.repeat(np.arange(len(off) - 1), np.diff(off))
        assert np.array_equal(ta, tb[np.lexsort((tb, seg))])
        path = a.save(str(tmp_path / 'idx.npz'))
        d = FilterIndex.load(path, 'cuda')
        assert torch.equal(d.keys, a.keys) and torch.equal(d.offsets, a.offsets) and torch.equal(d.targets[:n_t], a.targets[:n_t])
    # KnowledgeGraph uses the device build + the disk cache transparently
    kg = tk.KnowledgeGraph(kg={'heads': h, 'tails': t, 'relations': r}, ent2ix={i: i for i in range(n_ent)},
                           rel2ix={i: i for i in range(n_rel)})
    kg.set_filter_cache(str(tmp_path / 'cache'))
    i1 = kg.filter_index('tails', torch.device('cuda', 0))
    kg2 = tk.KnowledgeGraph(kg={'heads': h, 'tails': t, 'relations': r}, ent2ix=kg.ent2ix, rel2ix=kg.rel2ix)
    kg2.set_filter_cache(str(tmp_path / 'cache'))
    i2 = kg2.filter_index('tails', torch.device('cuda', 0))     # loaded from disk
    assert torch.equal(i1.keys, i2.keys) and torch.equal(i1.offsets, i2.offsets)

    # Wikidata5M size
    n_ent, n_rel, ntr, nv, nte = orc.DATASET_SHAPES['wikidata5m']
    n = ntr + nv + nte
    h, t, r = orc.synthetic_triples(n_ent, n_rel, n, 5)
    hc, tc, rc = h.cuda(), t.cuda(), r.cuda()
    FilterIndex.from_triples_torch(hc[:1000], rc[:1000], tc[:1000], 'cuda')
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    it = FilterIndex.from_triples_torch(hc, rc, tc, 'cuda')
    ih = FilterIndex.from_triples_torch(tc, rc, hc, 'cuda')
    torch.cuda.synchronize()
    dt_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    it2 = FilterIndex.from_triples_torch(h, r, t, 'cuda')       # host triples: the copy included
    torch.cuda.synchronize()
    dt_host = time.perf_counter() - t0
    print('\nfilter CSR build at Wikidata5M size (%d facts): both sides %.3f s from device-resident triples; '
          'one side %.3f s from host triples (H2D copy included); reference evaluate_dicts: ~23 us/fact = ~%.0f s'
          % (n, dt_dev, dt_host, 23e-6 * n))
    for ix in (it, ih):
        assert bool((ix.keys[1:] > ix.keys[:-1]).all())                      # sorted, unique keys
        assert bool((ix.offsets[1:] > ix.offsets[:-1]).all()) and int(ix.offsets[0]) == 0
        assert int(ix.offsets[-1]) <= n
    assert torch.equal(it.keys, it2.keys) and torch.equal(it.targets, it2.targets)
    # spot check against the numpy build on a 1 M-fact prefix, and membership of a few facts
    sub = slice(0, 1000000)
    a = FilterIndex.from_triples_torch(hc[sub], rc[sub], tc[sub], 'cuda')
    c = FilterIndex.from_triples(h[sub].numpy(), r[sub].numpy(), t[sub].numpy(), 'cuda')
    assert torch.equal(a.keys, c.keys) and torch.equal(a.offsets, c.offsets) and torch.equal(a.targets, c.targets)
    lo, hi = it.lookup(hc[:64], rc[:64])
    for i in range(64):
        assert int(t[i]) in it.targets[int(lo[i]):int(hi[i])].tolist()


def test_dissimilarities_are_differentiable(hip):
    """l1 / l2_dissimilarity carry an autograd graph like the reference's torch expressions
    (utils/dissimilarities.py:11-25): a user model may call them inside its scoring function."""
    from torchkge_amd.utils import l1_dissimilarity, l2_dissimilarity
    g = torch.Generator().manual_seed(0)
    a = torch.randn(37, 1, 24, generator=g).cuda().requires_grad_(True)
    b = torch.randn(37, 5, 24, generator=g).cuda().requires_grad_(True)
    for fn, ref in ((l1_dissimilarity, orc.l1_dissimilarity), (l2_dissimilarity, orc.l2_dissimilarity)):
        ac, bc = a.detach().cpu().requires_grad_(True), b.detach().cpu().requires_grad_(True)
        wgt = torch.randn(37, 5, generator=g)
        out = fn(a, b)
        want = ref(ac, bc)
        assert (out.detach().cpu() - want.detach()).abs().max().item() < 1e-4
        (out * wgt.cuda()).sum().backward()
        (want * wgt).sum().backward()
        assert (a.grad.cpu() - ac.grad).abs().max().item() < 1e-4
        assert (b.grad.cpu() - bc.grad).abs().max().item() < 1e-4
        a.grad = b.grad = None
    with torch.no_grad():
        assert not l2_dissimilarity(a, b).requires_grad
