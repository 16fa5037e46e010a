# -*- coding: utf-8 -*-
"""Generate the golden fixtures in this directory by RUNNING THE REAL REFERENCE
(torchkge v0.17.7 imported from /root/reference, CPU).  Run in the build
container only (the reference does not exist on the GPU box):

    python tests/golden/make_golden.py

Outputs (committed): tests/golden/ref_<kind>.npz, ref_sampler.npz, ref_toy.npz.
Each file holds the inputs (tables, index vectors, seeds) and the reference's
outputs for: scoring_function, inference_scoring_function on both sides,
LinkPredictionEvaluator ranks + metrics, BernoulliNegativeSampler.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, '/root/reference')
import pandas as pd  # noqa: E402
import torchkge  # noqa: E402
from torchkge.data_structures import KnowledgeGraph  # noqa: E402
from torchkge.evaluation import LinkPredictionEvaluator, RelationPredictionEvaluator  # noqa: E402
from torchkge.models import (TransEModel, TransHModel, TransDModel,  # noqa: E402
                             DistMultModel, ComplExModel)
from torchkge.sampling import BernoulliNegativeSampler, UniformNegativeSampler  # noqa: E402
from torchkge.utils import (get_rank, filter_scores, l1_dissimilarity,  # noqa: E402
                            l2_dissimilarity)
from torchkge.utils.operations import get_bernoulli_probs  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
assert torchkge.__version__ == '0.17.7'

N_ENT, N_REL, DIM, DIM_REL = 300, 7, 32, 24
N_FACTS, N_TEST, B = 1500, 64, 16


def make_kg(seed):
    g = torch.Generator().manual_seed(seed)
    # skew entity draws so that filter sets are not almost empty
    w = 1.0 / torch.arange(1, N_ENT + 1).float()
    heads = torch.multinomial(w, N_FACTS, replacement=True, generator=g)
    tails = torch.multinomial(w, N_FACTS, replacement=True, generator=g)
    rels = torch.randint(0, N_REL, (N_FACTS,), generator=g)
    # make sure every entity / relation id exists so n_ent / n_rel are as planned
    heads[:N_ENT] = torch.arange(N_ENT)
    rels[:N_REL] = torch.arange(N_REL)
    trip = torch.stack([heads, tails, rels], 1).unique(dim=0)
    perm = torch.randperm(trip.shape[0], generator=g)
    trip = trip[perm]
    df = pd.DataFrame(trip.numpy(), columns=['from', 'to', 'rel'])
    ent2ix = {i: i for i in range(N_ENT)}
    rel2ix = {i: i for i in range(N_REL)}
    kg = KnowledgeGraph(df=df, ent2ix=ent2ix, rel2ix=rel2ix)
    return kg


def sub_kg(kg, n):
    """Test split = last n facts, sharing the parent's full-graph filter dicts
    (what split_kg does, data_structures.py:236-238)."""
    return KnowledgeGraph(
        kg={'heads': kg.head_idx[-n:], 'tails': kg.tail_idx[-n:],
            'relations': kg.relations[-n:]},
        ent2ix=kg.ent2ix, rel2ix=kg.rel2ix, dict_of_heads=kg.dict_of_heads,
        dict_of_tails=kg.dict_of_tails, dict_of_rels=kg.dict_of_rels)


def build_model(kind, p):
    torch.manual_seed(0)
    if kind == 'transe':
        return TransEModel(DIM, N_ENT, N_REL, dissimilarity_type='L%d' % p)
    if kind == 'transh':
        return TransHModel(DIM, N_ENT, N_REL)
    if kind == 'transd':
        return TransDModel(DIM, DIM_REL, N_ENT, N_REL)
    if kind == 'distmult':
        return DistMultModel(DIM, N_ENT, N_REL)
    if kind == 'complex':
        return ComplExModel(DIM, N_ENT, N_REL)
    raise ValueError(kind)


def tables_of(kind, m):
    if kind == 'transe' or kind == 'distmult':
        t = [m.ent_emb.weight, m.rel_emb.weight]
    elif kind == 'transh':
        t = [m.ent_emb.weight, m.rel_emb.weight, m.norm_vect.weight]
    elif kind == 'transd':
        t = [m.ent_emb.weight, m.rel_emb.weight, m.ent_proj_vect.weight, m.rel_proj_vect.weight]
    else:
        t = [m.re_ent_emb.weight, m.im_ent_emb.weight, m.re_rel_emb.weight, m.im_rel_emb.weight]
    return [x.detach().clone().numpy() for x in t]


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None     # e.g. `relpred`: rewrite only ref_relpred.npz
    kg = make_kg(1234)
    kg_test = sub_kg(kg, N_TEST)
    common = dict(heads=kg.head_idx.numpy(), tails=kg.tail_idx.numpy(), rels=kg.relations.numpy(),
                  n_test=N_TEST, n_ent=N_ENT, n_rel=N_REL, b_size=B)

    for kind, p in [('transe', 2), ('transe', 1), ('transh', 2), ('transd', 2),
                    ('distmult', 2), ('complex', 2)]:
        if only is not None:
            break
        m = build_model(kind, p)
        # perturb tables a little so they are NOT exactly normalised: exercises
        # the on-the-fly normalisation of scoring_function vs raw tables at inference
        with torch.no_grad():
            for prm in m.parameters():
                if prm.requires_grad:
                    prm.mul_(1.0 + 0.05 * torch.sin(torch.arange(prm.numel()).float()).view_as(prm))
        tabs = tables_of(kind, m)
        h, t, r = kg_test.head_idx[:B], kg_test.tail_idx[:B], kg_test.relations[:B]
        with torch.no_grad():
            sf = m.scoring_function(h, t, r).numpy()
            # forward with n_neg = 2 (pos.repeat) -- interfaces.py:72-76
            g = torch.Generator().manual_seed(7)
            nh = torch.randint(0, N_ENT, (2 * B,), generator=g)
            nt = torch.randint(0, N_ENT, (2 * B,), generator=g)
            pos, neg = m(h, t, r, nh, nt)
            h_e, t_e, r_e, cand = m.inference_prepare_candidates(h, t, r, entities=True)
            s_tail = m.inference_scoring_function(h_e, cand, r_e).numpy()
            s_head = m.inference_scoring_function(cand, t_e, r_e).numpy()
            f_tail = filter_scores(torch.from_numpy(s_tail), kg.dict_of_tails, h, r, t).numpy()
            rk_tail = get_rank(torch.from_numpy(s_tail), t).numpy()
            frk_tail = get_rank(torch.from_numpy(f_tail), t).numpy()
            ev = LinkPredictionEvaluator(m, kg_test)
            ev.evaluate(b_size=B, verbose=False)
        out = dict(common)
        for i, tb in enumerate(tabs):
            out['table%d' % i] = tb
        out.update(p=p, sf=sf, fwd_pos=pos.numpy(), fwd_neg=neg.numpy(), neg_heads=nh.numpy(),
                   neg_tails=nt.numpy(), s_tail=s_tail, s_head=s_head, f_tail=f_tail,
                   rk_tail=rk_tail, frk_tail=frk_tail,
                   rank_true_heads=ev.rank_true_heads.numpy(),
                   rank_true_tails=ev.rank_true_tails.numpy(),
                   filt_rank_true_heads=ev.filt_rank_true_heads.numpy(),
                   filt_rank_true_tails=ev.filt_rank_true_tails.numpy(),
                   hit10=np.array(ev.hit_at_k(10)), mrr=np.array(ev.mrr()),
                   mean_rank=np.array(ev.mean_rank()))
        name = 'ref_%s%s.npz' % (kind, '_l1' if (kind == 'transe' and p == 1) else '')
        np.savez_compressed(os.path.join(HERE, name), **out)
        print(name, 'hit10', ev.hit_at_k(10), 'mrr', ev.mrr())

    # ---- relation prediction (evaluation.py:16-204) + relation-candidate scores ----
    rp = dict(common)
    for kind in ('transe', 'distmult', 'complex', 'transh', 'transd'):
        m = build_model(kind, 2)
        tabs = tables_of(kind, m)
        for i, tb in enumerate(tabs):
            rp['%s_table%d' % (kind, i)] = tb
        h, t, r = kg_test.head_idx[:B], kg_test.tail_idx[:B], kg_test.relations[:B]
        with torch.no_grad():
            h_e, t_e, r_e, cand = m.inference_prepare_candidates(h, t, r, entities=False)
            rp['%s_s_rel' % kind] = m.inference_scoring_function(h_e, t_e, cand).numpy()
            for directed in (True, False):
                ev = RelationPredictionEvaluator(m, kg_test, directed=directed)
                ev.evaluate(b_size=B, verbose=False)
                tag = '%s_%s' % (kind, 'dir' if directed else 'undir')
                rp[tag + '_rank'] = ev.rank_true_rels.numpy()
                rp[tag + '_frank'] = ev.filt_rank_true_rels.numpy()
                rp[tag + '_mrr'] = np.array(ev.mrr())
                rp[tag + '_hit3'] = np.array(ev.hit_at_k(3))
    np.savez_compressed(os.path.join(HERE, 'ref_relpred.npz'), **rp)
    if only is not None:
        print('done (only %s)' % only)
        return

    # ---- sampler ---------------------------------------------------------
    samp = BernoulliNegativeSampler(kg, n_neg=3)
    probs_dict = get_bernoulli_probs(kg)
    out = dict(common)
    out['bern_probs'] = samp.bern_probs.numpy()
    out['probs_keys'] = np.array(sorted(probs_dict.keys()), dtype=np.float64)
    h, t, r = kg.head_idx[:200], kg.tail_idx[:200], kg.relations[:200]
    for n_neg in (1, 3):
        torch.manual_seed(99)
        nh, nt = samp.corrupt_batch(h, t, r, n_neg=n_neg)
        out['bern_nh_%d' % n_neg] = nh.numpy()
        out['bern_nt_%d' % n_neg] = nt.numpy()
    usamp = UniformNegativeSampler(kg, n_neg=2)
    torch.manual_seed(99)
    nh, nt = usamp.corrupt_batch(h, t, r)
    out['unif_nh_2'] = nh.numpy()
    out['unif_nt_2'] = nt.numpy()
    torch.manual_seed(5)
    ch, ct = samp.corrupt_kg(batch_size=128, use_cuda=False)
    out['corrupt_kg_h'] = ch.numpy()
    out['corrupt_kg_t'] = ct.numpy()
    np.savez_compressed(os.path.join(HERE, 'ref_sampler.npz'), **out)

    # ---- the reference's own known-answer vectors (tests/test_utils.py) ---
    toy = dict(
        toy_heads=np.array([0, 0, 0, 0, 1, 1, 2, 3, 5]),
        toy_tails=np.array([1, 2, 3, 4, 2, 3, 4, 4, 4]),
        toy_rels=np.array([0, 0, 0, 0, 1, 2, 0, 4, 0]),
        diss_a=np.array([[1.4, 2, 3, 4], [5.4, 6, 7, 8]], dtype=np.float32),   # tests/test_utils.py:34
        diss_b=np.array([[1.3, 4, 2, 10], [5.9, 8, 6, 7]], dtype=np.float32),  # tests/test_utils.py:35
        rank_data=np.array([[1, 2, 3, 4, 0], [1, 2, 1, 3, 0]], dtype=np.float32),
        rank_true=np.array([4, 2]),
    )
    a, b = torch.from_numpy(toy['diss_a']), torch.from_numpy(toy['diss_b'])
    toy['l1'] = l1_dissimilarity(a, b).numpy()
    toy['l2'] = l2_dissimilarity(a, b).numpy()
    toy['ranks'] = get_rank(torch.from_numpy(toy['rank_data']), torch.from_numpy(toy['rank_true'])).numpy()
    toy['ranks_low'] = get_rank(torch.from_numpy(toy['rank_data']), torch.from_numpy(toy['rank_true']),
                                low_values=True).numpy()
    df = pd.DataFrame({'from': toy['toy_heads'], 'to': toy['toy_tails'], 'rel': toy['toy_rels']})
    tkg = KnowledgeGraph(df)
    pr = get_bernoulli_probs(tkg)
    toy['toy_probs_keys'] = np.array(sorted(pr.keys()), dtype=np.float64)
    toy['toy_probs_vals'] = np.array([pr[k] for k in sorted(pr.keys())], dtype=np.float64)
    toy['toy_rels_ix'] = tkg.relations.numpy()
    toy['toy_heads_ix'] = tkg.head_idx.numpy()
    toy['toy_tails_ix'] = tkg.tail_idx.numpy()
    np.savez_compressed(os.path.join(HERE, 'ref_toy.npz'), **toy)
    print('done')


if __name__ == '__main__':
    main()
