"""The reference's own unit tests for the in-scope API (tests/test_evaluation.py:18-36,
tests/test_utils.py:97-107,156-163, tests/test_data.py), re-stated against torchkge_amd.
Same toy graph, same assertions; the model lives on the MI355X."""
import pandas as pd
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def toy():
    import torchkge_amd as tk
    df = pd.DataFrame([[0, 1, 0], [0, 2, 0], [0, 3, 0], [0, 4, 0], [1, 2, 1], [1, 3, 2], [2, 4, 0], [3, 4, 4],
                       [5, 4, 0]], columns=['from', 'to', 'rel'])
    return tk, tk.KnowledgeGraph(df)


def test_LinkPredictionEvaluator_like_reference(toy):
    tk, kg = toy
    from torchkge_amd.models import TransEModel
    model = TransEModel(100, 6, 4, 'L1').cuda()                 # tests/test_evaluation.py:19
    evaluator = tk.LinkPredictionEvaluator(model, kg)
    evaluator.evaluate(b_size=len(kg), verbose=False)
    for r in (evaluator.rank_true_heads, evaluator.rank_true_tails, evaluator.filt_rank_true_heads,
              evaluator.filt_rank_true_tails):
        assert r.dtype == torch.long and len(r.shape) == 1 and r.shape[0] == len(kg)
    assert (evaluator.filt_rank_true_heads <= evaluator.rank_true_heads).all()
    assert int(evaluator.rank_true_tails.min()) >= 1 and int(evaluator.rank_true_tails.max()) <= kg.n_ent
    evaluator.print_results()
    for cls, args in ((tk.TransHModel, (20, 6, 4)), (tk.TransDModel, (20, 12, 6, 4)), (tk.DistMultModel, (20, 6, 4)),
                      (tk.ComplExModel, (20, 6, 4))):
        ev = tk.LinkPredictionEvaluator(cls(*args).cuda(), kg)
        ev.evaluate(b_size=4, verbose=False)
        assert ev.rank_true_heads.shape[0] == 9 and 0 < ev.mrr()[1] <= 1


def test_utils_like_reference(toy):
    tk, kg = toy
    from torchkge_amd.utils import get_rank, l1_dissimilarity, l2_dissimilarity
    a = torch.tensor([[1.4, 2, 3, 4], [5.4, 6, 7, 8]]).float().cuda()       # tests/test_utils.py:34-35
    b = torch.tensor([[1.3, 4, 2, 10], [5.9, 8, 6, 7]]).float().cuda()
    assert (l1_dissimilarity(a, b).cpu() - torch.tensor([9.1000, 4.5000])).abs().max() < 1e-5
    assert (l2_dissimilarity(a, b).cpu() - torch.tensor([41.0100, 6.2500])).sum() < 1e-03
    data = torch.tensor([[1, 2, 3, 4, 0], [1, 2, 1, 3, 0]]).float().cuda()  # tests/test_utils.py:157-163
    true = torch.tensor([4, 2]).cuda()
    assert torch.eq(get_rank(data, true).cpu(), torch.tensor([5, 4])).all()
    assert torch.eq(get_rank(data, true, low_values=True).cpu(), torch.tensor([1, 3])).all()


def test_training_loop_like_tutorial(toy):
    """docs/tutorials/transe.rst:23-58 on the toy graph: DataLoader('all') + sampler + MarginLoss + Adam."""
    tk, kg = toy
    from torchkge_amd.utils import DataLoader, MarginLoss
    model = tk.TransEModel(16, kg.n_ent, kg.n_rel, dissimilarity_type='L2').cuda()
    criterion = MarginLoss(0.5).cuda()
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-2, weight_decay=1e-5)
    sampler = tk.BernoulliNegativeSampler(kg)
    dataloader = DataLoader(kg, batch_size=4, use_cuda='all')
    first = None
    for epoch in range(30):
        running = 0.0
        for h, t, r in dataloader:
            n_h, n_t = sampler.corrupt_batch(h, t, r)
            optimizer.zero_grad()
            pos, neg = model(h, t, r, n_h, n_t)
            loss = criterion(pos, neg)
            loss.backward()
            optimizer.step()
            running += loss.item()
        model.normalize_parameters()
        first = running if first is None else first
    assert running < first                                       # it learns
    ent, rel = model.get_embeddings()
    assert (ent.norm(dim=1) - 1).abs().max().item() < 1e-5


@pytest.mark.gpu
def test_positional_sampler_and_triplet_classification():
    """PositionalNegativeSampler (sampling.py:330-505): a corrupted entity has already occupied
    that position for that relation (or is arbitrary when the relation has none), exactly one
    side changes per sample, attributes as the reference's.  TripletClassificationEvaluator
    (evaluation.py:428-585): thresholds / accuracy equal the reference's loop on the same
    negative samples."""
    import torchkge_amd as tk
    g = torch.Generator().manual_seed(12)
    n_ent, n_rel, n = 80, 6, 900
    h = torch.randint(0, n_ent, (n,), generator=g); t = torch.randint(0, n_ent, (n,), generator=g)
    r = torch.randint(0, n_rel - 1, (n,), generator=g)       # relation n_rel-1 never occurs in the main graph
    kw = dict(ent2ix={i: i for i in range(n_ent)}, rel2ix={i: i for i in range(n_rel)})
    kg = tk.KnowledgeGraph(kg={'heads': h, 'tails': t, 'relations': r}, **kw)
    kg_val, kg_test = kg.split_kg(sizes=(600, 300))
    kg_test.relations[:5] = n_rel - 1                         # a relation with no possible heads / tails
    s = tk.PositionalNegativeSampler(kg_val, kg_test=kg_test)
    assert set(s.possible_heads.keys()) == set(range(n_rel)) and s.n_poss_heads.shape == (n_rel,)
    for rr in range(n_rel):
        m = kg_val.relations == rr
        assert set(s.possible_heads[rr]) == set(kg_val.head_idx[m].tolist())
        assert set(s.possible_tails[rr]) == set(kg_val.tail_idx[m].tolist())
        assert int(s.n_poss_heads[rr]) == len(s.possible_heads[rr])
    hh, tt, rr_ = kg_test.head_idx.cuda(), kg_test.tail_idx.cuda(), kg_test.relations.cuda()
    torch.manual_seed(3)
    nh, nt = s.corrupt_batch(hh, tt, rr_)
    assert nh.dtype == torch.int64 and nh.is_cuda and nh.shape == hh.shape
    changed_h, changed_t = (nh != hh).cpu(), (nt != tt).cpu()
    assert not bool((changed_h & changed_t).any())
    nh_c, nt_c, rel_c = nh.cpu().tolist(), nt.cpu().tolist(), kg_test.relations.tolist()
    for i in range(len(rel_c)):
        if changed_h[i] and s.possible_heads[rel_c[i]]:
            assert nh_c[i] in s.possible_heads[rel_c[i]]
        if changed_t[i] and s.possible_tails[rel_c[i]]:
            assert nt_c[i] in s.possible_tails[rel_c[i]]
    torch.manual_seed(3)
    nh2, nt2 = s.corrupt_batch(hh, tt, rr_)
    assert torch.equal(nh, nh2) and torch.equal(nt, nt2)      # same seed, same samples

    m = tk.TransEModel(16, n_ent, n_rel, dissimilarity_type='L2').cuda()
    ev = tk.TripletClassificationEvaluator(m, kg_val, kg_test)
    torch.manual_seed(5)
    ev.evaluate(b_size=128)
    torch.manual_seed(5)
    negh, negt = ev.sampler.corrupt_kg(128, True, which='main')
    with torch.no_grad():
        neg_scores = m.scoring_function(negh.cuda(), negt.cuda(), kg_val.relations.cuda()).cpu()
    ref = torch.zeros(n_rel)
    for i in range(n_rel):                                     # the reference's loop (evaluation.py:533-538)
        mask = kg_val.relations == i
        ref[i] = neg_scores[mask].max() if mask.sum() > 0 else neg_scores.max()
    assert torch.equal(ev.thresholds.cpu(), ref)
    torch.manual_seed(9)
    acc = ev.accuracy(b_size=128)
    torch.manual_seed(9)
    negh, negt = ev.sampler.corrupt_kg(128, True, which='test')
    with torch.no_grad():
        sc = m.scoring_function(kg_test.head_idx.cuda(), kg_test.tail_idx.cuda(), kg_test.relations.cuda()).cpu()
        nsc = m.scoring_function(negh.cuda(), negt.cuda(), kg_test.relations.cuda()).cpu()
    thr = ref[kg_test.relations]
    assert acc == ((sc > thr).sum().item() + (nsc < thr).sum().item()) / (2 * kg_test.n_facts)
