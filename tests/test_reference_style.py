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


