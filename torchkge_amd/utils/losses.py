# -*- coding: utf-8 -*-
"""Training criteria on the two length-b score vectors Model.forward returns (positives, negatives), each a SUM over the
batch.  Written out as the elementwise formulas (what `torch.nn.MarginRankingLoss / SoftMarginLoss / BCELoss` with
`reduction='sum'` evaluate, operation for operation -- `tests/test_host_logic.py::test_training_criteria_equal_the_stock_torch_ones`
pins the bits) so that a training step is three elementwise kernels and a reduction, no criterion objects, no target tensors.
Not part of the HIP hot path: the scores come from K1 (`kge_score_triples`), the gradient goes back through it."""
import torch
from torch.nn import Module


class _PairCriterion(Module):
    """loss(positive scores, negative scores) -> scalar; subclasses give the per-pair / per-score terms."""

    def forward(self, positive_triplets, negative_triplets):
        return self.terms(positive_triplets, negative_triplets).sum()


class MarginLoss(_PairCriterion):
    """sum_i max(0, margin - f(pos_i) + f(neg_i)): a positive should out-score its negative by `margin`."""

    def __init__(self, margin):
        super().__init__()
        self.margin = float(margin)

    def terms(self, pos, neg):
        return (self.margin - (pos - neg)).clamp_min(0)


class LogisticLoss(Module):
    """sum_i log(1 + exp(-f(pos_i))) + sum_j log(1 + exp(f(neg_j))): labels +1 / -1 under the logistic model."""

    def forward(self, positive_triplets, negative_triplets):
        return torch.log1p(torch.exp(-positive_triplets)).sum() + torch.log1p(torch.exp(negative_triplets)).sum()


class BinaryCrossEntropyLoss(_PairCriterion):
    """Cross entropy of sigmoid(score) against the labels 1 (positives) / 0 (negatives), logarithms cut off at -100."""

    def terms(self, pos, neg):
        p = torch.sigmoid(torch.cat([pos, neg], dim=0))
        y = torch.cat([torch.ones_like(pos), torch.zeros_like(neg)], dim=0)
        return (y - 1) * torch.log(1 - p).clamp_min(-100) - y * torch.log(p).clamp_min(-100)
