# -*- coding: utf-8 -*-
"""Losses on the length-b score vectors (reference utils/losses.py:19-112):
stock torch.nn criteria with reduction='sum'; not part of the HIP hot path."""
from torch import ones_like, zeros_like, cat
from torch.nn import Module, Sigmoid, MarginRankingLoss, SoftMarginLoss, BCELoss


class MarginLoss(Module):
    """sum max(0, margin - f(pos) + f(neg))   (losses.py:19-44)."""

    def __init__(self, margin):
        super().__init__()
        self.loss = MarginRankingLoss(margin=margin, reduction='sum')

    def forward(self, positive_triplets, negative_triplets):
        return self.loss(positive_triplets, negative_triplets, target=ones_like(positive_triplets))


class LogisticLoss(Module):
    """sum log(1 + exp(-y f))   (losses.py:47-77)."""

    def __init__(self):
        super().__init__()
        self.loss = SoftMarginLoss(reduction='sum')

    def forward(self, positive_triplets, negative_triplets):
        targets = ones_like(positive_triplets)
        return self.loss(positive_triplets, targets) + self.loss(negative_triplets, -targets)


class BinaryCrossEntropyLoss(Module):
    """BCE on sigmoid(scores)   (losses.py:80-112)."""

    def __init__(self):
        super().__init__()
        self.sig = Sigmoid()
        self.loss = BCELoss(reduction='sum')

    def forward(self, positive_triplets, negative_triplets):
        scores = cat([positive_triplets, negative_triplets], dim=0)
        targets = cat([ones_like(positive_triplets), zeros_like(negative_triplets)], dim=0)
        return self.loss(self.sig(scores), targets)
