# -*- coding: utf-8 -*-
"""Negative samplers with the reference's interface (torchkge/sampling.py:16-327):
``NegativeSampler``, ``UniformNegativeSampler``, ``BernoulliNegativeSampler``
(``.bern_probs``, ``.corrupt_batch(heads, tails, relations, n_neg=None)``,
``.corrupt_kg(batch_size, use_cuda, which)``).

The random draws are issued with the same torch RNG calls, in the same order
and with the same sizes as the reference (bernoulli, randint(k),
randint(B*n_neg - k)), so under the same seed and device the samples are the
reference's; the masked index-puts (the integer part) run in one HIP scatter
(kge_corrupt_scatter).  ``sync_free=True`` draws B*n_neg replacements for both
sides instead, removing the device->host sync of ``mask.sum().item()`` at the
price of a different (equally distributed) random stream.
"""
import torch
from torch import bernoulli, cat, ones, randint, tensor

from . import _hip
from .exceptions import NotYetImplementedError
from .utils.data import DataLoader
from .utils.operations import get_bernoulli_probs


class NegativeSampler:
    """Interface (sampling.py:16-138)."""

    def __init__(self, kg, kg_val=None, kg_test=None, n_neg=1):
        self.kg = kg
        self.n_ent = kg.n_ent
        self.n_facts = kg.n_facts
        self.kg_val = kg_val
        self.kg_test = kg_test
        self.n_neg = n_neg
        self.n_facts_val = 0 if kg_val is None else kg_val.n_facts
        self.n_facts_test = 0 if kg_test is None else kg_test.n_facts
        self.sync_free = False

    def corrupt_batch(self, heads, tails, relations, n_neg):
        raise NotYetImplementedError('NegativeSampler is just an interface, please consider using '
                                     'a child class where this is implemented.')

    def corrupt_kg(self, batch_size, use_cuda, which='main'):
        """Corrupt a whole graph batch by batch with n_neg=1 (sampling.py:76-138)."""
        assert which in ['main', 'train', 'test', 'val']
        if which == 'val':
            assert self.n_facts_val > 0
        if which == 'test':
            assert self.n_facts_test > 0
        tmp_cuda = 'batch' if use_cuda else None
        kg = self.kg_val if which == 'val' else (self.kg_test if which == 'test' else self.kg)
        dataloader = DataLoader(kg, batch_size=batch_size, use_cuda=tmp_cuda)
        corr_heads, corr_tails = [], []
        for batch in dataloader:
            neg_heads, neg_tails = self.corrupt_batch(batch[0], batch[1], batch[2], n_neg=1)
            corr_heads.append(neg_heads)
            corr_tails.append(neg_tails)
        if use_cuda:
            return cat(corr_heads).long().cpu(), cat(corr_tails).long().cpu()
        return cat(corr_heads).long(), cat(corr_tails).long()

    # shared by the Uniform and Bernoulli samplers
    def _corrupt(self, heads, tails, probs, n_neg):
        device = heads.device
        assert device == tails.device
        _hip.require_cuda(heads, tails)
        batch_size = heads.shape[0]
        n = batch_size * n_neg
        mask = bernoulli(probs)                                  # RNG draw #1
        if self.sync_free:
            draws_h = randint(1, self.n_ent, (n,), device=device)
            draws_t = randint(1, self.n_ent, (n,), device=device)
            # position j consumes draw #(ones before j); any fixed assignment is
            # equally distributed, the scatter kernel keeps the prefix-sum rule
        else:
            n_h_cor = int(mask.sum().item())                     # the reference's sync (:319)
            draws_h = randint(1, self.n_ent, (n_h_cor,), device=device)       # draw #2
            draws_t = randint(1, self.n_ent, (n - n_h_cor,), device=device)   # draw #3
        return _hip.corrupt_scatter(heads, tails, mask.to(torch.uint8), draws_h, draws_t, n_neg)


class UniformNegativeSampler(NegativeSampler):
    """Head or tail replaced with probability 1/2 (sampling.py:141-223)."""

    def __init__(self, kg, kg_val=None, kg_test=None, n_neg=1):
        super().__init__(kg, kg_val, kg_test, n_neg)

    def corrupt_batch(self, heads, tails, relations=None, n_neg=None):
        if n_neg is None:
            n_neg = self.n_neg
        probs = ones(size=(heads.shape[0] * n_neg,), device=heads.device) / 2
        return self._corrupt(heads, tails, probs, n_neg)


class BernoulliNegativeSampler(NegativeSampler):
    """Head replaced with probability tph/(tph+hpt) of the relation
    (Wang et al. 2014; sampling.py:226-327)."""

    def __init__(self, kg, kg_val=None, kg_test=None, n_neg=1):
        super().__init__(kg, kg_val, kg_test, n_neg)
        self.bern_probs = self.evaluate_probabilities()

    def evaluate_probabilities(self):
        """fp32 (n_rel) vector, 0.5 for relations absent from the graph
        (sampling.py:263-276)."""
        bern_probs = get_bernoulli_probs(self.kg)
        tmp = []
        for i in range(self.kg.n_rel):
            tmp.append(bern_probs[i] if i in bern_probs.keys() else 0.5)
        return tensor(tmp).float()

    def corrupt_batch(self, heads, tails, relations, n_neg=None):
        if n_neg is None:
            n_neg = self.n_neg
        self.bern_probs = self.bern_probs.to(heads.device)
        return self._corrupt(heads, tails, self.bern_probs[relations].repeat(n_neg), n_neg)
