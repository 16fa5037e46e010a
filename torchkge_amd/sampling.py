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


def get_possible_heads_tails(kg, possible_heads=None, possible_tails=None):
    """{relation: set(heads)}, {relation: set(tails)} of a graph, optionally extending
    given ones (sampling.py:556-600).  Facts are inserted in graph order, like the
    reference, so ``list(set)`` enumerates each set in the same order."""
    from collections import defaultdict
    possible_heads = defaultdict(set) if possible_heads is None else defaultdict(set, possible_heads)
    possible_tails = defaultdict(set) if possible_tails is None else defaultdict(set, possible_tails)
    h, t, r = kg.head_idx.tolist(), kg.tail_idx.tolist(), kg.relations.tolist()
    for hi, ti, ri in zip(h, t, r):
        possible_heads[ri].add(hi)
        possible_tails[ri].add(ti)
    return dict(possible_heads), dict(possible_tails)


class PositionalNegativeSampler(BernoulliNegativeSampler):
    """Socher et al. 2013: the head (or tail, Bernoulli choice of Wang et al. 2014) is
    replaced by an entity that already occupies that position for the same relation
    (sampling.py:330-505).  Same attributes as the reference (``possible_heads``,
    ``possible_tails``, ``n_poss_heads``, ``n_poss_tails``) and the same KIND of random
    draws in the same order (bernoulli, rand(n_heads), rand(n_tails), one randint per sample
    of a relation without candidates) -- but issued on the DEVICE generator (the reference
    draws ``rand`` / ``randint`` on the CPU generator, :470-503), so the samples are equally
    distributed, not seed-identical to the reference's; the per-sample Python loop of the
    reference (:480-503) is a gather from a per-relation CSR on the device."""

    def __init__(self, kg, kg_val=None, kg_test=None):
        super().__init__(kg, kg_val, kg_test, 1)
        self.possible_heads, self.possible_tails, self.n_poss_heads, self.n_poss_tails = self.find_possibilities()
        self._csr = {}

    def find_possibilities(self):
        possible_heads, possible_tails = get_possible_heads_tails(self.kg)
        if self.n_facts_val > 0:
            possible_heads, possible_tails = get_possible_heads_tails(self.kg_val, possible_heads, possible_tails)
        n_poss_heads, n_poss_tails = [], []
        assert possible_heads.keys() == possible_tails.keys()
        for r in range(self.kg.n_rel):
            if r in possible_heads.keys():
                possible_heads[r] = list(possible_heads[r])
                possible_tails[r] = list(possible_tails[r])
            else:
                possible_heads[r] = list()
                possible_tails[r] = list()
            n_poss_heads.append(len(possible_heads[r]))
            n_poss_tails.append(len(possible_tails[r]))
        return possible_heads, possible_tails, tensor(n_poss_heads), tensor(n_poss_tails)

    def _device_csr(self, device):
        key = str(device)
        if key not in self._csr:
            out = []
            for poss, n_poss in ((self.possible_heads, self.n_poss_heads), (self.possible_tails, self.n_poss_tails)):
                off = torch.zeros(self.kg.n_rel + 1, dtype=torch.int64)
                off[1:] = torch.cumsum(n_poss, 0)
                flat = [e for r in range(self.kg.n_rel) for e in poss[r]]
                out += [off.to(device), tensor(flat + [0], dtype=torch.int64).to(device), n_poss.to(device),
                        bool((n_poss == 0).any())]
            self._csr[key] = out
        return self._csr[key]

    def _pick(self, rels, off, flat, n_poss, has_empty, device):
        n = rels.shape[0]
        npos = n_poss[rels]
        choice = (npos.float() * torch.rand((n,), device=device)).floor().long()
        choice = torch.minimum(choice, (npos - 1).clamp_min(0))      # n * rand can round up to n in fp32
        corr = flat[off[rels] + choice]
        return corr, npos

    def corrupt_batch(self, heads, tails, relations, n_neg=None):
        device = heads.device
        assert device == tails.device
        _hip.require_cuda(heads, tails, relations)
        batch_size = heads.shape[0]
        self.bern_probs = self.bern_probs.to(device)
        off_h, flat_h, np_h, empty_h, off_t, flat_t, np_t, empty_t = self._device_csr(device)
        neg_heads, neg_tails = heads.clone(), tails.clone()
        mask = bernoulli(self.bern_probs[relations]).double()
        n_heads_corrupted = int(mask.sum().item())
        m_h, m_t = mask == 1, mask == 0
        rel_h, rel_t = relations[m_h], relations[m_t]
        assert rel_h.shape[0] == n_heads_corrupted and rel_t.shape[0] == batch_size - n_heads_corrupted
        corr_h, npos_h = self._pick(rel_h, off_h, flat_h, np_h, empty_h, device)
        corr_t, npos_t = self._pick(rel_t, off_t, flat_t, np_t, empty_t, device)
        for corr, npos, has_empty in ((corr_h, npos_h, empty_h), (corr_t, npos_t, empty_t)):
            if has_empty:       # relation never seen at this position: any entity (sampling.py:484-487)
                e = npos == 0
                k = int(e.sum().item())
                if k:
                    corr[e] = cat([randint(0, self.n_ent, (1,), device=device) for _ in range(k)])
        neg_heads[m_h] = corr_h
        neg_tails[m_t] = corr_t
        return neg_heads.long(), neg_tails.long()
