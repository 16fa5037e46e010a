# -*- coding: utf-8 -*-
"""get_rank and the Bernoulli-probability precompute of the reference
(utils/operations.py:37-61, :116-179)."""
import numpy as np
import torch

from .. import _hip


def get_rank(data, true, low_values=False):
    """rank_i = #{c : data[i,c] >= data[i,true_i]} (or <= with low_values);
    ties and the true entity itself count, NaN never does
    (utils/operations.py:37-61).  int64, same device."""
    return _hip.get_rank(data, true, low_values)


def _group_mean_counts(a, rel, n_rel):
    """mean over distinct (a, rel) groups of the group size, per relation."""
    key = rel.astype(np.int64) * (int(a.max()) + 1 if a.size else 1) + a.astype(np.int64)
    _, first, counts = np.unique(key, return_index=True, return_counts=True)
    r_of_group = rel[first]
    sums = np.bincount(r_of_group, weights=counts, minlength=n_rel)
    ngroups = np.bincount(r_of_group, minlength=n_rel)
    return sums, ngroups


def get_tph(t):
    """Average number of tails per head, per relation (operations.py:116-131).
    ``t``: (n,3) long tensor [head, tail, rel].  Returns {rel: value}."""
    t = np.asarray(t.cpu() if torch.is_tensor(t) else t)
    n_rel = int(t[:, 2].max()) + 1 if t.shape[0] else 0
    sums, ng = _group_mean_counts(t[:, 0], t[:, 2], n_rel)
    return {float(r): float(sums[r] / ng[r]) for r in range(n_rel) if ng[r] > 0}


def get_hpt(t):
    """Average number of heads per tail, per relation (operations.py:134-149)."""
    t = np.asarray(t.cpu() if torch.is_tensor(t) else t)
    n_rel = int(t[:, 2].max()) + 1 if t.shape[0] else 0
    sums, ng = _group_mean_counts(t[:, 1], t[:, 2], n_rel)
    return {float(r): float(sums[r] / ng[r]) for r in range(n_rel) if ng[r] > 0}


def get_bernoulli_probs(kg):
    """{rel: tph / (tph + hpt)} as in Wang et al. 2014 (operations.py:152-179)."""
    t = torch.cat((kg.head_idx.view(-1, 1), kg.tail_idx.view(-1, 1),
                   kg.relations.view(-1, 1)), dim=1).cpu()
    hpt, tph = get_hpt(t), get_tph(t)
    assert hpt.keys() == tph.keys()
    return {k: tph[k] / (tph[k] + hpt[k]) for k in tph.keys()}


def get_mask(length, start, end):
    """Bool mask of `length`, True on [start, end) (operations.py:13-34)."""
    mask = torch.zeros(length, dtype=torch.bool)
    mask[start:end] = True
    return mask


def get_dictionaries(df, ent=True):
    """ent2ix / rel2ix from a [from, to, rel] DataFrame (operations.py:64-83)."""
    if ent:
        tmp = set(df['from'].unique()).union(set(df['to'].unique()))
    else:
        tmp = set(df['rel'].unique())
    return {x: i for i, x in enumerate(sorted(tmp))}
