# -*- coding: utf-8 -*-
"""filter_scores / get_true_targets / init_embedding of the reference
(utils/modeling.py:21-28, :53-102)."""
import torch
from torch.nn import Embedding
from torch.nn.init import xavier_uniform_

from .. import _hip
from ..filter_index import filter_index_for


def init_embedding(n_vectors, dim):
    """nn.Embedding initialised Xavier-uniform (modeling.py:21-28)."""
    emb = Embedding(n_vectors, dim)
    xavier_uniform_(emb.weight.data)
    return emb


def get_true_targets(dictionary, key1, key2, true_idx, i):
    """Entities e such that (key1[i], key2[i], e) is a known fact, minus
    true_idx[i] (host helper, modeling.py:53-88, same KeyError quirk: a key
    whose set lacks true_idx[i] yields None)."""
    try:
        true_targets = dictionary[key1[i].item(), key2[i].item()].copy()
        if true_idx is not None:
            true_targets.remove(true_idx[i].item())
            if len(true_targets) > 0:
                return torch.tensor(list(true_targets)).long()
            return None
        return torch.tensor(list(true_targets)).long()
    except KeyError:
        return None


def filter_scores(scores, dictionary, key1, key2, true_idx):
    """Copy of `scores` with -inf at every known true target other than
    true_idx[i] (modeling.py:91-102).  `dictionary` is a torchkge dict-of-sets
    (converted once to a device FilterIndex and cached) or a FilterIndex."""
    _hip.require_cuda(scores, key1, key2, true_idx)     # true_idx None: mask every known target
    index = dictionary if hasattr(dictionary, 'lookup') else filter_index_for(dictionary, scores.device)
    seg_lo, seg_hi = index.lookup(key1, key2)
    filt = _hip.f32c(scores).clone()
    return _hip.filter_scores_(filt, None if true_idx is None else _hip.i64c(true_idx), seg_lo, seg_hi,
                               index.targets)
