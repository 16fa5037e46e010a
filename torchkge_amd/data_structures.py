# -*- coding: utf-8 -*-
"""KnowledgeGraph: three int64 index vectors + the filter sets of the FULL
graph, with the reference's constructor / attributes
(torchkge/data_structures.py:70-136, :160-281, :386-397, :418-434).

Differences in mechanism, not in interface:
  * dict_of_heads / dict_of_tails / dict_of_rels are built lazily and
    vectorised (numpy group-by) instead of a per-fact Python loop with nine
    ``.item()`` calls (23 us/fact in the reference);
  * the graph also carries its full-graph triples (``_filter_src``), shared by
    the children of ``split_kg`` exactly like the dicts are
    (data_structures.py:236-238), so the device FilterIndex used by the
    link-prediction evaluator is built by sort/unique without touching the dicts.
"""
from collections import defaultdict

import numpy as np
import torch
from torch import Tensor
from torch.utils.data import Dataset

from .exceptions import SanityError, SizeMismatchError, WrongArgumentsError
from .filter_index import FilterIndex
from .utils.operations import get_dictionaries


class _LazyDicts(object):
    """Shared, lazily materialised dict-of-sets of one full graph."""

    def __init__(self, heads, tails, rels):
        self.src = (heads, tails, rels)   # CPU int64 tensors of the full graph
        self._dicts = None
        self._index = {}                  # (side, device) -> FilterIndex
        self.cache_dir = None             # optional directory for the on-disk CSR cache

    def _cache_file(self, side):
        """<cache_dir>/filter_<side>_<n_facts>_<checksum>.npz -- keyed on the triples themselves."""
        import hashlib
        import os
        h = hashlib.sha1()
        for x in self.src:
            h.update(np.ascontiguousarray(x.numpy()).tobytes())
        return os.path.join(self.cache_dir, 'filter_%s_%d_%s.npz' % (side, int(self.src[0].shape[0]), h.hexdigest()[:16]))

    @staticmethod
    def _group(k1, k2, v):
        d = defaultdict(set)
        if k1.size == 0:
            return d
        order = np.lexsort((k2, k1))
        k1, k2, v = k1[order], k2[order], v[order]
        brk = np.flatnonzero((k1[1:] != k1[:-1]) | (k2[1:] != k2[:-1])) + 1
        starts = np.concatenate([[0], brk])
        ends = np.concatenate([brk, [k1.size]])
        a, b, vl = k1[starts].tolist(), k2[starts].tolist(), v.tolist()
        for x, y, s, e in zip(a, b, starts.tolist(), ends.tolist()):
            d[(x, y)] = set(vl[s:e])
        return d

    def dicts(self):
        if self._dicts is None:
            h, t, r = (x.numpy() for x in self.src)
            self._dicts = (self._group(t, r, h), self._group(h, r, t), self._group(h, t, r))
        return self._dicts

    def index(self, side, device):
        """side 'heads': (t, r) -> {h};  side 'tails': (h, r) -> {t}."""
        key = (side, str(device))
        if key not in self._index:
            import os
            path = self._cache_file(side) if self.cache_dir else None
            if path and os.path.exists(path):
                self._index[key] = FilterIndex.load(path, device)
                return self._index[key]
            h, t, r = self.src
            if torch.device(device).type == 'cuda':
                n_ent, n_rel = getattr(self, 'n_ent', None), getattr(self, 'n_rel', None)
                build = lambda a, b, c, dv: FilterIndex.from_triples_torch(a, b, c, dv, n_ent, n_rel, n_ent)   # noqa: E731
            else:
                build = FilterIndex.from_triples
            if side == 'heads':
                self._index[key] = build(t, r, h, device)
            else:
                self._index[key] = build(h, r, t, device)
            if path:
                os.makedirs(self.cache_dir, exist_ok=True)
                self._index[key].save(path)
        return self._index[key]


class KnowledgeGraph(Dataset):
    """Same constructor contract as the reference (data_structures.py:70-136):
    build from a DataFrame ``df`` with columns [from, to, rel], or from
    ``kg={'heads','tails','relations'}`` + ``ent2ix`` + ``rel2ix``; optional
    precomputed filter dicts."""

    def __init__(self, df=None, kg=None, ent2ix=None, rel2ix=None, dict_of_heads=None,
                 dict_of_tails=None, dict_of_rels=None, _filter_src=None):
        if df is None:
            if kg is None:
                raise WrongArgumentsError("Please provide at least one argument of `df` and kg`")
            if not (type(kg) == dict and 'heads' in kg and 'tails' in kg and 'relations' in kg):
                raise WrongArgumentsError("Keys in the `kg` dict should contain `heads`, `tails`, "
                                          "`relations`.")
            if rel2ix is None or ent2ix is None:
                raise WrongArgumentsError("Please provide the two dictionaries ent2ix and rel2ix "
                                          "if building from `kg`.")
        elif kg is not None:
            raise WrongArgumentsError("`df` and kg` arguments should not both be provided.")

        self.ent2ix = get_dictionaries(df, ent=True) if ent2ix is None else ent2ix
        self.rel2ix = get_dictionaries(df, ent=False) if rel2ix is None else rel2ix
        self.n_ent = max(self.ent2ix.values()) + 1
        self.n_rel = max(self.rel2ix.values()) + 1

        if df is not None:
            self.n_facts = len(df)
            self.head_idx = torch.tensor(df['from'].map(self.ent2ix).values).long()
            self.tail_idx = torch.tensor(df['to'].map(self.ent2ix).values).long()
            self.relations = torch.tensor(df['rel'].map(self.rel2ix).values).long()
        else:
            self.n_facts = kg['heads'].shape[0]
            self.head_idx, self.tail_idx, self.relations = kg['heads'], kg['tails'], kg['relations']

        self._explicit_dicts = None
        if dict_of_heads is not None and dict_of_tails is not None and dict_of_rels is not None:
            self._explicit_dicts = (dict_of_heads, dict_of_tails, dict_of_rels)
        if _filter_src is not None:
            self._lazy = _filter_src
        elif self._explicit_dicts is None:
            self._lazy = _LazyDicts(self.head_idx.cpu(), self.tail_idx.cpu(), self.relations.cpu())
            self._lazy.n_ent, self._lazy.n_rel = self.n_ent, self.n_rel      # id bounds for the device-side index build
        else:
            self._lazy = None
        try:
            self.sanity_check()
        except AssertionError:
            raise SanityError("Please check the sanity of arguments.")

    # the three filter dicts of the FULL graph (data_structures.py:386-397)
    @property
    def dict_of_heads(self):
        return self._explicit_dicts[0] if self._explicit_dicts is not None else self._lazy.dicts()[0]

    @property
    def dict_of_tails(self):
        return self._explicit_dicts[1] if self._explicit_dicts is not None else self._lazy.dicts()[1]

    @property
    def dict_of_rels(self):
        return self._explicit_dicts[2] if self._explicit_dicts is not None else self._lazy.dicts()[2]

    def set_filter_cache(self, cache_dir):
        """Keep the device filter CSRs of this graph on disk under ``cache_dir``
        (file name keyed on a checksum of the triples): later runs load them
        instead of re-sorting the graph."""
        if self._lazy is not None:
            self._lazy.cache_dir = cache_dir

    def filter_index(self, side, device):
        """Device FilterIndex of dict_of_heads (side='heads') or dict_of_tails
        (side='tails'); built from the full-graph triples when they are known
        (on the device, by sort / unique), otherwise converted from the dict."""
        if self._lazy is not None:
            return self._lazy.index(side, device)
        from .filter_index import filter_index_for
        d = self.dict_of_heads if side == 'heads' else self.dict_of_tails
        return filter_index_for(d, device)

    def __len__(self):
        return self.n_facts

    def __getitem__(self, item):
        return (self.head_idx[item].item(), self.tail_idx[item].item(), self.relations[item].item())

    def sanity_check(self):
        assert type(self.ent2ix) == dict and type(self.rel2ix) == dict
        assert len(self.ent2ix) == self.n_ent and len(self.rel2ix) == self.n_rel
        assert type(self.head_idx) == Tensor and type(self.tail_idx) == Tensor and \
            type(self.relations) == Tensor
        assert self.head_idx.dtype == torch.int64 and self.tail_idx.dtype == torch.int64 and \
            self.relations.dtype == torch.int64
        assert len(self.head_idx) == len(self.tail_idx) == len(self.relations)
        if self._explicit_dicts is not None:
            assert all(type(d) == defaultdict for d in self._explicit_dicts)

    def _child(self, mask):
        return KnowledgeGraph(
            kg={'heads': self.head_idx[mask], 'tails': self.tail_idx[mask],
                'relations': self.relations[mask]},
            ent2ix=self.ent2ix, rel2ix=self.rel2ix,
            dict_of_heads=None if self._explicit_dicts is None else self._explicit_dicts[0],
            dict_of_tails=None if self._explicit_dicts is None else self._explicit_dicts[1],
            dict_of_rels=None if self._explicit_dicts is None else self._explicit_dicts[2],
            _filter_src=self._lazy)

    def split_kg(self, share=0.8, sizes=None, validation=False):
        """Train/(val)/test split; children share this graph's filter sets
        (data_structures.py:160-281).  ``sizes`` (len 2 or 3) takes contiguous
        prefixes; otherwise a per-relation random split of ratio ``share`` that
        keeps every entity in the training part."""
        n = self.n_facts
        if sizes is not None:
            if len(sizes) not in (2, 3):
                raise SizeMismatchError('Tuple `sizes` should be of length 2 or 3.')
            if sum(sizes) != n:
                raise WrongArgumentsError('Sizes should sum to the number of facts.')
            idx = torch.arange(n)
            bounds = np.cumsum([0] + list(sizes))
            return tuple(self._child((idx >= int(lo)) & (idx < int(hi)))
                         for lo, hi in zip(bounds[:-1], bounds[1:]))
        assert share < 1
        masks = self.get_mask(share, validation=validation)
        return tuple(self._child(m) for m in masks)

    def get_mask(self, share, validation=False):
        """Random per-relation masks (train[, val], test) -- data_structures.py:283-345."""
        rel = self.relations
        mask = torch.zeros_like(rel).bool()
        mask_val = torch.zeros_like(rel).bool()
        for r in rel.unique():
            sub = torch.eq(rel, r).nonzero(as_tuple=False)[:, 0]
            rand = torch.randperm(len(sub))
            szs = self.get_sizes(len(sub), share=share, validation=validation)
            mask[sub[rand[:szs[0]]]] = True
            if validation:
                mask_val[sub[rand[szs[0]:szs[0] + szs[1]]]] = True
        present = torch.cat((self.head_idx[mask], self.tail_idx[mask])).unique()
        every = torch.cat((self.head_idx, self.tail_idx)).unique()
        if len(present) < len(every):
            missing = set(every.tolist()) - set(present.tolist())
            for e in missing:
                sub = ((self.head_idx == e) | (self.tail_idx == e)).nonzero(as_tuple=False)[:, 0]
                rand = torch.randperm(len(sub))
                szs = self.get_sizes(mask.shape[0], share=share, validation=validation)
                mask[sub[rand[:szs[0]]]] = True
                if validation:
                    mask_val[sub[rand[:szs[0]]]] = False
        if validation:
            assert not (mask & mask_val).any().item()
            return mask, mask_val, ~(mask | mask_val)
        return mask, ~mask

    @staticmethod
    def get_sizes(count, share, validation=False):
        """How many of `count` samples go to train[/val]/test (data_structures.py:347-384)."""
        if count == 1:
            return (1, 0, 0) if validation else (1, 0)
        if count == 2:
            return (1, 1, 0) if validation else (1, 1)
        n_train = max(int(count * share), 1)
        assert n_train < count
        if not validation:
            return n_train, count - n_train
        if count - n_train == 1:
            return n_train - 1, 1, 1
        n_val = int(int(count - n_train) / 2)
        return n_train, n_val, count - n_train - n_val

    def get_df(self):
        """DataFrame with columns ['from', 'to', 'rel'] (data_structures.py:399-415)."""
        from pandas import DataFrame
        ix2ent = {v: k for k, v in self.ent2ix.items()}
        ix2rel = {v: k for k, v in self.rel2ix.items()}
        df = DataFrame({'from': self.head_idx.cpu().numpy(), 'to': self.tail_idx.cpu().numpy(),
                        'rel': self.relations.cpu().numpy()})
        df['from'] = df['from'].map(ix2ent)
        df['to'] = df['to'].map(ix2ent)
        df['rel'] = df['rel'].map(ix2rel)
        return df


class SmallKG(Dataset):
    """Minimal (heads, tails, relations) container (data_structures.py:418-434)."""

    def __init__(self, heads, tails, relations):
        assert heads.shape == tails.shape == relations.shape
        self.head_idx, self.tail_idx, self.relations = heads, tails, relations
        self.length = heads.shape[0]

    def __len__(self):
        return self.length

    def __getitem__(self, item):
        return self.head_idx[item].item(), self.tail_idx[item].item(), self.relations[item].item()
