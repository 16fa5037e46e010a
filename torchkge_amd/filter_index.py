# -*- coding: utf-8 -*-
"""Device-resident index of the link-prediction filter sets.

The reference keeps ``dict_of_heads[(t, r)] -> set(h)`` and
``dict_of_tails[(h, r)] -> set(t)`` as Python dict-of-sets
(data_structures.py:386-397) and walks them row by row in ``filter_scores``
(utils/modeling.py:91-102).  Here the same mapping lives in HBM as a sorted-key
CSR: ``keys`` (int64, sorted, key = k1 * 2**31 + k2), ``offsets`` (int64) and
``targets`` (int32); a batch looks its segments up with one kernel
(kge_filter_lookup).  Built once per (dictionary | knowledge graph, device).
"""
import itertools

import numpy as np
import torch

from . import _hip

KEY2_SPAN = 1 << 31   # k2 (relation id) < 2**31, k1 (entity id) < 2**32


class FilterIndex(object):
    def __init__(self, keys, offsets, targets, device):
        self.keys = torch.as_tensor(keys, dtype=torch.int64).to(device)
        self.offsets = torch.as_tensor(offsets, dtype=torch.int64).to(device)
        self.targets = torch.as_tensor(targets, dtype=torch.int32).to(device)
        if self.targets.numel() == 0:   # keep a valid pointer for the kernels
            self.targets = torch.zeros(1, dtype=torch.int32, device=device)
        self.device = torch.device(device)
        self.n_keys = int(self.keys.shape[0])

    # -- constructors --------------------------------------------------------
    @classmethod
    def from_dict(cls, dictionary, device):
        """From a torchkge-style ``{(k1, k2): set(int)}`` mapping."""
        n = len(dictionary)
        if n == 0:
            return cls(np.zeros(0, np.int64), np.zeros(1, np.int64), np.zeros(0, np.int32), device)
        keys = np.fromiter((k1 * KEY2_SPAN + k2 for (k1, k2) in dictionary.keys()), dtype=np.int64, count=n)
        counts = np.fromiter((len(v) for v in dictionary.values()), dtype=np.int64, count=n)
        flat = np.fromiter(itertools.chain.from_iterable(dictionary.values()), dtype=np.int64,
                           count=int(counts.sum()))
        order = np.argsort(keys, kind='stable')
        start = np.concatenate([[0], np.cumsum(counts)])
        new_counts = counts[order]
        offsets = np.concatenate([[0], np.cumsum(new_counts)])
        # permutation that moves each segment to its sorted-key position
        seg_of = np.repeat(np.arange(n), new_counts)
        within = np.arange(int(new_counts.sum())) - offsets[seg_of]
        src = start[order][seg_of] + within
        return cls(keys[order], offsets, flat[src].astype(np.int32), device)

    @classmethod
    def from_triples(cls, key1, key2, values, device):
        """From parallel id arrays: index[(key1_j, key2_j)] ∋ values_j (duplicates
        collapse, like set.add).  numpy sort/unique -- no Python loop over facts."""
        key1 = np.asarray(key1, dtype=np.int64)
        key2 = np.asarray(key2, dtype=np.int64)
        values = np.asarray(values, dtype=np.int64)
        if key1.size == 0:
            return cls(np.zeros(0, np.int64), np.zeros(1, np.int64), np.zeros(0, np.int32), device)
        k = key1 * KEY2_SPAN + key2
        order = np.lexsort((values, k))
        k, v = k[order], values[order]
        keep = np.ones(k.shape[0], dtype=bool)
        keep[1:] = (k[1:] != k[:-1]) | (v[1:] != v[:-1])
        k, v = k[keep], v[keep]
        ukeys, first = np.unique(k, return_index=True)
        offsets = np.concatenate([first, [k.shape[0]]])
        return cls(ukeys, offsets, v.astype(np.int32), device)

    @classmethod
    def from_triples_device(cls, key1, key2, values, device, n_key1=None, n_key2=None, n_values=None):
        """The same index built ON the GPU by the library's own kernels (kge_filter_index_build: ONE 64-bit radix sort
        of the composite (key, value), flags, a scan, a scatter -- no ATen sort / unique, whose first use in a process
        costs ~0.5 s of kernel loading).  n_key1 / n_key2 / n_values: exclusive upper bounds of the ids (n_ent, n_rel,
        n_ent for a KnowledgeGraph); found by one reduction launch when not given.  Bit-identical to from_triples."""
        dev = torch.device(device)
        key1 = torch.as_tensor(key1, dtype=torch.int64).to(dev)
        key2 = torch.as_tensor(key2, dtype=torch.int64).to(dev)
        v = torch.as_tensor(values, dtype=torch.int64).to(dev)
        if key1.numel() == 0:
            return cls(np.zeros(0, np.int64), np.zeros(1, np.int64), np.zeros(0, np.int32), device)
        # the ids are ALWAYS checked against the bounds (one reduction launch + one sync per index build): the pack kernel
        # ORs the value into the low bits of the key and the radix sort looks at exactly the declared bits, so an id outside
        # [0, bound) -- a kg whose tensors disagree with ent2ix -- would silently merge or mis-sort filter lists.
        # (kge_i64_max3 compares as unsigned: a negative id comes back as a negative Python int here)
        m1, m2, m3 = _hip.i64_max3(key1, key2, v)
        if n_key1 is None or n_key2 is None or n_values is None:
            n_key1, n_key2, n_values = m1 + 1, m2 + 1, m3 + 1
        if min(m1, m2, m3) < 0 or m1 >= n_key1 or m2 >= n_key2 or m3 >= n_values:
            raise ValueError('filter index: ids outside [0, bound): max ids (%d, %d, %d) against bounds (%d, %d, %d) '
                             '(negative ids read as negative maxima)' % (m1, m2, m3, n_key1, n_key2, n_values))
        if not (0 < n_key2 <= KEY2_SPAN and 0 < n_values < (1 << 31) and n_key1 > 0):
            raise ValueError('filter index: ids out of range (negative, or a relation id >= 2**31)')
        built = _hip.filter_index_build(key1, key2, v, int(n_key1), int(n_key2), int(n_values), KEY2_SPAN)
        if built is None:       # (key, value) does not fit 64 bits: the two-sort ATen composition
            return cls._from_triples_aten(key1, key2, v, device)
        return cls(built[0], built[1], built[2], device)

    @classmethod
    def from_triples_torch(cls, key1, key2, values, device, n_key1=None, n_key2=None, n_values=None):
        """The index built ON the target device: the library's own kernels on the GPU (from_triples_device), the ATen
        sort / unique_consecutive composition elsewhere.  At Wikidata5M scale (2e7 facts) this replaces minutes of
        per-fact Python (data_structures.py:386-397) by a few device-side passes.  Bit-identical to from_triples."""
        if torch.device(device).type == 'cuda':
            return cls.from_triples_device(key1, key2, values, device, n_key1, n_key2, n_values)
        return cls._from_triples_aten(key1, key2, values, device)

    @classmethod
    def _from_triples_aten(cls, key1, key2, values, device):
        """torch sort / unique_consecutive (two stable sorts = lexicographic (key, value) order)."""
        dev = torch.device(device)
        key1 = torch.as_tensor(key1, dtype=torch.int64).to(dev)
        key2 = torch.as_tensor(key2, dtype=torch.int64).to(dev)
        v = torch.as_tensor(values, dtype=torch.int64).to(dev)
        if key1.numel() == 0:
            return cls(np.zeros(0, np.int64), np.zeros(1, np.int64), np.zeros(0, np.int32), device)
        k = key1 * KEY2_SPAN + key2
        v, o = torch.sort(v, stable=True)
        k = k[o]
        k, o = torch.sort(k, stable=True)
        v = v[o]
        keep = torch.ones(k.shape[0], dtype=torch.bool, device=dev)
        keep[1:] = (k[1:] != k[:-1]) | (v[1:] != v[:-1])
        k, v = k[keep], v[keep]
        ukeys, counts = torch.unique_consecutive(k, return_counts=True)
        offsets = torch.zeros(ukeys.shape[0] + 1, dtype=torch.int64, device=dev)
        offsets[1:] = torch.cumsum(counts, 0)
        return cls(ukeys, offsets, v.to(torch.int32), device)

    # -- on-disk cache ---------------------------------------------------------
    def save(self, path):
        """Write the CSR (keys, offsets, targets) as one .npz file."""
        n_t = int(self.offsets[-1].item()) if self.offsets.numel() else 0
        np.savez(path, keys=self.keys.cpu().numpy(), offsets=self.offsets.cpu().numpy(),
                 targets=self.targets.cpu().numpy()[:n_t])
        return path

    @classmethod
    def load(cls, path, device):
        z = np.load(path)
        return cls(z['keys'], z['offsets'], z['targets'], device)

    # -- queries -------------------------------------------------------------
    def lookup(self, key1, key2):
        """(seg_lo, seg_hi) int64 device vectors; empty segment = key absent."""
        _hip.require_cuda(key1, key2)
        return _hip.filter_lookup(self.keys, self.offsets, key1, key2, KEY2_SPAN)


class FilterPlan(object):
    """What one batch of link-prediction queries needs from the filter index, computed ONCE: the
    filter segments and true ids of its 2B queries (tail side first), and the grouping that lets
    every distinct filter list be scored once (kge_lp_filter_sub_planned): queries that share a key
    share the list AND the query row, so only the first query of a key 'owns' the list's pairs.
    A pure function of the test facts and the index -- not of the model -- so an evaluator keeps it
    across evaluate() calls (like the index itself, built once per graph).

    seg_lo / seg_hi / true_idx: (2B) int64;  woff: (2B + 1) exclusive prefix sum of the owned lengths;
    n_pairs = woff[-1] (host int);  long_q: queries with more than 512 list entries."""

    LONG = 512

    def __init__(self, seg_lo, seg_hi, true_idx, targets):
        self.seg_lo, self.seg_hi, self.true_idx, self.targets = seg_lo, seg_hi, true_idx, targets
        n = seg_lo.shape[0]
        dev = seg_lo.device
        if seg_lo.is_cuda and n > 0:    # the library's own kernels (atomicMin claim, one packed scan): no ATen scatter_reduce
            self.woff, self.long_q, self.n_pairs = _hip.filter_plan_build(seg_lo, seg_hi, int(targets.shape[0]), self.LONG)
            self.n_long = int(self.long_q.shape[0])
            return
        ln = seg_hi - seg_lo
        nonempty = ln > 0
        # first query of every distinct segment start (segments of one index are disjoint: the start names the key)
        first = torch.full((int(targets.shape[0]) + 1,), n, dtype=torch.int64, device=dev)
        q = torch.arange(n, dtype=torch.int64, device=dev)
        first.scatter_reduce_(0, torch.where(nonempty, seg_lo, torch.full_like(seg_lo, targets.shape[0])), q,
                              reduce='amin', include_self=True)
        owner = nonempty & (first[seg_lo.clamp(max=targets.shape[0])] == q)
        owned = torch.where(owner, ln, torch.zeros_like(ln))
        self.woff = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        self.woff[1:] = torch.cumsum(owned, 0)
        self.long_q = torch.nonzero(ln > self.LONG).view(-1).contiguous()
        self.n_pairs = int(self.woff[-1].item())       # (one host sync, at plan build only)
        self.n_long = int(self.long_q.shape[0])


class ColumnPlan(object):
    """The distinct query ROWS of one both-sides link-prediction batch (2B queries, tail side first).  A query row is
    a function of its key -- (h, r) on the tail side, (t, r) on the head side -- so queries that share a key share the
    row and differ only in their true entity, i.e. in the thresholds of the rank count (on FB15k-237-like test splits
    a quarter of the queries repeat an earlier key).  The count kernel then sweeps the matrix cores once per COLUMN:

      * keys with one query            -> one column each, ``col_q[column] = query``;
      * keys with m > 1 queries        -> ceil(m / sets) columns of up to ``sets`` queries, ``members[column, j]``;
      * ``qs_row[query]``              -> the column (row of the split query table) the query's cells are written to,
                                          -1 for all but the first query of a column.

    Single-query columns come first, both parts padded to the kernel's query panel with -1.  A pure function of the
    facts (not of the model or the index): built once per batch by the evaluator, on the device, no host loop."""

    def __init__(self, h, t, r, n_ent, n_rel, sets, pad, relation_major=False):
        dev = h.device
        B = h.shape[0]
        n = 2 * B
        self.n_queries = n
        self.sets = sets
        if h.is_cuda and B > 0:     # the library's own kernels (radix sort of (key, query), scans, one scatter)
            built = _hip.column_plan_build(h, t, r, n_ent, n_rel, sets, pad, relation_major)
            if built is not None:
                for k_, v_ in built.items():
                    setattr(self, k_, v_)
                return
        if relation_major:      # columns come out in key order: relation-major for kernels that gather per-relation rows
            key = torch.cat([r * n_ent + h, (r * n_ent + t) + n_ent * n_rel])
        else:
            key = torch.cat([h * n_rel + r, (t * n_rel + r) + n_ent * n_rel])    # side bit: the two sides never share rows
        uniq, inv, cnt = torch.unique(key, return_inverse=True, return_counts=True)
        order = torch.argsort(inv, stable=True)                       # queries grouped by key, original order inside
        start = torch.cumsum(cnt, 0) - cnt
        pos = torch.empty(n, dtype=torch.int64, device=dev)
        pos[order] = torch.arange(n, device=dev) - start[inv[order]]  # position of a query inside its key group
        chunks_of = (cnt + sets - 1) // sets                          # columns per key
        chunk0 = torch.cumsum(chunks_of, 0) - chunks_of
        chunk = chunk0[inv] + pos // sets                             # column (before the single / multi split)
        slot = pos % sets
        n_chunks = int(chunks_of.sum().item())                        # (host syncs at plan build only)
        size = torch.zeros(n_chunks, dtype=torch.int64, device=dev).scatter_add_(0, chunk, torch.ones_like(chunk))
        single = size == 1
        n1 = int(single.sum().item())
        n2 = n_chunks - n1
        self.n_single, self.n_multi = n1, n2
        self.n_single_p = pad(n1) if n1 > 0 else 0
        self.n_multi_p = pad(n2) if n2 > 0 else 0
        col1 = torch.cumsum(single.to(torch.int64), 0) - 1
        # grouped columns in order of DEcreasing size: a panel's compare loop runs as many times as its fullest
        # column has queries, so panels of equally full columns waste no passes
        order2 = torch.argsort(torch.where(single, torch.zeros_like(size), size), descending=True, stable=True)
        col2 = torch.empty(n_chunks, dtype=torch.int64, device=dev)
        col2[order2] = torch.arange(n_chunks, device=dev)        # (multi-query chunks occupy ranks 0 .. n2-1)
        is_single_q = single[chunk]
        self.col_q = torch.full((max(self.n_single_p, 1),), -1, dtype=torch.int32, device=dev)
        self.members = torch.full((max(self.n_multi_p, 1) * sets,), -1, dtype=torch.int32, device=dev)
        q = torch.arange(n, device=dev)
        self.col_q[col1[chunk[is_single_q]]] = q[is_single_q].to(torch.int32)
        mq = ~is_single_q
        self.members[col2[chunk[mq]] * sets + slot[mq]] = q[mq].to(torch.int32)
        row = torch.where(is_single_q, col1[chunk], self.n_single_p + col2[chunk])
        self.qs_row = torch.where(slot == 0, row, torch.full_like(row, -1)).to(torch.int32).contiguous()
        # for operands built by gathers (the non-fused query paths): the query that provides a column's row (padding
        # columns: query 0 -- their thresholds are +inf) and, per query, its column
        self.col_of_q = row.contiguous()
        self.rep = torch.zeros(max(self.n_single_p + self.n_multi_p, 1), dtype=torch.int64, device=dev)
        first = slot == 0
        self.rep[row[first]] = q[first]
        self.sets = sets
        self.n_columns = n_chunks
        self.n_distinct_keys = int(uniq.shape[0])


_CACHE = []          # [(dictionary, len, device, FilterIndex)] -- small LRU
_CACHE_MAX = 8


def filter_index_for(dictionary, device):
    """Cached FilterIndex of a dict-of-sets (identity + size keyed)."""
    device = torch.device(device)
    for k, (d, n, dev, idx) in enumerate(_CACHE):
        if d is dictionary and n == len(dictionary) and dev == device:
            _CACHE.append(_CACHE.pop(k))
            return idx
    idx = FilterIndex.from_dict(dictionary, device)
    _CACHE.append((dictionary, len(dictionary), device, idx))
    if len(_CACHE) > _CACHE_MAX:
        _CACHE.pop(0)
    return idx
