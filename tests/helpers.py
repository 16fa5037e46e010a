"""Shared test helpers (oracle loading, golden fixtures).  Test-only."""
import ctypes
import os
import subprocess

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

KIND_FILES = {
    ('transe', 2): 'ref_transe.npz', ('transe', 1): 'ref_transe_l1.npz',
    ('transh', 2): 'ref_transh.npz', ('transd', 2): 'ref_transd.npz',
    ('distmult', 2): 'ref_distmult.npz', ('complex', 2): 'ref_complex.npz',
}
N_TABLES = {'transe': 2, 'transh': 3, 'transd': 4, 'distmult': 2, 'complex': 4}


def load_golden(kind, p=2):
    z = np.load(os.path.join(GOLDEN, KIND_FILES[(kind, p)]))
    tables = [torch.from_numpy(z['table%d' % i]) for i in range(N_TABLES[kind])]
    return z, tables


def oracle_clib():
    """Build (if needed) and load the C oracle."""
    so = os.path.join(ROOT, 'oracle', '_build', 'libkge_oracle.so')
    src = os.path.join(ROOT, 'oracle', 'kge_oracle.c')
    if (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle')])
    return ctypes.CDLL(so)


def fptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def dict_to_csr(dictionary, key1, key2):
    """Per-query CSR view of dictionary[(key1_i, key2_i)] (python, test-only)."""
    has, off, tgt = [], [0], []
    for a, b in zip(np.asarray(key1).tolist(), np.asarray(key2).tolist()):
        if (a, b) in dictionary:
            has.append(1)
            tgt.extend(sorted(dictionary[(a, b)]))
        else:
            has.append(0)
        off.append(len(tgt))
    return (np.array(has, dtype=np.uint8), np.array(off, dtype=np.int64),
            np.array(tgt, dtype=np.int64))


class OracleRankEngine(object):
    """CPU stand-in for torchkge_amd.evaluation.HipRankEngine built on the
    oracle (TEST ONLY): lets the sharding / exchange logic of
    LinkPredictionEvaluator run under gloo with world_size > 1 on CPU."""

    name = 'oracle'

    def __init__(self, kind, tables, p=2):
        self.kind, self.tables, self.p = kind, tables, p

    def check_device(self, device):
        pass

    def lookup(self, index, key1, key2):
        from torchkge_amd.filter_index import KEY2_SPAN
        key = key1 * KEY2_SPAN + key2
        pos = torch.searchsorted(index.keys, key)
        posc = pos.clamp(max=max(index.n_keys - 1, 0))
        hit = (pos < index.n_keys) & (index.keys[posc] == key) if index.n_keys else torch.zeros_like(key, dtype=torch.bool)
        lo = torch.where(hit, index.offsets[posc], torch.zeros_like(key))
        hi = torch.where(hit, index.offsets[(posc + 1).clamp(max=index.offsets.shape[0] - 1)], torch.zeros_like(key))
        return lo, hi

    def lookup_both(self, index_t, index_h, h, t, r):
        lo_t, hi_t = self.lookup(index_t, h, r)
        lo_h, hi_h = self.lookup(index_h, t, r)
        base = index_t.targets.shape[0]
        found = hi_h > lo_h
        return (torch.cat([lo_t, torch.where(found, lo_h + base, lo_h)]),
                torch.cat([hi_t, torch.where(found, hi_h + base, hi_h)]),
                torch.cat([t, h]), torch.cat([index_t.targets, index_h.targets]))

    def problem(self, model, h, t, r, side, lo, hi, exchange=None):
        from oracle import kge_oracle as orc
        if side == 'both':      # 2B queries: tail side first
            full = torch.cat([orc.lp_scores(self.kind, self.tables, h, t, r, 'tail', self.p),
                              orc.lp_scores(self.kind, self.tables, h, t, r, 'head', self.p)])
        else:
            full = orc.lp_scores(self.kind, self.tables, h, t, r, side, self.p)

        class P(object):
            pass
        pr = P()
        pr.B, pr.lo, pr.hi, pr.full = full.shape[0], lo, hi, full
        return pr

    def true_scores(self, prob, true_idx):
        own = (true_idx >= prob.lo) & (true_idx < prob.hi)
        st = prob.full.gather(1, true_idx.view(-1, 1)).view(-1)
        return torch.where(own, st, torch.zeros_like(st))

    def partial_counts(self, prob, s_true, true_idx, seg_lo, seg_hi, targets):
        out = torch.zeros(3, prob.B, dtype=torch.int32)
        loc = prob.full[:, prob.lo:prob.hi]
        out[0] = (loc >= s_true.view(-1, 1)).sum(1).int()
        for i in range(prob.B):
            tv = s_true[i]
            neg = 1 if (-float('inf') >= tv) else 0
            for c in targets[int(seg_lo[i]):int(seg_hi[i])].tolist():
                if c < prob.lo or c >= prob.hi:
                    continue
                if c == int(true_idx[i]):
                    out[2, i] = 1
                    continue
                out[1, i] += int(prob.full[i, c] >= tv) - neg
        return out

    def finalize(self, counts):
        raw = counts[0].long()
        filt = torch.where(counts[2] > 0, raw - counts[1].long(), raw)
        return raw, filt

    def finalize_both(self, counts, out, off):
        raw, filt = self.finalize(counts)
        B = raw.shape[0] // 2
        out[1, off:off + B], out[3, off:off + B] = raw[:B], filt[:B]
        out[0, off:off + B], out[2, off:off + B] = raw[B:], filt[B:]

    def local_scores(self, prob):
        return prob.full[:, prob.lo:prob.hi].contiguous()

    def score_rows(self, prob, q0, q1, out):
        loc = self.local_scores(prob)
        out[:q1 - q0, :loc.shape[1]] = loc[q0:q1]
        return out

    def rank_tiles(self, tiles, n_total, true_idx, seg_lo, seg_hi, targets, rows, q_first, B, out, off, pos=None, own=None,
                   own_rank=0):
        """Rank-major tiles (P, m, per) of the score all-to-all -> ranks of `rows` queries, written like finalize_both."""
        assert pos is None
        if own is not None:
            tiles = tiles.clone()
            tiles[own_rank] = own
        P, m, per = tiles.shape
        full = tiles.permute(1, 0, 2).reshape(m, P * per)[:rows, :n_total]
        rk, frk = self.ranks_from_scores(full, true_idx[:rows], seg_lo[:rows], seg_hi[:rows], targets)
        for i in range(rows):
            q = q_first + i
            tail = q < B
            f = off + (q if tail else q - B)
            out[1 if tail else 0, f], out[3 if tail else 2, f] = rk[i], frk[i]

    def ranks_from_scores(self, scores, true_idx, seg_lo, seg_hi, targets):
        B = scores.shape[0]
        rk = torch.empty(B, dtype=torch.long)
        frk = torch.empty(B, dtype=torch.long)
        for i in range(B):
            row = scores[i]
            tv = row[true_idx[i]]
            raw = int((row >= tv).sum())
            seg = targets[int(seg_lo[i]):int(seg_hi[i])].tolist()
            rk[i] = raw
            if int(true_idx[i]) in seg:
                neg = 1 if (-float('inf') >= tv) else 0
                sub = sum(int(row[c] >= tv) - neg for c in seg if c != int(true_idx[i]))
                frk[i] = raw - sub
            else:
                frk[i] = raw
        return rk, frk


class ShardedOracleEngine(OracleRankEngine):
    """CPU stand-in for the HIP engine on a ROW-SHARDED model (TEST ONLY): it reads the entity
    tables from the model itself -- which holds only rows [lo, hi) after distributed.shard_model_ --
    builds the rows of the queries whose entity it owns (zeros elsewhere), has the evaluator's
    exchange sum them over the ranks, and scores its own candidates with the reference's formulas
    (oracle.lp_scores, restated on explicit rows).  TransE / DistMult / ComplEx."""

    name = 'oracle-sharded'

    def __init__(self, kind, p=2):
        self.kind, self.p = kind, p

    @staticmethod
    def _rows(table, idx, lo, hi):
        own = (idx >= lo) & (idx < hi)
        out = torch.zeros(idx.shape[0], table.shape[1])
        out[own] = table[idx[own] - lo]
        return out

    def problem(self, model, h, t, r, side, lo, hi, exchange=None):
        from oracle import kge_oracle as orc
        assert model._row_shard == (lo, hi) and exchange is not None
        tabs = [x.data for x in model._tables()]
        n_et = 2 if self.kind == 'complex' else 1
        ent, rel = tabs[:n_et], tabs[n_et:]
        assert all(x.shape[0] == hi - lo for x in ent)
        rows_h = [self._rows(x, h, lo, hi) for x in ent]
        rows_t = [self._rows(x, t, lo, hi) for x in ent]
        exchange(rows_h + rows_t)                   # one SUM all-reduce per matrix: x + 0 is exact
        b = h.shape[0]
        rr = [x[r] for x in rel]

        def tile(sd):
            if self.kind == 'transe':               # oracle._translation_inference
                cand = ent[0].view(1, hi - lo, -1).expand(b, -1, -1)
                if sd == 'tail':
                    return orc._translation_inference(self.p, rows_h[0], cand, rr[0])
                return orc._translation_inference(self.p, cand, rows_t[0], rr[0])
            if self.kind == 'distmult':
                d = ent[0].shape[1]
                cand = ent[0].view(1, hi - lo, d).expand(b, -1, -1)
                if sd == 'tail':
                    return ((rows_h[0] * rr[0]).view(b, 1, d) * cand).sum(dim=2)
                return (cand * (rr[0] * rows_t[0]).view(b, 1, d)).sum(dim=2)
            d = ent[0].shape[1]
            re_c = ent[0].view(1, -1, d).expand(b, -1, -1)
            im_c = ent[1].view(1, -1, d).expand(b, -1, -1)
            (re_h, im_h), (re_t, im_t), (re_r, im_r) = rows_h, rows_t, rr
            if sd == 'tail':
                return ((re_h * re_r - im_h * im_r).view(b, 1, d) * re_c
                        + (re_h * im_r + im_h * re_r).view(b, 1, d) * im_c).sum(dim=2)
            return (re_c * (re_r * re_t + im_r * im_t).view(b, 1, d)
                    + im_c * (re_r * im_t - im_r * re_t).view(b, 1, d)).sum(dim=2)

        class P(object):
            pass
        pr = P()
        pr.local = torch.cat([tile('tail'), tile('head')]) if side == 'both' else tile(side)
        pr.B, pr.lo, pr.hi = pr.local.shape[0], lo, hi
        return pr

    def true_scores(self, prob, true_idx):
        own = (true_idx >= prob.lo) & (true_idx < prob.hi)
        st = prob.local.gather(1, (true_idx - prob.lo).clamp(0, prob.hi - prob.lo - 1).view(-1, 1)).view(-1)
        return torch.where(own, st, torch.zeros_like(st))

    def partial_counts(self, prob, s_true, true_idx, seg_lo, seg_hi, targets):
        out = torch.zeros(3, prob.B, dtype=torch.int32)
        out[0] = (prob.local >= s_true.view(-1, 1)).sum(1).int()
        for i in range(prob.B):
            tv = s_true[i]
            neg = 1 if (-float('inf') >= tv) else 0
            for c in targets[int(seg_lo[i]):int(seg_hi[i])].tolist():
                if c < prob.lo or c >= prob.hi:
                    continue
                if c == int(true_idx[i]):
                    out[2, i] = 1
                    continue
                out[1, i] += int(prob.local[i, c - prob.lo] >= tv) - neg
        return out

    def local_scores(self, prob):
        return prob.local.contiguous()
