# -*- coding: utf-8 -*-
"""TransE / TransH / TransD with the reference's constructors, attributes and
state_dict keys (torchkge/models/translation.py:18-652) on the HIP engine.

TransH / TransD never build the reference's (n_rel, n_ent, d) projection cache
(`projected_entities`, filled by an n_ent-iteration Python loop,
translation.py:260-284 / :629-652): one scalar per (entity, relation) [TransH:
a = E.W^T] or per entity [TransD: s = Ep.E] is enough, and the rank-1
correction is applied inside the all-candidates kernel.
"""
import os

import torch

from .. import _hip
from ..exceptions import NotYetImplementedError
from ..utils.modeling import init_embedding
from .interfaces import TranslationModel, EntityCandidates, RelationProjections, _table_of


def _projections(kind, tabs, d_ent, d_rel, h_idx, t_idx, r_idx):
    """p_r(h), p_r(t) for the index vectors that are present (top-k inference
    passes an empty tensor for the missing side, inference.py:230-238)."""
    empty = torch.zeros(0, d_rel, dtype=torch.float32, device=r_idx.device)
    any_idx = h_idx if h_idx.shape[0] else t_idx
    proj_h = _hip.lp_prep(kind, _hip.SIDE_PROJ_H, tabs, d_ent, d_rel, h_idx, any_idx, r_idx, want_w=True)[0] \
        if h_idx.shape[0] else empty
    proj_t = _hip.lp_prep(kind, _hip.SIDE_PROJ_T, tabs, d_ent, d_rel, any_idx, t_idx, r_idx, want_w=True)[0] \
        if t_idx.shape[0] else empty
    return proj_h, proj_t


def _shard(table, lo, hi):
    return table if (lo == 0 and hi == table.shape[0]) else table[lo:hi]


def _ent_range(model, ent_lo, ent_hi):
    """Default candidate range: the whole table -- or, for a row-sharded model, its own rows."""
    if ent_hi is None:
        return model._row_shard if model._row_shard is not None else (0, model.n_ent)
    return ent_lo, ent_hi


class TransEModel(TranslationModel):
    """TransE (translation.py:18-125).  ``TransEModel(emb_dim, n_entities,
    n_relations, dissimilarity_type='L2')``; parameters ``ent_emb``, ``rel_emb``."""

    def __init__(self, emb_dim, n_entities, n_relations, dissimilarity_type='L2'):
        super().__init__(n_entities, n_relations, dissimilarity_type)
        self.emb_dim = emb_dim
        self.ent_emb = init_embedding(self.n_ent, self.emb_dim)
        self.rel_emb = init_embedding(self.n_rel, self.emb_dim)
        # translation.py:66-67: entities and (once) relations L2-normalised
        self.ent_emb.weight.data = torch.nn.functional.normalize(self.ent_emb.weight.data, p=2, dim=1)
        self.rel_emb.weight.data = torch.nn.functional.normalize(self.rel_emb.weight.data, p=2, dim=1)
        self._sad_bounds = self._abs_bounds

    _ENT_TABLES = ('ent_emb',)
    # the evaluator may hand lp_problem(side='both') a ColumnPlan: the fused query pipeline then writes one split row
    # per DISTINCT query row of the batch and the count kernel sweeps columns instead of queries
    lp_dedupe_queries = True
    # ... but not on the one-product level of the split prefilter: at d = 200 a shared row saves 4 MFMA groups per tile,
    # less than the grouped columns' multi-pass epilogue costs (r04 kernel: 0.632 vs 0.648 ms per evaluate).  r06 taught the
    # free-running kernel grouped columns (lp_hi_stream_kernel<.., GS = 4>) and measured again, same box, alternating: per
    # query 0.487-0.491 ms, columns 0.532-0.544 (profiles/r06/dedupe_level1_stream_ab.txt: the single-query launch 187 us +
    # the grouped one 104 us -- 57 panels, 6 items per workgroup, 2.85 compare passes per column -- against 283 us for the
    # per-query sweep; -24 % MFMAs do not pay on a kernel whose matrix pipe is 42 % busy).  KGE_TRANSE_DEDUPE_L1=1 turns it on.
    lp_dedupe_level1 = os.environ.get('KGE_TRANSE_DEDUPE_L1', '0') == '1'
    # the count sweep is enqueued in front of the second stream's filter correction (evaluation.COUNT_FIRST: -3 % here)
    lp_count_first = True

    def _tables(self):
        return [self.ent_emb.weight, self.rel_emb.weight]

    def _abs_bounds(self):
        """Device scalars (max |E|, max |R|) of this evaluation (guard slots 3 / 4, cached per session): every query
        element of TransE is e +- r, so their sum bounds both operands of the L1 prefilter."""
        g = self._lp_guard
        E, R = _hip.f32c(self.ent_emb.weight.data), _hip.f32c(self.rel_emb.weight.data)
        self._cache.get('sad_emax', [E], lambda: _hip.absmax(E, g[3:4]))
        self._cache.get('sad_rmax', [R], lambda: _hip.absmax(R, g[4:5]))
        return g[3:4], g[4:5]

    def _hip_kind(self):
        return _hip.TRANSE_L1 if self.dissimilarity_type == 'L1' else _hip.TRANSE_L2

    def normalize_parameters(self):
        """L2-normalise entity embeddings (translation.py:83-90)."""
        self._normalize_weight_(self.ent_emb)

    def get_embeddings(self):
        self.normalize_parameters()
        return self.ent_emb.weight.data, self.rel_emb.weight.data

    def inference_prepare_candidates(self, h_idx, t_idx, r_idx, entities=True):
        """(h, t, r, candidates); candidates is a stride-0 (b, N, d) view of the
        table, as in the reference (translation.py:105-125)."""
        self._check_unsharded('inference_prepare_candidates')
        b_size = max(h_idx.shape[0], t_idx.shape[0], r_idx.shape[0])   # inference passes one empty index
        E, R = self.ent_emb.weight.data, self.rel_emb.weight.data
        h, t, r = _hip.gather_rows(E, h_idx), _hip.gather_rows(E, t_idx), _hip.gather_rows(R, r_idx)
        if entities:
            candidates = E.view(1, self.n_ent, self.emb_dim).expand(b_size, self.n_ent, self.emb_dim)
        else:
            candidates = R.view(1, self.n_rel, self.emb_dim).expand(b_size, self.n_rel, self.emb_dim)
        return h, t, r, candidates

    def lp_problem(self, h_idx, t_idx, r_idx, side, ent_lo=0, ent_hi=None, exchange=None, qtabs=None, cols=None):
        ent_lo, ent_hi = _ent_range(self, ent_lo, ent_hi)
        tabs = [x.data for x in self._tables()]
        sd = _hip.side_code(side)
        fusable = (self.dissimilarity_type == 'L2' and self.l2_mode == 'auto' and self._guard_on and self._expand_ok is None
                   and self.split_filter and self._split_ok and h_idx.shape[0] > 0 and self.emb_dim % 4 == 0)
        if fusable and ent_lo == 0 and ent_hi == self.n_ent and self._row_shard is None:
            return self._fused_query_problem(h_idx, t_idx, r_idx, sd, tabs, cols if sd == _hip.SIDE_BOTH else None)
        if fusable and qtabs is not None and self._row_shard == (ent_lo, ent_hi) and sd == _hip.SIDE_BOTH:
            # ROW-SHARDED table with the replicas of the query entities' rows at hand (h_idx / t_idx index THEM): the same
            # fused query side, fed from the replicas; candidates = this rank's rows
            return self._fused_query_problem(h_idx, t_idx, r_idx, sd, tabs, None, qrep=_hip.f32c(qtabs[0]), c_base=ent_lo)
        Q0, _, _, _ = self._lp_prep(sd, h_idx, t_idx, r_idx, exchange, qtabs=qtabs)
        prob = self._translational_problem(Q0, self._cand_rows(_hip.f32c(tabs[0]), ent_lo, ent_hi),
                                           c_base=ent_lo)
        # (L2: f16-split count over columns; L1: the SAD count; broadcast-subtract L2: the packed-FMA count)
        plain_l2_direct = int(prob.desc.mode) == _hip.LP_L2_DIRECT and not prob.desc.Wq    # (kge_lp_count_ge_cols)
        prob.cols = cols if (sd == _hip.SIDE_BOTH and (prob.split is not None or prob.sad is not None or plain_l2_direct)) \
            else None
        return prob

    def _fused_query_problem(self, h_idx, t_idx, r_idx, sd, tabs, cols=None, qrep=None, c_base=0):
        """Inside evaluate(): the whole query side of a batch (q, ||q||^2, true scores, split queries, thresholds) from
        ONE kernel; the evaluator's pair_scores(true_idx) / count_ge calls then find their inputs ready.
        ``qrep`` (row-sharded tables, r05): replicas of the rows of the entities the queries mention -- the source and
        true-entity rows are read from THEM (h_idx / t_idx index the replicas; same rows, same chains, same bits as the
        owner shard's), while the candidate side -- norms, split / hi table, the bounds of the error band -- is this
        rank's own rows ``tabs[0]`` = global entities [c_base, c_base + rows)."""
        E = _hip.f32c(tabs[0])
        g = self._lp_guard
        key = '%d_%d' % (c_base, E.shape[0])
        Eq = E if qrep is None else qrep            # where the query pipeline reads e_src / e_true
        lvl1 = self._use_level1()
        frag = lvl1 and self._level1_stream()       # (r06: grouped columns run on the free-running kernel too)
        prep = None
        if frag:
            # the candidate side of the free-running sweep in ONE launch: ||e||^2 (the reference chain), the fragment-major
            # hi table and its residual maximum (guard slot 7, zeroed with the guard) -- kge_lp_table_prep_l2
            # (its two maxima stay per block -- hundreds of same-address atomics would serialise -- and every query
            # pipeline launch folds them into guard[1] / guard[7] on its way in: idempotent, 2 x 912 floats)
            prep = self._cache.get('tp_' + key, [E], lambda: _hip.table_prep_l2(E, g[1:2], g[7:8], deferred_max=True))
        if prep is not None:
            en = self._cache.get('en_' + key, [E], lambda: prep[0])
        else:
            en = self._cache.get('en_' + key, [E], lambda: _hip.row_sqnorm(E, max_io=g[1:2]))

        def enq():      # ||.||^2 of the rows the true scores are read from (replicas: their own chain norms, no maximum)
            if qrep is None:
                return en
            return self._cache.get('en_replica', [qrep], lambda: _hip.row_sqnorm(qrep))
        if lvl1:
            # one-product level of the split prefilter (a fitted model: the true entities sit in the sparse upper tail,
            # the 8x wider band still holds few pairs): planar hi table, thresholds from the measured f16 residuals
            if prep is not None:
                Eh, de2 = prep[1], g[7:8]
            else:
                Eh, de2 = self._cache.get('eh%d_' % frag + key, [E], lambda: _hip.hi_table(E, aug=en, frag=frag))
            tp_bmax = prep[2] if prep is not None else None
            pre = _hip.lp_query_pipeline(sd, Eq, tabs[1], h_idx, t_idx, r_idx, enq(), g[1:2], g[0:1], cols=cols, level=1,
                                         de2max=de2, tp_bmax=tp_bmax, zero_counts=True, regions=bool(frag) and bool(getattr(self, '_lp_regions', False)))
            split = {'Es': Eh, 'e2pref': None, 'enmax': g[1:2], 'overflow': g[2:3], 'level': 1, 'de2max': de2,
                     'list_stat': g[6:7], 'es_frag': frag}
        else:
            Es, e2 = self._cache.get('es_' + key, [E], lambda: _hip.split_table(E, aug=en))
            pre = _hip.lp_query_pipeline(sd, Eq, tabs[1], h_idx, t_idx, r_idx, enq(), g[1:2], g[0:1], e2pref=e2, cols=cols,
                                         zero_counts=True)
            split = {'Es': Es, 'e2pref': e2, 'enmax': g[1:2], 'overflow': g[2:3], 'list_stat': g[6:7]}
        # (SIDE_BOTH: the evaluator fills in the concatenated true indices it gets from the filter lookup)
        pre['true_idx'] = t_idx if sd == _hip.SIDE_TAIL else (h_idx if sd == _hip.SIDE_HEAD else None)
        prob = _hip.LpProblem(_hip.LP_L2_EXPAND, pre['Q'], E, qn=pre['qn'], en=en, c_base=c_base)
        prob.split = split
        prob.pre = pre
        return prob


def _both_r(r_idx, sd, hint=None):
    """Relation id per query: the 2B queries of a both-sides batch are the B facts twice (``hint``: that vector, precomputed by
    the evaluator with the batch's FilterPlan)."""
    r_idx = _hip.i64c(r_idx)
    if sd == _hip.SIDE_BOTH and hint is not None and hint.shape[0] == 2 * r_idx.shape[0] and hint.device == r_idx.device:
        return hint
    return torch.cat([r_idx, r_idx]) if sd == _hip.SIDE_BOTH else r_idx


class TransHModel(TranslationModel):
    """TransH (translation.py:128-284).  ``TransHModel(emb_dim, n_entities,
    n_relations)``; parameters ``ent_emb``, ``rel_emb``, ``norm_vect``."""

    def __init__(self, emb_dim, n_entities, n_relations):
        super().__init__(n_entities, n_relations, dissimilarity_type='L2')
        self.emb_dim = emb_dim
        self.ent_emb = init_embedding(self.n_ent, self.emb_dim)
        self.rel_emb = init_embedding(self.n_rel, self.emb_dim)
        self.norm_vect = init_embedding(self.n_rel, self.emb_dim)
        self._host_normalize()
        self.evaluated_projections = False

    _kind = _hip.TRANSH
    _ENT_TABLES = ('ent_emb',)
    # the count kernel's epilogue gathers X[r_i, c]: queries processed in relation order share those rows
    lp_sort_queries_by_relation = True
    lp_dedupe_queries = 'relation-major'   # ColumnPlan with the columns in relation order (same reason)
    lp_stream_columns = False              # (the free-running kernel's projection epilogue sweeps per query)

    def _tables(self):
        return [self.ent_emb.weight, self.rel_emb.weight, self.norm_vect.weight]

    @staticmethod
    def project(ent, norm_vect):
        return ent - (ent * norm_vect).sum(dim=1).view(-1, 1) * norm_vect

    def _host_normalize(self):
        F = torch.nn.functional
        self.ent_emb.weight.data = F.normalize(self.ent_emb.weight.data, p=2, dim=1)
        self.norm_vect.weight.data = F.normalize(self.norm_vect.weight.data, p=2, dim=1)
        self.rel_emb.weight.data = self.project(self.rel_emb.weight.data, self.norm_vect.weight.data)

    def normalize_parameters(self):
        """Normalise entities and normal vectors, re-project relations onto
        their hyperplanes (translation.py:208-219)."""
        self._normalize_weight_(self.ent_emb)
        self._normalize_weight_(self.norm_vect)
        R, W = self.rel_emb.weight.data, self.norm_vect.weight.data
        a = _hip.row_dot(R, W)                                   # (R.W) per relation
        # r - (r.w) w   via the ewise kernel: a*w then r - that
        aw = _hip.ewise(_hip.EW_MUL, a.view(-1, 1).expand_as(W).contiguous(), W)
        self.rel_emb.weight.data = _hip.ewise(_hip.EW_SUB, R, aw)

    def get_embeddings(self):
        self.normalize_parameters()
        return self.ent_emb.weight.data, self.rel_emb.weight.data, self.norm_vect.weight.data

    def _a_matrix(self, lo, hi):
        """a[c, r] = E[c].W[r] for c in [lo, hi): the only thing the
        reference's projected_entities cache is needed for (translation.py:272-281)."""
        E, W = _hip.f32c(self.ent_emb.weight.data), _hip.f32c(self.norm_vect.weight.data)
        Es = self._cand_rows(E, lo, hi)
        return self._cache.get('transh_a_%d_%d' % (lo, hi), [E, W],
                               lambda: _hip.LpProblem(_hip.LP_DOT, Es, W).scores())

    def _proj_problem(self, q, table, Wq, r_idx, c_base, K0, qn, en):
        """-||u - (e_c - x w)||^2 with x = e_c.w_{r_i}, expanded around the GEMM term u.e_c
        (KGE_LP_L2_PROJH): X[r, c] = W[r].E[c] is one small GEMM per evaluation, the
        per-query scalars are p = 2 u.w and z = ||w||^2 - 2."""
        XT, _ = self._proj_side_tables(table, c_base, K0)
        pz = torch.stack([_hip.row_dot(q, Wq, scale=2.0), _hip.row_sqnorm(Wq) - 2.0], dim=1).contiguous()
        prob = _hip.LpProblem(_hip.LP_L2_PROJH, q, table, qn=qn, en=en, Wq=pz, scal=XT, r_idx=r_idx,
                              c_base=c_base)
        return self._attach_proj_split(prob, table, en, XT, None, K0)

    def _proj_side_tables(self, table, c_base, K0):
        """(X, None): X[r, c] = W[r].E[c] for the candidate rows, one small GEMM per evaluation."""
        W = _hip.f32c(self.norm_vect.weight.data)
        n = table.shape[0]

        def build():    # (n_rel, n) view of a row-padded buffer: the split kernel reads whole 256-candidate tiles
            # (only the padding columns need the zeros -- the GEMM writes the rest: a strided fill of < 256 columns instead of
            # one of the whole 13.8 MB buffer)
            buf = torch.empty(W.shape[0], _hip.padded_cols(n), dtype=torch.float32, device=table.device)
            if buf.shape[1] > n:
                buf[:, n:].zero_()
            return _hip.LpProblem(_hip.LP_DOT, W, table).scores(buf[:, :n])
        return self._cache.get('transh_aT_%d_%d' % (c_base, n), [table, W], build), None

    def evaluate_projections(self):
        """Kept for API compatibility (translation.py:260-284); the engine needs
        no (n_rel, n_ent, d) cache, so this only marks the projections valid."""
        self.evaluated_projections = True

    def inference_prepare_candidates(self, h_idx, t_idx, r_idx, entities=True):
        """(proj_h, proj_t, r, candidates) (translation.py:234-258); candidates
        is an EntityCandidates handle, not a (b, N, d) copy."""
        self._check_unsharded('inference_prepare_candidates')
        tabs = [x.data for x in self._tables()]
        d = self.emb_dim
        if not entities:    # translation.py:252-256: every entity projected under EVERY relation
            r = _hip.gather_rows(tabs[1], r_idx)
            cand = tabs[1].view(1, self.n_rel, d).expand(h_idx.shape[0], self.n_rel, d)
            return (RelationProjections(self, _hip.i64c(h_idx), _hip.SIDE_PROJ_H),
                    RelationProjections(self, _hip.i64c(t_idx), _hip.SIDE_PROJ_T), r, cand)
        proj_h, proj_t = _projections(_hip.TRANSH, tabs, d, d, h_idx, t_idx, r_idx)
        r = _hip.gather_rows(tabs[1], r_idx)
        return proj_h, proj_t, r, EntityCandidates(self, _hip.i64c(r_idx), max(h_idx.shape[0], t_idx.shape[0]))

    def _relation_scores_proj(self, proj_h, proj_t, r):
        R = self.rel_emb.weight.data
        tab = _table_of(r)
        if tab is None or tab.data_ptr() != R.data_ptr() or tab.shape != R.shape:
            return None
        return _hip.relation_scores_proj(_hip.TRANSH, self.ent_emb.weight.data, R, self.norm_vect.weight.data, None,
                                         self.emb_dim, self.emb_dim, proj_h.idx, proj_t.idx)

    def _handle_problem(self, q, cand, ent_lo=0, ent_hi=None):
        ent_hi = self.n_ent if ent_hi is None else ent_hi
        E, W = _hip.f32c(self.ent_emb.weight.data), self.norm_vect.weight.data
        Wq = _hip.gather_rows(W, cand.r_idx)
        return self._translational_problem(q, self._cand_rows(E, ent_lo, ent_hi), Wq=Wq,
                                           scal=lambda: self._a_matrix(ent_lo, ent_hi), r_idx=cand.r_idx,
                                           c_base=ent_lo)

    def lp_problem(self, h_idx, t_idx, r_idx, side, ent_lo=0, ent_hi=None, exchange=None, qtabs=None, cols=None):
        ent_lo, ent_hi = _ent_range(self, ent_lo, ent_hi)
        tabs = [x.data for x in self._tables()]
        sd = _hip.side_code(side)
        r_both = _both_r(r_idx, sd, getattr(self, '_lp_r_both', None))
        if self._proj_fast_ok() and h_idx.shape[0] > 0:
            W = _hip.f32c(self.norm_vect.weight.data)
            prob = self._proj_fast_problem(sd, h_idx, t_idx, r_idx, r_both, ent_lo, ent_hi, exchange, qtabs,
                                           (_hip.LP_L2_PROJH, W, 2.0, -2.0, None, self._proj_side_tables))
            if prob is not None:
                prob.cols = cols if (sd == _hip.SIDE_BOTH and prob.split is not None) else None
                return prob
        Q0, _, _, Wq = self._lp_prep(sd, h_idx, t_idx, r_idx, exchange, qtabs=qtabs, want_w=True)
        prob = self._translational_problem(Q0, self._cand_rows(_hip.f32c(tabs[0]), ent_lo, ent_hi), Wq=Wq,
                                           scal=lambda: self._a_matrix(ent_lo, ent_hi),
                                           r_idx=r_both, c_base=ent_lo)
        prob.cols = cols if (sd == _hip.SIDE_BOTH and prob.split is not None) else None
        return prob


class TransDModel(TranslationModel):
    """TransD (translation.py:461-652).  ``TransDModel(ent_emb_dim, rel_emb_dim,
    n_entities, n_relations)``; parameters ``ent_emb``, ``rel_emb``,
    ``ent_proj_vect``, ``rel_proj_vect``.  Needs ent_emb_dim >= rel_emb_dim
    (the reference's ``ent[:, :rel_emb_dim]`` slice, :568)."""

    _kind = _hip.TRANSD
    _ENT_TABLES = ('ent_emb', 'ent_proj_vect')
    _ENT_POS = (0, 2)
    lp_sort_queries_by_relation = True     # (as TransH: the epilogue gathers G[r_i, c])
    lp_dedupe_queries = 'relation-major'
    lp_stream_columns = False

    def __init__(self, ent_emb_dim, rel_emb_dim, n_entities, n_relations):
        super().__init__(n_entities, n_relations, 'L2')
        self.ent_emb_dim = ent_emb_dim
        self.rel_emb_dim = rel_emb_dim
        self.ent_emb = init_embedding(self.n_ent, self.ent_emb_dim)
        self.rel_emb = init_embedding(self.n_rel, self.rel_emb_dim)
        self.ent_proj_vect = init_embedding(self.n_ent, self.ent_emb_dim)
        self.rel_proj_vect = init_embedding(self.n_rel, self.rel_emb_dim)
        F = torch.nn.functional
        for emb in (self.ent_emb, self.rel_emb, self.ent_proj_vect, self.rel_proj_vect):
            emb.weight.data = F.normalize(emb.weight.data, p=2, dim=1)
        self.evaluated_projections = False

    def _tables(self):
        return [self.ent_emb.weight, self.rel_emb.weight, self.ent_proj_vect.weight,
                self.rel_proj_vect.weight]

    def normalize_parameters(self):
        """L2-normalise all four tables (translation.py:570-579)."""
        for emb in (self.ent_emb, self.rel_emb, self.ent_proj_vect, self.rel_proj_vect):
            self._normalize_weight_(emb)

    def get_embeddings(self):
        self.normalize_parameters()
        return (self.ent_emb.weight.data, self.rel_emb.weight.data,
                self.ent_proj_vect.weight.data, self.rel_proj_vect.weight.data)

    def _neg_sigma(self, lo, hi):
        """-s[c], s[c] = Ep[c].E[c]: all the reference's projected_entities
        cache depends on per entity (translation.py:641-646)."""
        E, Ep = _hip.f32c(self.ent_emb.weight.data), _hip.f32c(self.ent_proj_vect.weight.data)
        return self._cache.get('transd_s_%d_%d' % (lo, hi), [E, Ep],
                               lambda: _hip.row_dot(self._cand_rows(Ep, lo, hi), self._cand_rows(E, lo, hi), scale=-1.0))

    def evaluate_projectionss(self):
        """Kept for API compatibility (translation.py:629-652, reference spelling)."""
        self.evaluated_projections = True

    evaluate_projections = evaluate_projectionss

    def inference_prepare_candidates(self, h_idx, t_idx, r_idx, entities=True):
        """(proj_h, proj_t, r, candidates) (translation.py:603-627)."""
        self._check_unsharded('inference_prepare_candidates')
        tabs = [x.data for x in self._tables()]
        de, dr = self.ent_emb_dim, self.rel_emb_dim
        if not entities:    # translation.py:621-626
            r = _hip.gather_rows(tabs[1], r_idx)
            cand = tabs[1].view(1, self.n_rel, dr).expand(h_idx.shape[0], self.n_rel, dr)
            return (RelationProjections(self, _hip.i64c(h_idx), _hip.SIDE_PROJ_H),
                    RelationProjections(self, _hip.i64c(t_idx), _hip.SIDE_PROJ_T), r, cand)
        proj_h, proj_t = _projections(_hip.TRANSD, tabs, de, dr, h_idx, t_idx, r_idx)
        r = _hip.gather_rows(tabs[1], r_idx)
        return proj_h, proj_t, r, EntityCandidates(self, _hip.i64c(r_idx), max(h_idx.shape[0], t_idx.shape[0]))

    def _relation_scores_proj(self, proj_h, proj_t, r):
        R = self.rel_emb.weight.data
        tab = _table_of(r)
        if tab is None or tab.data_ptr() != R.data_ptr() or tab.shape != R.shape:
            return None
        return _hip.relation_scores_proj(_hip.TRANSD, self.ent_emb.weight.data, R, self.rel_proj_vect.weight.data,
                                         self.ent_proj_vect.weight.data, self.ent_emb_dim, self.rel_emb_dim,
                                         proj_h.idx, proj_t.idx)

    def _problem(self, q, Wq, ent_lo, ent_hi, r_idx=None):
        E = _hip.f32c(self.ent_emb.weight.data)
        # candidates use E[c, :d_r]: same rows, inner dim K0 = d_r, leading dim d_e
        return self._translational_problem(q, self._cand_rows(E, ent_lo, ent_hi), Wq=Wq,
                                           scal=lambda: self._neg_sigma(ent_lo, ent_hi), c_base=ent_lo,
                                           K0=self.rel_emb_dim, r_idx=r_idx)

    def _proj_problem(self, q, table, Wq, r_idx, c_base, K0, qn, en):
        """-||u - (e'_c + y_c w)||^2, e' = e[:d_r], y_c = ep_c.e_c, expanded around the GEMM
        term u.e'_c (KGE_LP_L2_PROJD): G[r, c] = Rp[r].e'_c is one small GEMM per evaluation,
        the per-query scalars are p = -2 u.w and z = ||w||^2."""
        GT, sigma = self._proj_side_tables(table, c_base, K0)
        pz = torch.stack([_hip.row_dot(q, Wq, scale=-2.0), _hip.row_sqnorm(Wq)], dim=1).contiguous()
        prob = _hip.LpProblem(_hip.LP_L2_PROJD, q, table, qn=qn, en=en, Wq=pz, scal=GT, r_idx=r_idx, yc=sigma,
                              c_base=c_base, K0=K0)
        return self._attach_proj_split(prob, table, en, GT, sigma, K0)

    def _proj_side_tables(self, table, c_base, K0):
        """(G, sigma): G[r, c] = Rp[r].E[c, :d_r] (one small GEMM per evaluation) and sigma[c] = Ep[c].E[c]."""
        lo, hi = c_base, c_base + table.shape[0]
        Rp = _hip.f32c(self.rel_proj_vect.weight.data)
        Ep = _hip.f32c(self.ent_proj_vect.weight.data)
        n = table.shape[0]

        def build_g():
            buf = torch.empty(Rp.shape[0], _hip.padded_cols(n), dtype=torch.float32, device=table.device)
            if buf.shape[1] > n:
                buf[:, n:].zero_()
            return _hip.LpProblem(_hip.LP_DOT, Rp, table, K0=K0).scores(buf[:, :n])

        def build_s():
            buf = torch.zeros(_hip.padded_cols(n), dtype=torch.float32, device=table.device)
            buf[:n] = _hip.row_dot(self._cand_rows(Ep, lo, hi), table, scale=1.0)
            return buf[:n]
        GT = self._cache.get('transd_gT_%d_%d' % (lo, hi), [table, Rp], build_g)
        sigma = self._cache.get('transd_sp_%d_%d' % (lo, hi), [table, Ep], build_s)
        return GT, sigma

    def _handle_problem(self, q, cand, ent_lo=0, ent_hi=None):
        ent_hi = self.n_ent if ent_hi is None else ent_hi
        Wq = _hip.gather_rows(self.rel_proj_vect.weight.data, cand.r_idx)
        return self._problem(q, Wq, ent_lo, ent_hi, r_idx=cand.r_idx)

    def lp_problem(self, h_idx, t_idx, r_idx, side, ent_lo=0, ent_hi=None, exchange=None, qtabs=None, cols=None):
        ent_lo, ent_hi = _ent_range(self, ent_lo, ent_hi)
        sd = _hip.side_code(side)
        r_both = _both_r(r_idx, sd, getattr(self, '_lp_r_both', None))
        if self._proj_fast_ok() and h_idx.shape[0] > 0:
            Rp = _hip.f32c(self.rel_proj_vect.weight.data)
            prob = self._proj_fast_problem(sd, h_idx, t_idx, r_idx, r_both, ent_lo, ent_hi, exchange, qtabs,
                                           (_hip.LP_L2_PROJD, Rp, -2.0, 0.0, self.rel_emb_dim, self._proj_side_tables))
            if prob is not None:
                prob.cols = cols if (sd == _hip.SIDE_BOTH and prob.split is not None) else None
                return prob
        Q0, _, _, Wq = self._lp_prep(sd, h_idx, t_idx, r_idx, exchange, qtabs=qtabs, want_w=True)
        prob = self._problem(Q0, Wq, ent_lo, ent_hi, r_idx=r_both)
        prob.cols = cols if (sd == _hip.SIDE_BOTH and prob.split is not None) else None
        return prob
