# -*- coding: utf-8 -*-
"""DistMult / ComplEx with the reference's constructors, attributes and
state_dict keys (torchkge/models/bilinear.py:146-267, :414-556) on the HIP
engine: the all-candidates score matrix is one fp32 MFMA GEMM
S = Q . E^T (ComplEx: K = 2d over the [Re | Im] tables, no concatenation)."""
import torch

from .. import _hip
from ..utils.modeling import init_embedding
from .interfaces import BilinearModel, _table_of
from .translation import _ent_range


class DistMultModel(BilinearModel):
    """DistMult (bilinear.py:146-267): ``DistMultModel(emb_dim, n_entities,
    n_relations)``; parameters ``ent_emb`` (L2-normalised rows), ``rel_emb``."""

    _kind = _hip.DISTMULT
    _ENT_TABLES = ('ent_emb',)

    def __init__(self, emb_dim, n_entities, n_relations):
        super().__init__(emb_dim, n_entities, n_relations)
        self.ent_emb = init_embedding(self.n_ent, self.emb_dim)
        self.rel_emb = init_embedding(self.n_rel, self.emb_dim)
        self.ent_emb.weight.data = torch.nn.functional.normalize(self.ent_emb.weight.data, p=2, dim=1)

    def _tables(self):
        return [self.ent_emb.weight, self.rel_emb.weight]

    def normalize_parameters(self):
        """L2-normalise entity embeddings (bilinear.py:201-208)."""
        self._normalize_weight_(self.ent_emb)

    def get_embeddings(self):
        self.normalize_parameters()
        return self.ent_emb.weight.data, self.rel_emb.weight.data

    def inference_scoring_function(self, h, t, r):
        """Rank dispatch of bilinear.py:224-245: the 3-D argument is the
        candidate set; s = (a * b) . cand."""
        if t.dim() == 3:
            assert h.dim() == 2 and r.dim() == 2
            q, cand = _hip.ewise(_hip.EW_MUL, h, r), t          # tail completion
        elif h.dim() == 3:
            assert t.dim() == 2 and r.dim() == 2
            q, cand = _hip.ewise(_hip.EW_MUL, r, t), h          # head completion
        else:
            assert r.dim() == 3 and h.dim() == 2 and t.dim() == 2
            q, cand = _hip.ewise(_hip.EW_MUL, h, t), r          # relation prediction
        table = _table_of(cand)
        if table is not None:
            return _hip.LpProblem(_hip.LP_DOT, q, _hip.f32c(table)).scores()
        return _hip.lp_scores_batched(_hip.LP_DOT, q, cand)

    def inference_prepare_candidates(self, h_idx, t_idx, r_idx, entities=True):
        """(h, t, r, candidates) with a stride-0 (b, N, d) candidates view
        (bilinear.py:247-267)."""
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
        if ent_lo == 0 and ent_hi == self.n_ent and self._row_shard is None and exchange is None and qtabs is None:
            prob = self._dot_fused_problem(sd, h_idx, t_idx, r_idx, [_hip.f32c(tabs[0])], [_hip.f32c(tabs[1])])
            if prob is not None:
                return prob
        Q0 = self._lp_prep(sd, h_idx, t_idx, r_idx, exchange, qtabs=qtabs)[0]
        T0 = self._cand_rows(_hip.f32c(tabs[0]), ent_lo, ent_hi)
        prob = self._attach_dot_split(_hip.LpProblem(_hip.LP_DOT, Q0, T0, c_base=ent_lo), T0, c_base=ent_lo)
        prob.cols = cols if (sd == _hip.SIDE_BOTH and prob.split is not None) else None
        return prob


class ComplExModel(BilinearModel):
    """ComplEx (bilinear.py:414-556): ``ComplExModel(emb_dim, n_entities,
    n_relations)``; parameters ``re_ent_emb``, ``im_ent_emb``, ``re_rel_emb``,
    ``im_rel_emb``; never normalised (:475-480)."""

    _kind = _hip.COMPLEX
    _ENT_TABLES = ('re_ent_emb', 'im_ent_emb')
    _ENT_POS = (0, 1)

    def __init__(self, emb_dim, n_entities, n_relations):
        super().__init__(emb_dim, n_entities, n_relations)
        self.re_ent_emb = init_embedding(self.n_ent, self.emb_dim)
        self.im_ent_emb = init_embedding(self.n_ent, self.emb_dim)
        self.re_rel_emb = init_embedding(self.n_rel, self.emb_dim)
        self.im_rel_emb = init_embedding(self.n_rel, self.emb_dim)

    def _tables(self):
        return [self.re_ent_emb.weight, self.im_ent_emb.weight, self.re_rel_emb.weight,
                self.im_rel_emb.weight]

    @property
    def _d_rel(self):
        return self.emb_dim

    def _lp_width(self):
        return 2 * self.emb_dim

    def normalize_parameters(self):
        """No normalisation for ComplEx (bilinear.py:475-480)."""
        pass

    def get_embeddings(self):
        return (self.re_ent_emb.weight.data, self.im_ent_emb.weight.data,
                self.re_rel_emb.weight.data, self.im_rel_emb.weight.data)

    def inference_scoring_function(self, h, t, r):
        """h, t, r are (re, im) tuples; the pair holding 3-D tensors is the
        candidate set (bilinear.py:501-528)."""
        re_h, im_h = h[0], h[1]
        re_t, im_t = t[0], t[1]
        re_r, im_r = r[0], r[1]
        E = _hip
        if re_t.dim() == 3:      # tail: (re_h re_r - im_h im_r).Re + (re_h im_r + im_h re_r).Im
            assert re_h.dim() == 2 and re_r.dim() == 2
            A = E.ewise(E.EW_MULSUB, re_h, re_r, im_h, im_r)
            Bq = E.ewise(E.EW_MULADD, re_h, im_r, im_h, re_r)
            cre, cim = re_t, im_t
        elif re_h.dim() == 3:    # head: Re.(re_r re_t + im_r im_t) + Im.(re_r im_t - im_r re_t)
            assert re_t.dim() == 2 and re_r.dim() == 2
            A = E.ewise(E.EW_MULADD, re_r, re_t, im_r, im_t)
            Bq = E.ewise(E.EW_MULSUB, re_r, im_t, im_r, re_t)
            cre, cim = re_h, im_h
        else:                    # relation: (re_h re_t + im_h im_t).Re_r + (re_h im_t - im_h re_t).Im_r
            assert re_r.dim() == 3 and re_h.dim() == 2 and re_t.dim() == 2
            A = E.ewise(E.EW_MULADD, re_h, re_t, im_h, im_t)
            Bq = E.ewise(E.EW_MULSUB, re_h, im_t, im_h, re_t)
            cre, cim = re_r, im_r
        tre, tim = _table_of(cre), _table_of(cim)
        if tre is not None and tim is not None:
            return E.LpProblem(E.LP_DOT, A, E.f32c(tre), A1=Bq, T1=E.f32c(tim)).scores()
        s = E.lp_scores_batched(E.LP_DOT, A, cre)
        return E.ewise(E.EW_ADD, s, E.lp_scores_batched(E.LP_DOT, Bq, cim))

    def inference_prepare_candidates(self, h_idx, t_idx, r_idx, entities=True):
        """((re_h, im_h), (re_t, im_t), (re_r, im_r), (re_cand, im_cand))
        (bilinear.py:530-556)."""
        self._check_unsharded('inference_prepare_candidates')
        b_size = max(h_idx.shape[0], t_idx.shape[0], r_idx.shape[0])   # inference passes one empty index
        Ere, Eim, Rre, Rim = [x.data for x in self._tables()]
        g = _hip.gather_rows
        h = (g(Ere, h_idx), g(Eim, h_idx))
        t = (g(Ere, t_idx), g(Eim, t_idx))
        r = (g(Rre, r_idx), g(Rim, r_idx))
        if entities:
            shape = (b_size, self.n_ent, self.emb_dim)
            cand = (Ere.view(1, self.n_ent, self.emb_dim).expand(*shape),
                    Eim.view(1, self.n_ent, self.emb_dim).expand(*shape))
        else:
            shape = (b_size, self.n_rel, self.emb_dim)
            cand = (Rre.view(1, self.n_rel, self.emb_dim).expand(*shape),
                    Rim.view(1, self.n_rel, self.emb_dim).expand(*shape))
        return h, t, r, cand

    def lp_problem(self, h_idx, t_idx, r_idx, side, ent_lo=0, ent_hi=None, exchange=None, qtabs=None, cols=None):
        ent_lo, ent_hi = _ent_range(self, ent_lo, ent_hi)
        tabs = [x.data for x in self._tables()]
        sd = _hip.side_code(side)
        if ent_lo == 0 and ent_hi == self.n_ent and self._row_shard is None and exchange is None and qtabs is None:
            prob = self._dot_fused_problem(sd, h_idx, t_idx, r_idx, [_hip.f32c(tabs[0]), _hip.f32c(tabs[1])],
                                           [_hip.f32c(tabs[2]), _hip.f32c(tabs[3])])
            if prob is not None:
                return prob
        Q0, Q1, _, _ = self._lp_prep(sd, h_idx, t_idx, r_idx, exchange, qtabs=qtabs, want_q1=True)
        T0, T1 = self._cand_rows(_hip.f32c(tabs[0]), ent_lo, ent_hi), self._cand_rows(_hip.f32c(tabs[1]), ent_lo, ent_hi)
        prob = self._attach_dot_split(_hip.LpProblem(_hip.LP_DOT, Q0, T0, A1=Q1, T1=T1, c_base=ent_lo), T0, T1,
                                      c_base=ent_lo)
        prob.cols = cols if (sd == _hip.SIDE_BOTH and prob.split is not None) else None
        return prob
