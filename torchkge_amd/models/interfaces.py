# -*- coding: utf-8 -*-
"""Model interfaces with the reference's public surface
(torchkge/models/interfaces.py:13-330) on top of the HIP engine.

Kept from the reference: ``forward`` (incl. the n_neg tiling),
``scoring_function``, ``normalize_parameters``, ``get_embeddings``,
``inference_prepare_candidates`` / ``inference_scoring_function`` with the
rank-based dispatch (the 3-D argument is the candidate set), and the pre-0.17
aliases ``lp_prep_cands`` / ``lp_scoring_function`` (docs/history.rst:153-157).

Added for the engine: ``lp_problem`` -- the descriptor of one all-candidates
scoring problem straight from index vectors, which is what
LinkPredictionEvaluator uses so that the (B, N) score matrix is never written.
"""
import os

import torch
from torch.nn import Module

from .. import _hip
from ..exceptions import NotYetImplementedError


class _ScoreTriples(torch.autograd.Function):
    """scoring_function as one fused HIP kernel, differentiable wrt the tables
    (kge_score_triples / kge_score_triples_bwd)."""

    @staticmethod
    def forward(ctx, kind, d_ent, d_rel, h, t, r, *tables):
        tabs = [x.detach() for x in tables]
        ctx.kind, ctx.d_ent, ctx.d_rel = kind, d_ent, d_rel
        ctx.save_for_backward(h, t, r, *tabs)
        return _hip.score_triples(kind, tabs, d_ent, d_rel, h, t, r)

    @staticmethod
    def backward(ctx, grad_out):
        h, t, r = ctx.saved_tensors[:3]
        tabs = list(ctx.saved_tensors[3:])
        needs = ctx.needs_input_grad[6:]
        grads = _hip.score_triples_bwd(ctx.kind, tabs, ctx.d_ent, ctx.d_rel, _hip.i64c(h),
                                       _hip.i64c(t), _hip.i64c(r), grad_out, needs)
        return (None,) * 6 + tuple(grads)


class _SessionCache(object):
    """Cache of per-entity precomputes (||E[c]||^2, E.W^T, Ep.E ...).  Only
    active inside ``with model.lp_session():`` (the evaluator opens one per
    ``evaluate`` call, during which the tables cannot change); outside a session
    every lookup rebuilds, so stale values are impossible."""

    def __init__(self):
        self.store = {}
        self.depth = 0

    def get(self, name, sources, build):
        if self.depth == 0:
            return build()
        sig = tuple((s.data_ptr(), tuple(s.shape), str(s.device)) for s in sources)
        hit = self.store.get(name)
        if hit is not None and hit[0] == sig:
            return hit[1]
        val = build()
        self.store[name] = (sig, val)
        return val

    def __enter__(self):
        self.depth += 1
        return self

    def __exit__(self, *exc):
        self.depth -= 1
        if self.depth == 0:
            self.store.clear()
        return False


class EntityCandidates(object):
    """Light handle standing for the reference's (b, N, d) ``candidates`` tensor
    when candidates are *projected* entities (TransH / TransD,
    translation.py:251, :620): it names the table and the per-query relation
    instead of materialising 4*b*N*d bytes."""

    def __init__(self, model, r_idx, b_size):
        self.model = model
        self.r_idx = r_idx
        self.shape = (b_size, model.n_ent, model._d_rel)

    def dim(self):
        return 3


class RelationProjections(object):
    """Light handle standing for the reference's (b, n_rel, d) ``proj_h`` / ``proj_t`` of relation
    prediction with relation-specific projections (TransH / TransD:
    ``projected_entities[:, idx].transpose(0, 1)``, translation.py:253-254, :622-623): the
    projection of each entity of ``idx`` under EVERY relation.  It names the entities instead of
    gathering b * n_rel * d floats from an (n_rel, n_ent, d) cache the engine never builds;
    ``materialize()`` gives the tensor itself."""

    def __init__(self, model, idx, side):
        self.model, self.idx, self.side = model, idx, side
        self.shape = (idx.shape[0], model.n_rel, model._d_rel)

    def dim(self):
        return 3

    def materialize(self):
        m = self.model
        b, n_rel, dr = self.shape
        tabs = [x.data for x in m._tables()]
        e = self.idx.repeat_interleave(n_rel)
        r = torch.arange(n_rel, device=self.idx.device).repeat(b)
        q = _hip.lp_prep(m._kind, self.side, tabs, m._d_ent, dr, e, e, r, want_w=True)[0]
        return q.view(b, n_rel, dr)


def _is_cand(x):
    if isinstance(x, EntityCandidates):
        return True
    if isinstance(x, (tuple, list)):
        return _is_cand(x[0])
    return torch.is_tensor(x) and x.dim() == 3


def _table_of(cand):
    """(N, d) table if ``cand`` is a batch-broadcast (stride-0) view of one
    matrix -- what inference_prepare_candidates returns -- else None."""
    if torch.is_tensor(cand) and cand.dim() == 3 and cand.stride(2) == 1 and \
            (cand.stride(0) == 0 or cand.shape[0] == 1):
        return cand[0]
    return None


# the DOT models' query side of a batch in one launch (kge_lp_dot_query_pipeline); KGE_DOT_FUSED=0: the separate kernels
DOT_FUSED = os.environ.get('KGE_DOT_FUSED', '1') == '1'
# TransH / TransD with the evaluator's second stream: which side of the preparation runs there (KGE_PREP_SIDE_SWAP=0: the candidate side; r06 same-box 0.565 -> 0.559 / 0.602 -> 0.597 ms with the query side there)
PREP_SIDE_SWAP = os.environ.get('KGE_PREP_SIDE_SWAP', '1') == '1'
# ... and their candidate table in one pass from the second evaluation on (kge_lp_dot_table_prep_fused); KGE_DOT_PREP_ONE_PASS=0: two
DOT_PREP_ONE_PASS = os.environ.get('KGE_DOT_PREP_ONE_PASS', '1') != '0'


class Model(Module):
    """Base interface (interfaces.py:13-174)."""

    _kind = None          # kge_hip.h model kind
    _ENT_TABLES = ()      # names of the entity-indexed nn.Embedding tables (row-sharded across GPUs)
    _ENT_POS = (0,)       # their positions in _tables()

    def __init__(self, n_entities, n_relations):
        super().__init__()
        self.n_ent = n_entities
        self.n_rel = n_relations
        self._cache = _SessionCache()
        self._row_shard = None      # (lo, hi): the entity tables hold only these rows (one shard per GPU)
        # evaluation-time guard scalars (device, float32 x 8): [0] max ||q||^2 (L2 norm guard),
        # [1] max ||e||^2 (segment 0), [2] split-list overflow flag, [3] max |X|, [4] max |y_c| (projection
        # modes), [5] max ||e||^2 (segment 1)
        self._lp_guard = None
        self._lp_guard_clean = False    # the guard vector is all zero (left so by the last evaluation's finalize launch)
        self._guard_on = False
        self._expand_ok = None      # TransE-L2 only: True/False forced by the evaluator, None: guarded
        self.split_filter = True    # fused rank counts via the f16-split prefilter + exact recheck
        self._split_ok = True       # cleared by the evaluator when the uncertain-pair list overflowed

    # ---- engine hooks (overridden by concrete models) ---------------------
    def _tables(self):
        raise NotImplementedError

    # ---- row-sharded entity tables (N/P rows per GPU, relation tables replicated) ----
    def shard_entities_(self, lo, hi):
        """Keep only rows [lo, hi) of every entity-indexed table (in place; ``n_ent`` stays the
        GLOBAL entity count).  From then on the model can only serve an entity-sharded
        LinkPredictionEvaluator (shard='entities'): query rows are built by the rank that owns
        the query's entity and summed over the ranks, each rank scores its own candidates."""
        assert self._row_shard is None, 'already sharded'
        assert 0 <= lo <= hi <= self.n_ent
        self._check_shardable(self.n_ent)
        for name in self._ENT_TABLES:
            emb = getattr(self, name)
            w = emb.weight.data[lo:hi].clone()
            emb.weight = torch.nn.Parameter(w, requires_grad=emb.weight.requires_grad)
            emb.num_embeddings = hi - lo
        self._row_shard = (lo, hi)
        return self

    def as_entity_shard_(self, n_total, lo, hi):
        """Declare a model that was CONSTRUCTED with n_entities = hi - lo to be the shard
        [lo, hi) of an n_total-entity model (tables too large to ever exist on one GPU)."""
        assert self._row_shard is None and self.n_ent == hi - lo and 0 <= lo <= hi <= n_total
        self._check_shardable(hi - lo)
        self.n_ent = n_total
        self._row_shard = (lo, hi)
        return self

    def _check_shardable(self, rows):
        """A model can only be marked row-sharded when it declares WHICH tables are entity-indexed: with an empty
        ``_ENT_TABLES`` (the base default, e.g. a user subclass with its own lp_problem) nothing would be sliced and
        every rank would score all N candidates as 'its shard' -- silently wrong summed counts."""
        if not self._ENT_TABLES:
            raise RuntimeError('torchkge_amd: %s declares no entity-indexed tables (_ENT_TABLES is empty); it cannot '
                               'be row-sharded' % type(self).__name__)
        for name in self._ENT_TABLES:
            n = getattr(self, name).weight.shape[0]
            if n != rows:
                raise RuntimeError('torchkge_amd: entity table %r has %d rows, expected %d' % (name, n, rows))

    def entity_table_bytes(self):
        return sum(getattr(self, n).weight.numel() * 4 for n in self._ENT_TABLES)

    def _check_unsharded(self, what):
        if self._row_shard is not None:
            raise RuntimeError('torchkge_amd: %s needs the whole entity tables; this model holds only rows '
                               '[%d, %d) (use LinkPredictionEvaluator(shard=\'entities\'))' % ((what,) + self._row_shard))

    def _cand_rows(self, table, lo, hi):
        """Rows [lo, hi) of an entity-indexed table as candidates."""
        if self._row_shard is not None:
            if (lo, hi) != self._row_shard:
                raise RuntimeError('torchkge_amd: candidate range [%d, %d) requested from a model that holds rows '
                                   '[%d, %d)' % ((lo, hi) + self._row_shard))
            return table
        return table if (lo == 0 and hi == table.shape[0]) else table[lo:hi]

    def lp_query_tables(self, qmap, exchange, gather=None):
        """Row-sharded model: compact replicas of the entity-table rows ``qmap['uniq']`` (the distinct entities
        the test queries mention, sorted) on EVERY rank, once per evaluate(); the batches then build their
        query rows locally from the replicas (``lp_problem(..., qtabs=...)`` with indices into ``uniq``).
        At FB15k-237 shape that moves ~12 k rows instead of the 2 x 20,466 query rows of every evaluation.

        ``gather`` (all-gather of equal-size row blocks) + ``qmap['sel']``: shards are contiguous id ranges and
        ``uniq`` is sorted, so the rows a rank owns are ONE slice of ``uniq`` -- every rank gathers its slice
        (padded to the longest), one all-gather per entity table, and a static index puts the blocks back
        in ``uniq`` order: (P - 1) / P of the replica's bytes received per rank.  Without it: every rank
        contributes the rows it owns to a zero matrix (kge_lp_prep_sharded as a gather) and ``exchange``
        sums them (x + 0 is exact) -- twice the bytes, any engine."""
        assert self._row_shard is not None
        lo, hi = self._row_shard
        tabs = [x.data for x in self._tables()]
        out = []
        if gather is not None and qmap.get('sel') is not None:
            for pos in self._ENT_POS:
                T = _hip.f32c(tabs[pos])
                block = torch.zeros(qmap['maxc'], T.shape[1], dtype=torch.float32, device=T.device)
                if qmap['mine_local'].shape[0]:
                    block[:qmap['mine_local'].shape[0]] = _hip.gather_rows(T, qmap['mine_local'])
                out.append(_hip.gather_rows(gather(block), qmap['sel']))
            return out
        ent_ids = qmap['uniq']
        zr = torch.zeros_like(ent_ids)
        for pos in self._ENT_POS:
            T = _hip.f32c(tabs[pos])
            d = T.shape[1]
            out.append(_hip.lp_prep(_hip.TRANSE_L2, _hip.SIDE_PROJ_H, [T, T], d, d, ent_ids, ent_ids, zr,
                                    ent_lo=lo, ent_n=hi - lo)[0])
        exchange(out)
        return out

    def lp_true_scores_replica(self, prob, qctx):
        """Row-sharded model, both-sides batch: the (2B) true scores computed by EVERY rank from the
        query-entity replicas ``qctx = (replica tables, h, t as indices into them)`` -- the true entity of a
        tail-side query is the fact's tail, of a head-side query its head, and both are rows of the
        replicas.  Same rows, same sequential chain (kge_lp_pair_scores) as the owner shard would run on
        its own table: identical bits, no collective.  None: this problem's mode needs per-candidate
        side tables (projection modes) -- the caller falls back to the owner's value summed over ranks."""
        A0, _, A1, _, qn, _, Wq, _, _, _ = prob.keep
        if Wq is not None:
            return None
        qt, hq, tq = qctx
        rep = [_hip.f32c(x) for x in qt]
        mode, K0 = int(prob.desc.mode), int(prob.desc.K0)
        true_q = torch.cat([tq, hq])
        if mode == _hip.LP_L2_EXPAND:
            en = self._cache.get('en_replica', rep, lambda: _hip.row_sqnorm(rep[0], K=K0))
            P = _hip.LpProblem(mode, A0, rep[0], qn=qn, en=en, K0=K0)
        elif mode == _hip.LP_DOT:
            P = _hip.LpProblem(mode, A0, rep[0], A1=A1, T1=rep[1] if A1 is not None else None, K0=K0)
        elif mode in (_hip.LP_L1_DIRECT, _hip.LP_L2_DIRECT):
            P = _hip.LpProblem(mode, A0, rep[0], K0=K0)
        else:
            return None
        return P.pair_scores(true_q)

    def _lp_prep(self, side, h_idx, t_idx, r_idx, exchange=None, qtabs=None, **want):
        """kge_lp_prep on this model's tables.  Row-sharded tables: rows of entities another rank owns
        come back as zeros and ``exchange`` (the evaluator's all-reduce SUM over the shards) completes
        the entity-derived outputs Q0 (/Q1); x + 0 is exact, so every rank ends up with the rows the
        owner computed.  ``qtabs`` (lp_query_tables): replicas of the query entities' rows are at hand --
        h_idx / t_idx then index THEM and no exchange is needed."""
        tabs = [x.data for x in self._tables()]
        if qtabs is not None:
            for pos, q in zip(self._ENT_POS, qtabs):
                tabs[pos] = q
            return _hip.lp_prep(self._hip_kind(), side, tabs, self._d_ent, self._d_rel, h_idx, t_idx, r_idx, **want)
        lo, n = (self._row_shard[0], self._row_shard[1] - self._row_shard[0]) if self._row_shard is not None else (0, -1)
        out = _hip.lp_prep(self._hip_kind(), side, tabs, self._d_ent, self._d_rel, h_idx, t_idx, r_idx,
                           ent_lo=lo, ent_n=n, **want)
        if self._row_shard is not None:
            if exchange is None:
                raise RuntimeError('torchkge_amd: a row-sharded model needs the evaluator\'s query exchange')
            exchange([x for x in (out[0], out[1]) if x is not None])
        return out

    L2_EXPAND_LIMIT = float('inf')      # TranslationModel: bound on ||q||^2 + ||e||^2 for the norm expansion

    def _uses_guard(self):
        """Does an evaluation of this model need the guard scalars?"""
        return bool(self.split_filter)

    def lp_guard_begin(self, device):
        """Start of an evaluation.  Returns the device guard vector (or None): the
        norm kernels fold max ||q||^2 / max ||e||^2 into it and the split prefilter
        raises its overflow flag there -- no extra launch, graph-capturable, no host
        sync; the evaluator reads it once with the ranks."""
        # (plain-Python state: object.__setattr__ skips nn.Module.__setattr__'s Parameter / Module / buffer checks, which
        # cost ~2.5 us per assignment -- five of them per evaluate() were 1.5 % of a cfg2 step)
        _set = object.__setattr__
        _set(self, '_expand_ok', None)
        if not self._uses_guard():
            return None
        # (whoever begins a guarded section may leave the vector dirty: the evaluator, which knows that its last finalize
        # launch zeroed it, restores the flag right after this call)
        _set(self, '_lp_guard_clean', False)
        if self._lp_guard is None or self._lp_guard.device != device:
            _set(self, '_lp_guard', torch.zeros(8, dtype=torch.float32, device=device))
            _set(self, '_lp_guard_clean', True)
        _set(self, '_guard_on', True)
        return self._lp_guard

    # One-product level of the split prefilter (kge_split_args.level): 'auto' = the evaluator decides from the number of
    # pairs the last evaluation re-scored (LinkPredictionEvaluator._level); 0 / 1 force a level.
    split_level = 'auto'
    _split_level = 0        # what the running evaluation uses (set by the evaluator)
    _lp_regions = False     # ... and whether the sweep's uncertain pairs go to regions of 32 queries (region recheck)
    _lp_r_both = None       # ... and, around one lp_problem('both') call, the batch's [r | r] vector (FilterPlan.r_both)

    # the one-product level on the FREE-RUNNING count kernel (lp_hi_stream.hip: fragment-major candidate table, resident
    # query panel, no block-wide barriers) wherever it handles the GEMM's width; it sweeps per query (no query columns)
    lp_hi_stream = True

    def _lp_width(self):
        """Columns of the all-candidates GEMM (ComplEx: 2 d)."""
        return self._d_rel

    def _level1_stream(self):
        return bool(self.lp_hi_stream) and _hip.hi_stream_ok(self._lp_width())

    # ... which CAN sweep query COLUMNS (one matrix sweep per distinct query row, the members' thresholds compared in the
    # epilogue: lp_hi_stream_kernel<.., GS = 4>, r06) for the plain-threshold counts -- TransE-L2, DistMult, ComplEx -- on rows
    # that fit its resident panel.  Off by default: measured slower than the per-query sweep at cfg2 (TransEModel.lp_dedupe_level1)
    lp_stream_columns = os.environ.get('KGE_STREAM_COLUMNS', '0') == '1'

    def _level1_stream_columns(self):
        return bool(self.lp_stream_columns) and (self._lp_width() + 2 + 15) // 16 <= 32

    def _use_level1(self):
        lv = self.split_level
        want = self._split_level == 1 if lv == 'auto' else int(lv) == 1
        return bool(want) and _hip.split_accum_model() == 1     # (the band assumes the measured MFMA accumulation)

    def lp_guard_end(self):
        _set = object.__setattr__
        _set(self, '_guard_on', False)
        _set(self, '_expand_ok', None)
        _set(self, '_split_ok', True)

    def _attach_dot_split(self, prob, T0, T1=None, c_base=0):
        """Rank counts of a KGE_LP_DOT problem through the certified f16-split
        prefilter (only inside an evaluation, where the guard vector exists)."""
        if not (self._guard_on and self.split_filter and self._split_ok):
            return prob
        g = self._lp_guard
        key = '%d_%d' % (c_base, T0.shape[0])

        def norms():
            en0 = _hip.row_sqnorm(T0, max_io=g[1:2], bound_only=True)      # (only the maxima are used: the operands' scale)
            if T1 is not None:
                _hip.row_sqnorm(T1, max_io=g[5:6], bound_only=True)
            del en0
            return True
        srcs = [T0] + ([T1] if T1 is not None else [])
        self._cache.get('esn_' + key, srcs, norms)       # the norm maxima fix the operands' scale: before either table
        nm1 = g[5:6] if T1 is not None else None
        if self._use_level1():
            # one-product level (see TransEModel._fused_query_problem): planar hi table + its residual maximum
            frag = self._level1_stream()
            Eh, de2 = self._cache.get('ehd%d_' % frag + key, srcs,
                                      lambda: _hip.hi_table(T0, X1=T1, dot=True, nmax0=g[1:2], nmax1=nm1, frag=frag))
            prob.split = {'Es': Eh, 'e2pref': None, 'enmax': g[1:2], 'enmax1': nm1, 'overflow': g[2:3], 'level': 1,
                          'de2max': de2, 'list_stat': g[6:7], 'es_frag': frag}
            return prob
        Es, e2 = self._cache.get('esd_' + key, srcs, lambda: _hip.split_table(T0, X1=T1, dot=True, nmax0=g[1:2], nmax1=nm1))
        prob.split = {'Es': Es, 'e2pref': e2, 'enmax': g[1:2], 'enmax1': nm1, 'overflow': g[2:3], 'list_stat': g[6:7]}
        return prob

    def _dot_fused_problem(self, sd, h_idx, t_idx, r_idx, ent, rel):
        """DistMult / ComplEx inside evaluate(), one-product level, whole tables on this GPU: the query side of the batch
        from ONE launch (kge_lp_dot_query_pipeline: q, the exact true scores, the planar hi operand with per-query scales,
        its residuals, the thresholds, zeroed rank counters) instead of nine; None when that path does not apply.
        ``ent`` / ``rel``: [E] / [R] (DistMult) or [Re, Im] / [Re_r, Im_r] (ComplEx)."""
        if not (DOT_FUSED and self._guard_on and self.split_filter and self._split_ok and self._use_level1()):
            return None
        d = ent[0].shape[1]
        if h_idx.shape[0] == 0 or d % 8 != 0 or sd not in (_hip.SIDE_TAIL, _hip.SIDE_HEAD, _hip.SIDE_BOTH):
            return None
        T0, T1 = ent[0], (ent[1] if len(ent) > 1 else None)
        g = self._lp_guard
        frag = self._level1_stream()
        nm1 = g[5:6] if T1 is not None else None
        # candidate side (cached per evaluation) in two launches: norm maxima per block, then the hi table whose blocks fold
        # them into guard[1] / guard[5] and leave their residual maxima per block -- folded into guard[7] by every query
        # pipeline launch on its way in (kge_lp_dot_table_prep; the guard vector is zeroed per evaluation)
        srcs = [T0] + ([T1] if T1 is not None else [])
        # r06: from the second evaluation on ONE launch and one pass -- the scale of the maxima the previous evaluation's
        # query pipeline left in `prev`; the pipeline folds this pass's maxima, and a table that has outgrown its scale
        # raises the overflow flag (that evaluation is redone on three products; the next one finds the new maxima)
        prev = self.__dict__.get('_lp_dot_prev')
        if prev is None or prev[0].device != T0.device:
            prev = [torch.zeros(2, dtype=torch.float32, device=T0.device), None]
            object.__setattr__(self, '_lp_dot_prev', prev)
        sig = (T0.data_ptr(), None if T1 is None else T1.data_ptr(), tuple(T0.shape))
        one_pass = bool(DOT_PREP_ONE_PASS and frag and prev[1] == sig and _hip.dot_table_prep_fusable(T0, T1))
        ckey = 'dtp%d%d_0_%d' % (frag, one_pass, T0.shape[0])
        Eh, dnb, ws = self._cache.get(ckey, srcs, lambda: _hip.dot_table_prep(T0, T1, g[1:2], nm1, frag,
                                                                              prev_nmax=prev[0] if one_pass else None))
        prev[1] = sig       # (the pipeline below stores this evaluation's maxima there)
        sp = {'Es': Eh, 'e2pref': None, 'enmax': g[1:2], 'enmax1': nm1, 'overflow': g[2:3], 'level': 1, 'de2max': g[7:8],
              'list_stat': g[6:7], 'es_frag': frag}
        pre = _hip.lp_dot_query_pipeline(sd, T0, T1, rel[0], rel[1] if len(rel) > 1 else None, h_idx, t_idx, r_idx,
                                         sp['enmax'], nm1, sp['de2max'], g[0:1], sp['overflow'], zero_counts=True,
                                         dn_bmax=dnb, regions=bool(frag) and bool(getattr(self, '_lp_regions', False)),
                                         nm_bmax=ws if one_pass else None, prev_nmax=prev[0])
        pre['true_idx'] = t_idx if sd == _hip.SIDE_TAIL else (h_idx if sd == _hip.SIDE_HEAD else None)
        prob = _hip.LpProblem(_hip.LP_DOT, pre['Q'], T0, A1=pre['Q1'], T1=T1)
        prob.split = sp
        prob.pre = pre
        return prob

    def lp_problem_both(self, h_idx, t_idx, r_idx):
        """Both sides of a batch as ONE problem of 2B queries (tail-side queries first)
        against the entity table: ``lp_problem(..., side='both')``.  Ranks are per
        query, so they do not depend on the batch composition; what changes is that
        every latency-bound short kernel of a batch runs once."""
        return self.lp_problem(h_idx, t_idx, r_idx, 'both')

    def lp_session(self):
        """Context inside which per-entity precomputes are cached (tables must
        not change while it is open)."""
        return self._cache

    @property
    def _d_ent(self):
        return self._tables()[0].shape[1]

    @property
    def _d_rel(self):
        return self._tables()[1].shape[1]

    def _hip_kind(self):
        return self._kind

    # ---- reference API ----------------------------------------------------
    def forward(self, heads, tails, relations, negative_heads, negative_tails,
                negative_relations=None):
        """(pos, neg) scores; several negatives per fact tile the positives
        (interfaces.py:39-82)."""
        if negative_relations is None:
            negative_relations = relations
        n_neg = 1
        if negative_heads.shape[0] > negative_relations.shape[0]:
            n_neg = int(negative_heads.shape[0] / negative_relations.shape[0])
            negative_relations = negative_relations.repeat(n_neg)
        # positives and negatives in ONE fused kernel launch (and one backward): per-triple
        # scores are independent, so this is the reference's two calls value for value
        b = heads.shape[0]
        scores = self.scoring_function(torch.cat([heads, negative_heads]), torch.cat([tails, negative_tails]),
                                       torch.cat([relations, negative_relations]))
        pos, neg = scores[:b], scores[b:]
        if n_neg > 1:
            pos = pos.repeat(n_neg)
        return pos, neg

    def scoring_function(self, h_idx, t_idx, r_idx):
        """Score of each triplet, one fused HIP kernel (K1), differentiable."""
        self._check_unsharded('scoring_function')
        tables = self._tables()
        _hip.require_cuda(h_idx, t_idx, r_idx, *tables)
        return _ScoreTriples.apply(self._hip_kind(), self._d_ent, self._d_rel, h_idx, t_idx,
                                   r_idx, *tables)

    def normalize_parameters(self):
        raise NotImplementedError

    def get_embeddings(self):
        raise NotImplementedError

    def inference_scoring_function(self, h, t, r):
        raise NotImplementedError

    def inference_prepare_candidates(self, h_idx, t_idx, r_idx, entities=True):
        raise NotImplementedError

    # pre-0.17 names used by BASELINE.json's north_star (docs/history.rst:153-157)
    def lp_scoring_function(self, h, t, r):
        return self.inference_scoring_function(h, t, r)

    def lp_prep_cands(self, h_idx, t_idx, r_idx, entities=True):
        return self.inference_prepare_candidates(h_idx, t_idx, r_idx, entities=entities)

    def lp_problem(self, h_idx, t_idx, r_idx, side, ent_lo=0, ent_hi=None, exchange=None, qtabs=None):
        """kge_lp_desc owner for scoring every entity in [ent_lo, ent_hi) as the
        tail (side='tail') or head (side='head') of each (h, r, t).  ``exchange`` / ``qtabs``: see _lp_prep."""
        raise NotImplementedError

    @staticmethod
    def _normalize_weight_(emb):
        w = emb.weight.data
        _hip.require_cuda(w)
        if not w.is_contiguous():
            w = w.contiguous()
            emb.weight.data = w
        _hip.normalize_rows_(w)


class TranslationModel(Model):
    """Translational models: score = -dissimilarity(p(h) + r, p(t))
    (interfaces.py:177-272)."""

    def __init__(self, n_entities, n_relations, dissimilarity_type):
        super().__init__(n_entities, n_relations)
        assert dissimilarity_type in ['L1', 'L2', 'torus_L1', 'torus_L2', 'torus_eL2']
        if dissimilarity_type not in ('L1', 'L2'):
            raise NotYetImplementedError('torus dissimilarities (TorusE) are outside the MI355X '
                                         'hot path (SURVEY.md section 2, row 5).')
        self.dissimilarity_type = dissimilarity_type
        from ..utils.dissimilarities import l1_dissimilarity, l2_dissimilarity
        self.dissimilarity = l1_dissimilarity if dissimilarity_type == 'L1' else l2_dissimilarity
        # 'expand': ||q-e||^2 = ||q||^2 + ||e||^2 - 2 q.e as an fp32 MFMA GEMM;
        # 'direct': broadcast-subtract on the VALU; 'auto': expand while the operands
        # are small enough for the cancellation error to stay inside the 1e-5 score
        # tolerance (||q||^2 + ||e||^2 <= L2_EXPAND_LIMIT), direct otherwise.
        self.l2_mode = 'auto'
        self._sad_bounds = None     # TransE-L1: callable giving the device scalars (max |E|, max |R|) of an evaluation

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # reference TransH/TransD state_dicts carry the (n_rel, n_ent, d)
        # `projected_entities` cache (translation.py:178-181); the engine has none.
        state_dict.pop(prefix + 'projected_entities', None)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _direct_mode(self):
        return _hip.LP_L1_DIRECT if self.dissimilarity_type == 'L1' else _hip.LP_L2_DIRECT

    # measured: |expand - reference| ~ 2.4e-7 * (||q||^2 + ||e||^2) at d = 200 (1.2e-6 at 5)
    L2_EXPAND_LIMIT = 16.0

    def _uses_guard(self):
        # the L2 models take the norm expansion (TransE: + the split prefilter) optimistically; TransE-L1's integer
        # prefilter needs the overflow flag of the guard vector
        return (self.dissimilarity_type == 'L2' and self.l2_mode == 'auto') or \
            (self.dissimilarity_type == 'L1' and bool(self.split_filter))

    def _proj_problem(self, q, table, Wq, r_idx, c_base, K0, qn, en):
        """TransH / TransD: the expansion around u.e with the per-pair projection term
        (KGE_LP_L2_PROJH / _PROJD); None = this model has no such form."""
        return None

    def _attach_proj_split(self, prob, table, en, X, yc, K0):
        """Rank counts of a projection-mode problem through the f16-split prefilter
        (inside an evaluation only): split table as for TransE-L2, plus device-side
        bounds on |X| (and |y_c|) for the error band."""
        if not (self._guard_on and self._expand_ok is None and self.l2_mode == 'auto' and self.split_filter
                and self._split_ok):
            return prob
        g = self._lp_guard
        key = '%d_%d' % (prob.desc.c_base, table.shape[0])
        Kq = table.shape[1] if K0 is None else K0
        self._cache.get('xmax_' + key, [X], lambda: _hip.absmax(X, g[3:4]))
        if yc is not None:
            self._cache.get('ymax_' + key, [yc], lambda: _hip.absmax(yc, g[4:5]))
        prob.split = {'enmax': g[1:2], 'overflow': g[2:3], 'xabsmax': g[3:4], 'yabsmax': g[4:5] if yc is not None else None,
                      'list_stat': g[6:7]}
        if self._use_level1():      # one-product level (see TransEModel._fused_query_problem)
            frag = self._level1_stream()
            Eh, de2 = self._cache.get('eh%d_' % frag + key, [table], lambda: _hip.hi_table(table, K=Kq, aug=en, frag=frag))
            prob.split.update({'Es': Eh, 'e2pref': None, 'level': 1, 'de2max': de2, 'es_frag': frag})
        else:
            Es, e2 = self._cache.get('es_' + key, [table], lambda: _hip.split_table(table, K=Kq, aug=en))
            prob.split.update({'Es': Es, 'e2pref': e2})
        return prob

    def _proj_fast_problem(self, sd, h_idx, t_idx, r_idx, r_both, ent_lo, ent_hi, exchange, qtabs, spec):
        """TransH / TransD inside evaluate() (guarded optimistic norm expansion + split prefilter), r05: the query side in
        TWO launches -- kge_lp_prep (the query rows; the gathered W rows are not materialised) and kge_proj_query_stats
        (||q||^2, p, z by the chains of the separate kernels: same bits) -- and, on the one-product level, the candidate
        side in ONE (kge_lp_table_prep_l2: norms + fragment-major hi table + residual maximum, its two maxima folded by
        the threshold kernel).  The measured max |X| (a kernel + a fill per evaluation) gives way to the per-query bound
        ||w_i|| max||e|| inside the threshold kernel.  ``spec`` = (mode, W table, scale, z_add, K0, tables builder).
        None: shapes / alignment need the general path."""
        mode, Wt, scale, z_add, K0, build_side = spec
        g = self._lp_guard
        tabs = [x.data for x in self._tables()]
        table = self._cand_rows(_hip.f32c(tabs[0]), ent_lo, ent_hi)
        Kq = table.shape[1] if K0 is None else K0
        if Kq % 4 or table.stride(0) % 4 or table.data_ptr() % 16 or table.shape[0] == 0:
            return None
        lvl1 = self._use_level1()
        frag = lvl1 and self._level1_stream()
        key = '%d_%d' % (ent_lo, table.shape[0])

        def cand_side():
            """norms + fragment-major hi table + residual maximum, X = W.E^T (TransD: G, sigma): per evaluation, cached"""
            prep_ = None
            if frag:
                prep_ = self._cache.get('tp_' + key, [table],
                                        lambda: _hip.table_prep_l2(table, g[1:2], g[7:8], deferred_max=True, K=K0))
            if prep_ is not None:
                en_ = self._cache.get('en_' + key, [table], lambda: prep_[0])
            else:
                en_ = self._cache.get('en_' + key, [table], lambda: _hip.row_sqnorm(table, K=K0, max_io=g[1:2]))
            XT_, yc_ = build_side(table, ent_lo, K0)
            if yc_ is not None:
                self._cache.get('ymax_' + key, [yc_], lambda: _hip.absmax(yc_, g[4:5]))
            return prep_, en_, XT_, yc_

        # (r06) the candidate side -- three to five launches that read tables only -- on the evaluator's second stream
        # beside the query side's two (LinkPredictionEvaluator hands the stream over for the session: fork / join by
        # events, two parallel branches of the captured graph).  They write different guard scalars.
        side = getattr(self, '_lp_side_stream', None) if table.is_cuda else None
        # (one-product level on unsharded tables / replicas: the query rows' planar hi operand rides the same launch)
        hi_too = frag and (self._row_shard is None or qtabs is not None)

        def query_side():
            out_ = self._lp_prep(sd, h_idx, t_idx, r_idx, exchange, qtabs=qtabs, **({'want_hi': True} if hi_too else {}))
            # (r06: the launch also zeroes the batch's (3, 2B) rank counters -- the evaluator's partial_counts takes them)
            zc_ = torch.empty(3, out_[0].shape[0], dtype=torch.int32, device=out_[0].device) if out_[0].is_cuda else None
            return out_, zc_, _hip.proj_query_stats(out_[0], Wt, r_both, scale, z_add, qmax_io=g[0:1], zero=zc_)

        def keep(xs):       # allocated on the side stream, consumed on the main one
            for x in xs:
                if torch.is_tensor(x):
                    x.record_stream(main)
                elif isinstance(x, (tuple, list)):
                    keep(x)
        if side is not None:
            main = torch.cuda.current_stream(table.device)
            side.wait_stream(main)
            if PREP_SIDE_SWAP:
                # the QUERY side on the second stream: what follows the join -- true scores, thresholds, sweep -- then stays on
                # the queue of the candidate side's branch (the graph executor continues a joined chain there: the other
                # way round the critical chain changed queues twice, ~11 us per change)
                with torch.cuda.stream(side):
                    out, zc, st = query_side()
                cand = cand_side()
                main.wait_stream(side)
                keep([out, zc, st])
            else:
                with torch.cuda.stream(side):
                    cand = cand_side()
                out, zc, st = query_side()
                main.wait_stream(side)
                keep(cand)
        else:
            out, zc, st = query_side()
        Q0 = out[0]
        if st is None:
            return None
        qn, pz = st
        prep, en, XT, yc = cand if side is not None else cand_side()
        prob = _hip.LpProblem(mode, Q0, table, qn=qn, en=en, Wq=pz, scal=XT, r_idx=r_both, yc=yc, c_base=ent_lo, K0=K0)
        split = {'enmax': g[1:2], 'overflow': g[2:3], 'xabsmax': None, 'yabsmax': None, 'list_stat': g[6:7]}
        if yc is not None:
            split['yabsmax'] = g[4:5]
        if lvl1:
            if prep is not None:
                split.update({'Es': prep[1], 'e2pref': None, 'level': 1, 'de2max': g[7:8], 'es_frag': True, 'tp_bmax': prep[2]})
            else:
                Eh, de2 = self._cache.get('eh%d_' % frag + key, [table], lambda: _hip.hi_table(table, K=Kq, aug=en, frag=frag))
                split.update({'Es': Eh, 'e2pref': None, 'level': 1, 'de2max': de2, 'es_frag': frag})
        else:
            Es, e2 = self._cache.get('es_' + key, [table], lambda: _hip.split_table(table, K=Kq, aug=en))
            split.update({'Es': Es, 'e2pref': e2})
        prob.split = split
        if hi_too:
            prob.pre_q = (out[4], out[5])
        prob.zero_counts = zc
        return prob

    def _proj_fast_ok(self):
        return bool(self.dissimilarity_type == 'L2' and self.l2_mode == 'auto' and self._guard_on and self._expand_ok is None
                    and self.split_filter and self._split_ok and getattr(self, 'lp_fast_proj', True))

    def _translational_problem(self, q, table, Wq=None, scal=None, r_idx=None, c_base=0, K0=None):
        """Problem for s[i,c] = -diss(q_i, table[c] (- a w_i))."""
        if self.dissimilarity_type == 'L2' and self.l2_mode in ('expand', 'auto') and \
                (Wq is None or (self._kind is not None and r_idx is not None)):
            # inside evaluate() the expansion is optimistic: the two norm kernels also
            # fold their maxima into the guard scalars, which are checked once at the end
            guarded = self.l2_mode == 'auto' and self._expand_ok is None and self._guard_on
            gq, ge = (self._lp_guard[0:1], self._lp_guard[1:2]) if guarded else (None, None)
            en = self._cache.get('en_%d_%d' % (c_base, table.shape[0]), [table],
                                 lambda: _hip.row_sqnorm(table, K=K0, max_io=ge))
            qn = _hip.row_sqnorm(q, max_io=gq)
            ok = True
            if self.l2_mode == 'auto' and not guarded:
                ok = self._expand_ok
                if ok is None:      # drop-in API call: decide now on the actual operands (one sync)
                    ok = q.shape[0] == 0 or float((qn.max() + en.max()).item()) <= self.L2_EXPAND_LIMIT
            if ok and Wq is not None:
                return self._proj_problem(q, table, Wq, r_idx, c_base, K0, qn, en)
            if ok:
                prob = _hip.LpProblem(_hip.LP_L2_EXPAND, q, table, qn=qn, en=en, c_base=c_base, K0=K0)
                if guarded and self.split_filter and self._split_ok:
                    # rank counts through the certified f16-split prefilter (16x MFMA rate)
                    Kq = q.shape[1] if K0 is None else K0
                    Es, e2 = self._cache.get('es_%d_%d' % (c_base, table.shape[0]), [table],
                                             lambda: _hip.split_table(table, K=Kq, aug=en))
                    prob.split = {'Es': Es, 'e2pref': e2, 'enmax': ge, 'overflow': self._lp_guard[2:3]}
                return prob
        if callable(scal):          # built only when the broadcast-subtract kernel is really taken
            scal = scal()
        prob = _hip.LpProblem(self._direct_mode(), q, table, Wq=Wq, scal=scal, r_idx=r_idx,
                              c_base=c_base, K0=K0)
        if (self.dissimilarity_type == 'L1' and Wq is None and self._guard_on and self.split_filter and self._split_ok
                and self._sad_bounds is not None and K0 is None and self._row_shard is None
                and c_base == 0 and table.shape[0] == self.n_ent):     # (whole table: the bounds are taken over it)
            # TransE-L1 inside an evaluation: rank counts through the certified 16-bit SAD prefilter (lp_l1_sad.hip).
            # Every query element is e +- r, so max|e| + max|r| (device scalars in the guard vector) bounds both operands.
            g = self._lp_guard
            emax, rmax = self._sad_bounds()
            Ei = self._cache.get('sad_%d_%d' % (c_base, table.shape[0]), [table], lambda: _hip.sad_rows(table, emax, rmax))
            prob.sad = {'Ei': Ei, 'emax': emax, 'rmax': rmax, 'overflow': g[2:3]}
        return prob

    def inference_scoring_function(self, proj_h, proj_t, r):
        """-dissimilarity(proj_h + r, proj_t) against every candidate; the 3-D
        argument (or EntityCandidates handle) is the candidate set
        (interfaces.py:240-272)."""
        if torch.is_tensor(r) and r.dim() == 3:
            if torch.is_tensor(proj_h) and proj_h.dim() == 2 and torch.is_tensor(proj_t) and proj_t.dim() == 2:
                # relation prediction without projections: -diss(h + r_c, t) = -diss(r_c, t - h)
                return self._score_against(_hip.ewise(_hip.EW_SUB, proj_t, proj_h), r)
            # relation-specific projections (interfaces.py:261-272): -diss(proj_h + r, proj_t) over (b, n_rel, d)
            if isinstance(proj_h, RelationProjections) and isinstance(proj_t, RelationProjections):
                fused = self._relation_scores_proj(proj_h, proj_t, r)
                if fused is not None:
                    return fused
            ph = proj_h.materialize() if isinstance(proj_h, RelationProjections) else proj_h
            pt = proj_t.materialize() if isinstance(proj_t, RelationProjections) else proj_t
            ph, pt = ph.view(ph.shape[0], -1, r.shape[2]), pt.view(pt.shape[0], -1, r.shape[2])
            ph, rr = torch.broadcast_tensors(ph, r)
            return - self.dissimilarity(_hip.ewise(_hip.EW_ADD, ph.contiguous(), rr.contiguous()), pt)
        assert r.dim() == 2
        if _is_cand(proj_t):
            assert torch.is_tensor(proj_h) and proj_h.dim() == 2   # tail completion
            return self._score_against(_hip.ewise(_hip.EW_ADD, proj_h, r), proj_t)
        assert _is_cand(proj_h) and proj_t.dim() == 2              # head completion
        return self._score_against(_hip.ewise(_hip.EW_SUB, proj_t, r), proj_h)

    def _score_against(self, q, cand):
        """s[i,c] = -diss(q_i, cand[i,c])."""
        if isinstance(cand, EntityCandidates):
            return self._handle_problem(q, cand).scores()
        table = _table_of(cand)
        if table is not None:
            return self._translational_problem(q, _hip.f32c(table)).scores()
        return _hip.lp_scores_batched(self._direct_mode(), q, cand)

    def _handle_problem(self, q, cand):
        raise NotImplementedError

    def _relation_scores_proj(self, proj_h, proj_t, r):
        """Fused relation-candidate scores when ``r`` is the (stride-0 expanded) relation table
        itself -- what inference_prepare_candidates(entities=False) returns; else None."""
        return None


class BilinearModel(Model):
    """Bilinear models (interfaces.py:275-330)."""

    # lp_problem(side='both', cols=ColumnPlan): the split count sweeps one column per distinct query row of the batch
    lp_dedupe_queries = True
    # the count sweep is enqueued in front of the second stream's filter correction (r06, same box: DistMult / FB15k, whose
    # filter lists hold 3.8 M entries, 2.063 -> 2.000 ms per evaluate; ComplEx / WN18RR +-0; TransH / TransD keep the filter
    # first: 0.594 against 0.617)
    lp_count_first = True

    def __init__(self, emb_dim, n_entities, n_relations):
        super().__init__(n_entities, n_relations)
        self.emb_dim = emb_dim
