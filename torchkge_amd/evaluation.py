# -*- coding: utf-8 -*-
"""LinkPredictionEvaluator with the reference's interface
(torchkge/evaluation.py:207-425): ``LinkPredictionEvaluator(model, kg)``,
``.evaluate(b_size, verbose)``, ``.mean_rank() / .hit_at_k(k) / .hit_at_k_heads
/ .hit_at_k_tails / .mrr() / .print_results()``, the four int64 rank vectors and
``NotYetEvaluatedError`` before ``evaluate``.

What differs is how a batch is ranked.  The reference materialises
scores (b, N) [through (b, N, d) temporaries], clones them, filters row by row
in Python and counts twice (evaluation.py:290-300).  Here, per side:

    problem  = model.lp_problem(h, t, r, side)          # query prep kernel
    s_true   = problem.pair_scores(true_idx)            # B scores
    raw      = problem.count_ge(s_true)                 # tiled scorer, counts only
    sub,fnd  = problem.filter_sub(...)                  # scores only the filter lists
    rank, filt_rank = rank_finalize(raw, sub, fnd)

so the (B, N) matrix never reaches HBM; every kernel scores a pair with the
same fp32 arithmetic, so the ranks equal get_rank / filter_scores applied to
the materialised matrix bit for bit (``fused=False`` runs exactly that).

Multi-GPU (one process per GPU, torch.distributed / RCCL): ``shard='entities'``
row-shards the candidate range; partial results are exchanged either as the
score tiles themselves (``exchange='scores'``: all-gather, the collective the
north star names) or as rank counts (``exchange='counts'``: one int32
all-reduce of 3*B values, bit-identical ranks).  ``shard='queries'`` splits the
facts instead (no data-path collective, ranks all-gathered once at the end).
"""
import ctypes
import os
import struct
import weakref

import torch
from tqdm.autonotebook import tqdm

from . import _hip
from . import distributed as kdist
from .exceptions import NotYetEvaluatedError
from .filter_index import ColumnPlan, filter_index_for, KEY2_SPAN, FilterPlan
from .utils.data import get_n_batches
from .utils.modeling import filter_scores
from .utils.operations import get_rank


# LinkPredictionEvaluator._internal_batch: facts per batch of the fused path / bound on a batch's uncertain-pair list
# (r05: 32768 -> 65536 -- FB15k's 59,071 test facts as ONE batch of 118 k queries instead of two: DistMult 2.21 -> 2.14 ms per
# evaluate, profiles/r05/dot_fused_ab.txt; the per-batch costs -- graph launch, read-back, the latency chains of the query side --
# are paid once)
COALESCE_BATCH = int(os.environ.get('KGE_COALESCE_BATCH', 65536))
COALESCE_LIST_BYTES = 2 << 30
# exchange='scores': bytes of the local (rows, N/P) fp32 score tile of one all-to-all (the receive buffer has the same
# size).  A batch is cut into as many row tiles as it takes, so b_size never decides whether the score exchange fits.
SCORE_TILE_BYTES = int(os.environ.get('KGE_SCORE_TILE_BYTES', 256 << 20))
# the count sweep enqueued IN FRONT of the filter correction of the second stream (its persistent workgroups then get every CU
# from the start and the short filter kernels fill the leftover slots): per model (Model.lp_count_first; measured r05, same box,
# alternating: TransE 0.502 -> 0.486 ms, DistMult 2.310 -> 2.330, TransH +-1 %: profiles/r05/count_first_ab.txt); the
# environment switch forces it on or off everywhere
COUNT_FIRST = {'0': False, '1': True}.get(os.environ.get('KGE_COUNT_FIRST', ''), None)
DEDUPE_QUERIES = os.environ.get('KGE_DEDUPE_QUERIES', '1') != '0'    # count kernel on distinct query rows (ColumnPlan)
# ... on the one-product level too (measured r04, same box, ms per evaluate with / without columns on level 1: TransE 0.648 /
# 0.632, ComplEx 0.479 / 0.479, DistMult 2.642 / 2.667, TransH 0.813 / 0.838 -- profiles/r04/dedupe_level1_ab.txt): yes,
# except where the model opts out (Model.lp_dedupe_level1; TransE does).  KGE_DEDUPE_LEVEL1=0 switches it off everywhere.
DEDUPE_LEVEL1 = os.environ.get('KGE_DEDUPE_LEVEL1', '1') != '0'
# One-product level of the split prefilter (model.split_level = 'auto'): an evaluation whose three-product sweep re-scored
# at most LEVEL1_ENTER pairs per query hands the NEXT one to the one-product sweep (a third of the matrix work, ~4x the
# re-scored pairs); one whose one-product sweep re-scored more than LEVEL1_LEAVE per query hands it back.  Break-even on
# cfg2: ~24 extra pairs per query (0.25 ms of count kernel against ~4 G exact pair scores / s).
LEVEL1_ENTER, LEVEL1_LEAVE = 4.0, 30.0
# (measured: the one-product sweep re-scores ~5.8x the pairs of the three-product one -- entering at <= 4 predicts <= 23 --;
#  an evaluator that had to LEAVE level 1 does not try again until the three-product count has halved: no flip-flopping
#  between two captured graphs on a model that sits at the boundary)
# Both numbers are cfg2's (N = 14,541 candidates).  What level 1 saves per query is two thirds of a three-product sweep over
# the N candidates, what it costs is one exact chain per extra pair -- both proportional to the row width, so the break-even
# number of extra pairs per query scales with N alone: 24 at N = 14,541, i.e. N / 600.  r06: at cfg5's shape (N = 4.59 M,
# K = 1024) the sweep goes 270 -> 84 ms for 10,266 queries (profiles/r06/hi_chunk_first_timing.txt), 18 us per query or
# thousands of exact pair scores: a fitted model there re-scores hundreds of pairs per query on three products and still
# belongs on level 1.
LEVEL1_CANDIDATES_REF = 14541.0


def level1_thresholds(n_cand):
    """(enter, leave): re-scored pairs per query on the three-product level up to which the next evaluation runs on the
    one-product level / on the one-product level above which it goes back, for `n_cand` candidates per query."""
    scale = max(1.0, float(n_cand) / LEVEL1_CANDIDATES_REF)
    return LEVEL1_ENTER * scale, LEVEL1_LEAVE * scale


class HipRankEngine(object):
    """Per-batch, per-side ranking on the HIP library (the product path)."""

    name = 'hip'

    @staticmethod
    def check_device(device):
        if device.type != 'cuda':
            raise RuntimeError('torchkge_amd.LinkPredictionEvaluator runs on MI355X (HIP) only: '
                               'move the model to `cuda` (there is no CPU fallback).')

    @staticmethod
    def lookup(index, key1, key2):
        return index.lookup(key1, key2)

    _targets_cat = None

    def lookup_both(self, index_t, index_h, h, t, r):
        """Both sides of a batch as 2B queries (tail side first): filter segments into the
        concatenated target arrays of the two indices, the 2B true ids, and that array."""
        key = (index_t.targets.data_ptr(), index_h.targets.data_ptr(), index_t.targets.shape[0],
               index_h.targets.shape[0])
        if self._targets_cat is None or self._targets_cat[0] != key:
            self._targets_cat = (key, torch.cat([index_t.targets, index_h.targets]))
        seg_lo, seg_hi, true_idx = _hip.filter_lookup_both(index_t.keys, index_t.offsets, index_h.keys,
                                                           index_h.offsets, index_t.targets.shape[0], h, t, r,
                                                           KEY2_SPAN)
        return seg_lo, seg_hi, true_idx, self._targets_cat[1]

    @staticmethod
    def problem(model, h, t, r, side, lo, hi, exchange=None, qctx=None, cols=None):
        """ROW-SHARDED model: `qctx` = (replicas of the query entities' rows, h and t as indices into them),
        exchanged once per evaluate(); or `exchange`: completes the query rows of this batch (sum over the shards)."""
        kw = {'cols': cols} if cols is not None else {}
        if getattr(model, '_row_shard', None) is not None:
            if qctx is not None:
                return model.lp_problem(qctx[1], qctx[2], r, side, ent_lo=lo, ent_hi=hi, qtabs=qctx[0], **kw)
            if exchange is not None:
                return model.lp_problem(h, t, r, side, ent_lo=lo, ent_hi=hi, exchange=exchange, **kw)
        return model.lp_problem(h, t, r, side, ent_lo=lo, ent_hi=hi, **kw)

    @staticmethod
    def true_scores(prob, true_idx):
        if prob.pre is not None and prob.pre.get('true_idx') is None:
            prob.pre['true_idx'] = true_idx     # both-sides batch: the fused query pipeline scored exactly these pairs
        st = prob.pair_scores(true_idx)
        if hasattr(prob, 'split_true'):
            prob.split_true = (st, true_idx)    # (the count sweep need not list the pairs whose score IS the threshold)
        return st

    uses_plans = True       # per-batch FilterPlan (segments + grouping), built once per evaluator

    def plan_both(self, index_t, index_h, h, t, r, model=None):
        """FilterPlan of one both-sides batch: the filter lookup and the grouping of its 2B queries; for models whose
        count kernel takes COLUMNS (distinct query rows, Model.lp_dedupe_queries) also the batch's ColumnPlan."""
        seg_lo, seg_hi, true_idx, targets = self.lookup_both(index_t, index_h, h, t, r)
        plan = FilterPlan(seg_lo, seg_hi, true_idx, targets)
        plan.cols = None
        plan.r_both = torch.cat([r, r]).contiguous() if (r.is_cuda and r.dtype == torch.int64) else None
        if model is not None and getattr(model, 'lp_dedupe_queries', False) and h.shape[0] > 0 and DEDUPE_QUERIES:
            plan.cols = ColumnPlan(h, t, r, model.n_ent, model.n_rel, _hip.split_group_sets(), _hip.split_query_rows_padded,
                                   relation_major=(model.lp_dedupe_queries == 'relation-major'))
        return plan

    flag_columns = True     # partial_counts(pad=k) appends k spare int32 columns (the guard flags ride the counts exchange)

    @staticmethod
    def partial_counts(prob, s_true, true_idx, seg_lo, seg_hi, targets, plan=None, pad=0, aux=None, count_first=False):
        """int32 (3, B [+ pad]): raw >= counts, filter correction, found-true flag for this shard.
        ``aux`` (a HIP stream): the filter correction -- which needs the true scores only -- runs there, beside the
        all-candidates count and its exact recheck on the current stream (fork / join by events: captured into the
        hipGraph of evaluate() as two parallel branches)."""
        pre_counts = prob.pre.pop('counts', None) if getattr(prob, 'pre', None) is not None else None
        if pre_counts is None and getattr(prob, 'zero_counts', None) is not None:
            # TransH / TransD: zeroed by kge_proj_query_stats' launch -- unless this sweep wants region counters zeroed too
            zc, prob.zero_counts = prob.zero_counts, None
            if pad == 0 and not (prob.wants_regions() if hasattr(prob, 'wants_regions') else 0):
                pre_counts = zc
        if pre_counts is not None and pad == 0 and tuple(pre_counts.shape) == (3, prob.B):
            out = pre_counts        # zeroed by the fused query pipeline's launch: no fill node
        else:
            # (+ the region counters of the sweep's uncertain-pair list where they apply -- TransH / TransD: the same fill)
            nreg = prob.wants_regions() if (pad == 0 and hasattr(prob, 'wants_regions')) else 0
            buf = torch.zeros(3 * (prob.B + pad) + nreg, dtype=torch.int32, device=s_true.device)
            out = buf[:3 * (prob.B + pad)].view(3, prob.B + pad)
            if nreg:
                prob.region_count = buf[3 * (prob.B + pad):]
        # (only beside the split-prefilter count kernel -- one persistent workgroup per CU that leaves 30 KB of LDS and a
        # fifth of the registers free; the fp32 tile kernel runs TWO workgroups per CU and loses one of them to a
        # co-resident kernel's LDS: measured 2.44 -> 3.37 ms per evaluate with --no-split)
        if aux is None or getattr(prob, 'split', None) is None:
            prob.count_ge(s_true, out[0])
            prob.filter_sub(s_true, true_idx, seg_lo, seg_hi, targets, out[1], out[2], grouped=True, plan=plan)
            return out
        main = torch.cuda.current_stream(s_true.device)
        if FILTER_BESIDE_RECHECK and prob.B > 0 and prob.N > 0:
            # the filter correction forked BEHIND the count sweep, beside its exact recheck (both are short, L2-bound kernels;
            # the sweep -- one or two persistent workgroups per CU -- then has the chip to itself)
            def fork():
                aux.wait_stream(main)
                with torch.cuda.stream(aux):
                    prob.filter_sub(s_true, true_idx, seg_lo, seg_hi, targets, out[1], out[2], grouped=True, plan=plan)
            prob._count_ge_split(s_true, out[0], between=fork)
            main.wait_stream(aux)
            return out
        aux.wait_stream(main)
        if count_first:
            prob.count_ge(s_true, out[0])
            with torch.cuda.stream(aux):
                prob.filter_sub(s_true, true_idx, seg_lo, seg_hi, targets, out[1], out[2], grouped=True, plan=plan)
        else:
            with torch.cuda.stream(aux):
                prob.filter_sub(s_true, true_idx, seg_lo, seg_hi, targets, out[1], out[2], grouped=True, plan=plan)
            prob.count_ge(s_true, out[0])
        main.wait_stream(aux)
        return out

    @staticmethod
    def finalize(counts):
        return _hip.rank_finalize(counts[0], counts[1], counts[2])

    writes_flags = True     # finalize_both(guard=, flags=): the guard decisions are written by the same launch

    @staticmethod
    def finalize_both(counts, out, off, pos=None, guard=None, flags=None, zero_guard=False, indirect=None):
        """Ranks of a 2B-query batch into the (4, n) result matrix at columns off..off+B-1 (or pos[off..]);
        ``flags`` (2 floats behind the ranks): [max ||q||^2 + max ||e||^2, list overflow] from the guard vector;
        ``zero_guard``: the launch leaves the guard vector zeroed for the next evaluation;
        ``indirect``: the launch writes to the matrix whose address it finds THERE (pinned host memory, r06)."""
        _hip.rank_finalize_both(counts[0], counts[1], counts[2], out, off, pos, guard, flags, zero_guard, indirect)

    writes_host = True      # finalize_both(indirect=) exists

    zeroes_guard = True     # finalize_both(zero_guard=True) exists

    @staticmethod
    def local_scores(prob):
        return prob.scores()

    @staticmethod
    def score_rows(prob, q0, q1, out):
        """Local scores of queries [q0, q1) into out[:q1 - q0] (one row block of the score all-to-all)."""
        return prob.scores_rows(q0, q1, out)

    @staticmethod
    def rank_tiles(tiles, n_total, true_idx, seg_lo, seg_hi, targets, rows, q_first, B, out, off, pos=None, own=None,
                   own_rank=0):
        """Ranks of `rows` queries from the rank-major tiles (P, m, per) the all-to-all delivered, into `out`
        (own: the block of this rank's own local tile, read in place instead of tiles[own_rank])."""
        return _hip.filtered_rank_from_tiles(tiles, n_total, true_idx, seg_lo, seg_hi, targets, rows, q_first, B,
                                             out, off, pos, own, own_rank)

    @staticmethod
    def ranks_from_scores(scores, true_idx, seg_lo, seg_hi, targets):
        return _hip.filtered_rank_from_scores(scores, true_idx, seg_lo, seg_hi, targets)


def _to_host(t):
    """Device -> host through a pinned buffer of torch's caching host allocator (a pageable
    destination is staged by the runtime in chunks: several times slower for the ~1 MB of ranks)."""
    if not t.is_cuda:
        return t
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host.copy_(t, non_blocking=True)
    # (polling an event instead -- `while not ev.query()` -- was measured 2.4 % SLOWER per evaluate, r05:
    # profiles/r05/host_wait_ab.txt; hipStreamSynchronize already spins)
    torch.cuda.current_stream(t.device).synchronize()
    return host


class _GraphSegments(object):
    """evaluate() of one entity shard as hipGraph segments with the collectives between them:
    the short launches of a batch replay without host gaps, the RCCL calls stay ordinary
    eager calls on the same stream (nothing collective is ever captured)."""

    def __init__(self):
        self.items = []                 # ('g', CUDAGraph) | ('c', callable)
        self.pool = torch.cuda.graph_pool_handle()   # one pool: later segments read earlier segments' tensors
        self._g = self._ctx = None

    def begin(self):
        self._g = torch.cuda.CUDAGraph()
        # thread-local capture mode: with a RCCL process group alive its watchdog thread keeps querying events, which
        # invalidates a capture in the default (global) mode (tools/probe/rccl_graph_probe.py)
        self._ctx = torch.cuda.graph(self._g, pool=self.pool, capture_error_mode='thread_local')
        self._ctx.__enter__()

    def end(self, exc=(None, None, None)):
        if self._ctx is not None:
            ctx, self._ctx = self._ctx, None
            ctx.__exit__(*exc)
            if exc[0] is None:
                self.items.append(('g', self._g))

    def cut(self, fn):
        self.end()
        self.items.append(('c', fn))
        self.begin()

    def replay(self):
        for kind, x in self.items:
            if kind == 'g':
                x.replay()
            else:
                x()


class _EvalState(object):
    """What an evaluator LEARNS about (model, knowledge graph): the static per-batch plans, the captured hipGraphs, the level
    of the split prefilter, the second stream.  The reference idiom builds ``LinkPredictionEvaluator(model, kg)`` anew for
    every validation (evaluation.py:252-262, docs/tutorials): the state is therefore kept per (model, kg, options) at
    module level -- weakly keyed on the model, so it dies with it -- and a NEW evaluator on the same pair replays the
    graph its predecessor captured instead of paying first-call prices again."""

    SHARED = ('_plans', '_plan_stamp', '_plan_refs', '_plan_gen', '_perm', '_qmap', '_graph', '_graph_static', '_graph_key',
              '_graph_src', '_graph_seen', '_graph_cache', '_aux_stream', '_n_evaluations', '_level', '_level1_max', '_level1_seen',
              '_level0_seen', '_mem_fit', '_graph_failed', '_ctimes', '_fast')

    def __init__(self):
        self._plans = self._plan_stamp = self._plan_refs = self._perm = self._qmap = None
        self._plan_gen = 0
        self._graph = self._graph_static = self._graph_key = self._graph_src = self._graph_seen = None
        self._graph_cache = {}  # key -> (graph, static state): the two levels' captures are both kept
        self._aux_stream = None
        self._n_evaluations = 0
        self._level = 0         # level of the split prefilter the next evaluation runs (see LEVEL1_ENTER)
        self._level1_max = None             # cap on the three-product re-scored pairs per query at which level 1 is RE-entered (None: no cap)
        self._level0_seen = None            # ... the last such count observed on level 0
        self._level1_seen = None            # re-scored pairs per query last observed on level 1 (the region decision's input)
        self._mem_fit = None
        self._graph_failed = False
        self._ctimes = None     # collective_timing(): (start, end) event pairs of the data-path collectives
        self._fast = None       # (signature, replay info) of the last steady-state graph replay: LinkPredictionEvaluator._fast_sig
        self.kg_ref = None
        self.group_ref = None   # the process group of the options key (held so that its id() cannot be reused while this state lives)


_STATES = weakref.WeakKeyDictionary()       # model -> {(id(kg), options): _EvalState}
SHARE_STATE = os.environ.get('KGE_SHARE_EVAL_STATE', '1') != '0'
# the filter correction of the second stream beside the exact recheck (1) instead of beside the count sweep (0)
FILTER_BESIDE_RECHECK = os.environ.get('KGE_FILTER_BESIDE_RECHECK', '0') == '1'
# steady-state evaluate() calls skip the full prologue (LinkPredictionEvaluator._fast_sig / _evaluate_fast, r06)
FAST_REPLAY = os.environ.get('KGE_FAST_REPLAY', '1') != '0'
_FLAGS4 = struct.Struct('4f')
# TransH / TransD: candidate-side preparation of an evaluation on the second stream, beside the query side (r06)
PREP_SIDE_STREAM = os.environ.get('KGE_PREP_SIDE_STREAM', '1') != '0'
# single-GPU both-sides evaluation: ranks + flags written by the finalize launches straight into pinned host memory (r06)
DIRECT_HOST_RANKS = os.environ.get('KGE_DIRECT_HOST_RANKS', '1') != '0'
# region recheck (one-product level): from this many re-scored pairs per query on the three-product level
REGION_MIN_LEVEL0 = float(os.environ.get('KGE_REGION_MIN_LEVEL0', '1.2'))
# ... and, once an evaluation on the one-product level has been seen, from this many re-scored pairs per query THERE (r06: rows
# longer than one LDS segment -- DistMult / ComplEx d = 400 -- gain 3.6 % at 9.6 pairs per query and lose 1-3 % at 2.0)
REGION_MIN_LEVEL1 = float(os.environ.get('KGE_REGION_MIN_LEVEL1', '5.0'))


# states kept per model: the least recently used one goes when a new (kg, options) pair would exceed it -- its captured
# hipGraphs, their memory pools, static fact / output buffers and plans are released then (at evaluator CONSTRUCTION: no
# capture of this thread is running).  Evaluators that still hold such a state keep working on it privately.
MAX_STATES_PER_MODEL = int(os.environ.get('KGE_EVAL_STATES_PER_MODEL', 4))


def _shared_state(model, kg, cfg, group=None):
    """The _EvalState of (model, kg, cfg) -- created on first use; None when the pair cannot be weakly referenced.
    ``group``: the process group whose id() is part of `cfg` -- held by the state, so that the id cannot be handed to
    another group object while the state lives."""
    import collections
    try:
        per_model = _STATES.get(model)
        if per_model is None:
            per_model = _STATES[model] = collections.OrderedDict()
        key = (id(kg), cfg)
        st = per_model.get(key)
        if st is not None and st.kg_ref() is kg and st.group_ref is group:
            per_model.move_to_end(key)
            return st
        for k in [k for k, v in per_model.items() if v.kg_ref() is None]:    # graphs that are gone (id() may be reused)
            del per_model[k]
        st = _EvalState()
        st.kg_ref = weakref.ref(kg)
        st.group_ref = group
        per_model[key] = st
        while len(per_model) > max(1, MAX_STATES_PER_MODEL):
            per_model.popitem(last=False)
        return st
    except TypeError:
        return None


def clear_eval_state(model=None):
    """Forget what evaluators have learned -- plans, split level, captured hipGraphs and their memory pools -- about
    `model` (None: about every model).  The state is shared by all LinkPredictionEvaluator objects built on the same
    (model, knowledge graph, options): a NEW evaluator inherits the level policy's state, the captured graphs and a
    failed-capture flag from its predecessors (``share_state=False`` or KGE_SHARE_EVAL_STATE=0 opt out).  Call this to
    release the device memory those graphs hold while the model stays alive, or to start from a clean slate after the
    tables were replaced by something unrelated.  Not while an evaluate() of the model is being captured."""
    if model is None:
        for m in list(_STATES.keys()):
            _STATES.pop(m, None)
    else:
        _STATES.pop(model, None)


class LinkPredictionEvaluator(object):
    """Evaluate a model by link prediction (evaluation.py:207-425).

    Parameters (beyond the reference's ``model, knowledge_graph``)
    ----------
    fused: bool -- rank without materialising the score matrix (default).
    shard: None | 'entities' | 'queries' -- multi-GPU partitioning (needs an
        initialised torch.distributed process group, one rank per GPU).
    exchange: 'counts' | 'scores' -- what entity shards exchange.
    graph: None | bool -- replay evaluate() as a hipGraph (removes the host launch
        gaps between the ~20 short kernels of a batch).  True: capture on the first
        call.  None (default, 'auto'): the first call with given shapes runs eagerly
        and serves as the warm-up, the second one captures, later ones replay -- a
        single evaluation never pays for a capture, a validation loop gets the graph.
        False: always eager.  A capture that fails (e.g. a user model that syncs)
        falls back to eager for good.
    group: torch.distributed process group (default: WORLD).
    query_exchange: 'evaluate' | 'batch' -- models whose entity tables are ROW-SHARDED
        (distributed.shard_model_): how the query rows reach every rank.  'evaluate'
        (default): the rows of the distinct entities the test facts mention are summed
        over the ranks once per evaluate() (owner contributes the row, the others zeros)
        and every batch builds its queries locally from those replicas; 'batch': each
        batch's (2B, K) query matrix is built by the owners and summed.  Same ranks.
    coalesce: None | int -- the batch the fused kernels see.  In the reference ``b_size`` bounds the
        (b, N, d) temporaries; here ranks are per query and a small b_size only means many small
        launches, so the facts are processed ``max(b_size, coalesce)`` at a time (None: the module
        default COALESCE_BATCH = 65536), never more than fits in a quarter of the FREE device memory
        (uncertain-pair list + query buffers; single-GPU evaluators).  ``coalesce=0`` takes
        ``b_size`` literally -- the opt-out when b_size is the script's memory knob.
    """

    def __init__(self, model, knowledge_graph, fused=True, shard=None, exchange='counts',
                 group=None, engine=None, graph=None, overlap=False, both_sides=True, query_exchange='evaluate',
                 coalesce=None, graph_collectives=None, share_state=None):
        self.model = model
        # entity shards + hipGraph: capture the RCCL collectives INSIDE the one graph of evaluate() (thread-local capture
        # mode: the process-group watchdog thread keeps querying events) instead of cutting the capture at every
        # collective.  Opt-in (None: env KGE_GRAPH_COLLECTIVES=1): verified on a world of one RCCL rank only.
        self.graph_collectives = (os.environ.get('KGE_GRAPH_COLLECTIVES') == '1') if graph_collectives is None \
            else bool(graph_collectives)
        # internal batch of the fused path (None: the module default COALESCE_BATCH, 0: exactly b_size)
        self.coalesce = coalesce
        self.kg = knowledge_graph
        n = knowledge_graph.n_facts
        self.rank_true_heads = torch.empty(size=(n,)).long()
        self.rank_true_tails = torch.empty(size=(n,)).long()
        self.filt_rank_true_heads = torch.empty(size=(n,)).long()
        self.filt_rank_true_tails = torch.empty(size=(n,)).long()
        self.evaluated = False
        assert shard in (None, 'entities', 'queries')
        assert exchange in ('counts', 'scores')
        self.fused, self.shard, self.exchange, self.group = fused, shard, exchange, group
        self.engine = engine if engine is not None else HipRankEngine()
        self.graph = graph if engine is None else False   # replay evaluate() as one hipGraph (see above)
        self.overlap = overlap                      # two-stream overlap of the short kernels (single GPU, fused)
        # both sides of a batch as ONE 2B-query problem (single GPU, fused): every latency-bound short
        # kernel of a batch runs once instead of twice, the all-candidates count kernel sees 2B queries
        self.both_sides = both_sides
        self._cut = None        # set while evaluate() is being captured as graph segments (see _GraphSegments)
        # per-batch FilterPlans (filter segments, true ids, grouping): a function of the test facts and the filter
        # index only, kept across evaluate() calls; _plan_stamp tells when they went stale
        self._fl, self._fl_done = None, False
        # row-sharded models: the distinct entities of the test facts and the facts re-indexed into that list
        # (static like the plans); their rows are exchanged ONCE per evaluate() (query_exchange='evaluate') instead
        # of the (2B, K) query rows of every batch ('batch')
        self.query_exchange = query_exchange
        self._qb = None
        self._shard_flags = None
        # models whose count kernel gathers a per-(relation, candidate) table in its epilogue (TransH / TransD) ask for
        # the facts of a batch to be PROCESSED sorted by relation: the queries of a wavefront then share one or two
        # relation rows of that table instead of 32 different ones.  _perm[j] = original position of the j-th processed
        # fact (the ranks are written straight to it); static like the plans.
        # the filter correction of a both-sides batch on a second stream, beside the count kernel and its recheck (single
        # GPU; measured r04, same box: TransE 0.629 -> 0.619 ms, DistMult / FB15k 2.607 -> 2.536, ComplEx / TransH +-0:
        # profiles/r04/overlap_filter_ab.txt).  KGE_OVERLAP_FILTER=0 keeps everything on one stream.
        self.overlap_filter = os.environ.get('KGE_OVERLAP_FILTER', '1') == '1'
        self.last_rescored_per_query = None
        # everything learned about (model, kg) -- plans, graphs, level, second stream: _EvalState, shared by the evaluators
        # of the same pair and options (share_state=False, a custom engine or KGE_SHARE_EVAL_STATE=0: private)
        share = SHARE_STATE if share_state is None else bool(share_state)
        st = None
        if share and engine is None:
            cfg = (fused, shard, exchange, id(group) if group is not None else None, graph, overlap, both_sides,
                   query_exchange, coalesce, self.graph_collectives, self.overlap_filter)
            st = _shared_state(model, knowledge_graph, cfg, group)
        self._st = st if st is not None else _EvalState()
        if self._st._graph_failed:
            self.graph = False
        # the second stream is created HERE, at construction (its creation costs tens of ms in a fresh process: better in
        # front of the first evaluation -- which pays the process's first-use costs anyway -- than as a spike inside the
        # second or third one)
        if self.overlap_filter and engine is None and self._st._aux_stream is None:
            try:
                dev0 = next(model.parameters()).device
                if dev0.type == 'cuda':
                    self._st._aux_stream = torch.cuda.Stream(dev0)
            except (StopIteration, RuntimeError):
                pass

    def _internal_batch(self, b_size, n_local):
        """Batch the fused kernels see.  In the reference ``b_size`` only bounds the (b, N, d) temporaries
        (evaluation.py:286-300); ranks are per query and do not depend on the batching, and the fused path holds
        O(b * d) bytes per batch plus the uncertain-pair list of the split prefilter.  A script written for the
        reference passes b_size = 32 .. 256: taken literally that is hundreds of launches that each fill a fraction
        of the GPU, so the facts are processed max(b_size, COALESCE_BATCH) at a time -- less when the list of a batch
        would pass COALESCE_LIST_BYTES (16 B per query and list slot, both sides).  Same formula on every rank."""
        target = COALESCE_BATCH if self.coalesce is None else int(self.coalesce)
        if target <= b_size or not self.fused or self._generic_model or not isinstance(self.engine, HipRankEngine):
            return b_size
        per_query = max(_hip.SPLIT_LIST_PER_QUERY, self.model.n_ent // 50)
        fit = max(1, int(COALESCE_LIST_BYTES // (16 * per_query)))
        if self.shard is None and torch.cuda.is_available():
            # next to a training job the device may be nearly full: the scratch of one internal batch (list slots of
            # both sides + ~4 (2B, K) fp32 query-side buffers) stays within a quarter of what is free.  Decided ONCE per
            # evaluator (b_size keys the plans and the hipGraph: a value that followed the allocator's jitter would
            # rebuild both on every call), as a power of two, counting the blocks torch's allocator holds cached as
            # free; re-decided only when the memory now free would shrink the batch by 2x or more.
            # (Sharded evaluators keep the rank-independent formula: every rank must cut the same batches.)
            try:
                kept = getattr(self, '_mem_fit', None)
                # (hipMemGetInfo + the allocator's statistics cost ~30 us of host time with the GPU idle in front of a
                # 0.5 ms step: asked on the first evaluation and every 32nd one after it)
                if kept is None or self._n_evaluations % 32 == 0:
                    dev = getattr(self, '_dev', None) or next(self.model.parameters()).device
                    free = torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
                    k_q = 2 * int(getattr(self.model, 'emb_dim', 0) or 0) + 64
                    now = max(1, int((free // 4) // (16 * per_query + 2 * 4 * 4 * k_q)))
                    now = 1 << (now.bit_length() - 1)
                    if kept is None or now * 2 <= kept:
                        self._mem_fit = kept = now
                fit = min(fit, kept)
            except Exception:
                pass
        return max(b_size, min(target, fit, max(n_local, 1)))

    def _ensure_plans(self, kg, f_lo, f_hi, b_size, index_t, index_h, device):
        """Build (or keep) the FilterPlans of every batch of this evaluation, OUTSIDE any graph capture."""
        want_sort = bool(getattr(self.model, 'lp_sort_queries_by_relation', False))
        stamp = (b_size, f_lo, f_hi, str(device), want_sort, DEDUPE_QUERIES, getattr(self.model, 'lp_dedupe_queries', False),
                 tuple((x.data_ptr(), x._version, x.shape[0]) for x in (kg.head_idx, kg.tail_idx, kg.relations)),
                 tuple((x.data_ptr(), x.shape[0]) for ix in (index_h, index_t) for x in (ix.keys, ix.offsets, ix.targets)))
        # ... and the stamped tensors themselves (strong references, compared by identity): a data_ptr / shape stamp
        # alone would accept a swapped kg or a rebuilt filter index that landed on a recycled address
        refs = (kg.head_idx, kg.tail_idx, kg.relations, index_h.keys, index_h.offsets, index_h.targets,
                index_t.keys, index_t.offsets, index_t.targets)
        if (self._plans is not None and self._plan_stamp == stamp and self._plan_refs is not None
                and len(refs) == len(self._plan_refs) and all(a is b for a, b in zip(refs, self._plan_refs))):
            return
        self._plan_refs = refs
        heads, tails, rels = (kg.head_idx[f_lo:f_hi].to(device), kg.tail_idx[f_lo:f_hi].to(device),
                              kg.relations[f_lo:f_hi].to(device))
        self._perm = None
        if want_sort and rels.shape[0] > 0:
            # stable sort by relation INSIDE every batch (batch membership is unchanged)
            batch_of = torch.div(torch.arange(rels.shape[0], device=device), b_size, rounding_mode='floor')
            n_rel = int(self.model.n_rel)
            n_keys = (int(rels.shape[0] - 1) // b_size + 1) * n_rel
            if rels.is_cuda and n_keys <= (1 << 32) and isinstance(self.engine, HipRankEngine):
                self._perm = _hip.sort_perm(batch_of * n_rel + rels, n_keys)   # the library's radix sort (no ATen sort)
            else:
                self._perm = torch.argsort(batch_of * n_rel + rels, stable=True).contiguous()
            heads, tails, rels = heads[self._perm], tails[self._perm], rels[self._perm]
        n = heads.shape[0]
        world, rank = kdist.world_and_rank(self.group) if self.shard else (1, 0)
        self._qmap = None
        if self.shard == 'entities' and kdist.multi(world):
            # (row-sharded tables only: the distinct entities of the test facts and the facts re-indexed into that list)
            uniq, inv = torch.unique(torch.cat([heads, tails]), return_inverse=True)
            self._qmap = {'uniq': uniq.contiguous(), 'hq': inv[:n].contiguous(), 'tq': inv[n:].contiguous()}
            # uniq is sorted and the shards are contiguous id ranges: rank p owns ONE slice [a_p, b_p) of uniq
            per = kdist.shard_size(self.model.n_ent, world)
            cuts = torch.searchsorted(uniq, torch.arange(world + 1, device=device, dtype=uniq.dtype) * per).tolist()
            cuts[-1] = int(uniq.shape[0])
            maxc = max(1, max(b - a for a, b in zip(cuts[:-1], cuts[1:])))
            owner = torch.div(uniq, per, rounding_mode='floor')
            first = torch.tensor(cuts[:-1], device=device, dtype=torch.int64)
            self._qmap.update({
                'maxc': maxc,
                'mine_local': (uniq[cuts[rank]:cuts[rank + 1]] - rank * per).contiguous(),
                # row of uniq[u] in the gathered (P * maxc, d) layout: its owner's block, its position in the slice
                'sel': (owner * maxc + (torch.arange(uniq.shape[0], device=device) - first[owner])).contiguous()})
        plans = {}
        for i in range(get_n_batches(f_hi - f_lo, b_size)):
            sl = slice(i * b_size, (i + 1) * b_size)
            plans[(i * b_size, heads[sl].shape[0])] = self.engine.plan_both(index_t, index_h, heads[sl], tails[sl], rels[sl],
                                                                            model=self.model)
        self._plans, self._plan_stamp = plans, stamp
        self._plan_gen += 1

    # -- filter indices ------------------------------------------------------
    def _filter_indices(self, device):
        kg = self.kg
        if hasattr(kg, 'filter_index'):
            return kg.filter_index('heads', device), kg.filter_index('tails', device)
        return (filter_index_for(kg.dict_of_heads, device), filter_index_for(kg.dict_of_tails, device))

    # -- one side of one batch ------------------------------------------------
    def _rank_side(self, h, t, r, side, index, lo, hi, sharded):
        eng = self.engine
        key1, true_idx = (h, t) if side == 'tail' else (t, h)
        seg_lo, seg_hi = eng.lookup(index, key1, r)
        if self._generic_model:
            return self._rank_side_generic(h, t, r, side, index, true_idx, key1)
        prob = eng.problem(self.model, h, t, r, side, lo, hi, **self._xkw(sharded))
        if self.fused and not (sharded and self.exchange == 'scores'):
            s_true = eng.true_scores(prob, true_idx)
            if sharded:
                self._timed(lambda: kdist.all_reduce_sum(s_true, self.group))()    # owner shard holds the value, others 0
            counts = eng.partial_counts(prob, s_true, true_idx, seg_lo, seg_hi, index.targets)
            if sharded:
                self._timed(lambda: kdist.all_reduce_sum(counts, self.group))()
            return eng.finalize(counts)
        scores = eng.local_scores(prob)
        if sharded:
            scores = self._timed(lambda: kdist.all_gather_columns(scores, self.model.n_ent, self.group))()
        return eng.ranks_from_scores(scores, true_idx, seg_lo, seg_hi, index.targets)

    def _rank_batch_both(self, h, t, r, index_t, index_h, out, off, lo, hi, sharded, last=False, guard=None):
        """Both sides of one batch through one problem of 2B queries (tail side first):
        one filter lookup, one query-side launch, one count (+ recheck), one filter
        correction, one finalize into columns off.. of the (4, n) result matrix.  Ranks
        are per query: identical to two _rank_side calls.  Entity-sharded: ONE collective per
        batch -- the (3, 2B) partial rank counts (+ the guard flags on the last batch); the (2B) true
        scores need a second one only when no query-entity replicas are at hand (the owner shard holds
        the value, the others 0; x + 0 is exact)."""
        eng = self.engine
        plan = self._plans.get((off, h.shape[0])) if self._plans is not None else None
        if plan is not None:    # filter segments, true ids and the grouping of the batch: precomputed (FilterPlan)
            seg_lo, seg_hi, true_idx, targets = plan.seg_lo, plan.seg_hi, plan.true_idx, plan.targets
        else:
            seg_lo, seg_hi, true_idx, targets = eng.lookup_both(index_t, index_h, h, t, r)
        xkw = self._xkw(sharded)
        by_scores = sharded and self.exchange == 'scores'
        lvl1 = hasattr(self.model, '_use_level1') and self.model._use_level1()      # (the policy's choice, or a forced level)
        # (one-product level: the matrix work a shared row saves is a third of what it was, the grouped columns' multi-pass
        # epilogue costs what it always did -- models say whether columns still pay there: lp_dedupe_level1)
        # (... the free-running one-product kernel takes columns since r06 where the model's count has plain thresholds and
        # its rows fit the resident panel: Model._level1_stream_columns)
        lvl1_cols = DEDUPE_LEVEL1 and getattr(self.model, 'lp_dedupe_level1', True) and \
            (not (hasattr(self.model, '_level1_stream') and self.model._level1_stream()) or
             (hasattr(self.model, '_level1_stream_columns') and self.model._level1_stream_columns()))
        if plan is not None and getattr(plan, 'cols', None) is not None and not by_scores and (lvl1_cols or not lvl1):
            xkw['cols'] = plan.cols     # (entity shards too: the columns are a property of the queries, not of the candidates)
        # (the relation id per query of a both-sides batch -- [r | r] -- precomputed with the plan: no concatenation kernel per batch)
        hint = getattr(plan, 'r_both', None) if plan is not None else None
        if hint is not None and hasattr(self.model, '_lp_r_both'):
            object.__setattr__(self.model, '_lp_r_both', hint)
        try:
            prob = eng.problem(self.model, h, t, r, 'both', lo, hi, **xkw)
        finally:
            if hint is not None and hasattr(self.model, '_lp_r_both'):
                object.__setattr__(self.model, '_lp_r_both', None)
        if by_scores:
            return self._exchange_score_tiles(prob, h.shape[0], true_idx, seg_lo, seg_hi, targets, out, off)
        s_true = None
        if sharded and self._qb is not None and getattr(prob, 'pre', None) is not None and prob.pre.get('true_idx') is None:
            # row-sharded tables, fused query side fed from the query-entity replicas (r05): the pipeline has scored
            # exactly the (query, true entity) pairs -- on every rank, from the same rows: no collective
            prob.pre['true_idx'] = true_idx
            s_true = prob.pre['s_true']
            if hasattr(prob, 'split_true'):
                prob.split_true = (s_true, true_idx)
        elif sharded and self._qb is not None and hasattr(self.model, 'lp_true_scores_replica'):
            # row-sharded tables: the true entities' rows are in the query-entity replicas, so every rank scores
            # the (query, true entity) pairs itself -- same rows, same chain, same bits: no collective
            s_true = self.model.lp_true_scores_replica(prob, self._qb)
        if s_true is None:
            s_true = eng.true_scores(prob, true_idx)
            if sharded:
                self._collective(lambda s_=s_true, g_=self.group: kdist.all_reduce_sum(s_, g_))
        n2 = s_true.shape[0]
        # entity shards, last batch: the two guard decisions ride the counts exchange as 0 / 1 columns (a SUM > 0
        # means "some rank says so"; max ||q||^2 is the same on every rank, so "max_q + max_e_p > limit on some
        # rank p" IS "max_q + max_p max_e_p > limit") -- no separate MAX all-reduce, no extra host sync
        ride = sharded and last and guard is not None and getattr(eng, 'flag_columns', False)
        kw = {'plan': plan} if plan is not None else {}
        if ride:
            kw['pad'] = 3
        if self._use_aux and not sharded and isinstance(eng, HipRankEngine) and s_true.is_cuda:
            kw['aux'] = self._aux_stream        # (created by evaluate(), outside any capture)
            kw['count_first'] = bool(getattr(self.model, 'lp_count_first', False)) if COUNT_FIRST is None else COUNT_FIRST
        counts = eng.partial_counts(prob, s_true, true_idx, seg_lo, seg_hi, targets, **kw)
        if ride:
            lim = float(self.model.L2_EXPAND_LIMIT)
            counts[0, n2:n2 + 1] = ((guard[0:1] + guard[1:2]) > lim).to(torch.int32) if lim != float('inf') else 0
            counts[0, n2 + 1:n2 + 2] = (guard[2:3] > 0).to(torch.int32)
            counts[0, n2 + 2:n2 + 3] = guard[6:7].to(torch.int32)      # pairs this shard re-scored (level policy: their SUM)
        if sharded:     # (the recorded call runs again at every graph replay: bind the tensor, not the name)
            self._collective(lambda c_=counts, g_=self.group: kdist.all_reduce_sum(c_, g_))
        if ride:
            self._shard_flags = counts[0, n2:n2 + 3]
        fkw = {}
        if (last and guard is not None and not sharded and self._fl is not None and getattr(eng, 'writes_flags', False)):
            fkw = {'guard': guard, 'flags': self._fl}       # the last finalize also writes the two guard flags
            self._fl_done = True
            if getattr(eng, 'zeroes_guard', False):         # ... and leaves the guard vector zeroed for the next evaluation
                fkw['zero_guard'] = True
                self._guard_zeroed = True
        direct = self.__dict__.get('_direct_ptr') if not sharded else None
        if direct and self._perm is None:
            fkw['indirect'] = direct            # ranks (and flags) straight into pinned host memory
        if self._perm is not None:
            eng.finalize_both(counts[:, :n2] if ride else counts, out, off, self._perm, **fkw)
            if direct and last:
                # facts processed in another order (sorted by relation): the finalize launches scatter into the device matrix;
                # the packed result (ranks + flags, `out` is its head) then leaves in ONE coalesced pass of the same graph
                _hip.copy_i64_indirect(out, 4 * out.shape[1] + 2, direct)
        else:
            eng.finalize_both(counts[:, :n2] if ride else counts, out, off, **fkw)

    def _exchange_score_tiles(self, prob, B, true_idx, seg_lo, seg_hi, targets, out, off):
        """exchange='scores' of one both-sides batch (2B queries): the collective north_star names, as a path that
        scales.  The queries are cut into row tiles of P * m rows; every rank scores a tile against ITS candidates
        (fp32 scores, (P * m, N/P) -- bounded by SCORE_TILE_BYTES whatever b_size is), ONE all-to-all hands rank j the
        m rows it ranks as P rank-major tiles, and kge_filtered_rank_from_tiles ranks them where they lie (true score
        read from the tiles like the reference reads it from the score matrix, evaluation.py:291-300).  Per rank and
        evaluation: (P-1)/P^2 * 2B*N*4 bytes on the fabric (the all-gather moved P times that and ranked all 2B rows on
        every rank).  Each rank writes only its queries' columns of the zero-initialised result matrix; one int64
        all-reduce at the end of evaluate() completes it."""
        eng = self.engine
        world, rank = kdist.world_and_rank(self.group)
        n_ent = self.model.n_ent
        per = kdist.shard_size(n_ent, world)
        n2 = 2 * B
        m_max = max(1, min(-(-n2 // world), SCORE_TILE_BYTES // (4 * per * world)))
        dev = true_idx.device
        for q0 in range(0, n2, world * m_max):
            q1 = min(n2, q0 + world * m_max)
            m = -(-(q1 - q0) // world)
            loc = torch.empty(world * m, per, dtype=torch.float32, device=dev)     # rows >= q1 - q0 / columns >= the
            eng.score_rows(prob, q0, q1, loc)                                      # shard's width: never read
            recv = torch.empty(world, m, per, dtype=torch.float32, device=dev)
            # (a world of one -- forced collectives -- exchanges nothing: the rank kernel reads the own block from `loc`;
            # a property of the process group, so it is the same when this call is recorded into graph segments)
            in_place = world == 1
            self._collective(lambda a_=loc, b_=recv, g_=self.group: kdist.all_to_all_rows(a_, b_, g_))
            my0 = q0 + rank * m
            rows = min(m, q1 - my0)
            if rows > 0:
                kw = {'own': loc[rank * m:(rank + 1) * m], 'own_rank': rank} if in_place else {}
                eng.rank_tiles(recv, n_ent, true_idx[my0:], seg_lo[my0:], seg_hi[my0:], targets, rows, my0, B, out, off,
                               self._perm, **kw)

    def _xkw(self, sharded):
        """Engine keyword for the query exchange of row-sharded entity tables: every rank builds the
        query rows whose entity it owns (zeros elsewhere) and ONE all-reduce SUM per query matrix
        hands the (2B, K) rows to every rank (x + 0 is exact)."""
        if not sharded:
            return {}
        kw = {'exchange': lambda tensors, g_=self.group: self._collective(
            lambda: [kdist.all_reduce_sum(x, g_) for x in tensors])}
        if self._qb is not None:
            kw['qctx'] = self._qb
        return kw

    def _timed(self, fn):
        """`fn` bracketed by two events on the current stream while collective timing is on (decided when the call
        RUNS: the calls recorded into graph segments at capture time are timed at replay too)."""
        st = self._st       # (NOT the evaluator: recorded calls live in the shared state of (model, kg) and are replayed by later
        # evaluator objects -- a closure over `self` would keep the first evaluator, and through it the model that keys the
        # weak dictionary of states, alive for ever)

        def call(*a):
            # (timing-enabled events must not be recorded into a stream capture: with the collectives captured inside
            # the graph -- graph_collectives -- the capture call runs untimed, and replays contain no Python calls)
            if st._ctimes is None or torch.cuda.is_current_stream_capturing():
                return fn(*a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a)
            e1.record()
            st._ctimes.append((e0, e1))
            return out
        return call

    def collective_timing(self, on=True):
        """Device time of the data-path collectives (RCCL calls on the evaluation stream, bracketed by events):
        ``collective_timing(True)`` starts collecting, ``collective_timing(False)`` returns
        ``{'collectives': n, 'ms': total}`` for the evaluate() calls in between and stops.  Collectives that were
        captured INSIDE the hipGraph of evaluate() (``graph_collectives``) are not Python calls at replay: they are
        not timed (0 collectives reported); graph segments and eager evaluations are."""
        if on:
            self._ctimes = []
            return None
        ev, self._ctimes = self._ctimes or [], None
        if ev:
            ev[-1][1].synchronize()
        return {'collectives': len(ev), 'ms': round(sum(a.elapsed_time(b) for a, b in ev), 4)}

    def _collective_value(self, fn, shape, like):
        """A collective that RETURNS a tensor: run it now -- or, while evaluate() is being captured, allocate the
        result in the graph's pool, cut the capture, and record a call that fills that buffer at replay."""
        buf = like.new_empty(shape)
        fn = self._timed(fn)
        if self._cut is None:
            fn(buf)
        else:
            self._cut(lambda: fn(buf))
        return buf

    def _collective(self, fn):
        """Run a collective now -- or, while evaluate() is being captured, close the current
        graph segment, record the call, and open the next segment."""
        fn = self._timed(fn)
        if self._cut is None:
            fn()
        else:
            self._cut(fn)

    def _rank_batch_overlapped(self, h, t, r, index_t, index_h):
        """Both sides of one batch on two HIP streams: the short kernels (filter
        lookup, query prep, true scores, filter correction) of one side run on an
        auxiliary stream underneath the other side's long all-candidates count
        kernel instead of in front of it.  Same kernels, same results."""
        dev = h.device
        main = torch.cuda.current_stream(dev)
        if self._aux_stream is None:
            self._aux_stream = torch.cuda.Stream(dev)
        aux = self._aux_stream
        B = h.shape[0]
        m = self.model
        # tail side prepared on the main stream
        lo_t, hi_t = index_t.lookup(h, r)
        prob_t = m.lp_problem(h, t, r, 'tail')
        st_t = prob_t.pair_scores(t)
        cnt_t = torch.zeros(3, B, dtype=torch.int32, device=dev)
        cnt_h = torch.zeros(3, B, dtype=torch.int32, device=dev)
        e_t = torch.cuda.Event()
        e_t.record(main)
        aux.wait_event(e_t)
        with torch.cuda.stream(aux):     # head-side prep + tail filter correction, under the tail GEMM
            lo_h, hi_h = index_h.lookup(t, r)
            prob_h = m.lp_problem(h, t, r, 'head')
            st_h = prob_h.pair_scores(h)
            e_h = torch.cuda.Event()
            e_h.record(aux)
            prob_t.filter_sub(st_t, t, lo_t, hi_t, index_t.targets, cnt_t[1], cnt_t[2])
        prob_t.count_ge(st_t, cnt_t[0])
        main.wait_event(e_h)
        prob_h.count_ge(st_h, cnt_h[0])
        with torch.cuda.stream(aux):     # head filter correction, under the head GEMM
            prob_h.filter_sub(st_h, h, lo_h, hi_h, index_h.targets, cnt_h[1], cnt_h[2])
            e_done = torch.cuda.Event()
            e_done.record(aux)
        main.wait_event(e_done)
        for x in [lo_h, hi_h, st_h] + [k for k in prob_h.keep if k is not None]:
            x.record_stream(main)        # allocated on aux, consumed on main
        for x in [lo_t, hi_t, st_t, cnt_t, cnt_h] + [k for k in prob_t.keep if k is not None]:
            x.record_stream(aux)         # allocated on main, consumed on aux
        rk_t, frk_t = _hip.rank_finalize(cnt_t[0], cnt_t[1], cnt_t[2])
        rk_h, frk_h = _hip.rank_finalize(cnt_h[0], cnt_h[1], cnt_h[2])
        return rk_t, frk_t, rk_h, frk_h

    def _rank_side_generic(self, h, t, r, side, index, true_idx, key1):
        """Reference composition (evaluation.py:290-300) for models that only
        implement the public inference_* API."""
        h_emb, t_emb, r_emb, cand = self.model.inference_prepare_candidates(h, t, r, entities=True)
        if side == 'tail':
            scores = self.model.inference_scoring_function(h_emb, cand, r_emb)
        else:
            scores = self.model.inference_scoring_function(cand, t_emb, r_emb)
        filt = filter_scores(scores, index, key1, r, true_idx)
        return get_rank(scores, true_idx).detach(), get_rank(filt, true_idx).detach()

    def evaluate(self, b_size, verbose=True):
        """Rank the true head and tail of every fact of ``kg`` among all
        entities, raw and filtered (evaluation.py:263-308)."""
        # STEADY STATE (r06): the previous evaluation of this (model, kg, options) was a plain single-GPU graph replay and
        # nothing that keys the capture has changed -- same b_size, same fact tensors (identity + version), same table
        # addresses, same kernel-choice switches, same split level: replay at once.  The GPU idles while the host prepares a
        # replay (~65 us of a 0.49 ms step at cfg2: profiles/r05/timeline_transe_fb15k237.txt); the full prologue below --
        # plan stamps, capture key, filter indices, guard / session bookkeeping -- is ~35 us of that.  Anything unusual
        # (a guard flag up, a list overflow, every 32nd evaluation's memory check) goes through the full path.
        user_b_size = b_size
        fast = self._st._fast if FAST_REPLAY else None
        if fast is not None:
            if self._n_evaluations % 32 != 0 and fast[0] == self._fast_sig(user_b_size) and self._evaluate_fast(fast[1]):
                return
            self._st._fast = None
        # (the tensors whose addresses the capture key holds: the model's own table list where it has one -- a walk over
        # Module.parameters() costs three times as much host time, and the GPU idles while the host prepares the replay)
        tables_of = getattr(self.model, '_tables', None)
        try:
            params = list(tables_of()) if callable(tables_of) else None
        except NotImplementedError:         # (a user model on the generic path: interfaces.Model._tables is abstract)
            params = None
        if not params:
            params = list(self.model.parameters())
        device = params[0].device
        self._dev = device
        self.engine.check_device(device)
        kg = self.kg
        from .models.interfaces import Model as _BaseModel
        impl = getattr(type(self.model), 'lp_problem', None)
        self._generic_model = impl is None or impl is _BaseModel.lp_problem

        world, rank = kdist.world_and_rank(self.group) if self.shard else (1, 0)
        world_n = world
        sharded = self.shard == 'entities' and kdist.multi(world)
        lo, hi = kdist.shard_range(self.model.n_ent, world, rank) if sharded else (0, self.model.n_ent)
        row_shard = getattr(self.model, '_row_shard', None)
        if row_shard is not None and (not sharded or row_shard != (lo, hi)):
            raise RuntimeError('torchkge_amd: the model holds only entity rows [%d, %d): evaluate it with '
                               "shard='entities' on the process group it was sharded over" % row_shard)
        if self.shard == 'queries' and kdist.multi(world):
            f_lo, f_hi = kdist.shard_range(kg.n_facts, world, rank)
        else:
            f_lo, f_hi = 0, kg.n_facts

        # the filter correction on a second stream beside the count kernel (single GPU): from the second evaluation on --
        # creating and first using a stream costs tens of ms in a fresh process -- or at once when the caller asked for an
        # immediate capture (graph=True); the stream is created HERE, never inside a capture
        self._use_aux = bool(self.overlap_filter and not sharded and device.type == 'cuda' and
                             isinstance(self.engine, HipRankEngine) and (self._n_evaluations > 0 or self.graph is True))
        if self._use_aux and self._aux_stream is None:
            self._aux_stream = torch.cuda.Stream(device)
        n_local = f_hi - f_lo
        b_size = self._internal_batch(b_size, n_local)
        index_h, index_t = self._filter_indices(device)
        guard = None
        if hasattr(self.model, 'lp_guard_begin') and not self._generic_model and device.type == 'cuda':
            was_clean = bool(getattr(self.model, '_lp_guard_clean', False))
            same_guard = getattr(self.model, '_lp_guard', None)
            guard = self.model.lp_guard_begin(device)   # TransE-L2: optimistic MFMA norm expansion
            if was_clean and guard is not None and guard is same_guard:
                object.__setattr__(self.model, '_lp_guard_clean', True)     # (zeroed by the last evaluation's finalize launch)
        session = self.model.lp_session() if hasattr(self.model, 'lp_session') else _NullCtx()
        # (TransH / TransD: the candidate-side preparation runs on the second stream beside the query side's launches --
        # Model._proj_fast_problem; KGE_PREP_SIDE_STREAM=0 keeps one stream)
        if self._use_aux and PREP_SIDE_STREAM and not self._generic_model:
            object.__setattr__(self.model, '_lp_side_stream', self._aux_stream)

        try:
            overlap = (self.overlap and self.fused and not sharded and not self._generic_model and
                       isinstance(self.engine, HipRankEngine) and device.type == 'cuda')

            # (exchange='scores' on entity shards: the all-to-all of score row tiles needs an engine that ranks tiles;
            # others keep the per-side all-gather of full score rows)
            by_scores = sharded and self.exchange == 'scores'
            both = (self.both_sides and self.fused and not self._generic_model and not overlap and
                    (not by_scores or hasattr(self.engine, 'rank_tiles')) and hasattr(self.engine, 'lookup_both'))
            by_scores = by_scores and both
            if both and getattr(self.engine, 'uses_plans', False) and n_local > 0:
                self._ensure_plans(kg, f_lo, f_hi, b_size, index_t, index_h, device)
            else:
                self._plans = self._qmap = self._perm = None
            use_qmap = (row_shard is not None and self._qmap is not None and self.query_exchange == 'evaluate')
            self._qb = None

            def facts():
                """The facts of this evaluation on the device, in processing order."""
                hh, tt, rr = (kg.head_idx[f_lo:f_hi].to(device), kg.tail_idx[f_lo:f_hi].to(device),
                              kg.relations[f_lo:f_hi].to(device))
                if self._perm is not None:
                    hh, tt, rr = hh[self._perm], tt[self._perm], rr[self._perm]
                return hh, tt, rr

            def alloc_out():
                # (4, n) ranks + two trailing int64 that carry the guard flags (4 floats: norm guard, list overflow,
                # re-scored pairs, spare): ONE device-to-host copy
                flat = torch.empty(4 * n_local + 2, dtype=torch.int64, device=device)
                return flat, flat[:4 * n_local].view(4, n_local), flat[4 * n_local:].view(torch.float32)

            def run(heads, tails, rels, out, fl):
                with session, torch.no_grad():
                    self._guard_zeroed = False
                    if guard is not None:
                        # (clean when the previous evaluation's last finalize zeroed it -- Model._lp_guard_clean; replays
                        # of a graph captured without this fill check the flag in front of the replay)
                        if self.model._expand_ok is None and not getattr(self.model, '_lp_guard_clean', False):
                            guard.zero_()
                        # whatever this run does, it dirties the vector: a redo that raises or ends without a zeroing
                        # finalize must not leave the flag standing over stale overflow / re-scored counts
                        object.__setattr__(self.model, '_lp_guard_clean', False)
                    n_batches = get_n_batches(n_local, b_size)
                    if by_scores:       # every rank writes only the columns of the queries it ranks
                        out.zero_()
                    self._shard_flags = None
                    self._fl, self._fl_done = fl, False
                    qt = None
                    if use_qmap:    # row-sharded tables: replicas of the rows of the query entities, once per evaluate()
                        gather = None
                        if getattr(self.engine, 'uses_plans', False):
                            gather = lambda blk, g_=self.group: self._collective_value(
                                lambda out_: kdist.all_gather_blocks(blk, out_, g_),
                                (blk.shape[0] * world_n, blk.shape[1]), blk)
                        qt = self.model.lp_query_tables(self._qmap, self._xkw(True)['exchange'], gather)
                    for i in tqdm(range(n_batches), total=n_batches, unit='batch', disable=(not verbose),
                                  desc='Link prediction evaluation'):
                        sl = slice(i * b_size, (i + 1) * b_size)
                        h, t, r = heads[sl], tails[sl], rels[sl]
                        self._qb = (qt, self._qmap['hq'][sl], self._qmap['tq'][sl]) if qt is not None else None
                        if overlap:
                            out[1, sl], out[3, sl], out[0, sl], out[2, sl] = \
                                self._rank_batch_overlapped(h, t, r, index_t, index_h)
                            continue
                        if both:
                            self._rank_batch_both(h, t, r, index_t, index_h, out, i * b_size, lo, hi, sharded,
                                                  last=(i == n_batches - 1), guard=guard)
                            continue
                        out[1, sl], out[3, sl] = self._rank_side(h, t, r, 'tail', index_t, lo, hi, sharded)
                        out[0, sl], out[2, sl] = self._rank_side(h, t, r, 'head', index_h, lo, hi, sharded)
                    if by_scores:       # ... and one SUM completes the (4, n) rank matrix on every rank
                        self._collective(lambda o_=out, g_=self.group: kdist.all_reduce_sum(o_, g_))
                    if guard is not None and self._shard_flags is not None:
                        # entity shards: the flags came back summed over the ranks with the last batch's counts
                        fl[0:1].copy_(torch.where(self._shard_flags[0:1] > 0, float('inf'), 0.0))
                        fl[1:2].copy_(self._shard_flags[1:2].to(torch.float32))
                        fl[2:3].copy_(self._shard_flags[2:3].to(torch.float32))
                    elif guard is not None and not self._fl_done:
                        # [max ||q||^2 + max ||e||^2, split-prefilter overflow] behind the ranks (the both-sides path
                        # has the last batch's finalize write them)
                        torch.add(guard[0:1], guard[1:2], out=fl[0:1])
                        fl[1:2].copy_(guard[2:3])
                        fl[2:3].copy_(guard[6:7])
                    self._fl = None
                    if guard is not None and self._guard_zeroed:
                        object.__setattr__(self.model, '_lp_guard_clean', True)

            # one hipGraph when run() contains no collective (single GPU, query shards); graph segments with
            # the collectives between them for entity shards exchanging counts; eager otherwise
            # level of the split prefilter for THIS evaluation (single GPU, fused): decided by the previous one
            # (entity shards exchanging counts: the re-scored pair count rides the counts all-reduce like the guard flags, so
            # every rank takes the same decision -- r05; other multi-rank forms stay on the three-product level)
            # (r06: query shards too -- their ranks meet only in the final all-gather, the re-scored pair counts are summed over
            # the ranks beside the guard flags so that every rank takes the same decision and captures the same graphs)
            level_ok = (not kdist.multi(world)) or self.shard == 'queries' or (sharded and both and not by_scores and
                                                                                getattr(self.engine, 'flag_columns', False))
            level_now = self._level if (guard is not None and both and level_ok) else 0
            forced_level = getattr(self.model, 'split_level', 'auto')
            if forced_level != 'auto' and guard is not None and both:
                # (a forced level is what Model._use_level1 runs whatever the policy's state says: the capture key, the region
                # decision and Model._split_level name THAT level -- r06: a fresh shared state used to report 0 here)
                level_now = 1 if (int(forced_level) == 1 and self.model._use_level1()) else 0
            # (the WHOLE candidate range: entity shards sum their re-scored pairs over the ranks -- same number, same decision)
            lvl_enter, lvl_leave = level1_thresholds(self.model.n_ent)
            if hasattr(self.model, '_split_level'):
                object.__setattr__(self.model, '_split_level', level_now)
            # the sweep's uncertain pairs in regions of 32 queries (kge_lp_split_recheck_regions) pay when a region holds
            # enough of them to amortise its query rows: ~5 x the three-product level's count on the one-product level
            # -> from REGION_MIN_LEVEL0 re-scored pairs per query there (TransE cfg2: 1.8 -> 9.5 per query: 0.50 -> 0.48 ms;
            # TransH at 5 per query: 0.64 -> 0.66)
            regions_now = self._regions_for(level_now, kdist.multi(world))
            if hasattr(self.model, '_split_level'):
                object.__setattr__(self.model, '_lp_regions', regions_now)
            multi = kdist.multi(world)
            segmented = multi and sharded and both
            one_graph = segmented and self.graph_collectives and kdist.backend_name(self.group) == 'nccl'
            use_graph = (self.graph is not False and device.type == 'cuda' and n_local > 0 and
                         not self._generic_model and
                         (not multi or self.shard == 'queries' or segmented))
            if segmented and os.environ.get('KGE_EAGER_COLLECTIVES') == '1':
                # escape hatch for a multi-GPU box (no code change): no graph segments, no captured collectives --
                # every kernel and every RCCL call of a sharded evaluate() is an ordinary eager launch
                use_graph = one_graph = False
            # (r06) single GPU, both-sides batches: the finalize launches write ranks and flags straight into a pinned host
            # buffer (its address reaches them through a mailbox the host fills before every run / replay): no rank copy
            host_buf = None
            self.__dict__['_direct_ptr'] = None
            # (where the facts are processed in another order -- TransH / TransD sort them by relation -- the finalize launches
            # keep scattering on the device, scattered 8-byte stores across PCIe cost 0.60 -> 0.66 ms, and one coalesced pass
            # at the end of the graph carries the packed result over: _rank_batch_both)
            if (DIRECT_HOST_RANKS and not multi and both and guard is not None and n_local > 0 and not by_scores and not overlap
                    and device.type == 'cuda' and getattr(self.engine, 'writes_host', False)
                    and getattr(self.engine, 'writes_flags', False)):
                host_buf = self._arm_host_out(n_local)
                if host_buf is not None:
                    self.__dict__['_direct_ptr'] = self._st.__dict__['_mailbox'][1]
            direct_now = host_buf is not None
            key = None
            if use_graph:
                # capture is keyed on everything that fixes shapes and ADDRESSES (tables, filter index); table
                # VALUES may change freely.  The filter indices are kept alive with the graph (their pointers
                # are baked into it).
                key = (b_size, n_local, str(device), self.fused, overlap, both, segmented, one_graph, lo, hi, f_lo, f_hi,
                       getattr(self.model, 'l2_mode', None), getattr(self.model, 'split_filter', None),   # kernel choice is baked in
                       level_now, regions_now, getattr(self.model, 'split_level', None), self.overlap_filter, direct_now,
                       tuple(p_.data_ptr() for p_ in params), self._plan_gen, use_qmap,
                       tuple((x.data_ptr(), x.shape[0]) for ix in (index_h, index_t)
                             for x in (ix.keys, ix.offsets, ix.targets)))
                if self.graph is None and self._graph_key != key and self._graph_seen != key and key not in self._graph_cache \
                        and self._n_evaluations == 0:
                    # 'auto', the FIRST evaluation of this (model, kg): eager -- it is the warm-up (and pays the process's
                    # first-use costs once), the next one captures.  A later change of key (the other level of the split
                    # prefilter, another b_size) warms up and captures in the same call: no second eager evaluation.
                    self._graph_seen = key
                    use_graph = False
            if not use_graph:
                heads, tails, rels = facts()
                flat, out, fl = alloc_out()
                run(heads, tails, rels, out, fl)
            else:
                # the whole evaluate() as ONE hipGraph: ~20 short launches per batch replayed without host gaps
                if self._graph_key != key and key in self._graph_cache:
                    # (e.g. the other level of the split prefilter, captured earlier: switch back without a new capture)
                    if self._graph_key is not None and self._graph is not None:
                        self._graph_cache[self._graph_key] = (self._graph, self._graph_static)
                    self._graph, self._graph_static = self._graph_cache.pop(key)
                    self._graph_key, self._graph_src = key, None
                if self._graph_key != key:
                    hh_, tt_, rr_ = facts()
                    st = {'h': hh_.clone(), 't': tt_.clone(), 'r': rr_.clone(),
                          'out': alloc_out(), 'index': (index_h, index_t), 'engine': self.engine, 'plans': self._plans,
                          'qmap': self._qmap}
                    # A hipGraph must not be DESTROYED while a stream is capturing (hipErrorStreamCaptureUnsupported, and
                    # ~CUDAGraph throwing takes the process down): garbage that holds old graphs -- e.g. a previous
                    # evaluator, which sits in a reference cycle with its graph segments -- is collected now, and the
                    # cyclic collector stays off until the capture is over.
                    import gc
                    gc_was_on = gc.isenabled()
                    gc.collect()
                    gc.disable()
                    try:
                        if self.graph is not None or self._graph_seen != key:
                            side = self._aux_stream if self._aux_stream is not None else torch.cuda.Stream(device)
                            side.wait_stream(torch.cuda.current_stream(device))
                            with torch.cuda.stream(side):            # warm-up outside capture (lazy inits, attribute sets)
                                run(st['h'], st['t'], st['r'], st['out'][1], st['out'][2])
                            torch.cuda.current_stream(device).wait_stream(side)
                        if one_graph:       # collectives captured with the kernels (see graph_collectives)
                            g = torch.cuda.CUDAGraph()
                            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                                run(st['h'], st['t'], st['r'], st['out'][1], st['out'][2])
                        elif segmented:
                            g = _GraphSegments()
                            self._cut = g.cut
                            g.begin()
                            try:
                                run(st['h'], st['t'], st['r'], st['out'][1], st['out'][2])
                            except BaseException:
                                import sys
                                g.end(sys.exc_info())
                                raise
                            finally:
                                self._cut = None
                            g.end()
                        else:
                            g = torch.cuda.CUDAGraph()
                            with torch.cuda.graph(g):
                                run(st['h'], st['t'], st['r'], st['out'][1], st['out'][2])
                    except Exception as exc:
                        if self.graph is True:
                            raise
                        import warnings
                        warnings.warn('torchkge_amd: hipGraph capture of evaluate() failed (%s); running eagerly' % (exc,))
                        self.graph = False
                        self._graph_failed = True
                        self._graph = self._graph_static = self._graph_key = None
                        self._graph_cache = {}
                        torch.cuda.synchronize(device)
                        flat, out, fl = alloc_out()
                        run(*facts(), out, fl)
                        g = None
                    finally:
                        if gc_was_on:
                            gc.enable()
                    if g is not None:
                        st['zeroes_guard'] = bool(self._guard_zeroed)     # the captured run() ended with a guard-zeroing finalize
                        st['shard_flags'] = self._shard_flags             # (entity shards: where the summed flags land)
                        st['needs_clean_guard'] = guard is not None
                        st['targets_cat'] = getattr(self.engine, '_targets_cat', None)   # baked into the graph too
                        if self._graph_key is not None and self._graph is not None:
                            # keep ONE earlier capture (the other split level); older ones go -- outside any capture
                            self._graph_cache = {self._graph_key: (self._graph, self._graph_static)}
                        self._graph, self._graph_static, self._graph_key = g, st, key
                        self._graph_src = None
                if self._graph_key == key:
                    st = self._graph_static
                    src = tuple((x.data_ptr(), x._version, f_lo, f_hi) for x in (kg.head_idx, kg.tail_idx, kg.relations))
                    if src != self._graph_src:      # refresh the graph's static inputs only when the facts changed
                        hh_, tt_, rr_ = facts()
                        st['h'].copy_(hh_, non_blocking=True)
                        st['t'].copy_(tt_, non_blocking=True)
                        st['r'].copy_(rr_, non_blocking=True)
                        self._graph_src = src
                    if guard is not None and st.get('needs_clean_guard'):
                        # the graph holds no fill of the guard vector: it relies on the previous evaluation's zeroing finalize
                        if not getattr(self.model, '_lp_guard_clean', False):
                            guard.zero_()
                        object.__setattr__(self.model, '_lp_guard_clean', False)
                    self._graph.replay()
                    self._shard_flags = st.get('shard_flags')
                    if guard is not None and st.get('zeroes_guard'):
                        object.__setattr__(self.model, '_lp_guard_clean', True)
                    flat, out, fl = st['out']

            res = None
            n_pol = None
            any_redo = False
            for attempt in (0, 1):
                if guard is None:
                    break
                # the expansion was safe iff ||q||^2 + ||e||^2 stayed small; otherwise its
                # cancellation error could exceed the score tolerance -> redo on the VALU kernel
                if kdist.multi(world) and self._shard_flags is None:
                    flags = fl[:2].clone()
                    kdist.all_reduce_max(flags, self.group)     # every rank must take the same branch
                    worst, overflow = flags.tolist()
                    rescored = 0.0
                    if self.shard == 'queries' and level_ok:
                        resc = fl[2:3].clone()
                        kdist.all_reduce_sum(resc, self.group)
                        rescored = float(resc.item())
                        n_pol = kg.n_facts          # (the sum covers every rank's facts)
                else:   # one device-to-host transfer for the ranks and the flags (16 bytes = two int64)
                    if host_buf is not None:    # ... which the last finalize launch has made itself
                        torch.cuda.current_stream(device).synchronize()
                        packed = host_buf
                    else:
                        packed = _to_host(flat)
                    worst, overflow, rescored, _ = packed[-2:].view(torch.float32).tolist()
                    res = packed[:-2].view(4, n_local)
                redo = check_again = False
                if not worst <= self.model.L2_EXPAND_LIMIT:
                    self.model._expand_ok = False
                    redo = True
                elif overflow == 2.0 and attempt == 0:
                    # (r06) not the list: a DOT candidate table converted in one pass under the scale of the PREVIOUS
                    # evaluation's norm maxima has left that binade (Model._dot_fused_problem).  The query pipeline has stored
                    # the new maxima: the same path again -- a replay of the same graph -- is consistent now
                    redo = check_again = True
                elif overflow > 0 and level_now == 1 and attempt == 0 and getattr(self.model, 'split_level', 0) == 'auto':
                    # the one-product level's wider band overflowed the list: this evaluation again on the three-product
                    # sweep (whose own flags are then checked like a first run's)
                    self._level = level_now = 0
                    # (... and do not come back before the three-product count has halved -- as when LEAVING level 1: an
                    # evaluator whose level-0 count sits below LEVEL1_ENTER would otherwise enter, overflow and redo on
                    # every other evaluation)
                    self._level1_max = 0.5 * (self._level0_seen if self._level0_seen is not None else lvl_enter)
                    object.__setattr__(self.model, '_split_level', 0)
                    redo = check_again = True
                elif overflow > 0:      # more near-ties than the split prefilter's list holds: exact fp32 counts
                    self.model._split_ok = False
                    redo = True
                elif n_local > 0 and (not kdist.multi(world) or (level_ok and (self._shard_flags is not None or n_pol is not None))):
                    # level policy for the NEXT evaluation, from the pairs this one re-scored (flags[2])
                    per_q = rescored / (2.0 * (n_pol if n_pol is not None else n_local))
                    self.last_rescored_per_query = per_q
                    if level_now == 0 and rescored > 0:
                        self._level0_seen = per_q
                        cap_ = lvl_enter if self._level1_max is None else min(lvl_enter, self._level1_max)
                        if per_q <= cap_ and getattr(self.model, 'split_level', 0) == 'auto':
                            self._level = 1
                    elif level_now == 1 and per_q > lvl_leave:
                        self._level = 0
                        if self._level0_seen is not None:       # do not come back before the model has changed a lot
                            self._level1_max = 0.5 * self._level0_seen
                    if level_now == 1 and rescored > 0:
                        self._level1_seen = per_q
                if not redo:
                    break
                any_redo = True
                res = None
                flat, out, fl = alloc_out()
                run(*facts(), out, fl)
                if not check_again:
                    break
        finally:
            self._qb = None
            self.__dict__['_direct_ptr'] = None
            object.__setattr__(self.model, '_lp_side_stream', None)
            if guard is not None:      # never leave the model in guarded mode (exceptions included)
                self.model.lp_guard_end()
        if self.shard == 'queries' and kdist.multi(world):
            out = kdist.all_gather_facts(out, kg.n_facts, self.group)
        if res is None and host_buf is not None:     # (a redone evaluation: its finalize launches wrote there too)
            torch.cuda.current_stream(device).synchronize()
            res = host_buf[:-2].view(4, n_local)
        if res is None:
            res = _to_host(out)
        self.rank_true_heads, self.rank_true_tails = res[0], res[1]
        self.filt_rank_true_heads, self.filt_rank_true_tails = res[2], res[3]
        self.evaluated = True
        self._n_evaluations += 1
        # (this evaluation ran twice: guard flag, list overflow, stale table scale)
        self.__dict__['_last_redo'] = any_redo or self.__dict__.pop('_fast_flagged', False)
        # the next call may replay at once if THIS one was an undisturbed single-GPU replay of the current capture
        self._st._fast = None
        if (FAST_REPLAY and use_graph and key is not None and self._graph_key == key and not kdist.multi(world)
                and guard is not None and not any_redo and isinstance(self._graph, torch.cuda.CUDAGraph)
                and (forced_level != 'auto' or self._level == level_now) and self._regions_for(level_now, False) == regions_now
                and self._fast_sig(user_b_size) is not None):
            gst = self._graph_static
            self._st._fast = (self._fast_sig(user_b_size),
                              {'graph': self._graph, 'static': gst, 'guard': guard, 'n_local': n_local, 'level': level_now,
                               'needs_clean_guard': bool(gst.get('needs_clean_guard')), 'zeroes_guard': bool(gst.get('zeroes_guard')),
                               'direct': direct_now, 'regions': regions_now})

    def _fast_sig(self, b_size):
        """What must be unchanged for the last captured graph to be replayed without the full prologue (cheap to compute:
        identities, versions, addresses)."""
        kg, m, st = self.kg, self.model, self._st
        h, t, r = kg.head_idx, kg.tail_idx, kg.relations
        try:
            tabs = m._tables()
        except Exception:
            return None
        return (b_size, id(kg), id(h), h._version, id(t), t._version, id(r), r._version, kg.n_facts,
                tuple(p.data_ptr() for p in tabs), getattr(m, 'l2_mode', None), getattr(m, 'split_filter', None),
                getattr(m, 'split_level', None), st._level, st._level1_max, self.graph, id(getattr(m, '_lp_guard', None)),
                id(st._graph), st._plan_gen, id(getattr(kg, '_lazy', None)), self.coalesce, st._mem_fit)

    def _regions_for(self, level, multi):
        """The sweep's uncertain pairs in regions of 32 queries?  From REGION_MIN_LEVEL0 re-scored pairs per query seen on the
        three-product level -- and, once the one-product level has been measured, from REGION_MIN_LEVEL1 there."""
        r = bool(level == 1 and not multi and self._level0_seen is not None and self._level0_seen >= REGION_MIN_LEVEL0)
        if r and self._level1_seen is not None and self._level1_seen < REGION_MIN_LEVEL1:
            r = False       # (too few pairs to amortise a region's query rows)
        return r

    def _arm_host_out(self, n_local):
        """Direct host results (r06): a fresh pinned (4 n + 2) int64 buffer whose device-visible address goes into the
        state's mailbox -- the finalize launches of the coming run / replay write there.  None: not available."""
        st = self._st
        mb = st.__dict__.get('_mailbox')
        try:
            if mb is None:
                box = torch.zeros(1, dtype=torch.int64, pin_memory=True)
                dev = _hip.host_device_pointer(box)
                mb = st.__dict__['_mailbox'] = (box, dev, ctypes.c_int64.from_address(box.data_ptr()))
            if mb[1] is None:
                return None
            host = torch.empty(4 * n_local + 2, dtype=torch.int64, pin_memory=True)
        except RuntimeError:        # (no pinned memory to be had: the copy path needs none of this)
            st.__dict__['_mailbox'] = (None, None, None)
            return None
        dev = _hip.host_device_pointer(host)
        if dev is None:
            return None
        mb[2].value = dev
        return host

    def _evaluate_fast(self, info):
        """One steady-state evaluation: guard hygiene, graph replay, ONE device-to-host copy (ranks + flags), the level
        policy.  False: something needs the full path (nothing has been changed that it would not redo)."""
        m = self.model
        guard, n_local = info['guard'], info['n_local']
        if info['needs_clean_guard']:
            if not getattr(m, '_lp_guard_clean', False):
                guard.zero_()
            object.__setattr__(m, '_lp_guard_clean', False)
        host = None
        if info.get('direct'):
            host = self._arm_host_out(n_local)
            if host is None:
                return False
        info['graph'].replay()
        if info['zeroes_guard']:
            object.__setattr__(m, '_lp_guard_clean', True)
        if host is not None:        # the last finalize of the graph wrote ranks and flags there
            torch.cuda.current_stream(info['static']['out'][0].device).synchronize()
            packed = host
        else:
            packed = _to_host(info['static']['out'][0])
        # (the four flag floats straight from the pinned buffer: slicing + view + tolist cost 4 us of GPU idle time per call)
        worst, overflow, rescored, _ = _FLAGS4.unpack(ctypes.string_at(packed.data_ptr() + 8 * (packed.numel() - 2), 16))
        if not worst <= m.L2_EXPAND_LIMIT or overflow > 0:
            # (norm guard / list overflow: the full path replays, sees the same flags and redoes; a stale table scale -- flag
            # value 2 -- is gone in that replay: the query pipeline has stored the new maxima)
            self.__dict__['_fast_flagged'] = True
            return False
        level_now = info['level']
        if n_local > 0:
            lvl_enter, lvl_leave = level1_thresholds(m.n_ent)
            per_q = rescored / (2.0 * n_local)
            self.last_rescored_per_query = per_q
            if level_now == 0 and rescored > 0:
                self._level0_seen = per_q
                cap_ = lvl_enter if self._level1_max is None else min(lvl_enter, self._level1_max)
                if per_q <= cap_ and getattr(m, 'split_level', 0) == 'auto':
                    self._level = 1
            elif level_now == 1 and per_q > lvl_leave:
                self._level = 0
                if self._level0_seen is not None:
                    self._level1_max = 0.5 * self._level0_seen
            if level_now == 1 and rescored > 0:
                self._level1_seen = per_q
                if self._regions_for(1, False) != info.get('regions'):
                    self._st._fast = None       # (the next call decides about the regions again: full path, another capture)
        # (the four rank vectors are rows of this host tensor, handed out on access -- _rank_row: four view objects built
        # here cost 5 us between two replays)
        self.__dict__['_rank_rows'] = packed.as_strided((4, n_local), (n_local, 1))
        self.__dict__['_rank_set'] = {}
        self.__dict__['_last_redo'] = False
        self.evaluated = True
        self._n_evaluations += 1
        return True

    # -- metrics (evaluation.py:310-425) --------------------------------------
    def _check(self):
        if not self.evaluated:
            raise NotYetEvaluatedError('Evaluator not evaluated call LinkPredictionEvaluator.evaluate')

    def mean_rank(self):
        """(mean rank, filtered mean rank), each the mean of the head-side and
        tail-side means (evaluation.py:310-331)."""
        self._check()
        sum_ = (self.rank_true_heads.float().mean() + self.rank_true_tails.float().mean()).item()
        filt_sum = (self.filt_rank_true_heads.float().mean() +
                    self.filt_rank_true_tails.float().mean()).item()
        return sum_ / 2, filt_sum / 2

    def hit_at_k_heads(self, k=10):
        self._check()
        return ((self.rank_true_heads <= k).float().mean().item(),
                (self.filt_rank_true_heads <= k).float().mean().item())

    def hit_at_k_tails(self, k=10):
        self._check()
        return ((self.rank_true_tails <= k).float().mean().item(),
                (self.filt_rank_true_tails <= k).float().mean().item())

    def hit_at_k(self, k=10):
        """(Hit@k, filtered Hit@k) averaged over head and tail replacement
        (evaluation.py:352-374)."""
        self._check()
        head_hit, filt_head_hit = self.hit_at_k_heads(k=k)
        tail_hit, filt_tail_hit = self.hit_at_k_tails(k=k)
        return (head_hit + tail_hit) / 2, (filt_head_hit + filt_tail_hit) / 2

    def mrr(self):
        """(MRR, filtered MRR) averaged over head and tail replacement
        (evaluation.py:376-397)."""
        self._check()
        head_mrr = (self.rank_true_heads.float() ** (-1)).mean()
        tail_mrr = (self.rank_true_tails.float() ** (-1)).mean()
        filt_head_mrr = (self.filt_rank_true_heads.float() ** (-1)).mean()
        filt_tail_mrr = (self.filt_rank_true_tails.float() ** (-1)).mean()
        return ((head_mrr + tail_mrr).item() / 2, (filt_head_mrr + filt_tail_mrr).item() / 2)

    def print_results(self, k=None, n_digits=3):
        """Same report as the reference (evaluation.py:399-425)."""
        if k is None:
            k = 10
        ks = [k] if type(k) == int else list(k)
        for i in ks:
            print('Hit@{} : {} \t\t Filt. Hit@{} : {}'.format(
                i, round(self.hit_at_k(k=i)[0], n_digits), i, round(self.hit_at_k(k=i)[1], n_digits)))
        print('Mean Rank : {} \t Filt. Mean Rank : {}'.format(
            int(self.mean_rank()[0]), int(self.mean_rank()[1])))
        print('MRR : {} \t\t Filt. MRR : {}'.format(
            round(self.mrr()[0], n_digits), round(self.mrr()[1], n_digits)))


def _rank_row(i, name):
    """rank_true_heads / rank_true_tails / filt_rank_true_heads / filt_rank_true_tails: a tensor assigned to the attribute, or
    row i of the (4, n) host tensor the last steady-state evaluation left."""
    def get(self):
        d = self.__dict__
        v = d.get('_rank_set', {}).get(i)
        if v is None and d.get('_rank_rows') is not None:
            v = d['_rank_set'][i] = d['_rank_rows'][i]
        return v

    def set_(self, v):
        d = self.__dict__
        d.setdefault('_rank_set', {})[i] = v
    return property(get, set_, doc=name)


for _i, _n in enumerate(('rank_true_heads', 'rank_true_tails', 'filt_rank_true_heads', 'filt_rank_true_tails')):
    setattr(LinkPredictionEvaluator, _n, _rank_row(_i, _n))


def _forward(name):
    return property(lambda self: getattr(self._st, name), lambda self, v: setattr(self._st, name, v))


for _n in _EvalState.SHARED:        # evaluator attributes that live in the (possibly shared) _EvalState
    setattr(LinkPredictionEvaluator, _n, _forward(_n))


class _NullCtx(object):
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class RelationPredictionEvaluator(object):
    """Rank the true relation of every fact among all relations, raw and
    filtered, with the reference's interface (torchkge/evaluation.py:16-204):
    ``RelationPredictionEvaluator(model, kg, directed=True)``, ``.evaluate``,
    ``.mean_rank / .hit_at_k / .mrr / .print_results``, ``rank_true_rels`` and
    ``filt_rank_true_rels``.  Candidates are the relation table (at most a few
    thousand rows), so the drop-in composition prepare -> score -> filter_scores
    -> get_rank is used as is, every step on the HIP kernels."""

    def __init__(self, model, knowledge_graph, directed=True):
        self.model = model
        self.kg = knowledge_graph
        self.directed = directed
        self.rank_true_rels = torch.empty(size=(knowledge_graph.n_facts,)).long()
        self.filt_rank_true_rels = torch.empty(size=(knowledge_graph.n_facts,)).long()
        self.evaluated = False

    def evaluate(self, b_size, verbose=True):
        device = next(self.model.parameters()).device
        HipRankEngine.check_device(device)
        kg = self.kg
        heads, tails, rels = kg.head_idx.to(device), kg.tail_idx.to(device), kg.relations.to(device)
        index = filter_index_for(kg.dict_of_rels, device)
        ranks = torch.empty(2, kg.n_facts, dtype=torch.int64, device=device)
        n_batches = get_n_batches(kg.n_facts, b_size)
        with torch.no_grad():
            for i in tqdm(range(n_batches), total=n_batches, unit='batch', disable=(not verbose),
                          desc='Relation prediction evaluation'):
                sl = slice(i * b_size, (i + 1) * b_size)
                h_idx, t_idx, r_idx = heads[sl], tails[sl], rels[sl]
                h_emb, t_emb, r_emb, candidates = self.model.inference_prepare_candidates(
                    h_idx, t_idx, r_idx, entities=False)
                scores = self.model.inference_scoring_function(h_emb, t_emb, candidates)
                filt_scores = filter_scores(scores, index, h_idx, t_idx, r_idx)
                if not self.directed:      # evaluation.py:99-104: also score (t, _, h)
                    scores_bis = self.model.inference_scoring_function(t_emb, h_emb, candidates)
                    filt_scores_bis = filter_scores(scores_bis, index, h_idx, t_idx, r_idx)
                    scores = torch.cat((scores, scores_bis), dim=1)
                    filt_scores = torch.cat((filt_scores, filt_scores_bis), dim=1)
                ranks[0, sl] = get_rank(scores, r_idx)
                ranks[1, sl] = get_rank(filt_scores, r_idx)
        res = ranks.cpu()
        self.rank_true_rels, self.filt_rank_true_rels = res[0], res[1]
        self.evaluated = True

    def _check(self):
        if not self.evaluated:
            raise NotYetEvaluatedError('Evaluator not evaluated call LinkPredictionEvaluator.evaluate')

    def mean_rank(self):
        self._check()
        return self.rank_true_rels.float().mean().item(), self.filt_rank_true_rels.float().mean().item()

    def hit_at_k(self, k=10):
        self._check()
        return ((self.rank_true_rels <= k).float().mean().item(),
                (self.filt_rank_true_rels <= k).float().mean().item())

    def mrr(self):
        self._check()
        return ((self.rank_true_rels.float() ** (-1)).mean().item(),
                (self.filt_rank_true_rels.float() ** (-1)).mean().item())

    def print_results(self, k=None, n_digits=3):
        if k is None:
            k = 10
        for i in ([k] if type(k) == int else list(k)):
            print('Hit@{} : {} \t\t Filt. Hit@{} : {}'.format(
                i, round(self.hit_at_k(k=i)[0], n_digits), i, round(self.hit_at_k(k=i)[1], n_digits)))
        print('Mean Rank : {} \t Filt. Mean Rank : {}'.format(int(self.mean_rank()[0]), int(self.mean_rank()[1])))
        print('MRR : {} \t\t Filt. MRR : {}'.format(round(self.mrr()[0], n_digits), round(self.mrr()[1], n_digits)))
