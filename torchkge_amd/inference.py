# -*- coding: utf-8 -*-
"""Top-k inference of missing entities with the reference's
interface (torchkge/inference.py:156-250): ``EntityInference(model,
known_entities, known_relations, top_k=1, missing='tails', dictionary=None)``,
``.evaluate(b_size, verbose)`` filling ``.predictions`` (n, top_k) long and
``.scores`` (n, top_k) float on the CPU.

Same score kernels as link prediction; the reference's full
``scores.sort(descending=True)`` + slice (:148-150, :243-245) is a top-k kernel
(kge_topk, order: score descending, index ascending).  Two reference bugs are
not reproduced: scores are stored by slice (the reference indexes
``self.scores[i*b, (i+1)*b]`` with a tuple), and ``missing='heads'`` works (the
reference derives the batch size from the empty head index).
"""
import torch
from tqdm.autonotebook import tqdm

from . import _hip
from .exceptions import WrongArgumentsError
from .utils.data import get_n_batches
from .utils.modeling import filter_scores


class DataLoader_:
    """Sequential batches over two aligned index vectors (inference.py:14-75)."""

    def __init__(self, a, b, batch_size, use_cuda=None):
        self.a, self.b, self.batch_size, self.use_cuda = a, b, batch_size, use_cuda
        if use_cuda is not None and use_cuda == 'all':
            self.a, self.b = self.a.cuda(), self.b.cuda()

    def __len__(self):
        return get_n_batches(len(self.a), self.batch_size)

    def __iter__(self):
        for i in range(len(self)):
            sl = slice(i * self.batch_size, (i + 1) * self.batch_size)
            if self.use_cuda is not None and self.use_cuda == 'batch':
                yield self.a[sl].cuda(), self.b[sl].cuda()
            else:
                yield self.a[sl], self.b[sl]


def _device_of(model):
    dev = next(model.parameters()).device
    if dev.type != 'cuda':
        raise RuntimeError('torchkge_amd inference runs on MI355X (HIP) only: move the model to `cuda`.')
    return dev


class EntityInference(object):
    """Infer the top_k most plausible missing heads or tails (inference.py:156-250).

    The reference scores every entity, sorts the (b, N) matrix and keeps k columns (:243-245).  Here the candidates
    are processed TILE BY TILE: a (b, C) tile of scores is written into a scratch buffer that is reused
    for every tile, ``kge_topk_chunk`` masks the known targets that fall into the tile (``dictionary``) and keeps the
    tile's k best as (score, global id); the partial lists are merged by the same kernel.  Memory is O(b * C + b * k *
    N / C) instead of O(b * N), and a ROW-SHARDED model (distributed.shard_model_) runs the same code on its own rows:
    per-shard top-k, one all-gather of the P x (b, k) partial lists, merge -- identical predictions on every rank.
    Order: score descending, entity id ascending (the reference's sort leaves ties unspecified).

    Extra keywords: ``tile`` (candidates per tile, default: so that the scratch stays below 256 MB), ``group``
    (torch.distributed process group of a row-sharded model)."""

    def __init__(self, model, known_entities, known_relations, top_k=1, missing='tails', dictionary=None, tile=None,
                 group=None):
        if missing not in ['heads', 'tails']:
            raise WrongArgumentsError("missing entity should either be 'heads' or 'tails'")
        self.model = model
        self.known_entities = known_entities
        self.known_relations = known_relations
        self.missing = missing
        self.top_k = top_k
        self.dictionary = dictionary
        self.tile, self.group = tile, group
        self.predictions = torch.empty(size=(len(known_entities), top_k)).long()
        self.scores = torch.empty(size=(len(known_entities), top_k))

    def _tile(self, b, n_local):
        if self.tile is not None:
            c = int(self.tile)
        else:
            c = max(4096, (256 << 20) // (4 * max(b, 1)))
        c = max(256, (c // 256) * 256)          # tile starts stay 16-byte aligned for every kernel's vector loads
        return min(c, max(n_local, 1))

    def evaluate(self, b_size, verbose=True):
        from . import distributed as kdist
        from .filter_index import filter_index_for
        dev = _device_of(self.model)
        model = self.model
        impl = getattr(type(model), 'lp_problem', None)
        from .models.interfaces import Model as _BaseModel
        if impl is None or impl is _BaseModel.lp_problem:
            return self._evaluate_materialised(b_size, verbose)      # user-defined model: the reference composition
        ents, rels = self.known_entities.to(dev), self.known_relations.to(dev)
        side = 'head' if self.missing == 'heads' else 'tail'
        row_shard = getattr(model, '_row_shard', None)
        world, rank = kdist.world_and_rank(self.group) if row_shard is not None else (1, 0)
        sharded = row_shard is not None and kdist.multi(world)
        if row_shard is not None and not sharded:
            raise RuntimeError('torchkge_amd: a row-sharded model needs its process group for inference')
        lo, hi = row_shard if sharded else (0, model.n_ent)
        n_local = hi - lo
        k = min(self.top_k, model.n_ent)
        index = None
        if self.dictionary is not None:
            index = self.dictionary if hasattr(self.dictionary, 'lookup') else filter_index_for(self.dictionary, dev)
        xkw = {}
        if sharded:   # owner-built query rows summed over the ranks (x + 0 is exact), as the evaluator's 'batch' exchange
            xkw['exchange'] = lambda tensors: [kdist.all_reduce_sum(x, self.group) for x in tensors]
        preds, vals = [], []
        n_batches = get_n_batches(len(ents), b_size)
        session = model.lp_session() if hasattr(model, 'lp_session') else None
        with torch.no_grad():
            if session is not None:
                session.__enter__()
            try:
                for i in tqdm(range(n_batches), total=n_batches, unit='batch', disable=(not verbose), desc='Inference'):
                    sl = slice(i * b_size, (i + 1) * b_size)
                    e, r = ents[sl], rels[sl]
                    b = e.shape[0]
                    prob = model.lp_problem(e, e, r, side, ent_lo=lo, ent_hi=hi, **xkw)
                    seg_lo = seg_hi = targets = None
                    if index is not None:
                        seg_lo, seg_hi = index.lookup(e, r)
                        targets = index.targets
                    C = self._tile(b, n_local)
                    n_tiles = (n_local + C - 1) // C
                    tile = torch.empty(b, C, dtype=torch.float32, device=dev)
                    pv = torch.empty(b, n_tiles * k, dtype=torch.float32, device=dev)
                    pi = torch.empty(b, n_tiles * k, dtype=torch.int64, device=dev)
                    for ti in range(n_tiles):
                        c0, c1 = ti * C, min(n_local, (ti + 1) * C)
                        view = tile[:, :c1 - c0]
                        prob.scores_chunk(c0, c1, view)
                        _hip.topk_chunk(view, lo + c0, k, pv, pi, ti * k, seg_lo, seg_hi, targets)
                    if sharded:     # the P per-shard lists of every row, rank-major = id-ascending shard order
                        pv, pi = _gather_partials(pv, pi, k, n_local, model.n_ent, world, self.group)
                    if pv.shape[1] == k and not sharded:
                        v, ix = pv, pi
                    else:
                        v = torch.empty(b, k, dtype=torch.float32, device=dev)
                        ix = torch.empty(b, k, dtype=torch.int64, device=dev)
                        _hip.topk_chunk(pv, 0, k, v, ix, 0, ids_in=pi)
                    preds.append(ix)
                    vals.append(v)
            finally:
                if session is not None:
                    session.__exit__(None, None, None)
        self.predictions = torch.cat(preds).cpu() if preds else self.predictions
        self.scores = torch.cat(vals).cpu() if vals else self.scores

    def _evaluate_materialised(self, b_size, verbose=True):
        """The reference composition on HIP ops: prepare -> score (b, N) -> filter -> top-k."""
        dev = _device_of(self.model)
        ents, rels = self.known_entities.to(dev), self.known_relations.to(dev)
        none = torch.zeros(0, dtype=torch.long, device=dev)
        preds, vals = [], []
        n_batches = get_n_batches(len(ents), b_size)
        with torch.no_grad():
            for i in tqdm(range(n_batches), total=n_batches, unit='batch', disable=(not verbose), desc='Inference'):
                sl = slice(i * b_size, (i + 1) * b_size)
                known_ents, known_rels = ents[sl], rels[sl]
                if self.missing == 'heads':
                    _, t_emb, rel_emb, candidates = self.model.inference_prepare_candidates(
                        none, known_ents, known_rels, entities=True)
                    scores = self.model.inference_scoring_function(candidates, t_emb, rel_emb)
                else:
                    h_emb, _, rel_emb, candidates = self.model.inference_prepare_candidates(
                        known_ents, none, known_rels, entities=True)
                    scores = self.model.inference_scoring_function(h_emb, candidates, rel_emb)
                if self.dictionary is not None:
                    scores = filter_scores(scores, self.dictionary, known_ents, known_rels, None)
                v, ix = _hip.topk(scores, min(self.top_k, scores.shape[1]))
                preds.append(ix)
                vals.append(v)
        self.predictions = torch.cat(preds).cpu() if preds else self.predictions
        self.scores = torch.cat(vals).cpu() if vals else self.scores


def _gather_partials(pv, pi, k, n_local, n_ent, world, group):
    """Per-shard partial lists (b, m_p) of every rank -> (b, P * m) with the shards in rank (= id) order.  Shards
    may hold different numbers of tiles (uneven last shard): lists are padded to the longest with (-inf, -1)."""
    import torch.distributed as dist
    from . import distributed as kdist
    b, m = pv.shape
    per = kdist.shard_size(n_ent, world)
    m_max = m
    if True:      # every rank needs the same bound: the tiles of the largest shard
        mm = torch.tensor([m], device=pv.device, dtype=torch.int64)
        kdist.all_reduce_max(mm, group)
        m_max = int(mm.item())
    if m_max != m:
        pv = torch.cat([pv, pv.new_full((b, m_max - m), float('-inf'))], 1)
        pi = torch.cat([pi, pi.new_full((b, m_max - m), -1)], 1)
    gv = pv.new_empty(world, b, m_max)
    gi = pi.new_empty(world, b, m_max)
    dist.all_gather_into_tensor(gv.view(world * b, m_max), pv.contiguous(), group=group)
    dist.all_gather_into_tensor(gi.view(world * b, m_max), pi.contiguous(), group=group)
    return (gv.permute(1, 0, 2).reshape(b, world * m_max).contiguous(),
            gi.permute(1, 0, 2).reshape(b, world * m_max).contiguous())
