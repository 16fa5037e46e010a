# -*- coding: utf-8 -*-
"""Top-k inference of missing entities / relations with the reference's
interface (torchkge/inference.py:77-250): ``EntityInference(model,
known_entities, known_relations, top_k=1, missing='tails', dictionary=None)``
and ``RelationInference(model, entities1, entities2, top_k=1, dictionary=None)``,
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


class RelationInference(object):
    """Infer the top_k most plausible relations between entity pairs
    (inference.py:77-153)."""

    def __init__(self, model, entities1, entities2, top_k=1, dictionary=None):
        self.model = model
        self.entities1 = entities1
        self.entities2 = entities2
        self.topk = top_k
        self.dictionary = dictionary
        self.predictions = torch.empty(size=(len(entities1), top_k)).long()
        self.scores = torch.empty(size=(len(entities2), top_k))

    def evaluate(self, b_size, verbose=True):
        dev = _device_of(self.model)
        e1, e2 = self.entities1.to(dev), self.entities2.to(dev)
        none = torch.zeros(0, dtype=torch.long, device=dev)
        preds, vals = [], []
        n_batches = get_n_batches(len(e1), b_size)
        with torch.no_grad():
            for i in tqdm(range(n_batches), total=n_batches, unit='batch', disable=(not verbose), desc='Inference'):
                sl = slice(i * b_size, (i + 1) * b_size)
                ents1, ents2 = e1[sl], e2[sl]
                h_emb, t_emb, _, candidates = self.model.inference_prepare_candidates(ents1, ents2, none,
                                                                                      entities=False)
                scores = self.model.inference_scoring_function(h_emb, t_emb, candidates)
                if self.dictionary is not None:
                    scores = filter_scores(scores, self.dictionary, ents1, ents2, None)
                v, ix = _hip.topk(scores, min(self.topk, scores.shape[1]))
                preds.append(ix)
                vals.append(v)
        self.predictions = torch.cat(preds).cpu() if preds else self.predictions
        self.scores = torch.cat(vals).cpu() if vals else self.scores


class EntityInference(object):
    """Infer the top_k most plausible missing heads or tails
    (inference.py:156-250)."""

    def __init__(self, model, known_entities, known_relations, top_k=1, missing='tails', dictionary=None):
        if missing not in ['heads', 'tails']:
            raise WrongArgumentsError("missing entity should either be 'heads' or 'tails'")
        self.model = model
        self.known_entities = known_entities
        self.known_relations = known_relations
        self.missing = missing
        self.top_k = top_k
        self.dictionary = dictionary
        self.predictions = torch.empty(size=(len(known_entities), top_k)).long()
        self.scores = torch.empty(size=(len(known_entities), top_k))

    def evaluate(self, b_size, verbose=True):
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
