# -*- coding: utf-8 -*-
from ..exceptions import NotProvidedError
from .data import DataLoader, get_n_batches
from .dissimilarities import l1_dissimilarity, l2_dissimilarity
from .losses import MarginLoss, LogisticLoss, BinaryCrossEntropyLoss
from .modeling import init_embedding, get_true_targets, filter_scores
from .operations import get_rank, get_mask, get_bernoulli_probs, get_tph, get_hpt, get_dictionaries


def __getattr__(name):
    if name in ('Trainer', 'TrainDataLoader'):
        raise NotProvidedError('torchkge_amd.utils does not provide %s (torchkge/utils/training.py is host glue outside the hot '
                          'path): write the tutorial loop around Model.forward -- sampler.corrupt_batch, model(h, t, r, nh, '
                          'nt), criterion, backward, optimizer.step -- see tests/test_reference_style.py and INTEGRATION.md'
                          % name)
    raise AttributeError('module %r has no attribute %r' % (__name__, name))
