# -*- coding: utf-8 -*-
"""torchkge_amd: MI355X-native engine for the torchkge scoring / link-prediction
hot path, behind torchkge's own Python interfaces.  See DESIGN.md."""
__version__ = '0.1.0'

from .exceptions import NotYetEvaluatedError, NotProvidedError
from .utils import MarginLoss, LogisticLoss
from .utils import l1_dissimilarity, l2_dissimilarity
from .data_structures import KnowledgeGraph, SmallKG
from .evaluation import LinkPredictionEvaluator, RelationPredictionEvaluator, clear_eval_state
from .inference import EntityInference
from .models import TransEModel, TransHModel, TransDModel, DistMultModel, ComplExModel
from .sampling import BernoulliNegativeSampler, UniformNegativeSampler


# Names of the reference that are OUT OF SCOPE of this engine: a clear
# error instead of an AttributeError, so a ported script says what to do.
_NOT_PROVIDED = {
    'TripletClassificationEvaluator': 'triplet classification is outside the link-prediction hot path',
    'PositionalNegativeSampler': 'use BernoulliNegativeSampler / UniformNegativeSampler (the samplers on the hot path)',
    'RelationInference': 'use RelationPredictionEvaluator, or EntityInference for missing entities',
}


def __getattr__(name):
    if name in _NOT_PROVIDED:
        raise NotProvidedError('torchkge_amd does not provide torchkge.%s: %s (see INTEGRATION.md)' % (name, _NOT_PROVIDED[name]))
    raise AttributeError('module %r has no attribute %r' % (__name__, name))
