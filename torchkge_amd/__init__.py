# -*- coding: utf-8 -*-
"""torchkge_amd: MI355X-native engine for the torchkge scoring / link-prediction
hot path, behind torchkge's own Python interfaces.  See DESIGN.md."""
__version__ = '0.1.0'

from .exceptions import NotYetEvaluatedError
from .utils import MarginLoss, LogisticLoss
from .utils import l1_dissimilarity, l2_dissimilarity
from .data_structures import KnowledgeGraph, SmallKG
from .evaluation import LinkPredictionEvaluator, RelationPredictionEvaluator, TripletClassificationEvaluator
from .inference import EntityInference, RelationInference
from .models import TransEModel, TransHModel, TransDModel, DistMultModel, ComplExModel
from .sampling import BernoulliNegativeSampler, UniformNegativeSampler, PositionalNegativeSampler
