# -*- coding: utf-8 -*-
from .interfaces import Model, TranslationModel, BilinearModel, EntityCandidates
from .translation import TransEModel, TransHModel, TransDModel
from .bilinear import DistMultModel, ComplExModel
