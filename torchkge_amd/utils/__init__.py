# -*- coding: utf-8 -*-
from .data import DataLoader, get_n_batches
from .dissimilarities import l1_dissimilarity, l2_dissimilarity
from .losses import MarginLoss, LogisticLoss, BinaryCrossEntropyLoss
from .modeling import init_embedding, get_true_targets, filter_scores
from .operations import get_rank, get_mask, get_bernoulli_probs, get_tph, get_hpt, get_dictionaries
