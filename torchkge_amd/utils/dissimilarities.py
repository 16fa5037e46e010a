# -*- coding: utf-8 -*-
"""l1/l2 dissimilarities of the reference (utils/dissimilarities.py:11-25) on
the HIP row kernels.  ``l2_dissimilarity`` is the SQUARED L2 norm."""
import torch

from .. import _hip


def _rowwise(a, b, mode):
    assert len(a.shape) == len(b.shape)
    _hip.require_cuda(a, b)
    a, b = torch.broadcast_tensors(a, b)
    shape = a.shape[:-1]
    K = a.shape[-1]
    q = a.reshape(-1, K)
    c = b.reshape(-1, 1, K)
    out = _hip.lp_scores_batched(mode, q, c)     # one candidate per row
    return (-out).reshape(shape)


def l1_dissimilarity(a, b):
    """||a - b||_1 along the last dim (dissimilarities.py:11-16)."""
    return _rowwise(a, b, _hip.LP_L1_DIRECT)


def l2_dissimilarity(a, b):
    """||a - b||_2^2 along the last dim (dissimilarities.py:19-25)."""
    return _rowwise(a, b, _hip.LP_L2_DIRECT)
