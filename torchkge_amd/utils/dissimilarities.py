# -*- coding: utf-8 -*-
"""l1/l2 dissimilarities of the reference (utils/dissimilarities.py:11-25) on
the HIP row kernels.  ``l2_dissimilarity`` is the SQUARED L2 norm.  Both are
differentiable like the reference's torch expressions (user-defined models call
``model.dissimilarity`` inside their own scoring functions): the forward value
comes from the HIP kernel, the backward is the closed form
d/da ||a-b||_1 = sign(a-b), d/da ||a-b||_2^2 = 2 (a-b)."""
import torch

from .. import _hip


def _rowwise(a, b, mode):
    a, b = torch.broadcast_tensors(a, b)
    shape = a.shape[:-1]
    K = a.shape[-1]
    q = a.reshape(-1, K)
    c = b.reshape(-1, 1, K)
    out = _hip.lp_scores_batched(mode, q, c)     # one candidate per row
    return (-out).reshape(shape)


class _Dissimilarity(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, mode):
        ctx.mode = mode
        ctx.save_for_backward(a, b)
        return _rowwise(a.detach(), b.detach(), mode)

    @staticmethod
    def backward(ctx, grad_out):
        a, b = ctx.saved_tensors
        diff = a - b                                   # broadcast shape
        g = torch.sign(diff) if ctx.mode == _hip.LP_L1_DIRECT else 2.0 * diff
        g = g * grad_out.unsqueeze(-1)
        ga = g.sum_to_size(a.shape) if ctx.needs_input_grad[0] else None
        gb = (-g).sum_to_size(b.shape) if ctx.needs_input_grad[1] else None
        return ga, gb, None


def _diss(a, b, mode):
    assert len(a.shape) == len(b.shape)
    _hip.require_cuda(a, b)
    if torch.is_grad_enabled() and (a.requires_grad or b.requires_grad):
        return _Dissimilarity.apply(a, b, mode)
    return _rowwise(a, b, mode)


def l1_dissimilarity(a, b):
    """||a - b||_1 along the last dim (dissimilarities.py:11-16)."""
    return _diss(a, b, _hip.LP_L1_DIRECT)


def l2_dissimilarity(a, b):
    """||a - b||_2^2 along the last dim (dissimilarities.py:19-25)."""
    return _diss(a, b, _hip.LP_L2_DIRECT)
