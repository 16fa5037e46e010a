# -*- coding: utf-8 -*-
"""ctypes binding of libkge_hip_coll.so (include/kge_hip_coll.h): the RCCL exchange step of the entity-sharded
link-prediction path for hosts that do NOT go through torch.distributed (the Python evaluator does:
torchkge_amd/distributed.py).  Used by INTEGRATION.md's recipe and by tests/test_gpu_collectives.py; loaded on demand
so that importing torchkge_amd never needs librccl."""
import ctypes
import os

import torch

from . import _hip

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc', 'libkge_hip_coll.so')
UNIQUE_ID_BYTES = 128
_vp, _i64, _int = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
_SIGNATURES = {
    'kge_comm_unique_id': [_vp],
    'kge_comm_init': [_vp, _int, _int, _vp],
    'kge_comm_destroy': [_vp],
    'kge_allgather_scores': [_vp, _int, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp],
    'kge_alltoall_scores': [_vp, _int, _int, _vp, _vp, _i64, _i64, _int, _vp],
    'kge_allreduce_ranks': [_vp, _vp, _i64, _vp],
    'kge_allreduce_counts': [_vp, _vp, _i64, _vp],
    'kge_allreduce_sum_f32': [_vp, _vp, _i64, _vp],
}
_lib = None


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('torchkge_amd: %s not found -- build it with `python -m torchkge_amd.csrc.build`' % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, args in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = args, _int
        _lib = lib
    return _lib


def _check(rc, name):
    if rc != 0:
        raise RuntimeError('torchkge_amd: %s failed (%s)' % (name, 'ncclResult_t %d' % (rc - 1000) if rc >= 1000 else
                                                         ('bad arguments' if rc < 0 else 'hipError_t %d' % rc)))


def unique_id():
    buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
    _check(load_library().kge_comm_unique_id(buf), 'kge_comm_unique_id')
    return buf.raw


class Comm(object):
    """One RCCL communicator (one per GPU / process)."""

    def __init__(self, world, rank, uid):
        self.world, self.rank = int(world), int(rank)
        self._h = ctypes.c_void_p()
        _check(load_library().kge_comm_init(ctypes.byref(self._h), self.world, self.rank, ctypes.c_char_p(uid)),
               'kge_comm_init')

    def close(self):
        if self._h:
            load_library().kge_comm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def allgather_scores(self, local, n_total):
        """(B, n_per) partial score tile of this rank -> (B, n_total) on every rank.  n_per must be
        ceil(n_total / world) on EVERY rank (the last shard zero-padded by the caller): ncclAllGather takes one count."""
        _hip.require_cuda(local)
        local = _hip.f32c(local)
        B, per = local.shape
        need = -(-int(n_total) // self.world)
        if per < need:      # a short last shard: zero-pad it here (ncclAllGather takes ONE count for every rank; a caller
            padded = local.new_zeros(B, need)      # that pads further must pad to the same width on every rank)
            padded[:, :per] = local
            local, per = padded, need
        gathered = local.new_empty(self.world, B, per)
        full = local.new_empty(B, n_total)
        _check(load_library().kge_allgather_scores(self._h, self.world, _hip._p(local), _hip._p(gathered), _hip._p(full),
                                                   n_total, B, per, n_total, _hip._stream()), 'kge_allgather_scores')
        return full

    def alltoall_scores(self, local, recv=None, recv_own=True):
        """(world * m, n_per) local score tile -> (world, m, n_per) rank-major tiles of THIS rank's m queries
        (recv_own=False: block `rank` of the result is left unwritten -- rank it from local[rank * m:] instead)."""
        _hip.require_cuda(local)
        assert local.dtype == torch.float32 and local.is_contiguous() and local.shape[0] % self.world == 0
        m, per = local.shape[0] // self.world, local.shape[1]
        if recv is None:
            recv = local.new_empty(self.world, m, per)
        _check(load_library().kge_alltoall_scores(self._h, self.world, self.rank, _hip._p(local), _hip._p(recv), m, per,
                                                  1 if recv_own else 0, _hip._stream()), 'kge_alltoall_scores')
        return recv

    def allreduce_ranks(self, ranks):
        _hip.require_cuda(ranks)
        assert ranks.dtype == torch.int64 and ranks.is_contiguous()
        _check(load_library().kge_allreduce_ranks(self._h, _hip._p(ranks), ranks.numel(), _hip._stream()),
               'kge_allreduce_ranks')
        return ranks

    def allreduce_counts(self, counts):
        _hip.require_cuda(counts)
        assert counts.dtype == torch.int32 and counts.is_contiguous()
        _check(load_library().kge_allreduce_counts(self._h, _hip._p(counts), counts.numel(), _hip._stream()),
               'kge_allreduce_counts')
        return counts

    def allreduce_sum(self, x):
        _hip.require_cuda(x)
        assert x.dtype == torch.float32 and x.is_contiguous()
        _check(load_library().kge_allreduce_sum_f32(self._h, _hip._p(x), x.numel(), _hip._stream()),
               'kge_allreduce_sum_f32')
        return x
