# -*- coding: utf-8 -*-
"""ctypes binding of libkge_hip.so (include/kge_hip.h) and thin tensor-level
wrappers.  PyTorch is plumbing here: it owns device memory and the stream; all
arithmetic of the hot path happens inside the HIP library.

There is NO CPU fallback: every wrapper raises if the library is missing or a
tensor is not a HIP (``cuda``) tensor.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libkge_hip.so')

# enums of include/kge_hip.h
TRANSE_L1, TRANSE_L2, TRANSH, TRANSD, DISTMULT, COMPLEX = range(6)
SIDE_TAIL, SIDE_HEAD, SIDE_PROJ_H, SIDE_PROJ_T, SIDE_BOTH = range(5)
EW_ADD, EW_SUB, EW_MUL, EW_MULSUB, EW_MULADD = range(5)
LP_DOT, LP_L2_EXPAND, LP_L1_DIRECT, LP_L2_DIRECT, LP_L2_PROJH, LP_L2_PROJD = range(6)

_vp = ctypes.c_void_p
_i64 = ctypes.c_int64
_int = ctypes.c_int


class LpDesc(ctypes.Structure):
    """struct kge_lp_desc (include/kge_hip.h)."""
    _fields_ = [
        ('mode', ctypes.c_int32), ('K0', ctypes.c_int32), ('K1', ctypes.c_int32),
        ('reserved', ctypes.c_int32),
        ('B', _i64), ('N', _i64), ('c_base', _i64),
        ('A0', _vp), ('lda0', _i64), ('T0', _vp), ('ldt0', _i64),
        ('A1', _vp), ('lda1', _i64), ('T1', _vp), ('ldt1', _i64),
        ('qn', _vp), ('en', _vp),
        ('Wq', _vp), ('ldw', _i64), ('scal', _vp), ('scal_ld', _i64),
        ('r_idx', _vp), ('yc', _vp),
    ]


class SplitArgs(ctypes.Structure):
    """struct kge_split_args (include/kge_hip.h)."""
    _fields_ = [
        ('Qs', _vp), ('Es', _vp), ('qn0', _vp), ('qn1', _vp), ('qmax0', _vp), ('qmax1', _vp),
        ('emax0', _vp), ('emax1', _vp), ('xabsmax', _vp), ('yabsmax', _vp), ('accum_model', ctypes.c_int32),
        ('eps_scale', ctypes.c_float), ('thr_ready', ctypes.c_int32), ('q_cell_ss', _vp), ('e2pref', _vp),
        ('thr', _vp), ('list', _vp), ('cap', ctypes.c_int32), ('list_count', _vp), ('overflow', _vp),
        ('col_q', _vp), ('n_single_p', _i64), ('members', _vp), ('n_multi_p', _i64),
        ('q_cell_ss_index', _vp), ('q_cell_ss_ld', _i64),
        ('level', ctypes.c_int32), ('q_dn2', _vp), ('q_dn2_index', _vp), ('de2max', _vp),
        ('es_frag', ctypes.c_int32), ('true_idx', _vp), ('tp_block_max', _vp), ('tp_blocks', ctypes.c_int32),
        ('q_scale_per_query', ctypes.c_int32), ('region_count', _vp),
    ]


class SadArgs(ctypes.Structure):
    """struct kge_sad_args (include/kge_hip.h)."""
    _fields_ = [
        ('Qi', _vp), ('Ei', _vp), ('emax', _vp), ('rmax', _vp), ('eps_scale', ctypes.c_float),
        ('thr', _vp), ('list', _vp), ('cap', ctypes.c_int32), ('list_count', _vp), ('overflow', _vp),
        ('col_q', _vp), ('n_single_p', _i64), ('members', _vp), ('n_multi_p', _i64),
    ]


_SIGNATURES = {
    'kge_score_triples': [_int, _vp, _vp, _vp, _vp, _int, _int, _vp, _vp, _vp, _i64, _vp, _vp],
    'kge_score_triples_bwd': [_int, _vp, _vp, _vp, _vp, _int, _int, _vp, _vp, _vp, _i64, _vp,
                              _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    'kge_segment_sum_rows': [_vp, _i64, _int, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp],
    'kge_key_hist': [_vp, _i64, _vp, _i64, _vp, _vp],
    'kge_key_scatter': [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp],
    'kge_key_sort': [_vp, _i64, _vp, _i64, _int, _vp, _vp, _i64, _vp],
    'kge_lp_prep': [_int, _int, _vp, _vp, _vp, _vp, _int, _int, _vp, _vp, _vp, _i64, _vp, _vp,
                    _vp, _vp, _vp],
    'kge_lp_prep_sharded': [_int, _int, _vp, _vp, _vp, _vp, _int, _int, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp,
                            _vp, _vp, _vp],
    'kge_lp_prep_hi': [_int, _int, _vp, _vp, _vp, _vp, _int, _int, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp,
                       _vp, _vp, _vp, _int, _i64, _vp, _vp],
    'kge_relation_scores_proj': [_int, _vp, _vp, _vp, _vp, _int, _int, _vp, _vp, _i64, _i64, _vp, _i64, _vp],
    'kge_ewise': [_int, _vp, _vp, _vp, _vp, _i64, _vp, _vp],
    'kge_proj_query_stats': [_vp, _i64, _vp, _i64, _vp, _i64, _int, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp, _vp, _i64, _vp],
    'kge_row_sqnorm': [_vp, _i64, _i64, _int, _vp, _vp, _vp],
    'kge_row_sqnorm_any_order': [_vp, _i64, _i64, _int, _vp, _vp, _vp],
    'kge_row_dot': [_vp, _vp, _i64, _i64, _int, ctypes.c_float, _vp, _vp],
    'kge_gather_rows': [_vp, _i64, _vp, _i64, _int, _vp, _vp],
    'kge_normalize_rows': [_vp, _i64, _i64, _int, _vp],
    'kge_lp_scores': [ctypes.POINTER(LpDesc), _vp, _i64, _vp],
    'kge_lp_pair_scores': [ctypes.POINTER(LpDesc), _vp, _vp, _i64, _vp, _vp],
    'kge_lp_count_ge': [ctypes.POINTER(LpDesc), _vp, _vp, _vp],
    'kge_lp_split_units': [_int, _int],
    'kge_lp_split_rows': [_vp, _i64, _int, _vp, _i64, _int, _i64, _int, _int, _vp, ctypes.c_float, _vp, _vp, _vp,
                          _vp, _vp, _vp],
    'kge_lp_split_prefix_max': [_vp, _i64, _int, _int, _vp, _vp],
    'kge_lp_hi_units': [_int],
    'kge_lp_hi_rows': [_vp, _i64, _int, _vp, _i64, _int, _i64, _int, _int, _vp, ctypes.c_float, _vp, _vp, _vp, _vp, _vp,
                       _vp, _vp],
    'kge_lp_hi_rows_frag': [_vp, _i64, _int, _vp, _i64, _int, _i64, _int, _vp, ctypes.c_float, _vp, _vp, _vp, _vp, _vp, _vp],
    'kge_lp_hi_stream_supported': [_int],
    'kge_lp_table_prep_l2': [_vp, _i64, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp],
    'kge_lp_table_prep_blocks': [_i64],
    'kge_lp_dot_table_prep_blocks': [_i64, _int],
    'kge_lp_split_count': [ctypes.POINTER(LpDesc), ctypes.POINTER(SplitArgs), _vp, _vp, _vp],
    'kge_lp_split_recheck': [ctypes.POINTER(LpDesc), _vp, _vp, ctypes.c_int32, _vp, _vp, _vp, _vp],
    'kge_absmax': [_vp, _i64, _vp, _vp],
    'kge_lp_count_ge_cols': [ctypes.POINTER(LpDesc), _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp],
    'kge_topk_chunk': [_vp, _i64, _i64, _i64, _i64, _int, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _vp],
    'kge_lp_sad_rows': [_vp, _i64, _i64, _int, _vp, _vp, _vp, _vp, _vp],
    'kge_lp_split_recheck_regions': [ctypes.POINTER(LpDesc), _vp, _vp, ctypes.c_int32, _vp, _vp, _vp, _vp, _vp],
    'kge_lp_split_regions': [_i64],
    'kge_lp_split_regions_supported': [ctypes.POINTER(LpDesc)],
    'kge_lp_sad_count': [ctypes.POINTER(LpDesc), ctypes.POINTER(SadArgs), _vp, _vp, _vp],
    'kge_lp_sad_recheck': [ctypes.POINTER(LpDesc), _vp, _vp, ctypes.c_int32, _vp, _vp, _vp],
    'kge_lp_query_pipeline': [_int, _vp, _vp, _int, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _int, ctypes.c_float, _vp, _vp,
                              _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _int, _vp, _i64, _vp],
    'kge_lp_dot_query_pipeline': [_int, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _int, ctypes.c_float,
                                  _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _int, _vp, _int, _vp, _vp],
    'kge_lp_dot_table_prep': [_vp, _i64, _int, _vp, _i64, _int, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp],
    'kge_lp_dot_table_prep_fused': [_vp, _i64, _int, _vp, _i64, _int, _i64, _vp, _vp, _vp, _vp, _vp],
    'kge_mfma_f16_selftest': [],
    'kge_lp_filter_sub': [ctypes.POINTER(LpDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    'kge_lp_filter_sub_grouped': [ctypes.POINTER(LpDesc), _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp],
    'kge_lp_filter_sub_planned': [ctypes.POINTER(LpDesc), _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp,
                                  _vp, _vp],
    'kge_rank_finalize': [_vp, _vp, _vp, _i64, _vp, _vp, _vp],
    'kge_rank_finalize_both': [_vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _int, _vp, _vp],
    'kge_host_device_pointer': [_vp, _vp],
    'kge_copy_i64_indirect': [_vp, _i64, _vp, _vp],
    'kge_lp_scores_batched': [_int, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _int, _vp, _i64, _vp],
    'kge_get_rank': [_vp, _i64, _vp, _i64, _i64, _int, _vp, _vp],
    'kge_filter_lookup': [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp],
    'kge_filter_lookup_both': [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp],
    'kge_filter_scores': [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _vp],
    'kge_filtered_rank_from_scores': [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp],
    'kge_filtered_rank_from_tiles': [_vp, _i64, _i64, _int, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _i64,
                                     _vp, _vp, _int, _vp],
    'kge_corrupt_scatter': [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp],
    'kge_i64_max3': [_vp, _vp, _vp, _i64, _vp, _vp],
    'kge_filter_index_build': [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    'kge_filter_plan_build': [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp],
    'kge_column_plan_build': [_vp, _vp, _vp, _i64, _i64, _i64, _int, _int, _vp, _vp, _i64, _vp],
    'kge_column_plan_emit': [_i64, _i64, _i64, _int, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    'kge_topk': [_vp, _i64, _i64, _i64, _int, _vp, _vp, _vp],
}
# every symbol include/kge_hip.h declares
EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + ['kge_corrupt_ws_elems', 'kge_abi_version', 'kge_lp_split_rows_padded',
                                               'kge_build_arch', 'kge_lp_filter_sub_ws_bytes', 'kge_lp_sad_cols_padded',
                                               'kge_key_sort_ws_bytes', 'kge_lp_split_group_sets',
                                               'kge_filter_index_ws_bytes', 'kge_filter_plan_ws_bytes',
                                               'kge_column_plan_ws_bytes'])

_lib = None
ABI_VERSION = 32        # kge_abi_version() of the library this binding was written against


def load_library():
    """dlopen libkge_hip.so (built in-tree by torchkge_amd/csrc/build.py).
    Fails loudly: the HIP extension IS the product path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'torchkge_amd: %s not found -- build it with `python -m torchkge_amd.csrc.build` '
            '(or __graft_entry__.build()); there is no CPU / PyTorch fallback.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = _int
    lib.kge_corrupt_ws_elems.argtypes = [_i64]
    lib.kge_corrupt_ws_elems.restype = _i64
    lib.kge_lp_split_rows_padded.argtypes = [_i64, _int]
    lib.kge_lp_split_rows_padded.restype = _i64
    lib.kge_lp_filter_sub_ws_bytes.argtypes = [_i64, _i64]
    lib.kge_lp_filter_sub_ws_bytes.restype = _i64
    lib.kge_key_sort_ws_bytes.argtypes = [_i64, _int]
    lib.kge_key_sort_ws_bytes.restype = _i64
    lib.kge_filter_index_ws_bytes.argtypes = [_i64]
    lib.kge_filter_index_ws_bytes.restype = _i64
    lib.kge_filter_plan_ws_bytes.argtypes = [_i64, _i64]
    lib.kge_filter_plan_ws_bytes.restype = _i64
    lib.kge_column_plan_ws_bytes.argtypes = [_i64, _i64, _i64]
    lib.kge_column_plan_ws_bytes.restype = _i64
    lib.kge_lp_sad_cols_padded.argtypes = [_int]
    lib.kge_lp_sad_cols_padded.restype = _i64
    lib.kge_lp_split_group_sets.argtypes = []
    lib.kge_lp_split_group_sets.restype = _int
    lib.kge_abi_version.argtypes = []
    lib.kge_abi_version.restype = _int
    lib.kge_build_arch.argtypes = []
    lib.kge_build_arch.restype = ctypes.c_char_p
    if lib.kge_abi_version() != ABI_VERSION:
        raise RuntimeError('torchkge_amd: libkge_hip.so ABI version mismatch')
    _lib = lib
    return lib


def _check(rc, name):
    if rc != 0:
        if rc < 0:
            raise RuntimeError('torchkge_amd: %s rejected its arguments (code %d)' % (name, rc))
        raise RuntimeError('torchkge_amd: %s failed with hipError_t %d' % (name, rc))


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                'torchkge_amd runs only on MI355X (HIP) tensors; got a %s tensor. There is no '
                'CPU fallback -- move the model / indices to `cuda`.' % t.device)


def _stream():
    return _vp(torch.cuda.current_stream().cuda_stream)


class _NoSwitch(object):
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def _on(device):
    """Context that makes `device` current for a launch -- a no-op object when it already is
    (torch.cuda.device costs ~10 us of host time per entry; there are ~30 launches per batch)."""
    if device.index is None or device.index == torch.cuda.current_device():
        return _NO_SWITCH
    return torch.cuda.device(device)


def _p(t):
    return None if t is None else _vp(t.data_ptr())


def f32c(t):
    """fp32 contiguous view/copy (precondition of every table / matrix argument)."""
    if t.dtype != torch.float32:
        raise RuntimeError('torchkge_amd: expected a float32 tensor, got %s' % t.dtype)
    return t if t.is_contiguous() else t.contiguous()


def i64c(t):
    if t.dtype != torch.int64:
        t = t.long()
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------
# tensor-level wrappers
# ---------------------------------------------------------------------------
def score_triples(kind, tables, d_ent, d_rel, h, t, r):
    lib = load_library()
    require_cuda(h, t, r, *tables)
    tabs = [f32c(x) for x in tables] + [None] * (4 - len(tables))
    h, t, r = i64c(h), i64c(t), i64c(r)
    B = h.shape[0]
    out = torch.empty(B, dtype=torch.float32, device=h.device)
    with _on(h.device):
        _check(lib.kge_score_triples(kind, _p(tabs[0]), _p(tabs[1]), _p(tabs[2]), _p(tabs[3]),
                                     d_ent, d_rel, _p(h), _p(t), _p(r), B, _p(out), _stream()),
               'kge_score_triples')
    return out


# gradient-row streams of kge_score_triples_bwd's row mode: per model kind, for every table
# (index into `tables`) the first stream, the number of adjacent streams and the index they use
_BWD_STREAMS = {
    TRANSE_L1: [(0, 0, 2, 'ht'), (1, 2, 1, 'r')], TRANSE_L2: [(0, 0, 2, 'ht'), (1, 2, 1, 'r')],
    DISTMULT: [(0, 0, 2, 'ht'), (1, 2, 1, 'r')],
    TRANSH: [(0, 0, 2, 'ht'), (1, 2, 1, 'r'), (2, 3, 1, 'r')],
    COMPLEX: [(0, 0, 2, 'ht'), (1, 2, 2, 'ht'), (2, 4, 1, 'r'), (3, 5, 1, 'r')],
    TRANSD: [(0, 0, 2, 'ht'), (2, 2, 2, 'ht'), (1, 4, 1, 'r'), (3, 5, 1, 'r')],
}
BWD_SORTED_MIN_BATCH = 2048     # below this the plain atomic scatter is as fast


_KEY_SORT_WS = {}
BWD_PERM = os.environ.get('KGE_BWD_PERM', 'sort')      # 'sort' (kge_key_sort) | 'torch' (torch.sort) | 'count' (kge_key_hist / _scatter): how score_triples_bwd orders the ids of a large batch


def sort_perm(keys, n_keys):
    """Stable ascending order of small non-negative int64 keys (< n_keys <= 2**32): perm[j] = position of the j-th
    key (kge_key_sort: one device radix sort of (key, position) pairs)."""
    lib = load_library()
    require_cuda(keys)
    keys = i64c(keys)
    n, dev = keys.shape[0], keys.device
    bits = max(1, int(n_keys - 1).bit_length())
    if bits > 32:
        raise RuntimeError('sort_perm: keys need more than 32 bits')
    nb = int(lib.kge_key_sort_ws_bytes(n, bits))
    ws = torch.empty(max(nb, 8), dtype=torch.uint8, device=dev)
    perm = torch.empty(n, dtype=torch.int64, device=dev)
    if n > 0:
        with _on(dev):
            _check(lib.kge_key_sort(_p(keys), n, None, 0, bits, _p(perm), _p(ws), nb, _stream()), 'kge_key_sort')
    return perm


def score_triples_bwd(kind, tables, d_ent, d_rel, h, t, r, grad_out, needs):
    """Returns a list of gradient tensors (or None) matching ``tables``.  Large
    batches take the sorted reduction (per-triple gradient rows, then one atomic
    row-add per run of equal target rows) instead of one atomic per element."""
    lib = load_library()
    tabs = [f32c(x) for x in tables] + [None] * (4 - len(tables))
    grads = [torch.zeros_like(x) for x in tabs[:len(tables)]] + [None] * (4 - len(tables))
    go = f32c(grad_out)
    B = h.shape[0]
    dev = h.device
    if B < BWD_SORTED_MIN_BATCH:
        with _on(dev):
            _check(lib.kge_score_triples_bwd(kind, _p(tabs[0]), _p(tabs[1]), _p(tabs[2]), _p(tabs[3]),
                                             d_ent, d_rel, _p(h), _p(t), _p(r), B, _p(go),
                                             _p(grads[0]), _p(grads[1]), _p(grads[2]), _p(grads[3]),
                                             None, 0, _stream()), 'kge_score_triples_bwd')
        return [g if n else None for g, n in zip(grads[:len(tables)], needs)]
    streams = _BWD_STREAMS[kind]
    n_streams = max(s0 + ns for _, s0, ns, _ in streams)
    rows = torch.empty(n_streams * B * d_ent, dtype=torch.float32, device=dev)
    with _on(dev):
        _check(lib.kge_score_triples_bwd(kind, _p(tabs[0]), _p(tabs[1]), _p(tabs[2]), _p(tabs[3]),
                                         d_ent, d_rel, _p(h), _p(t), _p(r), B, _p(go), None, None, None, None,
                                         _p(rows), d_ent, _stream()), 'kge_score_triples_bwd')
        perms = {}
        for ti, s0, ns, key in streams:
            if not needs[ti]:
                continue
            g = grads[ti]
            k0, n0, k1, n1 = (h, B, t, B) if key == 'ht' else (r, B, None, 0)
            if key not in perms:    # the ids (they index g's rows) in sorted order: runs of equal target rows
                if BWD_PERM == 'sort':
                    # device radix sort of (id, position) over the id's bits (kge_key_sort): ~4x cheaper than the counting
                    # sort below at B = 32768, whose wave-aggregated atomics walk up to 64 distinct ids per wavefront
                    bits = max(1, int(g.shape[0] - 1).bit_length())
                    nb = _KEY_SORT_WS.get((n0 + n1, bits))
                    if nb is None:      # (the size query walks rocPRIM's host-side configuration: once per shape)
                        nb = _KEY_SORT_WS[(n0 + n1, bits)] = int(lib.kge_key_sort_ws_bytes(n0 + n1, bits))
                    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
                    perm = torch.empty(n0 + n1, dtype=torch.int64, device=dev)
                    _check(lib.kge_key_sort(_p(k0), n0, _p(k1), n1, bits, _p(perm), _p(ws), nb, _stream()), 'kge_key_sort')
                elif BWD_PERM == 'torch':
                    perm = torch.sort(k0 if k1 is None else torch.cat([k0, k1])).indices
                else:               # counting sort: hist, cumsum, scatter
                    cnt = torch.zeros(2, g.shape[0], dtype=torch.int32, device=dev)
                    _check(lib.kge_key_hist(_p(k0), n0, _p(k1), n1, _p(cnt[0]), _stream()), 'kge_key_hist')
                    off = torch.cumsum(cnt[0], 0, dtype=torch.int64) - cnt[0]
                    perm = torch.empty(n0 + n1, dtype=torch.int64, device=dev)
                    _check(lib.kge_key_scatter(_p(k0), n0, _p(k1), n1, _p(off), _p(cnt[1]), _p(perm), _stream()),
                           'kge_key_scatter')
                perms[key] = perm
            _check(lib.kge_segment_sum_rows(rows.data_ptr() + s0 * B * d_ent * 4, d_ent, g.shape[1], _p(k0), n0,
                                            _p(k1), n1, _p(perms[key]), _p(g), g.stride(0), _stream()),
                   'kge_segment_sum_rows')
    return [g if n else None for g, n in zip(grads[:len(tables)], needs)]


def side_code(side):
    """'tail' / 'head' / 'both' (both sides of a batch as one 2B-query problem, tail side first)."""
    return {'tail': SIDE_TAIL, 'head': SIDE_HEAD, 'both': SIDE_BOTH}[side]


def lp_prep(kind, side, tables, d_ent, d_rel, h, t, r, want_qn=False, want_w=False,
            want_q1=False, ent_lo=0, ent_n=-1, want_hi=False):
    """kge_lp_prep; ent_n >= 0: the entity tables hold only rows [ent_lo, ent_lo + ent_n) (kge_lp_prep_sharded:
    rows of entities this shard does not own come back as zeros, to be summed over the shards).
    want_hi (TransH / TransD, unsharded or replica tables): the same launch also writes the planar f16 hi operand of the
    query rows and their residuals (kge_lp_prep_hi); returns (Q0, Q1, qn, Wq, Qh, q_dn2)."""
    lib = load_library()
    require_cuda(h, t, r, *tables)
    tabs = [f32c(x) for x in tables] + [None] * (4 - len(tables))
    h, t, r = i64c(h), i64c(t), i64c(r)
    B = h.shape[0]
    dev = h.device
    n_facts = B
    if side == SIDE_BOTH:       # 2B queries: the tail-side queries of the B facts, then the head-side ones
        B = 2 * B
    Q0 = torch.empty(B, d_rel, dtype=torch.float32, device=dev)
    Q1 = torch.empty(B, d_rel, dtype=torch.float32, device=dev) if want_q1 else None
    qn = torch.empty(B, dtype=torch.float32, device=dev) if want_qn else None
    Wq = torch.empty(B, d_rel, dtype=torch.float32, device=dev) if want_w else None
    if want_hi:
        assert ent_n < 0 and kind in (TRANSH, TRANSD)
        units_p = int(lib.kge_lp_hi_units(d_rel))
        Bp = int(lib.kge_lp_split_rows_padded(B, 1))
        Qh = torch.empty(Bp * units_p * 32, dtype=torch.uint8, device=dev)
        dn2 = torch.empty(B, dtype=torch.float32, device=dev)
        with _on(dev):
            _check(lib.kge_lp_prep_hi(kind, side, _p(tabs[0]), _p(tabs[1]), _p(tabs[2]), _p(tabs[3]), d_ent, d_rel, _p(h), _p(t),
                                      _p(r), n_facts, ent_lo, ent_n, _p(Q0), _p(Q1), _p(qn), _p(Wq), _p(Qh), units_p, Bp,
                                      _p(dn2), _stream()), 'kge_lp_prep_hi')
        return Q0, Q1, qn, Wq, Qh, dn2
    with _on(dev):
        _check(lib.kge_lp_prep_sharded(kind, side, _p(tabs[0]), _p(tabs[1]), _p(tabs[2]), _p(tabs[3]),
                                       d_ent, d_rel, _p(h), _p(t), _p(r), n_facts, ent_lo, ent_n, _p(Q0), _p(Q1),
                                       _p(qn), _p(Wq), _stream()), 'kge_lp_prep_sharded')
    return Q0, Q1, qn, Wq


def proj_query_stats(Q, W, r_idx, scale, z_add, qmax_io=None, zero=None):
    """kge_proj_query_stats: (qn, pz) of a projection-mode problem in one launch -- ||Q_i||^2 and (scale * Q_i . W[r_i],
    ||W[r_i]||^2 + z_add), the chains of row_sqnorm / row_dot (same bits).  None when shapes / alignment need the
    separate kernels."""
    lib = load_library()
    require_cuda(Q, W, r_idx, qmax_io)
    Q, W, r_idx = f32c(Q), f32c(W), i64c(r_idx)
    rows, K = Q.shape
    qn = torch.empty(rows, dtype=torch.float32, device=Q.device)
    pz = torch.empty(rows, 2, dtype=torch.float32, device=Q.device)
    with _on(Q.device):
        rc = int(lib.kge_proj_query_stats(_p(Q), Q.stride(0), _p(W), W.stride(0), _p(r_idx), rows, K, float(scale), float(z_add),
                                          _p(qn), _p(pz), _p(qmax_io), _p(zero), 0 if zero is None else zero.numel(),
                                          _stream()))
    if rc == KGE_EUNSUPPORTED:
        return None
    _check(rc, 'kge_proj_query_stats')
    return qn, pz


def relation_scores_proj(kind, E, R, Wt, Ep, d_ent, d_rel, h, t):
    """(B, n_rel) scores of every relation for (h_i, ?, t_i) under the relation-specific projections of
    TransH / TransD (kge_relation_scores_proj)."""
    lib = load_library()
    require_cuda(E, R, Wt, Ep, h, t)
    E, R, Wt = f32c(E), f32c(R), f32c(Wt)
    Ep = None if Ep is None else f32c(Ep)
    h, t = i64c(h), i64c(t)
    B, n_rel = h.shape[0], R.shape[0]
    out = torch.empty(B, n_rel, dtype=torch.float32, device=E.device)
    with _on(E.device):
        _check(lib.kge_relation_scores_proj(kind, _p(E), _p(R), _p(Wt), _p(Ep), d_ent, d_rel, _p(h), _p(t), B, n_rel,
                                            _p(out), out.stride(0), _stream()), 'kge_relation_scores_proj')
    return out


def ewise(op, a, b, c=None, d=None):
    """Elementwise query-side algebra (kge_ewise); all operands same shape."""
    lib = load_library()
    require_cuda(a, b, c, d)
    a, b = f32c(a), f32c(b)
    c = None if c is None else f32c(c)
    d = None if d is None else f32c(d)
    assert a.shape == b.shape
    out = torch.empty_like(a)
    with _on(a.device):
        _check(lib.kge_ewise(op, _p(a), _p(b), _p(c), _p(d), a.numel(), _p(out), _stream()), 'kge_ewise')
    return out


_ACCUM_MODEL = None


def split_accum_model():
    """1 if this GPU's f16 MFMA passed the accumulation self-test (tighter error band of the
    split prefilter), else 0 (band valid for any fp32 adder).  Tested once per process, the first
    time a split count is prepared -- i.e. in evaluate()'s eager warm-up, never inside a capture."""
    global _ACCUM_MODEL
    if _ACCUM_MODEL is None:
        rc = int(load_library().kge_mfma_f16_selftest())
        _ACCUM_MODEL = 1 if rc == 1 else 0
    return _ACCUM_MODEL


SPLIT_EPS_SCALE = 1.0          # multiplies the proven error band of the f16-split prefilter (tests shrink it)
SPLIT_LIST_PER_QUERY = 64      # capacity of the uncertain-pair list per query of the batch (floor; grows with N)


def split_rows(X, K=None, is_query=False, aug=None, X1=None, K1=None, dot=False, nmax0=None, nmax1=None,
               cell_ss=False, row_index=None):
    """f16 hi/lo split operand of kge_lp_split_count (uint8 tensor holding
    [rows_p][units_p][64 bytes]) of [X | X1].  L2 mode (dot=False): candidates
    carry -||e||^2/2 in the extra column (aug = ||e||^2), queries carry 1, fixed
    scale.  DOT mode: queries carry their guard column (aug = ||q||^2), the
    scale comes from the device scalars nmax0 (+ nmax1)."""
    lib = load_library()
    require_cuda(X, X1, aug, nmax0, nmax1)
    X = f32c(X)
    rows, ld = X.shape[0], X.stride(0)
    if row_index is not None:       # output row r <- source row row_index[r] (gather inside the kernel)
        rows = int(row_index.shape[0])
    K = X.shape[1] if K is None else K
    ld1 = 0
    if X1 is not None:
        X1 = f32c(X1)
        K1 = X1.shape[1] if K1 is None else K1
        ld1 = X1.stride(0)
    else:
        K1 = 0
    if dot:
        aug_mode, aug_mul = (3, 1.0) if is_query else (4, 0.0)
    else:
        aug_mode, aug_mul = (2, 1.0) if is_query else (1, -0.5)
    units_p = int(lib.kge_lp_split_units(K + K1, 1))
    rows_p = int(lib.kge_lp_split_rows_padded(rows, 1 if is_query else 0))
    out = torch.empty(max(rows_p, 1) * units_p * 64, dtype=torch.uint8, device=X.device)
    css = torch.empty(units_p, max(rows_p, 1), dtype=torch.float32, device=X.device) if cell_ss else None
    with _on(X.device):
        _check(lib.kge_lp_split_rows(_p(X), ld, K, _p(X1), ld1, K1, rows, 1 if is_query else 0, aug_mode, _p(aug),
                                     aug_mul, _p(nmax0), _p(nmax1), _p(out), _p(css), _p(row_index), _stream()),
               'kge_lp_split_rows')
    return (out, css) if cell_ss else out


def split_table(X, K=None, aug=None, X1=None, K1=None, dot=False, nmax0=None, nmax1=None):
    """Candidate operand of the split prefilter + the prefix squared-norm maxima that tighten its error
    band (kge_lp_split_rows with cell sums, kge_lp_split_prefix_max): (Es, e2pref)."""
    lib = load_library()
    Es, css = split_rows(X, K=K, aug=aug, X1=X1, K1=K1, dot=dot, nmax0=nmax0, nmax1=nmax1, cell_ss=True)
    units_p = css.shape[0]
    e2 = torch.zeros(units_p, dtype=torch.float32, device=X.device)
    with _on(X.device):
        _check(lib.kge_lp_split_prefix_max(_p(css), X.shape[0], 0, units_p, _p(e2), _stream()),
               'kge_lp_split_prefix_max')
    return Es, e2


def hi_rows(X, K=None, is_query=False, aug=None, X1=None, K1=None, dot=False, nmax0=None, nmax1=None, want_dn2=False,
            dn2max=None, row_index=None, frag=False):
    """PLANAR hi operand of the one-product level (kge_lp_hi_rows): uint8 tensor [rows_p][hi_units][32 bytes] of
    [X | X1] (+ the per-row residuals ||x - hi(x)||^2 when want_dn2; their maximum folded into the device scalar
    dn2max).  Augmentation / scale conventions as split_rows."""
    lib = load_library()
    require_cuda(X, X1, aug, nmax0, nmax1, dn2max)
    X = f32c(X)
    rows, ld = X.shape[0], X.stride(0)
    if row_index is not None:
        rows = int(row_index.shape[0])
    K = X.shape[1] if K is None else K
    ld1 = 0
    if X1 is not None:
        X1 = f32c(X1)
        K1 = X1.shape[1] if K1 is None else K1
        ld1 = X1.stride(0)
    else:
        K1 = 0
    if dot:
        aug_mode, aug_mul = (3, 1.0) if is_query else (4, 0.0)
    else:
        aug_mode, aug_mul = (2, 1.0) if is_query else (1, -0.5)
    units_p = int(lib.kge_lp_hi_units(K + K1))
    rows_p = int(lib.kge_lp_split_rows_padded(rows, 1 if is_query else 0))
    out = torch.empty(max(rows_p, 1) * units_p * 32, dtype=torch.uint8, device=X.device)
    dn2 = torch.empty(max(rows, 1), dtype=torch.float32, device=X.device) if want_dn2 else None
    if frag:        # candidates of the free-running sweep: the same values in fragment-major order (kge_lp_hi_rows_frag)
        assert not is_query and row_index is None
        with _on(X.device):
            _check(lib.kge_lp_hi_rows_frag(_p(X), ld, K, _p(X1), ld1, K1, rows, aug_mode, _p(aug), aug_mul, _p(nmax0),
                                           _p(nmax1), _p(out), _p(dn2), _p(dn2max), _stream()), 'kge_lp_hi_rows_frag')
        return (out, dn2) if want_dn2 else out
    with _on(X.device):
        _check(lib.kge_lp_hi_rows(_p(X), ld, K, _p(X1), ld1, K1, rows, 1 if is_query else 0, aug_mode, _p(aug), aug_mul,
                                  _p(nmax0), _p(nmax1), _p(out), _p(dn2), _p(dn2max), _p(row_index), _stream()),
               'kge_lp_hi_rows')
    return (out, dn2) if want_dn2 else out


HI_STREAM = os.environ.get('KGE_HI_STREAM', '1') != '0'    # one-product level on the free-running kernel (lp_hi_stream.hip)


def hi_stream_ok(K):
    """Does the free-running one-product sweep (fragment-major candidate table, kge_split_args.es_frag) handle K columns?"""
    return HI_STREAM and int(load_library().kge_lp_hi_stream_supported(int(K))) == 1


def hi_stream_groups_ok(mode, K):
    """Does the free-running sweep take GROUPED query columns (ColumnPlan.members) for this problem?  r06: the plain-threshold
    modes (KGE_LP_DOT / KGE_LP_L2_EXPAND) on rows short enough for the resident panel (<= 32 k16 units)."""
    return mode in (LP_DOT, LP_L2_EXPAND) and (int(K) + 2 + 15) // 16 <= 32


def hi_table(X, K=None, aug=None, X1=None, K1=None, dot=False, nmax0=None, nmax1=None, frag=False):
    """Candidate operand of the one-product level: (Eh, de2max) -- the hi table (planar, or fragment-major for the
    free-running sweep) and the device scalar max_c ||e_c - hi(e_c)||^2 of its error band."""
    de2 = torch.zeros(1, dtype=torch.float32, device=X.device)
    Eh = hi_rows(X, K=K, aug=aug, X1=X1, K1=K1, dot=dot, nmax0=nmax0, nmax1=nmax1, dn2max=de2, frag=frag)
    return Eh, de2


def table_prep_l2(E, emax_io, de2max_io, deferred_max=False, K=None):
    """Candidate side of the L2 one-product sweep in ONE launch (kge_lp_table_prep_l2): (en, Ef) -- the squared row norms
    (kge_row_sqnorm's chain, same bits; maximum folded into emax_io) and the fragment-major hi table (residual maximum
    folded into de2max_io).  ``deferred_max``: (en, Ef, block_max) -- the two maxima stay per block (no same-address
    atomics) until lp_query_pipeline(tp_bmax=block_max) folds them into the scalars.  None when the table's shape /
    alignment needs the separate kernels."""
    lib = load_library()
    require_cuda(E, emax_io, de2max_io)
    E = f32c(E)
    rows, ld = E.shape[0], E.stride(0)
    K = E.shape[1] if K is None else int(K)     # (TransD: the first d_r columns of the (N, d_e) table)
    if K % 4 or ld % 4 or E.data_ptr() % 16 or rows == 0:
        return None
    units_p = int(lib.kge_lp_hi_units(K))
    rows_p = int(lib.kge_lp_split_rows_padded(rows, 0))
    en = torch.empty(rows, dtype=torch.float32, device=E.device)
    out = torch.empty(rows_p * units_p * 32, dtype=torch.uint8, device=E.device)
    bm = torch.empty(2 * int(lib.kge_lp_table_prep_blocks(rows)), dtype=torch.float32, device=E.device) if deferred_max else None
    with _on(E.device):
        rc = int(lib.kge_lp_table_prep_l2(_p(E), ld, rows, K, _p(en), _p(emax_io), _p(out), _p(de2max_io), _p(bm), _stream()))
    if rc == KGE_EUNSUPPORTED:
        return None
    _check(rc, 'kge_lp_table_prep_l2')
    return (en, out, bm) if deferred_max else (en, out)


# the uncertain-pair list of the free-running sweep cut into regions of 32 queries, re-scored with the region's query rows
# resident in LDS (kge_lp_split_recheck_regions); KGE_REGION_RECHECK=0: the one global list
# the count sweep of a fused-query-side batch as TWO launches over the halves of the queries, the first half's region recheck on
# a side stream beside the second half's sweep (LpProblem._count_ge_split_halves)
SWEEP_HALVES = os.environ.get('KGE_SWEEP_HALVES', '0') == '1'
_HALVES_STREAMS = {}


def _halves_stream(device):
    key = str(device)
    s = _HALVES_STREAMS.get(key)
    if s is None:
        s = _HALVES_STREAMS[key] = torch.cuda.Stream(device)
    return s


REGION_RECHECK = os.environ.get('KGE_REGION_RECHECK', '1') == '1'
# ... for the projection models too (TransH / TransD, ~5 listed pairs per query: a region's fixed cost -- 26 KB of query rows
# for ~170 pairs -- outweighs the rows it saves: 0.64 -> 0.66 ms, profiles/r05/region_recheck_ab.txt; off)
REGION_RECHECK_PROJ = os.environ.get('KGE_REGION_RECHECK_PROJ', '0') == '1'


def _counts_and_regions(Bq, dev, regions):
    """(counts (3, Bq) int32, region counters or None, int32 elements to zero): ONE buffer, zeroed by the pipeline's launch."""
    nreg = int(load_library().kge_lp_split_regions(Bq)) if regions else 0
    buf = torch.empty(3 * Bq + nreg, dtype=torch.int32, device=dev)
    return buf[:3 * Bq].view(3, Bq), (buf[3 * Bq:] if nreg else None), 3 * Bq + nreg


def lp_query_pipeline(side, E, R, h, t, r, en, emax, qmax_io, e2pref=None, cols=None, level=0, de2max=None, tp_bmax=None,
                      zero_counts=False, regions=False):
    """TransE-L2 query side of one batch in one launch (kge_lp_query_pipeline): dict with Q, qn,
    s_true, Qs, thr, n_list -- bit-identical to lp_prep + row_sqnorm + pair_scores + split_rows +
    the threshold kernel.  ``cols`` (filter_index.ColumnPlan): the split rows are written per COLUMN
    (one row per distinct query row of the batch) instead of per query."""
    lib = load_library()
    require_cuda(E, R, h, t, r, en, emax, qmax_io)
    E, R = f32c(E), f32c(R)
    h, t, r = i64c(h), i64c(t), i64c(r)
    B, d, dev = h.shape[0], E.shape[1], E.device
    Bq = 2 * B if side == SIDE_BOTH else B          # SIDE_BOTH: tail-side queries, then head-side queries
    Bp = int(lib.kge_lp_split_rows_padded(Bq, 1))
    # (level 1: the PLANAR hi operand of the one-product sweep, 32 bytes per unit)
    units_p, cell = (int(lib.kge_lp_hi_units(d)), 32) if level == 1 else (int(lib.kge_lp_split_units(d, 1)), 64)
    qrows = Bp if cols is None else cols.n_single_p + cols.n_multi_p
    assert cols is None or cols.n_queries == Bq
    out = {'Q': torch.empty(Bq, d, dtype=torch.float32, device=dev), 'qn': torch.empty(Bq, dtype=torch.float32, device=dev),
           's_true': torch.empty(Bq, dtype=torch.float32, device=dev),
           'Qs': torch.empty(max(qrows, 1) * units_p * cell, dtype=torch.uint8, device=dev), 'cols': cols, 'level': level,
           'thr': torch.empty(4 * Bp, dtype=torch.float32, device=dev),
           'n_list': torch.empty(1, dtype=torch.int32, device=dev)}
    if level == 1:      # (the residuals ||q - hi(q)||^2: a later split_count that recomputes the thresholds needs them)
        out['q_dn2'] = torch.empty(Bq, dtype=torch.float32, device=dev)
    zero_n = 0
    if zero_counts:     # the batch's (3, Bq) rank counters (+ the list's region counters), zeroed by this launch (no fill node)
        out['counts'], out['region_count'], zero_n = _counts_and_regions(Bq, dev, regions and level == 1 and cols is None
                                                                         and REGION_RECHECK)
    with _on(dev):
        _check(lib.kge_lp_query_pipeline(side, _p(E), _p(R), d, _p(h), _p(t), _p(r), B, _p(en), _p(emax), _p(qmax_io),
                                         split_accum_model(), SPLIT_EPS_SCALE, _p(out['Q']), _p(out['qn']),
                                         _p(out['s_true']), _p(out['Qs']), _p(out['thr']), _p(out['n_list']),
                                         _p(None if level == 1 else e2pref), _p(None if cols is None else cols.qs_row),
                                         level, _p(de2max), _p(out.get('q_dn2')), _p(tp_bmax),
                                         0 if tp_bmax is None else tp_bmax.shape[0] // 2, _p(out.get('counts')),
                                         zero_n, _stream()),
               'kge_lp_query_pipeline')
    return out


def dot_table_prep_fusable(X0, X1):
    """kge_lp_dot_table_prep_fused takes these tables (float4-readable rows)."""
    ok = X0.shape[1] % 4 == 0 and X0.stride(0) % 4 == 0 and X0.data_ptr() % 16 == 0
    if X1 is not None:
        ok = ok and X1.shape[1] % 4 == 0 and X1.stride(0) % 4 == 0 and X1.data_ptr() % 16 == 0
    return ok


def dot_table_prep(X0, X1, nmax0_io, nmax1_io, frag, prev_nmax=None):
    """Candidate side of a DOT problem on the one-product level in two launches (kge_lp_dot_table_prep): (Eh, dn_block_max, ws)
    -- the hi table (fragment-major when ``frag``) and its residual maxima per block, to be folded by lp_dot_query_pipeline;
    the squared-norm maxima of the table's segment(s) end up in nmax0_io / nmax1_io.
    ``prev_nmax`` (2 device floats, r06): ONE launch, one pass (kge_lp_dot_table_prep_fused) -- the scale of the maxima a
    previous evaluation left there; ws then holds THIS pass's squared-norm block maxima: hand it to
    lp_dot_query_pipeline(nm_bmax=ws, prev_nmax=prev_nmax), which folds them into nmax0_io / nmax1_io."""
    lib = load_library()
    require_cuda(X0, X1, nmax0_io, nmax1_io)
    X0 = f32c(X0)
    rows, K0 = X0.shape
    K1, ld1 = 0, 0
    if X1 is not None:
        X1 = f32c(X1)
        K1, ld1 = X1.shape[1], X1.stride(0)
    units_p = int(lib.kge_lp_hi_units(K0 + K1))
    rows_p = int(lib.kge_lp_split_rows_padded(rows, 0))
    out = torch.empty(rows_p * units_p * 32, dtype=torch.uint8, device=X0.device)
    if prev_nmax is not None:
        assert frag, 'the one-pass table preparation writes the fragment-major table'
        nb = int(lib.kge_lp_dot_table_prep_blocks(rows, 1))
        nmb = torch.empty(2 * nb, dtype=torch.float32, device=X0.device)
        dnb = torch.empty(nb, dtype=torch.float32, device=X0.device)
        with _on(X0.device):
            _check(lib.kge_lp_dot_table_prep_fused(_p(X0), X0.stride(0), K0, _p(X1), ld1, K1, rows, _p(prev_nmax), _p(out),
                                                   _p(dnb), _p(nmb), _stream()), 'kge_lp_dot_table_prep_fused')
        return out, dnb, nmb
    ws = torch.empty(2 * int(lib.kge_lp_dot_table_prep_blocks(rows, 0)), dtype=torch.float32, device=X0.device)
    dnb = torch.empty(int(lib.kge_lp_dot_table_prep_blocks(rows, 1)), dtype=torch.float32, device=X0.device)
    with _on(X0.device):
        _check(lib.kge_lp_dot_table_prep(_p(X0), X0.stride(0), K0, _p(X1), ld1, K1, rows, 1 if frag else 0, _p(nmax0_io),
                                         _p(nmax1_io), _p(out), _p(dnb), _p(ws), _stream()), 'kge_lp_dot_table_prep')
    return out, dnb, ws


def lp_dot_query_pipeline(side, E0, E1, R0, R1, h, t, r, emax0, emax1, de2max, qmax_io, overflow, zero_counts=False,
                          dn_bmax=None, regions=False, nm_bmax=None, prev_nmax=None):
    """DistMult (E1 = R1 = None) / ComplEx query side of one batch on the one-product level in one launch
    (kge_lp_dot_query_pipeline): dict with Q (and Q1), qn, s_true, Qs (planar hi operand, PER-QUERY scales), thr, q_dn2,
    n_list, counts -- Q / Q1 / s_true bit-identical to lp_prep + pair_scores."""
    lib = load_library()
    require_cuda(E0, R0, h, t, r, emax0, de2max, overflow)
    E0, R0 = f32c(E0), f32c(R0)
    E1, R1 = (None, None) if E1 is None else (f32c(E1), f32c(R1))
    h, t, r = i64c(h), i64c(t), i64c(r)
    B, d, dev = h.shape[0], E0.shape[1], E0.device
    K = d if E1 is None else 2 * d
    Bq = 2 * B if side == SIDE_BOTH else B
    Bp = int(lib.kge_lp_split_rows_padded(Bq, 1))
    units_p = int(lib.kge_lp_hi_units(K))
    out = {'Q': torch.empty(Bq, d, dtype=torch.float32, device=dev),
           'Q1': None if E1 is None else torch.empty(Bq, d, dtype=torch.float32, device=dev),
           'qn': torch.empty(Bq, dtype=torch.float32, device=dev), 's_true': torch.empty(Bq, dtype=torch.float32, device=dev),
           'Qs': torch.empty(Bp * units_p * 32, dtype=torch.uint8, device=dev), 'cols': None, 'level': 1,
           'thr': torch.empty(4 * Bp, dtype=torch.float32, device=dev),
           'q_dn2': torch.empty(Bq, dtype=torch.float32, device=dev),
           'n_list': torch.empty(1, dtype=torch.int32, device=dev), 'q_scale_per_query': True}
    zero_n = 0
    if zero_counts:
        out['counts'], out['region_count'], zero_n = _counts_and_regions(Bq, dev, regions and REGION_RECHECK)
    with _on(dev):
        _check(lib.kge_lp_dot_query_pipeline(side, _p(E0), _p(E1), _p(R0), _p(R1), d, _p(h), _p(t), _p(r), B, _p(emax0),
                                             _p(emax1), _p(de2max), _p(qmax_io), split_accum_model(), SPLIT_EPS_SCALE,
                                             _p(out['Q']), _p(out['Q1']), _p(out['qn']), _p(out['s_true']), _p(out['Qs']),
                                             _p(out['thr']), _p(out['q_dn2']), _p(out['n_list']), _p(overflow),
                                             _p(out.get('counts')), zero_n, _p(dn_bmax),
                                             0 if dn_bmax is None else dn_bmax.shape[0], _p(nm_bmax),
                                             0 if nm_bmax is None else nm_bmax.shape[0] // 2, _p(prev_nmax), _stream()),
               'kge_lp_dot_query_pipeline')
    return out


def sad_rows(X, emax, rmax, K=None, row_index=None):
    """16-bit fixed-point operand of the L1 prefilter (kge_lp_sad_rows): (rows, K padded to 8) uint16, scaled by
    32700 / (*emax + *rmax) (device scalars: max |x| of the entity and of the relation table)."""
    lib = load_library()
    require_cuda(X, emax, rmax)
    X = f32c(X)
    rows, ld = X.shape[0], X.stride(0)
    if row_index is not None:       # output row r <- source row row_index[r] (query columns)
        rows = int(row_index.shape[0])
    K = X.shape[1] if K is None else K
    Kp = int(lib.kge_lp_sad_cols_padded(K))
    out = torch.empty(max(rows, 1), Kp, dtype=torch.int16, device=X.device)
    with _on(X.device):
        _check(lib.kge_lp_sad_rows(_p(X), ld, rows, K, _p(emax), _p(rmax), _p(out), _p(row_index), _stream()),
               'kge_lp_sad_rows')
    return out


def absmax(x, max_io):
    """max_io[0] = max(max_io[0], max |x|) on the device (no sync)."""
    lib = load_library()
    require_cuda(x, max_io)
    x = f32c(x)
    with _on(x.device):
        _check(lib.kge_absmax(_p(x), x.numel(), _p(max_io), _stream()), 'kge_absmax')
    return max_io


def split_query_rows_padded(n):
    """Rows of a split QUERY operand for n queries / columns (a multiple of the count kernel's query panel)."""
    return int(load_library().kge_lp_split_rows_padded(n, 1))


def split_group_sets():
    """Threshold sets per grouped column of kge_lp_split_count (kge_split_args.members)."""
    return int(load_library().kge_lp_split_group_sets())


def padded_cols(n):
    """Row length (in floats) of a per-candidate matrix that the split count kernel may read
    up to the edge of its last 256-candidate tile."""
    return int(load_library().kge_lp_split_rows_padded(n, 0))


FAST_BOUND_NORMS = os.environ.get('KGE_FAST_BOUND_NORMS', '1') == '1'


def row_sqnorm(X, K=None, max_io=None, bound_only=False):
    """Squared row norms by the sequential fmaf chain (the contract of every value that enters a score) -- or, with
    ``bound_only`` (the result only bounds an error / fixes an operand scale: the DOT modes of the split prefilter), in
    any summation order by the bandwidth-bound kernel (kge_row_sqnorm_any_order)."""
    lib = load_library()
    require_cuda(X)
    X = f32c(X)
    rows, ld = X.shape[0], X.stride(0)
    K = X.shape[1] if K is None else K
    out = torch.empty(rows, dtype=torch.float32, device=X.device)
    with _on(X.device):
        if bound_only and FAST_BOUND_NORMS:
            _check(lib.kge_row_sqnorm_any_order(_p(X), ld, rows, K, _p(out), _p(max_io), _stream()), 'kge_row_sqnorm_any_order')
        else:
            _check(lib.kge_row_sqnorm(_p(X), ld, rows, K, _p(out), _p(max_io), _stream()), 'kge_row_sqnorm')
    return out


def row_dot(X, Y, scale=1.0):
    lib = load_library()
    require_cuda(X, Y)
    X, Y = f32c(X), f32c(Y)
    assert X.shape == Y.shape
    rows, K = X.shape
    out = torch.empty(rows, dtype=torch.float32, device=X.device)
    with _on(X.device):
        _check(lib.kge_row_dot(_p(X), _p(Y), K, rows, K, ctypes.c_float(scale), _p(out), _stream()),
               'kge_row_dot')
    return out


def gather_rows(X, idx):
    lib = load_library()
    require_cuda(X, idx)
    X, idx = f32c(X), i64c(idx)
    rows, K = idx.shape[0], X.shape[1]
    out = torch.empty(rows, K, dtype=torch.float32, device=X.device)
    with _on(X.device):
        _check(lib.kge_gather_rows(_p(X), X.stride(0), _p(idx), rows, K, _p(out), _stream()),
               'kge_gather_rows')
    return out


def normalize_rows_(X):
    """In-place F.normalize(X, p=2, dim=1)."""
    lib = load_library()
    require_cuda(X)
    if not X.is_contiguous() or X.dtype != torch.float32:
        raise RuntimeError('normalize_rows_: need a contiguous float32 matrix')
    with _on(X.device):
        _check(lib.kge_normalize_rows(_p(X), X.stride(0), X.shape[0], X.shape[1], _stream()),
               'kge_normalize_rows')
    return X


class LpProblem(object):
    """Python owner of a kge_lp_desc: keeps the tensors alive and exposes the
    descriptor entry points."""

    def __init__(self, mode, A0, T0, A1=None, T1=None, qn=None, en=None, Wq=None, scal=None,
                 r_idx=None, c_base=0, K0=None, yc=None):
        require_cuda(A0, T0, A1, T1, qn, en, Wq, scal, r_idx, yc)
        self.keep = [A0, T0, A1, T1, qn, en, Wq, scal, r_idx, yc]
        self.device = A0.device
        d = LpDesc()
        d.mode = mode
        d.K0 = A0.shape[1] if K0 is None else K0
        d.K1 = 0 if A1 is None else A1.shape[1]
        d.B, d.N, d.c_base = A0.shape[0], T0.shape[0], c_base
        d.A0, d.lda0 = A0.data_ptr(), A0.stride(0)
        d.T0, d.ldt0 = T0.data_ptr(), T0.stride(0)
        if A1 is not None:
            d.A1, d.lda1 = A1.data_ptr(), A1.stride(0)
            d.T1, d.ldt1 = T1.data_ptr(), T1.stride(0)
        if qn is not None:
            d.qn, d.en = qn.data_ptr(), en.data_ptr()
        if Wq is not None:
            d.Wq, d.ldw = Wq.data_ptr(), Wq.stride(0)
            d.scal = scal.data_ptr()
            d.scal_ld = 1 if scal.dim() == 1 else scal.stride(0)
            if r_idx is not None:
                d.r_idx = r_idx.data_ptr()
        if yc is not None:
            d.yc = yc.data_ptr()
        self.desc = d
        self.B, self.N = int(d.B), int(d.N)
        self.split = None
        self.sad = None         # TransE-L1: {'Ei', 'emax', 'rmax', 'overflow'} -> counts via the u16 SAD prefilter
        self.pre = None         # outputs of the fused query pipeline (true scores, split queries, thresholds)
        self.cols = None        # filter_index.ColumnPlan of a both-sides batch: split count over distinct query rows
        self.pre_q = None       # (Qh, q_dn2): the queries' planar hi operand, already built (projection models, level 1)
        self.region_count = None    # zeroed region counters of the sweep's uncertain-pair list (wants_regions)
        self.split_true = None  # (s_true tensor, true ids): the thresholds are the exact scores of these pairs (evaluator)

    def scores(self, out=None):
        lib = load_library()
        if out is None:
            out = torch.empty(self.B, self.N, dtype=torch.float32, device=self.device)
        with _on(self.device):
            _check(lib.kge_lp_scores(ctypes.byref(self.desc), _p(out), out.stride(0), _stream()),
                   'kge_lp_scores')
        return out

    def scores_chunk(self, c0, c1, out):
        """Scores of LOCAL candidates [c0, c1) only, into out[:, :c1 - c0] -- the same descriptor with its
        candidate-side pointers advanced (kge_lp_desc is plain pointers + sizes)."""
        lib = load_library()
        d = LpDesc.from_buffer_copy(self.desc)
        n = c1 - c0
        d.N, d.c_base = n, self.desc.c_base + c0
        d.T0 = self.desc.T0 + 4 * c0 * self.desc.ldt0
        if self.desc.T1:
            d.T1 = self.desc.T1 + 4 * c0 * self.desc.ldt1
        if self.desc.en:
            d.en = self.desc.en + 4 * c0
        if self.desc.yc:
            d.yc = self.desc.yc + 4 * c0
        if self.desc.scal:
            # DIRECT modes: per-candidate scalars (N, scal_ld) -> advance rows; projection modes: X (n_rel, scal_ld >= N)
            # -> advance columns
            proj = int(self.desc.mode) in (LP_L2_PROJH, LP_L2_PROJD)
            d.scal = self.desc.scal + 4 * c0 * (1 if proj else self.desc.scal_ld)
        with _on(self.device):
            _check(lib.kge_lp_scores(ctypes.byref(d), _p(out), out.stride(0), _stream()), 'kge_lp_scores')
        return out

    def scores_rows(self, q0, q1, out):
        """Scores of QUERIES [q0, q1) against every local candidate, into out[:q1 - q0] (leading dimension out.stride(0)
        >= N) -- the same descriptor with its query-side pointers advanced (the score tiles of the sharded path's
        all-to-all exchange are produced a row block at a time)."""
        lib = load_library()
        d = LpDesc.from_buffer_copy(self.desc)
        d.B = q1 - q0
        d.A0 = self.desc.A0 + 4 * q0 * self.desc.lda0
        if self.desc.A1:
            d.A1 = self.desc.A1 + 4 * q0 * self.desc.lda1
        if self.desc.qn:
            d.qn = self.desc.qn + 4 * q0
        if self.desc.Wq:
            d.Wq = self.desc.Wq + 4 * q0 * self.desc.ldw
        if self.desc.r_idx:
            d.r_idx = self.desc.r_idx + 8 * q0
        if q1 > q0 and self.N > 0:
            with _on(self.device):
                _check(lib.kge_lp_scores(ctypes.byref(d), _p(out), out.stride(0), _stream()), 'kge_lp_scores')
        return out

    def pair_scores(self, ci, qi=None):
        if self.pre is not None and qi is None and ci is self.pre['true_idx']:
            return self.pre['s_true']       # already computed by the fused query pipeline (same chain)
        lib = load_library()
        ci = i64c(ci)
        P = ci.shape[0]
        out = torch.empty(P, dtype=torch.float32, device=self.device)
        with _on(self.device):
            _check(lib.kge_lp_pair_scores(ctypes.byref(self.desc), _p(qi), _p(ci), P, _p(out),
                                          _stream()), 'kge_lp_pair_scores')
        return out

    def count_ge(self, s_true, raw=None):
        lib = load_library()
        if raw is None:
            raw = torch.zeros(self.B, dtype=torch.int32, device=self.device)
        if self.split is not None and self.B > 0 and self.N > 0:
            return self._count_ge_split(s_true, raw)
        if self.sad is not None and self.B > 0 and self.N > 0:
            return self._count_ge_sad(s_true, raw)
        cols = self.cols
        if (cols is not None and int(self.desc.mode) == LP_L2_DIRECT and not self.desc.Wq and self.N > 0
                and int(self.desc.K0) % 4 == 0 and self.desc.lda0 % 4 == 0 and self.desc.ldt0 % 4 == 0
                and self.desc.A0 % 16 == 0 and self.desc.T0 % 16 == 0):
            # plain L2 broadcast-subtract counts over the batch's distinct query rows (packed-FMA kernel)
            with _on(self.device):
                _check(lib.kge_lp_count_ge_cols(ctypes.byref(self.desc), _p(s_true), _p(raw), _p(cols.rep), _p(cols.col_q),
                                                cols.n_single_p, _p(cols.members), cols.n_multi_p, _stream()),
                       'kge_lp_count_ge_cols')
            return raw
        with _on(self.device):
            _check(lib.kge_lp_count_ge(ctypes.byref(self.desc), _p(s_true), _p(raw), _stream()),
                   'kge_lp_count_ge')
        return raw

    def wants_regions(self):
        """Region counters the caller may zero for this problem's sweep (set them as ``self.region_count``): the
        free-running sweep without a fused query pipeline (TransH / TransD) -- 0 when the regions do not apply."""
        sp = self.split
        if (not REGION_RECHECK or not REGION_RECHECK_PROJ or sp is None or self.pre is not None or self.pre_q is None or not sp.get('es_frag')
                or int(sp.get('level', 0)) != 1 or self.B == 0 or self.N == 0):
            return 0
        if self.cols is not None and self.cols.n_multi_p == 0:
            return 0        # (single-query columns keep their column -> query map; grouped ones fall back to per query)
        lib = load_library()
        if not int(lib.kge_lp_split_regions_supported(ctypes.byref(self.desc))):
            return 0
        return int(lib.kge_lp_split_regions(self.B))

    def split_prepare(self):
        """Per-batch operands of the f16-split prefilter: the split query matrix and
        the scratch buffers (thresholds, uncertain-pair list, its counter)."""
        lib = load_library()
        K = int(self.desc.K0)
        A0, A1 = self.keep[0], self.keep[2]
        extra = {}
        level = int(self.split.get('level', 0))
        if self.split.get('es_frag') and self.pre is None and self.cols is not None and self.cols.n_multi_p > 0 \
                and not hi_stream_groups_ok(int(self.desc.mode), K + int(self.desc.K1)):
            # grouped columns on the free-running sweep (r06): plain thresholds, resident panel only -- the projection
            # epilogues and the chunked-panel kernel of long rows sweep per query (always valid)
            self.cols = None
        want_ss = self.split.get('e2pref') is not None and level == 0    # prefix-norm magnitude bound of the error band
        if self.pre is not None:
            assert int(self.pre.get('level', 0)) == level
            Qs, extra = self.pre['Qs'], {'thr_pre': self.pre['thr'], 'n_list_pre': self.pre['n_list'],
                                         's_true_pre': self.pre['s_true'], 'cols': self.pre.get('cols')}
            if level == 1:      # per QUERY (the fused pipeline computes every query's residual itself)
                extra['q_dn2'], extra['q_dn2_per_query'] = self.pre['q_dn2'], True
            if self.pre.get('region_count') is not None and self.split.get('es_frag') and extra['cols'] is None:
                extra['region_count'] = self.pre['region_count']
            if self.pre.get('q_scale_per_query'):       # DOT pipeline: the hi operand carries per-query scales
                extra['q_scale_per_query'], extra['qn0'] = True, self.pre['qn']
        elif int(self.desc.mode) == LP_DOT:
            qmax = torch.zeros(2, dtype=torch.float32, device=self.device)
            # (DOT mode: the norms bound the error band and fix the operands' scale; no score contains them)
            qn0 = row_sqnorm(A0, K=K, max_io=qmax[0:1], bound_only=True)
            qn1 = row_sqnorm(A1, max_io=qmax[1:2], bound_only=True) if A1 is not None else None
            qn = qn0 if qn1 is None else qn0 + qn1
            cols = self.cols
            if level == 1:      # one-product level: planar hi rows (per column when the batch has a ColumnPlan) + residuals
                Qs, dn2 = hi_rows(A0, K=K, is_query=True, aug=qn, X1=A1, dot=True, nmax0=qmax[0:1],
                                  nmax1=qmax[1:2] if A1 is not None else None, want_dn2=True,
                                  row_index=None if cols is None else cols.rep)
                extra = {'qn0': qn0, 'qn1': qn1, 'qmax': qmax, 'cols': cols, 'q_dn2': dn2}
            elif cols is not None:
                # COLUMNS: one split row per distinct query row of the batch (ColumnPlan) -- gathered from the query that
                # provides it; the per-query cell sums of the error band are read back through the query -> column map
                Qs = split_rows(A0, K=K, is_query=True, aug=qn, X1=A1, dot=True, nmax0=qmax[0:1],
                                nmax1=qmax[1:2] if A1 is not None else None, cell_ss=want_ss, row_index=cols.rep)
                extra = {'qn0': qn0, 'qn1': qn1, 'qmax': qmax, 'cols': cols}
            else:
                Qs = split_rows(A0, K=K, is_query=True, aug=qn, X1=A1, dot=True, nmax0=qmax[0:1],
                                nmax1=qmax[1:2] if A1 is not None else None, cell_ss=want_ss)
                extra = {'qn0': qn0, 'qn1': qn1, 'qmax': qmax}
        elif level == 1 and getattr(self, 'pre_q', None) is not None and self.cols is None:
            Qs, dn2 = self.pre_q        # (the projection models' query preparation wrote the hi operand in its own launch)
            extra = {'cols': None, 'q_dn2': dn2}
            if getattr(self, 'region_count', None) is not None and self.split.get('es_frag'):
                extra['region_count'] = self.region_count
        elif level == 1:                # L2 on the one-product level (non-fused query path)
            cols = self.cols
            Qs, dn2 = hi_rows(A0, K=K, is_query=True, want_dn2=True, row_index=None if cols is None else cols.rep)
            extra = {'cols': cols, 'q_dn2': dn2}
        elif self.cols is not None:     # L2 / projection modes on COLUMNS (gathered rows, as the DOT branch above)
            cols = self.cols
            Qs = split_rows(A0, K=K, is_query=True, cell_ss=want_ss, row_index=cols.rep)
            extra = {'cols': cols}
        else:
            Qs = split_rows(A0, K=K, is_query=True, cell_ss=want_ss)
        if isinstance(Qs, tuple):
            Qs, extra['q_cell_ss'] = Qs
        Bp = int(lib.kge_lp_split_rows_padded(self.B, 1))
        thr = extra['thr_pre'] if 'thr_pre' in extra else torch.empty(4 * Bp, dtype=torch.float32, device=self.device)
        # the band holds ~1e-3 of a query's candidates for an untrained model (far fewer for a trained one)
        cap = int(min(max(SPLIT_LIST_PER_QUERY, self.N // 100) * self.B, 2 ** 31 - 1))
        lst = torch.empty(2 * cap, dtype=torch.int32, device=self.device)
        n_list = extra['n_list_pre'] if 'n_list_pre' in extra else torch.empty(1, dtype=torch.int32, device=self.device)
        prep = {'Qs': Qs, 'thr': thr, 'cap': cap, 'list': lst, 'n_list': n_list}
        prep.update(extra)
        return prep

    def split_count(self, prep, s_true, raw):
        """kge_lp_split_count: thresholds + the f16 MFMA count kernel (raw += #{acc >= a_lo})."""
        lib = load_library()
        sp = self.split
        a = SplitArgs()
        a.Qs, a.Es = _p(prep['Qs']), _p(sp['Es'])
        a.qn0, a.qn1 = _p(prep.get('qn0')), _p(prep.get('qn1'))
        qmax = prep.get('qmax')
        if qmax is not None:
            a.qmax0 = qmax.data_ptr()
            a.qmax1 = qmax.data_ptr() + 4 if prep.get('qn1') is not None else None
        a.emax0, a.emax1 = _p(sp['enmax']), _p(sp.get('enmax1'))
        a.xabsmax, a.yabsmax = _p(sp.get('xabsmax')), _p(sp.get('yabsmax'))
        a.accum_model = split_accum_model()
        a.eps_scale = SPLIT_EPS_SCALE
        # thresholds written by the fused query pipeline are valid for its own true scores, once
        a.q_cell_ss, a.e2pref = _p(prep.get('q_cell_ss')), _p(sp.get('e2pref') if prep.get('q_cell_ss') is not None else None)
        if prep.get('q_cell_ss') is not None and prep.get('cols') is not None and self.pre is None:
            # the cell sums are per COLUMN (rows gathered inside kge_lp_split_rows): query i reads its column's
            a.q_cell_ss_index, a.q_cell_ss_ld = _p(prep['cols'].col_of_q), prep['q_cell_ss'].shape[1]
        a.thr_ready = 1 if (prep.get('s_true_pre') is s_true and not prep.get('thr_used')) else 0
        prep['thr_used'] = True
        a.thr, a.list, a.cap = _p(prep['thr']), _p(prep['list']), prep['cap']
        a.list_count, a.overflow = _p(prep['n_list']), _p(sp['overflow'])
        cols = prep.get('cols')
        if cols is not None:    # the split query rows are per COLUMN (distinct query rows), not per query
            a.col_q, a.n_single_p = _p(cols.col_q), cols.n_single_p
            a.members, a.n_multi_p = _p(cols.members), cols.n_multi_p
        a.level = int(sp.get('level', 0))
        tpb = sp.get('tp_bmax')
        if tpb is not None and not a.thr_ready:     # block maxima of the fused table preparation: folded by the threshold kernel
            a.tp_block_max, a.tp_blocks = _p(tpb), tpb.shape[0] // 2
        a.q_scale_per_query = 1 if prep.get('q_scale_per_query') else 0
        a.es_frag = 1 if sp.get('es_frag') else 0
        rcnt = prep.get('region_count') if a.es_frag else None
        if rcnt is not None and not int(lib.kge_lp_split_regions_supported(ctypes.byref(self.desc))):
            rcnt = prep['region_count'] = None
        if rcnt is not None:
            # a second sweep on the same operands: the counters start from zero again (zeroed by the threshold kernel when it
            # runs -- thr_ready = 0 -- like *list_count; by two fills otherwise)
            if getattr(self, '_regions_used', False) and a.thr_ready:
                rcnt.zero_()
                prep['n_list'].zero_()
            self._regions_used = True
            a.region_count = _p(rcnt)
        if a.es_frag:
            assert a.level == 1 and (cols is None or cols.n_multi_p == 0 or
                                     hi_stream_groups_ok(int(self.desc.mode), int(self.desc.K0) + int(self.desc.K1))), \
                'grouped columns on the free-running sweep: plain-threshold modes on resident panels only'
            # s_true IS the exact score of (query, split_true entity): the sweep need not list that pair
            st_true = getattr(self, 'split_true', None)
            if st_true is not None and st_true[0] is s_true:
                a.true_idx = _p(i64c(st_true[1]))
        if a.level == 1:        # one-product level: the band needs the operands' measured f16 residuals
            a.de2max = _p(sp['de2max'])
            a.q_dn2 = _p(prep.get('q_dn2'))
            if prep.get('q_dn2') is not None and cols is not None and not prep.get('q_dn2_per_query'):
                a.q_dn2_index = _p(cols.col_of_q)
        with _on(self.device):
            _check(lib.kge_lp_split_count(ctypes.byref(self.desc), ctypes.byref(a), _p(s_true), _p(raw), _stream()),
                   'kge_lp_split_count')
        return raw

    def split_recheck(self, prep, s_true, raw):
        """kge_lp_split_recheck: exact re-scoring of the pairs inside the error band."""
        lib = load_library()
        if prep.get('region_count') is not None:        # the sweep left its pairs region by region
            with _on(self.device):
                _check(lib.kge_lp_split_recheck_regions(ctypes.byref(self.desc), _p(s_true), _p(prep['list']), prep['cap'],
                                                        _p(prep['region_count']), _p(raw), _p(self.split.get('list_stat')),
                                                        _p(prep['n_list']), _stream()), 'kge_lp_split_recheck_regions')
            return raw
        with _on(self.device):
            _check(lib.kge_lp_split_recheck(ctypes.byref(self.desc), _p(s_true), _p(prep['list']), prep['cap'],
                                            _p(prep['n_list']), _p(raw), _p(self.split.get('list_stat')), _stream()),
                   'kge_lp_split_recheck')
        return raw

    def _count_ge_sad(self, s_true, raw):
        """Same counts as kge_lp_count_ge for a plain L1 problem through the certified 16-bit
        sum-of-absolute-differences prefilter + exact recheck of the pairs inside the error band
        (kge_lp_sad_count / kge_lp_sad_recheck); self.sad is set by the model."""
        lib = load_library()
        sd = self.sad
        cols = self.cols
        Qi = sad_rows(self.keep[0], sd['emax'], sd['rmax'], K=int(self.desc.K0),
                      row_index=None if cols is None else cols.rep)
        cap = int(min(max(SPLIT_LIST_PER_QUERY, self.N // 50) * self.B, 2 ** 31 - 1))
        a = SadArgs()
        thr = torch.empty(2 * self.B, dtype=torch.int32, device=self.device)
        lst = torch.empty(2 * cap, dtype=torch.int32, device=self.device)
        n_list = torch.empty(1, dtype=torch.int32, device=self.device)
        a.Qi, a.Ei, a.emax, a.rmax = _p(Qi), _p(sd['Ei']), _p(sd['emax']), _p(sd['rmax'])
        a.eps_scale = SPLIT_EPS_SCALE
        a.thr, a.list, a.cap, a.list_count, a.overflow = _p(thr), _p(lst), cap, _p(n_list), _p(sd['overflow'])
        if cols is not None:    # the fixed-point query rows are per COLUMN (distinct query rows of the batch)
            a.col_q, a.n_single_p = _p(cols.col_q), cols.n_single_p
            a.members, a.n_multi_p = _p(cols.members), cols.n_multi_p
        with _on(self.device):
            _check(lib.kge_lp_sad_count(ctypes.byref(self.desc), ctypes.byref(a), _p(s_true), _p(raw), _stream()),
                   'kge_lp_sad_count')
            _check(lib.kge_lp_sad_recheck(ctypes.byref(self.desc), _p(s_true), _p(lst), cap, _p(n_list), _p(raw),
                                          _stream()), 'kge_lp_sad_recheck')
        self.last_split = (n_list, (Qi, thr, lst))      # kept alive until the launches have run; tests read n_list
        return raw

    def _count_ge_split(self, s_true, raw, between=None):
        """Same counts as kge_lp_count_ge through the certified f16-split
        prefilter + exact recheck of the pairs inside the error band;
        self.split = {'Es', 'enmax', 'overflow'} is set by the model.
        ``between``: called after the sweep is enqueued and before the recheck (the evaluator forks its filter correction
        there: beside the recheck instead of beside the sweep)."""
        prep = self.split_prepare()
        if self._sweep_in_halves(prep, s_true):
            return self._count_ge_split_halves(prep, s_true, raw, between)
        self.split_count(prep, s_true, raw)
        if between is not None:
            between()
        self.split_recheck(prep, s_true, raw)
        self.last_split = (prep['n_list'], prep)     # kept alive until the launches have run; tests read n_list
        return raw

    # ---- the sweep in two halves, the first half's exact recheck beside the second half's sweep (r06) -------------------------
    def _sweep_in_halves(self, prep, s_true):
        """Applies where the sweep's uncertain pairs go to REGIONS (free-running kernel, fused query side with ready thresholds,
        K <= 256): a region belongs to 32 consecutive queries, so the queries [0, H) and [H, B) are two independent problems
        on the same operands -- same kernels, pointers advanced -- and the recheck of the first need not wait for the second."""
        if not SWEEP_HALVES or prep.get('region_count') is None or not self.split.get('es_frag') or prep.get('cols') is not None:
            return False
        if prep.get('s_true_pre') is not s_true or prep.get('thr_used') or self.desc.K1 or int(self.desc.mode) not in (LP_L2_EXPAND, LP_DOT):
            return False
        return self.B >= 8 * 192 and s_true.is_cuda

    def _count_ge_split_halves(self, prep, s_true, raw, between=None):
        lib = load_library()
        if not int(lib.kge_lp_split_regions_supported(ctypes.byref(self.desc))):
            prep['region_count'] = None
            self.split_count(prep, s_true, raw)
            if between is not None:
                between()
            self.split_recheck(prep, s_true, raw)
            self.last_split = (prep['n_list'], prep)
            return raw
        import copy as _copy
        Bp = int(lib.kge_lp_split_rows_padded(self.B, 1))
        H = (Bp // 2) // 192 * 192                      # whole 192-row padding units: the halves pad like the whole
        K = int(self.desc.K0) + int(self.desc.K1)
        row_bytes = int(lib.kge_lp_hi_units(K)) * 32
        cap_h = int(prep['cap']) // 2
        st_true = getattr(self, 'split_true', None)
        halves = []
        for q0, q1, li in ((0, H, 0), (H, self.B, 1)):
            sp = _copy.copy(self)
            d = LpDesc.from_buffer_copy(self.desc)
            d.B = q1 - q0
            d.A0 = self.desc.A0 + 4 * q0 * self.desc.lda0
            if self.desc.qn:
                d.qn = self.desc.qn + 4 * q0
            sp.desc, sp.B = d, q1 - q0
            s_sub = s_true[q0:q1]
            sp.split_true = (s_sub, st_true[1][q0:q1]) if (st_true is not None and st_true[0] is s_true) else None
            sp._regions_used = False
            pp = {'Qs': prep['Qs'][q0 * row_bytes:], 'thr': prep['thr'][2 * q0:], 'cap': cap_h,
                  'list': prep['list'][2 * cap_h * li:2 * cap_h * (li + 1)], 'n_list': prep['n_list'], 'cols': None,
                  's_true_pre': s_sub, 'region_count': prep['region_count'][(q0 // 32):], 'q_dn2': prep.get('q_dn2'),
                  'q_dn2_per_query': prep.get('q_dn2_per_query')}
            halves.append((sp, pp, s_sub, raw[q0:q1]))
        main = torch.cuda.current_stream(self.device)
        side = _halves_stream(self.device)
        (p0, pp0, s0, r0), (p1, pp1, s1, r1) = halves
        p0.split_count(pp0, s0, r0)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            p0.split_recheck(pp0, s0, r0)
        p1.split_count(pp1, s1, r1)
        if between is not None:
            between()
        p1.split_recheck(pp1, s1, r1)
        main.wait_stream(side)
        prep['thr_used'] = True
        self.last_split = (prep['n_list'], (prep, pp0, pp1))
        return raw

    def filter_sub(self, s_true, true_idx, seg_lo, seg_hi, targets, sub=None, found=None, grouped=False, plan=None):
        """Filter correction of every query.  ``grouped=True`` (link-prediction batches: queries that
        share a filter segment share their query row): kge_lp_filter_sub_grouped -- each distinct
        list scored once, pairs flattened over the grid (heavy-tailed lists stay cheap).  ``plan``
        (filter_index.FilterPlan): the same with the grouping precomputed (kge_lp_filter_sub_planned)."""
        lib = load_library()
        if sub is None:
            sub = torch.empty(self.B, dtype=torch.int32, device=self.device)
        if found is None:
            found = torch.empty(self.B, dtype=torch.int32, device=self.device)
        if plan is not None and self.B > 0:
            n_t = int(targets.shape[0])
            fs = torch.empty(max(n_t, 1), dtype=torch.float32, device=self.device)
            with _on(self.device):
                _check(lib.kge_lp_filter_sub_planned(ctypes.byref(self.desc), _p(s_true), _p(true_idx), _p(seg_lo),
                                                     _p(seg_hi), _p(targets), n_t, _p(plan.woff), plan.n_pairs,
                                                     _p(plan.long_q) if plan.n_long else None, plan.n_long, _p(fs),
                                                     _p(sub), _p(found), _stream()), 'kge_lp_filter_sub_planned')
            return sub, found
        if grouped and self.B > 0:
            n_t = int(targets.shape[0])
            nb = int(lib.kge_lp_filter_sub_ws_bytes(self.B, n_t))
            ws = torch.empty(max(nb, 8), dtype=torch.uint8, device=self.device)
            with _on(self.device):
                _check(lib.kge_lp_filter_sub_grouped(ctypes.byref(self.desc), _p(s_true), _p(true_idx), _p(seg_lo),
                                                     _p(seg_hi), _p(targets), n_t, _p(sub), _p(found), _p(ws), nb,
                                                     _stream()), 'kge_lp_filter_sub_grouped')
            return sub, found
        with _on(self.device):
            _check(lib.kge_lp_filter_sub(ctypes.byref(self.desc), _p(s_true), _p(true_idx),
                                         _p(seg_lo), _p(seg_hi), _p(targets), _p(sub), _p(found),
                                         _stream()), 'kge_lp_filter_sub')
        return sub, found


def rank_finalize(raw, sub, found):
    lib = load_library()
    B = raw.shape[0]
    rank = torch.empty(B, dtype=torch.int64, device=raw.device)
    filt = torch.empty(B, dtype=torch.int64, device=raw.device)
    with _on(raw.device):
        _check(lib.kge_rank_finalize(_p(raw), _p(sub), _p(found), B, _p(rank), _p(filt), _stream()),
               'kge_rank_finalize')
    return rank, filt


def rank_finalize_both(raw, sub, found, out, off, pos=None, guard=None, flags=None, zero_guard=False, indirect=None):
    """kge_rank_finalize_both: the ranks of a 2B-query batch into the (4, n) int64 result matrix
    `out` (rows: head raw, tail raw, head filtered, tail filtered) at columns off .. off + B - 1.
    ``indirect`` (r06): address of a device-visible int64 holding the address of the result matrix of THIS launch (pinned host
    memory: host_device_pointer) -- `out` then only gives the shape, the flags go behind the four rows."""
    lib = load_library()
    require_cuda(raw, out)
    B = raw.shape[0] // 2
    if out.dtype != torch.int64 or out.dim() != 2 or out.shape[0] != 4 or out.stride(1) != 1:
        raise RuntimeError('rank_finalize_both: out must be a (4, n) int64 matrix with unit column stride')
    with _on(raw.device):
        _check(lib.kge_rank_finalize_both(_p(raw), _p(sub), _p(found), B, _p(out), out.stride(0), off, _p(pos),
                                          _p(guard if flags is not None else None), _p(flags),
                                          1 if (zero_guard and flags is not None) else 0,
                                          ctypes.c_void_p(int(indirect)) if indirect else None, _stream()),
               'kge_rank_finalize_both')
    return out


def copy_i64_indirect(src, n, indirect):
    """kge_copy_i64_indirect: n int64 from the device tensor ``src`` to the buffer whose address the device finds at ``indirect``."""
    lib = load_library()
    require_cuda(src)
    with _on(src.device):
        _check(lib.kge_copy_i64_indirect(_p(src), int(n), ctypes.c_void_p(int(indirect)), _stream()), 'kge_copy_i64_indirect')


def host_device_pointer(t):
    """Device-visible address of a PINNED host tensor (kge_host_device_pointer); None when it is not mapped."""
    lib = load_library()
    dev = ctypes.c_void_p()
    if lib.kge_host_device_pointer(ctypes.c_void_p(t.data_ptr()), ctypes.byref(dev)) != 0 or not dev.value:
        return None
    return int(dev.value)


def lp_scores_batched(mode, q, cand):
    """q (B,K); cand (B,N,K) with arbitrary batch / row strides (inner stride 1)."""
    lib = load_library()
    require_cuda(q, cand)
    q = f32c(q)
    if cand.dtype != torch.float32:
        raise RuntimeError('expected float32 candidates')
    if cand.stride(2) != 1:
        cand = cand.contiguous()
    B, N, K = cand.shape
    out = torch.empty(B, N, dtype=torch.float32, device=q.device)
    with _on(q.device):
        _check(lib.kge_lp_scores_batched(mode, _p(q), q.stride(0), _p(cand), cand.stride(0),
                                         cand.stride(1), B, N, K, _p(out), out.stride(0), _stream()),
               'kge_lp_scores_batched')
    return out


def get_rank(scores, true_idx, low_values=False):
    lib = load_library()
    require_cuda(scores, true_idx)
    scores = f32c(scores)
    true_idx = i64c(true_idx)
    B, N = scores.shape
    rank = torch.empty(B, dtype=torch.int64, device=scores.device)
    with _on(scores.device):
        _check(lib.kge_get_rank(_p(scores), scores.stride(0), _p(true_idx), B, N,
                                1 if low_values else 0, _p(rank), _stream()), 'kge_get_rank')
    return rank


def filter_lookup(keys, offsets, key1, key2, n_key2):
    lib = load_library()
    key1, key2 = i64c(key1), i64c(key2)
    B = key1.shape[0]
    lo = torch.empty(B, dtype=torch.int64, device=key1.device)
    hi = torch.empty(B, dtype=torch.int64, device=key1.device)
    with _on(key1.device):
        _check(lib.kge_filter_lookup(_p(keys), keys.shape[0], _p(offsets), _p(key1), _p(key2),
                                     n_key2, B, _p(lo), _p(hi), _stream()), 'kge_filter_lookup')
    return lo, hi


def filter_lookup_both(keys_t, offsets_t, keys_h, offsets_h, targets_base_h, h, t, r, n_key2):
    """Both sides of B facts as one 2B-query batch (kge_filter_lookup_both):
    (seg_lo, seg_hi, true_idx), tail-side queries first; head-side segments are
    shifted by targets_base_h (the two target arrays are kept concatenated)."""
    lib = load_library()
    require_cuda(keys_t, keys_h, h, t, r)
    h, t, r = i64c(h), i64c(t), i64c(r)
    B = h.shape[0]
    lo = torch.empty(2 * B, dtype=torch.int64, device=h.device)
    hi = torch.empty(2 * B, dtype=torch.int64, device=h.device)
    true = torch.empty(2 * B, dtype=torch.int64, device=h.device)
    with _on(h.device):
        _check(lib.kge_filter_lookup_both(_p(keys_t), keys_t.shape[0], _p(offsets_t), _p(keys_h), keys_h.shape[0],
                                          _p(offsets_h), targets_base_h, _p(h), _p(t), _p(r), n_key2, B,
                                          _p(lo), _p(hi), _p(true), _stream()), 'kge_filter_lookup_both')
    return lo, hi, true


def filter_scores_(scores, true_idx, seg_lo, seg_hi, targets):
    lib = load_library()
    B, N = scores.shape
    with _on(scores.device):
        _check(lib.kge_filter_scores(_p(scores), scores.stride(0), _p(true_idx), _p(seg_lo),
                                     _p(seg_hi), _p(targets), B, N, _stream()), 'kge_filter_scores')
    return scores


def filtered_rank_from_scores(scores, true_idx, seg_lo, seg_hi, targets):
    lib = load_library()
    scores = f32c(scores)
    B, N = scores.shape
    rank = torch.empty(B, dtype=torch.int64, device=scores.device)
    filt = torch.empty(B, dtype=torch.int64, device=scores.device)
    with _on(scores.device):
        _check(lib.kge_filtered_rank_from_scores(_p(scores), scores.stride(0), _p(true_idx),
                                                 _p(seg_lo), _p(seg_hi), _p(targets), B, N,
                                                 _p(rank), _p(filt), _stream()),
               'kge_filtered_rank_from_scores')
    return rank, filt


def filtered_rank_from_tiles(tiles, n_total, true_idx, seg_lo, seg_hi, targets, rows, q_first, B, out, off, pos=None,
                             own=None, own_rank=0):
    """kge_filtered_rank_from_tiles: ranks of `rows` queries whose score rows lie in the rank-major tiles
    (P, m, per) of the score all-to-all; written into the (4, n) result matrix like rank_finalize_both.
    ``own`` (m, per): tile `own_rank` is read from there instead (the block the caller scored itself)."""
    lib = load_library()
    require_cuda(tiles, true_idx, out)
    assert tiles.dtype == torch.float32 and tiles.is_contiguous() and tiles.dim() == 3
    if out.dtype != torch.int64 or out.dim() != 2 or out.shape[0] != 4 or out.stride(1) != 1:
        raise RuntimeError('filtered_rank_from_tiles: out must be a (4, n) int64 matrix with unit column stride')
    P, m, per = tiles.shape
    if own is not None:
        require_cuda(own)
        assert own.dtype == torch.float32 and own.is_contiguous() and tuple(own.shape) == (m, per)
    with _on(tiles.device):
        _check(lib.kge_filtered_rank_from_tiles(_p(tiles), m, per, P, n_total, _p(true_idx), _p(seg_lo), _p(seg_hi),
                                                _p(targets), rows, q_first, B, _p(out), out.stride(0), off, _p(pos),
                                                _p(own), own_rank, _stream()), 'kge_filtered_rank_from_tiles')
    return out


def topk(scores, k):
    """(values (B,k), indices (B,k)) in the order (score desc, index asc)."""
    lib = load_library()
    require_cuda(scores)
    scores = f32c(scores)
    B, N = scores.shape
    idx = torch.empty(B, k, dtype=torch.int64, device=scores.device)
    val = torch.empty(B, k, dtype=torch.float32, device=scores.device)
    with _on(scores.device):
        _check(lib.kge_topk(_p(scores), scores.stride(0), B, N, k, _p(idx), _p(val), _stream()), 'kge_topk')
    return val, idx


def topk_chunk(scores, c_base, k, out_val, out_idx, col_off, seg_lo=None, seg_hi=None, targets=None, ids_in=None):
    """kge_topk_chunk: the k best of every row of the (B, C) tile `scores` (candidates c_base ..; masked in place by the
    rows' filter segments when `targets` is given) into columns [col_off, col_off + k) of out_val / out_idx (B, ldo);
    ``ids_in`` (B, C) int64: merge mode, the columns' candidate ids."""
    lib = load_library()
    require_cuda(scores, out_val, out_idx)
    assert scores.dtype == torch.float32 and scores.stride(1) == 1 and out_val.stride(0) == out_idx.stride(0)
    B, C = scores.shape
    with _on(scores.device):
        _check(lib.kge_topk_chunk(_p(scores), scores.stride(0), B, C, c_base, k, _p(seg_lo), _p(seg_hi), _p(targets),
                                  _p(ids_in), 0 if ids_in is None else ids_in.stride(0), _p(out_idx), _p(out_val),
                                  out_val.stride(0), col_off, _stream()), 'kge_topk_chunk')


KGE_EUNSUPPORTED = -3


def _host_counts(counts):
    """The element counts a builder left in a small device array -> Python ints (one pinned copy, one stream sync)."""
    host = torch.empty(counts.shape, dtype=counts.dtype, pin_memory=True)
    host.copy_(counts, non_blocking=True)
    torch.cuda.current_stream(counts.device).synchronize()
    return [int(x) for x in host.tolist()]


def i64_max3(a, b, c):
    """(max a, max b, max c) of three int64 id vectors of equal length, one launch + one sync."""
    lib = load_library()
    require_cuda(a, b, c)
    out = torch.zeros(3, dtype=torch.int64, device=a.device)
    with _on(a.device):
        _check(lib.kge_i64_max3(_p(a), _p(b), _p(c), a.shape[0], _p(out), _stream()), 'kge_i64_max3')
    return _host_counts(out)


def filter_index_build(key1, key2, values, n_key1, n_key2, n_values, key2_span):
    """kge_filter_index_build: (keys, offsets, targets) of the sorted-key CSR, or None when the composite
    (key, value) does not fit one 64-bit radix key (the caller then takes the two-sort path)."""
    lib = load_library()
    require_cuda(key1, key2, values)
    key1, key2, values = i64c(key1), i64c(key2), i64c(values)
    n, dev = key1.shape[0], key1.device
    keys = torch.empty(n, dtype=torch.int64, device=dev)
    offsets = torch.empty(n + 1, dtype=torch.int64, device=dev)
    targets = torch.empty(n, dtype=torch.int32, device=dev)
    counts = torch.empty(2, dtype=torch.int64, device=dev)
    nb = int(lib.kge_filter_index_ws_bytes(n))
    ws = torch.empty(max(nb, 8), dtype=torch.uint8, device=dev)
    with _on(dev):
        rc = lib.kge_filter_index_build(_p(key1), _p(key2), _p(values), n, n_key1, n_key2, n_values, key2_span, _p(keys),
                                        _p(offsets), _p(targets), _p(counts), _p(ws), nb, _stream())
    if rc == KGE_EUNSUPPORTED:
        return None
    _check(rc, 'kge_filter_index_build')
    n_keys, n_t = _host_counts(counts)
    # (compact copies: the capacity-n buffers would pin 20 bytes per fact for the lifetime of the index)
    return keys[:n_keys].clone(), offsets[:n_keys + 1].clone(), targets[:n_t].clone()


def filter_plan_build(seg_lo, seg_hi, n_targets, long_len):
    """kge_filter_plan_build: (woff (n + 1), long_q (n_long), n_pairs)."""
    lib = load_library()
    require_cuda(seg_lo, seg_hi)
    n, dev = seg_lo.shape[0], seg_lo.device
    woff = torch.empty(n + 1, dtype=torch.int64, device=dev)
    long_q = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    counts = torch.empty(2, dtype=torch.int64, device=dev)
    nb = int(lib.kge_filter_plan_ws_bytes(n, n_targets))
    ws = torch.empty(max(nb, 8), dtype=torch.uint8, device=dev)
    with _on(dev):
        _check(lib.kge_filter_plan_build(_p(seg_lo), _p(seg_hi), n, n_targets, long_len, _p(woff), _p(long_q), _p(counts),
                                         _p(ws), nb, _stream()), 'kge_filter_plan_build')
    n_pairs, n_long = _host_counts(counts)
    return woff, long_q[:n_long], n_pairs


def column_plan_build(h, t, r, n_ent, n_rel, sets, pad, relation_major):
    """kge_column_plan_build + _emit: the tensors of filter_index.ColumnPlan, or None when the keys do not fit 64 bits."""
    lib = load_library()
    require_cuda(h, t, r)
    h, t, r = i64c(h), i64c(t), i64c(r)
    B, dev = h.shape[0], h.device
    n = 2 * B
    nb = int(lib.kge_column_plan_ws_bytes(n, n_ent, n_rel))
    ws = torch.empty(max(nb, 8), dtype=torch.uint8, device=dev)
    counts = torch.empty(3, dtype=torch.int64, device=dev)
    with _on(dev):
        rc = lib.kge_column_plan_build(_p(h), _p(t), _p(r), B, n_ent, n_rel, sets, 1 if relation_major else 0, _p(counts),
                                       _p(ws), nb, _stream())
    if rc == KGE_EUNSUPPORTED:
        return None
    _check(rc, 'kge_column_plan_build')
    n_chunks, n1, n_distinct = _host_counts(counts)
    n2 = n_chunks - n1
    n1p, n2p = (pad(n1) if n1 > 0 else 0), (pad(n2) if n2 > 0 else 0)
    col_q = torch.empty(max(n1p, 1), dtype=torch.int32, device=dev)
    members = torch.empty(max(n2p, 1) * sets, dtype=torch.int32, device=dev)
    qs_row = torch.empty(n, dtype=torch.int32, device=dev)
    col_of_q = torch.empty(n, dtype=torch.int64, device=dev)
    rep = torch.empty(max(n1p + n2p, 1), dtype=torch.int64, device=dev)
    with _on(dev):
        _check(lib.kge_column_plan_emit(B, n_ent, n_rel, sets, n1p, n2p, _p(col_q), _p(members), _p(qs_row), _p(col_of_q),
                                        _p(rep), _p(ws), nb, _stream()), 'kge_column_plan_emit')
    return {'n_single': n1, 'n_multi': n2, 'n_single_p': n1p, 'n_multi_p': n2p, 'col_q': col_q, 'members': members,
            'qs_row': qs_row, 'col_of_q': col_of_q, 'rep': rep, 'n_columns': n_chunks, 'n_distinct_keys': n_distinct}


def corrupt_scatter(heads, tails, mask_u8, draws_h, draws_t, n_neg):
    lib = load_library()
    require_cuda(heads, tails, mask_u8, draws_h, draws_t)
    heads, tails = i64c(heads), i64c(tails)
    draws_h, draws_t = i64c(draws_h), i64c(draws_t)
    B = heads.shape[0]
    n = B * n_neg
    dev = heads.device
    nh = torch.empty(n, dtype=torch.int64, device=dev)
    nt = torch.empty(n, dtype=torch.int64, device=dev)
    ws = torch.empty(int(lib.kge_corrupt_ws_elems(n)), dtype=torch.int32, device=dev)
    with _on(dev):
        _check(lib.kge_corrupt_scatter(_p(heads), _p(tails), _p(mask_u8), _p(draws_h), _p(draws_t),
                                       B, n_neg, _p(nh), _p(nt), _p(ws), _stream()),
               'kge_corrupt_scatter')
    return nh, nt
