"""Timing of the candidate-table preparation kernels alone (HIP events): kge_lp_table_prep_l2 against row_sqnorm + hi_rows_frag."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchkge_amd import _hip  # noqa: E402

N, K = int(os.environ.get('N', 14541)), int(os.environ.get('K', 200))
E = torch.nn.functional.normalize(torch.randn(N, K), dim=1).cuda()
g = torch.zeros(8, device='cuda')


def fused():
    _hip.table_prep_l2(E, g[1:2], g[7:8])


def separate():
    en = _hip.row_sqnorm(E, max_io=g[1:2])
    _hip.hi_table(E, aug=en, frag=True)


def sqnorm():
    _hip.row_sqnorm(E, max_io=g[1:2])


for name, fn in (('fused', fused), ('separate', separate), ('row_sqnorm', sqnorm)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        fn()
    b.record()
    torch.cuda.synchronize()
    print('%s N=%d K=%d dbg=%s: %.1f us per call (eager launches back to back)' % (
        name, N, K, os.environ.get('KGE_TP_DBG', '0'), a.elapsed_time(b) * 1e3 / 50))
