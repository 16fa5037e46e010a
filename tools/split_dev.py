"""Development probe for the f16-split rank prefilter: exact vs split counts,
band occupancy, timing.  Run on the GPU box: python tools/split_dev.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchkge_amd import _hip  # noqa: E402


def problem(B, N, K, seed=0, scale_e=1.0):
    g = torch.Generator().manual_seed(seed)
    E = torch.nn.functional.normalize(torch.randn(N, K, generator=g), dim=1) * scale_e
    R = torch.nn.functional.normalize(torch.randn(50, K, generator=g), dim=1)
    h = torch.randint(0, N, (B,), generator=g)
    r = torch.randint(0, 50, (B,), generator=g)
    t = torch.randint(0, N, (B,), generator=g)
    E, q, t = E.cuda(), (E[h] + R[r]).cuda().contiguous(), t.cuda()
    return E, q, t


def run(B, N, K, eps_scale=1.0, timing=False, scale_e=1.0):
    E, q, t = problem(B, N, K, scale_e=scale_e)
    guard = torch.zeros(4, device='cuda')
    en = _hip.row_sqnorm(E, max_io=guard[1:2])
    qn = _hip.row_sqnorm(q, max_io=guard[0:1])
    prob = _hip.LpProblem(_hip.LP_L2_EXPAND, q, E, qn=qn, en=en)
    st = prob.pair_scores(t)
    exact = prob.count_ge(st)
    Es = _hip.split_rows(E, aug=en)
    prob.split = {'Es': Es, 'enmax': guard[1:2], 'overflow': guard[2:3]}
    _hip.SPLIT_EPS_SCALE = eps_scale
    got = prob.count_ge(st)
    torch.cuda.synchronize()
    n_list = int(prob.last_split[0].item())
    bad = int((got != exact).sum().item())
    print('B=%d N=%d K=%d eps_scale=%g: mismatching queries %d / %d, uncertain pairs %d (%.2f per query), '
          'overflow %g, max|diff| %d' % (B, N, K, eps_scale, bad, B, n_list, n_list / max(B, 1),
                                         float(guard[2]), int((got - exact).abs().max().item()) if B else 0))
    if timing:
        for name, sp in (('exact fp32 MFMA', None), ('f16 split + recheck', prob.split)):
            prob.split = sp
            for _ in range(3):
                prob.count_ge(st)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                prob.count_ge(st)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10
            print('   %-22s %.3f ms  (%.1f Gpairs/s)' % (name, dt * 1e3, B * N / dt / 1e9))
    return bad


if __name__ == '__main__':
    mode = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if mode in ('all', 'small'):
        run(64, 300, 32)
        run(1000, 3000, 200)
        run(193, 257, 17)
        run(5, 2, 1)
        run(1000, 3000, 200, scale_e=1.7)
    if mode in ('all', 'eps'):
        for e in (1.0, 0.25, 1 / 16, 1 / 64, 1 / 256):
            run(4096, 14541, 200, eps_scale=e)
    if mode in ('all', 'time'):
        run(32768, 14541, 200, timing=True)
