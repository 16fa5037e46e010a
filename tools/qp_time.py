"""Timing of the fused query-side launches alone (HIP events): the TransE-L2 pipeline (KGE_QP_DBG probes) and the DOT
pipeline against the launches it replaces.  KIND=transe|distmult|complex B=.. N=.. D=.."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchkge_amd import _hip as hip  # noqa: E402

KIND = os.environ.get('KIND', 'transe')
B, N, D, R = int(os.environ.get('B', 20466)), int(os.environ.get('N', 14541)), int(os.environ.get('D', 200)), 237
g = torch.Generator().manual_seed(0)
E = torch.nn.functional.normalize(torch.randn(N, D, generator=g), dim=1).cuda()
E1 = torch.nn.functional.normalize(torch.randn(N, D, generator=g), dim=1).cuda()
Rl = (0.1 * torch.randn(R, D, generator=g)).cuda()
R1 = (0.1 * torch.randn(R, D, generator=g)).cuda()
h = torch.randint(0, N, (B,), generator=g).cuda(); t = torch.randint(0, N, (B,), generator=g).cuda()
r = torch.randint(0, R, (B,), generator=g).cuda()
guard = torch.zeros(8, device='cuda')


def timed(name, fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    print('%-40s %s B=%d N=%d D=%d dbg=%s: %.1f us' % (name, KIND, B, N, D, os.environ.get('KGE_QP_DBG', '0'),
                                                     1e3 * a.elapsed_time(b) / n))


if KIND == 'transe':
    en, Eh, tpb = hip.table_prep_l2(E, guard[1:2], guard[7:8], deferred_max=True)
    timed('table_prep_l2', lambda: hip.table_prep_l2(E, guard[1:2], guard[7:8], deferred_max=True))
    timed('query_pipeline(level 1, both)', lambda: hip.lp_query_pipeline(hip.SIDE_BOTH, E, Rl, h, t, r, en, guard[1:2], guard[0:1],
                                                                         level=1, de2max=guard[7:8], tp_bmax=tpb,
                                                                         zero_counts=True))
else:
    cplx = KIND == 'complex'
    T1 = E1 if cplx else None
    hip.row_sqnorm(E, max_io=guard[1:2], bound_only=True)
    if cplx:
        hip.row_sqnorm(E1, max_io=guard[5:6], bound_only=True)
    nm1 = guard[5:6] if cplx else None
    Eh, de2 = hip.hi_table(E, X1=T1, dot=True, nmax0=guard[1:2], nmax1=nm1, frag=True)
    timed('dot_query_pipeline(both)', lambda: hip.lp_dot_query_pipeline(hip.SIDE_BOTH, E, T1, Rl, R1 if cplx else None, h, t, r,
                                                                        guard[1:2], nm1, de2, guard[0:1], guard[2:3],
                                                                        zero_counts=True))
    k = hip.COMPLEX if cplx else hip.DISTMULT
    tabs = [E, E1, Rl, R1] if cplx else [E, Rl]
    true = torch.cat([t, h])

    def separate():
        Q0, Q1, _, _ = hip.lp_prep(k, hip.SIDE_BOTH, tabs, D, D, h, t, r, want_q1=cplx) if cplx else \
            hip.lp_prep(k, hip.SIDE_BOTH, tabs, D, D, h, t, r)
        prob = hip.LpProblem(hip.LP_DOT, Q0, E, A1=Q1 if cplx else None, T1=T1)
        prob.pair_scores(true)
        qmax = torch.zeros(2, device='cuda')
        qn0 = hip.row_sqnorm(Q0, max_io=qmax[0:1], bound_only=True)
        qn = qn0 + hip.row_sqnorm(Q1, max_io=qmax[1:2], bound_only=True) if cplx else qn0
        hip.hi_rows(Q0, is_query=True, aug=qn, X1=Q1 if cplx else None, dot=True, nmax0=qmax[0:1],
                    nmax1=qmax[1:2] if cplx else None, want_dn2=True)
    timed('separate launches (without thresholds)', separate)
