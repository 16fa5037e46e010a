"""Epilogue variants of the free-running one-product kernel (KGE_HS_PROBE = 32 / 64 / 96: VALID results): counts against the
exact fp32 counts, then the count kernel's time (HIP events, alternating)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchkge_amd import _hip  # noqa: E402
from tools.split_dev import problem  # noqa: E402

B, N, K = int(os.environ.get('B', 40932)), 14541, 200
E, q, t = problem(B, N, K)
q = (E[t] + 0.5 * torch.nn.functional.normalize(torch.randn_like(q), dim=1)).contiguous()
guard = torch.zeros(8, device='cuda')
en = _hip.row_sqnorm(E, max_io=guard[1:2]); qn = _hip.row_sqnorm(q, max_io=guard[0:1])
prob = _hip.LpProblem(_hip.LP_L2_EXPAND, q, E, qn=qn, en=en)
st = prob.pair_scores(t)
exact = prob.count_ge(st)
Eh, de2 = _hip.hi_table(E, aug=en, frag=True)
prob.split = {'Es': Eh, 'e2pref': None, 'enmax': guard[1:2], 'overflow': guard[2:3], 'level': 1, 'de2max': de2,
              'list_stat': guard[6:7], 'es_frag': True}
prob.split_true = (st, t)
variants = ('0', '32', '64', '96')
for v in variants:
    os.environ['KGE_HS_PROBE'] = v
    got = prob.count_ge(st)
    print('variant %s: mismatching queries %d, listed %d' % (v, int((got != exact).sum()), int(prob.last_split[0].item())))
prep = prob.split_prepare()
raw = torch.zeros(B, dtype=torch.int32, device='cuda')
res = {v: [] for v in variants}
for rnd in range(4):
    for v in variants:
        os.environ['KGE_HS_PROBE'] = v
        for _ in range(3):
            prob.split_count(prep, st, raw)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            prob.split_count(prep, st, raw)
        b.record()
        torch.cuda.synchronize()
        res[v].append(a.elapsed_time(b) / 20)
for v in variants:
    print('variant %s: count %.4f ms (rounds: %s)' % (v, sum(res[v]) / len(res[v]), ' '.join('%.4f' % x for x in res[v])))
