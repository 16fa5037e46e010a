"""Timing of the split count kernel alone (HIP events), for KGE_SPLIT_DBG probes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchkge_amd import _hip  # noqa: E402
from tools.split_dev import problem  # noqa: E402

B, N, K = int(os.environ.get('B', 32768)), int(os.environ.get('N', 14541)), int(os.environ.get('K', 200))
E, q, t = problem(B, N, K)
guard = torch.zeros(8, device='cuda')
if os.environ.get('TAIL'):      # thresholds in the sparse upper tail (a fitted model): the true entity is the query's neighbour
    q = (E[t] + 0.5 * torch.nn.functional.normalize(torch.randn_like(q), dim=1)).contiguous()
en = _hip.row_sqnorm(E, max_io=guard[1:2])
qn = _hip.row_sqnorm(q, max_io=guard[0:1])
prob = _hip.LpProblem(_hip.LP_L2_EXPAND, q, E, qn=qn, en=en)
st = prob.pair_scores(t)
LEVEL = int(os.environ.get('LEVEL', '0'))      # 1: the one-product level (planar hi operands)
if LEVEL == 1:
    FRAG = os.environ.get('FRAG', '1') == '1'   # the free-running kernel (fragment-major candidate table)
    _Eh, _de2 = _hip.hi_table(E, aug=en, frag=FRAG)
    prob.split = {'Es': _Eh, 'e2pref': None, 'enmax': guard[1:2], 'overflow': guard[2:3], 'level': 1, 'de2max': _de2,
                  'list_stat': guard[6:7], 'es_frag': FRAG}
else:
    _Es, _e2 = _hip.split_table(E, aug=en)
    prob.split = {'Es': _Es, 'e2pref': None if os.environ.get('NO_PREF') else _e2, 'enmax': guard[1:2], 'overflow': guard[2:3]}
_hip.SPLIT_EPS_SCALE = float(os.environ.get('EPS', '1'))
prep = prob.split_prepare()
raw = torch.zeros(B, dtype=torch.int32, device='cuda')
nl = prep['n_list']


def count():
    prob.split_count(prep, st, raw)


def recheck():
    prob.split_recheck(prep, st, raw)


for name, fn in (('count', count), ('recheck', recheck)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        fn()
    b.record()
    torch.cuda.synchronize()
    print('%s level=%d frag=%s B=%d N=%d K=%d dbg=%s hs_waves=%s hs_qg=%s: %.3f ms (pairs listed %d)' % (
        name, LEVEL, os.environ.get('FRAG', '1') if LEVEL == 1 else '-', B, N, K, os.environ.get('KGE_SPLIT_DBG', '0'),
        os.environ.get('KGE_HS_WAVES', 'auto'), os.environ.get('KGE_HS_QG', 'auto'), a.elapsed_time(b) / 10, int(nl.item())))
