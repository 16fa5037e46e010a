"""The chunked-panel one-product count kernel (lp_hi_chunk.hip) alone: counts against the exact fp32 kernel, then HIP-event
timings of every count kernel that can run the shape (chunked NT = 3 / 4, r04's block-synchronous planar kernel, the
three-product kernel) and of the exact recheck.  Operands are generated on the GPU (cfg5 shape: 18.8 GB of fp32 table).

    B=10266 N=4594485 K=1024 python tools/hc_time.py            # cfg5 shape, L2 proxy (same sweep as DOT: 65 units)
    CHECK=1 B=2048 N=200000 K=1024 python tools/hc_time.py      # + counts == exact
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchkge_amd import _hip  # noqa: E402

B, N, K = int(os.environ.get('B', 10266)), int(os.environ.get('N', 4594485)), int(os.environ.get('K', 1024))
CHECK = os.environ.get('CHECK', '0') == '1'
NOISE = float(os.environ.get('NOISE', '0.5'))      # |q - e_true| relative to 1: smaller = thresholds further out in the tail
REPS = int(os.environ.get('REPS', '5'))
VARIANTS = os.environ.get('VARIANTS', 'hc4,hc3,lv1,lv0').split(',')
dev = torch.device('cuda')
g = torch.Generator(device=dev).manual_seed(0)


def unit_rows(n, k):
    x = torch.empty(n, k, device=dev)
    step = max(1, (1 << 28) // k)
    for i in range(0, n, step):
        blk = torch.randn(min(step, n - i), k, device=dev, generator=g)
        x[i:i + step] = torch.nn.functional.normalize(blk, dim=1)
    return x


E = unit_rows(N, K)
t = torch.randint(0, N, (B,), device=dev, generator=g)
q = (E[t] + NOISE * unit_rows(B, K)).contiguous()
guard = torch.zeros(8, device=dev)
en = _hip.row_sqnorm(E, max_io=guard[1:2])
qn = _hip.row_sqnorm(q, max_io=guard[0:1])
prob = _hip.LpProblem(_hip.LP_L2_EXPAND, q, E, qn=qn, en=en)
st = prob.pair_scores(t)
exact = prob.count_ge(st) if CHECK else None
torch.cuda.synchronize()
print('problem B=%d N=%d K=%d units=%d noise=%g  (table %.1f GB fp32)' % (B, N, K, (K + 2 + 15) // 16, NOISE, N * K * 4 / 1e9),
      flush=True)


def timed(fn, reps=REPS):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


tables = {}


def split_for(variant):
    if variant in ('hc3', 'hc4', 'lv1', 'hs', 'hs3', 'hs4'):
        frag = variant != 'lv1'
        key = 'hi%d' % frag
        if key not in tables:
            tables[key] = _hip.hi_table(E, aug=en, frag=frag)
        Eh, de2 = tables[key]
        return {'Es': Eh, 'e2pref': None, 'enmax': guard[1:2], 'overflow': guard[2:3], 'level': 1, 'de2max': de2,
                'list_stat': guard[6:7], 'es_frag': frag}
    if 'lv0' not in tables:
        tables['lv0'] = _hip.split_table(E, aug=en)
    Es, e2 = tables['lv0']
    return {'Es': Es, 'e2pref': e2, 'enmax': guard[1:2], 'overflow': guard[2:3]}


for variant in VARIANTS:
    os.environ.pop('KGE_HC_FORCE', None)    # 'hs': the resident-panel free-running kernel (rows <= 32 units)
    os.environ.pop('KGE_HS_NT', None)
    if variant in ('hs3', 'hs4'):           # ... with 96- / 128-query panels
        os.environ['KGE_HS_NT'] = variant[2]
    if variant in ('hc3', 'hc4'):
        os.environ['KGE_HC_NT'] = variant[2]
        os.environ['KGE_HC_FORCE'] = '1'
    guard[2] = 0
    prob.split = split_for(variant)
    prep = prob.split_prepare()
    raw = torch.zeros(B, dtype=torch.int32, device=dev)
    prob.split_count(prep, st, raw)
    prob.split_recheck(prep, st, raw)
    torch.cuda.synchronize()
    n_list = int(prep['n_list'].item())
    msg = ''
    if CHECK:
        bad = int((raw != exact).sum().item())
        msg = '  counts != exact: %d / %d (max |diff| %d)' % (bad, B, int((raw - exact).abs().max().item()))
    ms_c = timed(lambda: prob.split_count(prep, st, raw))
    ms_r = timed(lambda: prob.split_recheck(prep, st, raw))
    units = (K + 2 + 15) // 16
    mfma_flop = 2.0 * 16 * units * B * N * (3 if variant == 'lv0' else 1)
    print('%-4s count %.3f ms (%.0f TF executed, %.0f TF algorithmic = %.3f of 2.5 PF)  recheck %.3f ms  listed %d (%.1f per query) '
          'overflow %g%s' % (variant, ms_c, mfma_flop / ms_c / 1e9, 2.0 * K * B * N / ms_c / 1e9,
                             2.0 * K * B * N / ms_c / 1e9 / 2500.0, ms_r, n_list, n_list / B, float(guard[2]), msg), flush=True)
    del prep, raw
    if variant == 'lv1':
        tables.pop('hi0', None)
    torch.cuda.empty_cache()
