#!/usr/bin/env python
"""Cycle stamps of the phases of the one-product count kernel's stage body (probe 4096 of the DBG instantiation): block 0,
waves 0 and 4 (partners on one SIMD), the first 48 stages = 12 tiles at K = 200.  Prints the mean cycles per phase and a raw
excerpt (profiles/r04/split_timeline.txt).     TAIL=1 python tools/split_timeline.py"""
import os
import sys

os.environ['KGE_SPLIT_DBG'] = str(4096 | 64 | int(os.environ.get('EXTRA_DBG', '0')))
os.environ['LEVEL'] = '1'
os.environ.setdefault('TAIL', '1')
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'split_time.py')).read().split("for name, fn in")[0]
exec(compile(src, 'split_time_head', 'exec'))
for _ in range(5):
    count()
torch.cuda.synchronize()
NS = 48
ts = prep['list'].view(-1)[:2 * NS * 16 * 2].view(torch.int64).view(2, NS, 16).cpu().double()
names = ['wait frag u0', 'MFMA u0 (6)', 'DMA E0 E1', 'issue frag u1', 'wait frag u1', 'MFMA u1', 'DMA E2 E3',
         'frag u2 issue+wait', 'MFMA u2', 'DMA Q0-2 + next item', 'frag u3 issue+wait', 'vmcnt(0)', 'barrier',
         'frag u0 next + MFMA u3', 'epilogue / tile change']
S = 4
print('clock cycles (s_memtime), LEVEL=1, dbg=%s' % os.environ['KGE_SPLIT_DBG'])
for w in (0, 1):
    d = ts[w, :, 1:] - ts[w, :, :-1]
    gap = ts[w, 1:, 0] - ts[w, :-1, 15]
    stage_len = ts[w, 1:, 0] - ts[w, :-1, 0]
    last = torch.arange(NS) % S == S - 1
    print('wave %d: mean stage length %.0f cycles (tile = 4 stages: %.0f); by stage of the tile: %s' % (
        w * 4, stage_len.mean(), stage_len.mean() * S,
        ' '.join('%.0f' % stage_len[(torch.arange(NS - 1) % S) == k].mean() for k in range(S))))
    for i, n in enumerate(names):
        print('   %-28s all stages %7.0f   last stage of a tile %7.0f   others %7.0f' % (
            n, d[:, i].mean(), d[last, i].mean(), d[~last, i].mean()))
    print('   %-28s %7.0f' % ('loop back-edge', gap.mean()))
skew = ts[1, :, 0] - ts[0, :, 0]
print('wave 4 enters a stage %.0f cycles after wave 0 on average (min %.0f, max %.0f)' % (skew.mean(), skew.min(), skew.max()))
print('raw: stage, then cycles since the stage top of wave 0 at each stamp, wave 0 / wave 4')
for g in range(8, 16):
    base = ts[0, g, 0]
    print(g, ' '.join('%5.0f' % (x - base) for x in ts[0, g]), '|', ' '.join('%5.0f' % (x - base) for x in ts[1, g]))
