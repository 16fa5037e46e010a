"""Random shapes through the one-product level of the split prefilter (free-running kernel with 128- and 96-query panels,
chunked-panel kernel for long rows): counts after the exact recheck == the exact fp32 counts, no overflow.

    python tools/fuzz_level1.py [--cases 80] [--seed 1]
"""
import argparse
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchkge_amd import _hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=80)
    ap.add_argument('--seed', type=int, default=1)
    a = ap.parse_args()
    rnd = random.Random(a.seed)
    dev = torch.device('cuda')
    g = torch.Generator(device=dev).manual_seed(a.seed)
    bad = 0
    for case in range(a.cases):
        K = rnd.choice([8, 16, 40, 64, 100, 128, 200, 204, 208, 256, 300, 400, 416, 500, 512, 1024])
        B = rnd.choice([1, 5, 31, 32, 33, 95, 96, 97, 127, 128, 129, 191, 192, 193, 255, 257, 400, 700])
        N = rnd.choice([64, 65, 200, 255, 256, 257, 511, 513, 1000, 2047, 3000, 5000])
        nt = rnd.choice(['3', '4'])
        os.environ['KGE_HS_NT'] = nt
        os.environ['KGE_HC_NT'] = nt
        E = torch.nn.functional.normalize(torch.randn(N, K, device=dev, generator=g), dim=1) * (0.5 + rnd.random())
        t = torch.randint(0, N, (B,), device=dev, generator=g)
        q = (E[t] + (0.2 + rnd.random()) * torch.nn.functional.normalize(torch.randn(B, K, device=dev, generator=g), dim=1)).contiguous()
        guard = torch.zeros(8, device=dev)
        en = _hip.row_sqnorm(E, max_io=guard[1:2])
        qn = _hip.row_sqnorm(q, max_io=guard[0:1])
        prob = _hip.LpProblem(_hip.LP_L2_EXPAND, q, E, qn=qn, en=en)
        st = prob.pair_scores(t)
        exact = prob.count_ge(st)
        units = (K + 2 + 15) // 16
        frag = units <= 32 or units in (33, 65)
        Eh, de2 = _hip.hi_table(E, aug=en, frag=frag)
        prob.split = {'Es': Eh, 'e2pref': None, 'enmax': guard[1:2], 'overflow': guard[2:3], 'level': 1, 'de2max': de2,
                      'list_stat': guard[6:7], 'es_frag': frag}
        prep = prob.split_prepare()
        raw = torch.zeros(B, dtype=torch.int32, device=dev)
        prob.split_count(prep, st, raw)
        prob.split_recheck(prep, st, raw)
        torch.cuda.synchronize()
        ov = float(guard[2])
        nbad = int((raw != exact).sum())
        if nbad and ov == 0:
            bad += 1
        print('case %3d K=%4d B=%4d N=%5d NT=%s frag=%d: %s (overflow %g)' % (case, K, B, N, nt, frag,
              'ok' if nbad == 0 else ('%d COUNTS DIFFER' % nbad if ov == 0 else 'list overflow (counts not final)'), ov), flush=True)
    print('FUZZ', 'FAILED: %d cases' % bad if bad else 'passed', flush=True)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
