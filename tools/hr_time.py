"""Time the DOT candidate-table preparation (kge_lp_dot_table_prep: norm maxima + fragment-major hi table) at a given
shape, the coalesced kernel (r06) against the general one (KGE_HIROWS_OLD=1), and check that both write the same table.

    python tools/hr_time.py [--n 4594485] [--d 512] [--cplx 1]
"""
import argparse
import os
import subprocess
import sys
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def child(a):
    from torchkge_amd import _hip as hip
    g = torch.Generator(device='cuda').manual_seed(1)
    T0 = torch.randn(a.n, a.d, device='cuda', generator=g) * 0.05
    T1 = torch.randn(a.n, a.d, device='cuda', generator=g) * 0.05 if a.cplx else None
    ms = []
    for i in range(a.reps):
        g2 = torch.zeros(8, device='cuda')
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        Eh, dnb, _ws = hip.dot_table_prep(T0, T1, g2[1:2], g2[5:6] if a.cplx else None, True)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    v = Eh.view(torch.int32)
    # a checksum of the table (int64 sum of the words, order-free) + the residual maximum
    print('RESULT', os.environ.get('KGE_HIROWS_OLD', '0'), ' '.join('%.3f' % m for m in ms), int(v.to(torch.int64).sum()),
          int((v.to(torch.int64) * (torch.arange(v.numel(), device='cuda') % 1021 + 1).view_as(v)).sum()),
          '%.9e' % float(dnb.max()), flush=True)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=4594485)
    ap.add_argument('--d', type=int, default=512)
    ap.add_argument('--cplx', type=int, default=1)
    ap.add_argument('--reps', type=int, default=6)
    ap.add_argument('--child', action='store_true')
    a = ap.parse_args()
    if a.child:
        child(a)
    else:
        for old in ('0', '1', '0', '1'):
            env = dict(os.environ, KGE_HIROWS_OLD=old)
            subprocess.run([sys.executable, os.path.abspath(__file__), '--child', '--n', str(a.n), '--d', str(a.d),
                            '--cplx', str(a.cplx), '--reps', str(a.reps)], env=env, check=True)
