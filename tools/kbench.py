#!/usr/bin/env python
"""Micro-benchmark of single HIP entry points (HIP events on the launch stream).
    python tools/kbench.py [--B 20466 --N 14541 --d 200 --mode expand|dot|complex|l1|l2|axpy --what count|scores]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchkge_amd import _hip  # noqa: E402


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, default=20466)
    ap.add_argument('--N', type=int, default=14541)
    ap.add_argument('--d', type=int, default=200)
    ap.add_argument('--mode', default='expand')
    ap.add_argument('--what', default='count')
    ap.add_argument('--reps', type=int, default=10)
    a = ap.parse_args()
    g = torch.Generator().manual_seed(0)
    dev = 'cuda'
    Q = (torch.rand(a.B, a.d, generator=g) * 2 - 1).to(dev)
    T = (torch.rand(a.N, a.d, generator=g) * 2 - 1).to(dev)
    flops = 2 * a.d
    if a.mode == 'expand':
        prob = _hip.LpProblem(_hip.LP_L2_EXPAND, Q, T, qn=_hip.row_sqnorm(Q), en=_hip.row_sqnorm(T))
    elif a.mode == 'dot':
        prob = _hip.LpProblem(_hip.LP_DOT, Q, T)
    elif a.mode == 'complex':
        Q1 = (torch.rand(a.B, a.d, generator=g) * 2 - 1).to(dev)
        T1 = (torch.rand(a.N, a.d, generator=g) * 2 - 1).to(dev)
        prob = _hip.LpProblem(_hip.LP_DOT, Q, T, A1=Q1, T1=T1)
        flops = 4 * a.d
    elif a.mode in ('l1', 'l2'):
        prob = _hip.LpProblem(_hip.LP_L1_DIRECT if a.mode == 'l1' else _hip.LP_L2_DIRECT, Q, T)
        flops = 3 * a.d
    elif a.mode == 'axpy':
        W = (torch.rand(a.B, a.d, generator=g) - 0.5).to(dev)
        scal = (torch.rand(a.N, generator=g) - 0.5).to(dev)
        prob = _hip.LpProblem(_hip.LP_L2_DIRECT, Q, T, Wq=W, scal=scal)
        flops = 5 * a.d
    else:
        raise SystemExit('bad mode')
    ci = torch.randint(0, a.N, (a.B,), generator=g).to(dev)
    s_true = prob.pair_scores(ci)
    if a.what == 'count':
        raw = torch.zeros(a.B, dtype=torch.int32, device=dev)
        t = timeit(lambda: prob.count_ge(s_true, raw), a.reps)
    else:
        out = torch.empty(a.B, a.N, device=dev)
        t = timeit(lambda: prob.scores(out), a.reps)
    pairs = a.B * a.N
    print('%s %s B=%d N=%d d=%d env=%s : %.4f ms  %.3e pairs/s  %.1f TFLOP/s (%d flop/pair)' % (
        a.mode, a.what, a.B, a.N, a.d, os.environ.get('KGE_LP_TARGET_BLOCKS', '-'), t * 1e3, pairs / t,
        flops * pairs / t / 1e12, flops))


if __name__ == '__main__':
    main()
