"""Where the host time between two steady-state evaluate() calls goes (cfg2 by default): the pieces of
LinkPredictionEvaluator._evaluate_fast timed one by one with perf_counter_ns, the whole call, and the round trip of an
EMPTY graph replay + a 655 KB device-to-host copy + stream synchronisation (what no evaluate can go below).

    python tools/host_gap.py [--workload transe_fb15k237] [--reps 300]
"""
import argparse
import os
import sys
import time
import statistics as st
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench                                    # noqa: E402
import torchkge_amd as tk                       # noqa: E402
from torchkge_amd import evaluation as ev_mod   # noqa: E402


def med(xs):
    return st.median(xs) / 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='transe_fb15k237')
    ap.add_argument('--reps', type=int, default=300)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    model, _, kg, kg_test, _ = bench.build_workload(a.workload, dev)
    e = tk.LinkPredictionEvaluator(model, kg_test)
    for _ in range(12):
        e.evaluate(b_size=256, verbose=False)
    now = time.perf_counter_ns
    whole = []
    for _ in range(a.reps):
        t0 = now()
        e.evaluate(b_size=256, verbose=False)
        whole.append(now() - t0)
    fast = e._st._fast
    print('steady-state evaluate(): median %.1f us per call (fast path %s)' % (med(whole), 'armed' if fast else 'NOT armed'))
    if not fast:
        return
    info = fast[1]
    out = info['static']['out'][0]
    T = {k: [] for k in ('sig', 'replay_call', 'pinned_empty', 'copy_call', 'sync_wait', 'tolist_views')}
    for _ in range(a.reps):
        t0 = now()
        e._fast_sig(256)
        t1 = now()
        info['graph'].replay()
        t2 = now()
        host = torch.empty(out.shape, dtype=out.dtype, pin_memory=True)
        t3 = now()
        host.copy_(out, non_blocking=True)
        t4 = now()
        torch.cuda.current_stream(dev).synchronize()
        t5 = now()
        host[-2:].view(torch.float32).tolist()
        r = host[:-2].view(4, -1)
        _ = (r[0], r[1], r[2], r[3])
        t6 = now()
        for k, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
            T[k].append(v)
    print('pieces (median us): ' + ', '.join('%s %.1f' % (k, med(v)) for k, v in T.items()))
    # the floor: an empty graph, the same copy, the same synchronisation
    g = torch.cuda.CUDAGraph()
    z = torch.zeros(64, device=dev)
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        z.add_(1.0)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            z.add_(1.0)
    rt, rt_nocopy = [], []
    for _ in range(a.reps):
        t0 = now()
        g.replay()
        host = torch.empty(out.shape, dtype=out.dtype, pin_memory=True)
        host.copy_(out, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
        rt.append(now() - t0)
    for _ in range(a.reps):
        t0 = now()
        g.replay()
        torch.cuda.current_stream(dev).synchronize()
        rt_nocopy.append(now() - t0)
    print('empty graph replay + 655 KB D2H + sync: median %.1f us; without the copy %.1f us' % (med(rt), med(rt_nocopy)))
    # device time of one replay (events) against the host's period
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dts = []
    for _ in range(50):
        e0.record()
        info['graph'].replay()
        e1.record()
        torch.cuda.synchronize()
        dts.append(e0.elapsed_time(e1) * 1e3)
    print('device time of one graph replay (events): median %.1f us' % st.median(dts))


if __name__ == '__main__':
    main()
