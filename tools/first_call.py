#!/usr/bin/env python
"""What a reference-style script pays per evaluate(): the first call of the process, the first call of a FRESH
evaluator on the same graph (a script that builds its evaluator per epoch), and the steady state.
    python tools/first_call.py [workload]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchkge_amd as tk   # noqa: E402
import bench                # noqa: E402


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) * 1e3, 3)


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else 'transe_fb15k237'
    dev = torch.device('cuda', 0)
    model, tables, kg, kg_test, info = bench.build_workload(wl, dev, weights='xavier')
    out = {'workload': wl}
    ev = tk.LinkPredictionEvaluator(model, kg_test)
    out['first_evaluate_of_the_process_ms'] = timed(lambda: ev.evaluate(256, verbose=False))
    # graph=None ('auto'): a call whose capture key is new runs eagerly, the next one with the same key captures, later
    # ones replay.  Since r04 the second evaluation of an evaluator moves its filter correction to a second stream (new
    # key): call 2 is eager once more, call 3 captures, call 4 replays.
    out['second_call_ms'] = timed(lambda: ev.evaluate(256, verbose=False))
    out['third_call_ms'] = timed(lambda: ev.evaluate(256, verbose=False))
    out['fourth_call_ms'] = timed(lambda: ev.evaluate(256, verbose=False))
    fresh = []
    for _ in range(3):
        ev2 = tk.LinkPredictionEvaluator(model, kg_test)
        fresh.append(timed(lambda: ev2.evaluate(256, verbose=False)))
    out['first_evaluate_of_a_fresh_evaluator_ms'] = fresh
    ev3 = tk.LinkPredictionEvaluator(model, kg_test, graph=False)
    ev3.evaluate(256, verbose=False)
    out['eager_steady_ms'] = timed(lambda: [ev3.evaluate(256, verbose=False) for _ in range(10)]) / 10
    print(json.dumps(out))


if __name__ == '__main__':
    main()
