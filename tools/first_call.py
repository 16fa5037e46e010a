"""Wall time of the first, second and third LinkPredictionEvaluator.evaluate() on a fresh evaluator (cfg2 shape):
filter index + plans + eager warm-up, hipGraph capture, replay."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torchkge_amd as tk  # noqa: E402

dev = torch.device('cuda:0')
model, tables, kg, kg_test, info = bench.build_workload(sys.argv[1] if len(sys.argv) > 1 else 'transe_fb15k237', dev,
                                                        weights='xavier', kg_kind='zipf')
torch.cuda.synchronize()
for b in (256, 32768):
    t0 = time.perf_counter()
    ev = tk.LinkPredictionEvaluator(model, kg_test)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ts = []
    for _ in range(4):
        ta = time.perf_counter()
        ev.evaluate(b, verbose=False)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - ta) * 1e3)
    print('b_size=%d: constructor %.1f ms, evaluate calls %s ms' % (b, (t1 - t0) * 1e3, ', '.join('%.2f' % x for x in ts)))
