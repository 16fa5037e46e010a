#!/usr/bin/env python
"""Parts of the first evaluate() of a process (wall ms, one MI355X): with WARM=1 one tiny library launch (+ one ATen fill)
comes first, so the module load / runtime start-up is timed apart from the evaluate itself.
    [WARM=1] python tools/first_call_parts.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchkge_amd as tk   # noqa: E402
import bench                # noqa: E402
from torchkge_amd import _hip   # noqa: E402


def ms(t0):
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) * 1e3, 2)


dev = torch.device('cuda', 0)
model, tables, kg, kg_test, info = bench.build_workload('transe_fb15k237', dev, weights='xavier')
torch.cuda.synchronize()
out = {'warm': os.environ.get('WARM', '0')}
if out['warm'] == '1':
    t0 = time.perf_counter(); x = torch.ones(4, 8, device=dev); _hip.row_sqnorm(x); out['first_library_launch_ms'] = ms(t0)
if out['warm'] == '2':      # an ATen fill only
    t0 = time.perf_counter(); x = torch.ones(4, 8, device=dev); out['aten_fill_ms'] = ms(t0)
if out['warm'] == '3':      # the library loaded, nothing launched
    t0 = time.perf_counter(); _hip.load_library(); out['load_library_ms'] = ms(t0)
if out['warm'] == '4':      # a library launch on an existing tensor (no ATen kernel)
    x = next(model.parameters()).detach()
    t0 = time.perf_counter(); _hip.row_sqnorm(x); out['first_library_launch_ms'] = ms(t0)
t0 = time.perf_counter(); hd = kg.head_idx.to(dev); out['first_h2d_copy_of_2.5MB_ms'] = ms(t0)
t0 = time.perf_counter(); hd = kg.tail_idx.to(dev); out['second_h2d_copy_ms'] = ms(t0)
t0 = time.perf_counter(); fi = kg.filter_index('heads', dev); out['filter_index_heads_ms'] = ms(t0)
t0 = time.perf_counter(); fi = kg.filter_index('tails', dev); out['filter_index_tails_ms'] = ms(t0)
ev = tk.LinkPredictionEvaluator(model, kg_test)
t0 = time.perf_counter(); ev.evaluate(256, verbose=False); out['first_evaluate_ms'] = ms(t0)
print(out)
