#!/usr/bin/env python
"""Cycle breakdown of lp_gemm_kernel from an instrumented build (libkge_hip_timing.so,
built ad hoc from lp_gemm_mfma.hip with s_memtime brackets; not part of the product)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchkge_amd import _hip
_hip.LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), 'libkge_hip_timing.so')
lib = _hip.load_library()
B, N, d = 20466, 14541, int(os.environ.get('D', 200))
g = torch.Generator().manual_seed(0)
Q = (torch.rand(B, d, generator=g) * 2 - 1).cuda(); T = (torch.rand(N, d, generator=g) * 2 - 1).cuda()
prob = _hip.LpProblem(_hip.LP_L2_EXPAND, Q, T, qn=_hip.row_sqnorm(Q), en=_hip.row_sqnorm(T))
ci = torch.randint(0, N, (B,), generator=g).cuda()
st = prob.pair_scores(ci)
raw = torch.zeros(B, dtype=torch.int32, device='cuda')
for _ in range(3): prob.count_ge(st, raw)
torch.cuda.synchronize(); lib.kge_timing_reset()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); prob.count_ge(st, raw); e1.record(); torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)(); lib.kge_timing_read(out)
tot, stage, bar, epi, waves, steps = [out[i] for i in range(6)]
ms = e0.elapsed_time(e1)
print('kernel %.3f ms; waves %d; steps/wave %.1f' % (ms, waves, steps / waves))
print('per wave cycles: total %.0f  stage(vmcnt+ds_write) %.0f (%.1f%%)  barrier %.0f (%.1f%%)  epilogue+flush %.0f (%.1f%%)' % (
    tot / waves, stage / waves, 100 * stage / tot, bar / waves, 100 * bar / tot, epi / waves, 100 * epi / tot))
print('per step: total %.0f  stage %.0f  barrier %.0f  epi %.0f ; cycle counter rate = %.3f GHz-equivalent' % (
    tot / steps, stage / steps, bar / steps, epi / steps, tot / waves / (ms * 1e6)))
