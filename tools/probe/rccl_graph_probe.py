"""Can RCCL collectives be captured in a hipGraph with torch.distributed here?  (world of one, backend nccl)"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29655')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
dist.init_process_group('nccl', rank=0, world_size=1)
dev = torch.device('cuda:0')
x = torch.arange(1 << 16, dtype=torch.int32, device=dev)
y = torch.empty_like(x)
blk = torch.randn(1024, 200, device=dev)
out = torch.empty(1024, 200, device=dev)
dist.all_reduce(x)                       # warm-up: communicator creation outside capture
dist.all_gather_into_tensor(out, blk)
torch.cuda.synchronize()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    dist.all_reduce(x)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        y.copy_(x)
        dist.all_reduce(y)
        dist.all_gather_into_tensor(out, blk)
        y.add_(1)
    torch.cuda.synchronize()
    x.fill_(3)
    g.replay()
    torch.cuda.synchronize()
    print('capture + replay OK:', int(y[0]), '(expect 4)', bool(torch.equal(out, blk)))
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    print('graph replay with 2 collectives: %.1f us' % ((time.perf_counter() - t0) / 200 * 1e6))
    t0 = time.perf_counter()
    for _ in range(200):
        y.copy_(x); dist.all_reduce(y); dist.all_gather_into_tensor(out, blk); y.add_(1)
    torch.cuda.synchronize()
    print('eager: %.1f us' % ((time.perf_counter() - t0) / 200 * 1e6))
except Exception as e:  # noqa
    print('capture FAILED:', type(e).__name__, str(e)[:300])
dist.destroy_process_group()
