"""scoring_function (K1) time per model kind at the training batch size:  python tools/k1_time.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchkge_amd as tk  # noqa: E402
from oracle import kge_oracle as orc  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
n_ent, n_rel = 14541, 237
for kind, d in (('transe', 200), ('transe_l1', 200), ('transh', 200), ('transd', 200), ('distmult', 200), ('complex', 200),
                ('transe', 100), ('transe', 400)):
    k = 'transe' if kind.startswith('transe') else kind
    tables = orc.init_tables(k, n_ent, n_rel, d, seed=0)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import bench
    m = bench.make_model(k, 1 if kind == 'transe_l1' else 2, tables, n_ent, n_rel).cuda()
    h, t, r = orc.synthetic_triples(n_ent, n_rel, B, seed=3, device='cuda')
    with torch.no_grad():
        for _ in range(5):
            m.scoring_function(h, t, r)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()          # device time: eager back-to-back calls are host bound (~20 us per call)
        with torch.cuda.graph(g):
            for _ in range(50):
                m.scoring_function(h, t, r)
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(4):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    nrows = {'transe': 3, 'transh': 4, 'transd': 5, 'distmult': 3, 'complex': 6}[k]
    byt = B * (nrows * d * 4 + 28)
    print('%-10s d=%d B=%d blocks=%s: %.1f us  %.0f GB/s (%.3f of 8 TB/s)' % (
        kind, d, B, os.environ.get('KGE_K1_BLOCKS', 'default'), us, byt / us / 1e3, byt / us / 1e3 / 8000))
