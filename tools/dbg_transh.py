import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torchkge_amd as tk
from torchkge_amd import _hip
dev = torch.device('cuda', 0)
wl = sys.argv[1] if len(sys.argv) > 1 else 'transh_fb15k237'
model, tables, kg, kg_test, info = bench.build_workload(wl, dev, weights='trained', kg_kind='zipf', train_cfg={'steps': 300})
def ranks(ev):
    return torch.stack([ev.rank_true_heads, ev.rank_true_tails, ev.filt_rank_true_heads, ev.filt_rank_true_tails])
def run(level, stream, graph=False, split=True):
    model.split_level = level
    model.lp_hi_stream = stream
    model.split_filter = split
    ev = tk.LinkPredictionEvaluator(model, kg_test, graph=graph, share_state=False)
    ev.evaluate(32768, verbose=False)
    if graph:
        ev.evaluate(32768, verbose=False)
    return ranks(ev), ev.last_rescored_per_query
exact, _ = run(0, False, split=False)
for name, args in (('level0', (0, False)), ('level1 old kernel', (1, False)), ('level1 stream', (1, True)),
                   ('level1 stream graph', (1, True, True))):
    r, rs = run(*args)
    d = (r != exact)
    print('%-22s ranks differing from exact fp32 counts: %d (raw %d, filt %d)  rescored/query %s  max|d| %d' % (
        name, int(d.sum()), int(d[:2].sum()), int(d[2:].sum()), rs, int((r - exact).abs().max())))
    if int(d.sum()):
        idx = d.nonzero()[:8]
        print('   first:', [(int(a), int(b), int(r[a, b]), int(exact[a, b])) for a, b in idx])
