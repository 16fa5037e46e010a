"""cProfile of the STEADY-STATE evaluate() (graph replay) -- where the host time between two replays goes.
WL=workload (default transe_fb15k237), bench.py's trained-like weights."""
import cProfile, pstats, sys, os, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torchkge_amd as tk, bench
dev = torch.device('cuda', 0)
wl = os.environ.get('WL', 'transe_fb15k237')
model, tables, kg, kg_test, info = bench.build_workload(wl, dev)
if hasattr(bench, 'train_like'):
    pass
kg_test.head_idx = kg_test.head_idx.to(dev); kg_test.tail_idx = kg_test.tail_idx.to(dev); kg_test.relations = kg_test.relations.to(dev)
ev = tk.LinkPredictionEvaluator(model, kg_test)
for _ in range(8): ev.evaluate(32768, verbose=False)
torch.cuda.synchronize()
N = 400
t0 = time.perf_counter()
for _ in range(N): ev.evaluate(32768, verbose=False)
torch.cuda.synchronize()
print('ms/step', (time.perf_counter() - t0) / N * 1e3, 'level', getattr(ev, '_level', None))
pr = cProfile.Profile(); pr.enable()
for _ in range(N): ev.evaluate(32768, verbose=False)
pr.disable()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(32)
