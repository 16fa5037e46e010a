import cProfile, pstats, sys, os, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torchkge_amd as tk, bench
dev = torch.device('cuda', 0)
model, tables, kg, kg_test, info = bench.build_workload('transe_fb15k237', dev, weights='xavier')
torch.cuda.synchronize()
ev = tk.LinkPredictionEvaluator(model, kg_test)
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
ev.evaluate(256, verbose=False)
torch.cuda.synchronize()
print('first evaluate ms', (time.perf_counter() - t0) * 1e3)
pr.disable()
st = pstats.Stats(pr); st.sort_stats('cumtime').print_stats(28)
