"""evaluate() time of any model kind on a dataset-shaped synthetic KG:  python tools/eval_time.py transh fb15k237 200"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchkge_amd as tk  # noqa: E402
from oracle import kge_oracle as orc  # noqa: E402

kind, shape, d = sys.argv[1], sys.argv[2], int(sys.argv[3])
n_ent, n_rel, n_train, n_valid, n_test = orc.DATASET_SHAPES[shape]
if kind == 'transh':
    m = tk.TransHModel(d, n_ent, n_rel)
elif kind == 'transd':
    m = tk.TransDModel(d, d, n_ent, n_rel)
elif kind == 'transe_l1':
    m = tk.TransEModel(d, n_ent, n_rel, 'L1')
elif kind == 'transe':
    m = tk.TransEModel(d, n_ent, n_rel, 'L2')
elif kind == 'distmult':
    m = tk.DistMultModel(d, n_ent, n_rel)
else:
    m = tk.ComplExModel(d, n_ent, n_rel)
m = m.cuda()
h, t, r = orc.synthetic_triples(n_ent, n_rel, n_train + n_valid + n_test, 1001)
kg = tk.KnowledgeGraph(kg={'heads': h, 'tails': t, 'relations': r}, ent2ix={i: i for i in range(n_ent)},
                       rel2ix={i: i for i in range(n_rel)})
_, _, kg_test = kg.split_kg(sizes=(n_train, n_valid, n_test))
ev = tk.LinkPredictionEvaluator(m, kg_test, graph=True)
for _ in range(2):
    ev.evaluate(32768, verbose=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    ev.evaluate(32768, verbose=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print('%s d=%d on %s shape: evaluate %.3f ms, %.3g triples scored/s, filt MRR %.6f' % (
    kind, d, shape, dt * 1e3, n_test * 2 * n_ent / dt, ev.mrr()[1]))
