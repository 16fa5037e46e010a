"""cProfile of one eager evaluate() (host-side overhead of the launch path)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchkge_amd as tk  # noqa: E402
from oracle import kge_oracle as orc  # noqa: E402

n_ent, n_rel, n_train, n_valid, n_test = orc.DATASET_SHAPES['fb15k237']
m = tk.TransEModel(200, n_ent, n_rel, 'L2').cuda()
h, t, r = orc.synthetic_triples(n_ent, n_rel, n_train + n_valid + n_test, 1001)
kg = tk.KnowledgeGraph(kg={'heads': h, 'tails': t, 'relations': r}, ent2ix={i: i for i in range(n_ent)},
                       rel2ix={i: i for i in range(n_rel)})
_, _, kg_test = kg.split_kg(sizes=(n_train, n_valid, n_test))
kg_test.head_idx, kg_test.tail_idx, kg_test.relations = (x.cuda() for x in (kg_test.head_idx, kg_test.tail_idx, kg_test.relations))
ev = tk.LinkPredictionEvaluator(m, kg_test)
for _ in range(3):
    ev.evaluate(32768, verbose=False)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    ev.evaluate(32768, verbose=False)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
