#!/usr/bin/env python
"""Same-container CPU timing of the REAL reference (torchkge v0.17.7 imported from /root/reference)
against the oracle (oracle/kge_oracle.py, the `port` that bench.py's cpu_baseline leg times on the GPU
box, where the reference does not exist): same synthetic FB15k-237-shaped KG, same weights, same sample
of test triples, same b_size sweep.  Run in the BUILD container only:

    python tools/ref_vs_oracle_cpu.py [n_sample]

Prints one JSON object (copied into BASELINE.md section 2)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')
import pandas as pd                                              # noqa: E402
from torchkge.data_structures import KnowledgeGraph             # noqa: E402  (the reference)
from torchkge.evaluation import LinkPredictionEvaluator         # noqa: E402
from torchkge.models import TransEModel                         # noqa: E402
from oracle import kge_oracle as orc                            # noqa: E402


def main():
    n_sample = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    torch.set_num_threads(os.cpu_count())
    n_ent, n_rel, n_train, n_valid, n_test = orc.DATASET_SHAPES['fb15k237']
    d = 200
    h, t, r = orc.synthetic_triples_zipf(n_ent, n_rel, n_train + n_valid + n_test, 1003)
    tables = orc.init_tables('transe', n_ent, n_rel, d, seed=0)
    t0 = time.perf_counter()
    kg = KnowledgeGraph(kg={'heads': h, 'tails': t, 'relations': r}, ent2ix={i: i for i in range(n_ent)},
                        rel2ix={i: i for i in range(n_rel)})
    dict_s = time.perf_counter() - t0
    sel = slice(n_train + n_valid, n_train + n_valid + n_sample)
    kg_test = KnowledgeGraph(kg={'heads': h[sel], 'tails': t[sel], 'relations': r[sel]}, ent2ix=kg.ent2ix,
                             rel2ix=kg.rel2ix, dict_of_heads=kg.dict_of_heads, dict_of_tails=kg.dict_of_tails,
                             dict_of_rels=kg.dict_of_rels)
    m = TransEModel(d, n_ent, n_rel, dissimilarity_type='L2')
    m.load_state_dict({'ent_emb.weight': tables[0], 'rel_emb.weight': tables[1]})
    out = {'cores': os.cpu_count(), 'threads': torch.get_num_threads(), 'workload': 'TransE d=200 L2, FB15k-237-shaped Zipf KG',
           'n_sample': n_sample, 'reference_filter_dict_build_s': round(dict_s, 2), 'sweep': {}}
    for b in (32, 64, 128, 256):
        ref_s = orc_s = float('inf')
        for rep in range(3):        # interleaved, best of 3: the first touch of the (b, N, d) temporaries is page-fault bound
            ev = LinkPredictionEvaluator(m, kg_test)
            t0 = time.perf_counter()
            ev.evaluate(b_size=b, verbose=False)
            ref_s = min(ref_s, time.perf_counter() - t0)
            t0 = time.perf_counter()
            rh, rt, frh, frt = orc.lp_evaluate('transe', tables, h[sel], t[sel], r[sel], kg.dict_of_heads, kg.dict_of_tails, b, 2)
            orc_s = min(orc_s, time.perf_counter() - t0)
        same = bool(torch.equal(rh, ev.rank_true_heads) and torch.equal(rt, ev.rank_true_tails) and
                    torch.equal(frh, ev.filt_rank_true_heads) and torch.equal(frt, ev.filt_rank_true_tails))
        out['sweep'][b] = {'reference_s': round(ref_s, 2), 'oracle_s': round(orc_s, 2),
                           'reference_triples_per_s': round(n_sample * 2 * n_ent / ref_s, 1),
                           'oracle_triples_per_s': round(n_sample * 2 * n_ent / orc_s, 1),
                           'ranks_identical': same}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
