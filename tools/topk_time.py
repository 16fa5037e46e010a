#!/usr/bin/env python
"""Time EntityInference (tiled top-k) at cfg2 and cfg5 shapes on one MI355X:
    python tools/topk_time.py [--cfg5]
prints one JSON line per shape: ms per evaluate(), candidates scored per second, scratch bytes."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchkge_amd as tk          # noqa: E402
from oracle import kge_oracle as orc   # noqa: E402  (synthetic-KG generator only)
import bench                        # noqa: E402


def run(name, model, ents, rels, k, b_size, dictionary, reps=3, tile=None):
    dev = next(model.parameters()).device
    inf = tk.EntityInference(model, ents, rels, top_k=k, missing='tails', dictionary=dictionary, tile=tile)
    inf.evaluate(b_size, verbose=False)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        inf.evaluate(b_size, verbose=False)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / reps
    n = len(ents)
    b = min(b_size, n)
    C = inf._tile(b, model.n_ent)
    out = {'shape': name, 'queries': n, 'n_ent': model.n_ent, 'top_k': k, 'b_size': b_size, 'tile': C,
           'ms_per_evaluate': round(dt * 1e3, 3), 'candidates_scored_per_s': round(n * model.n_ent / dt, 1),
           'scratch_bytes': 4 * b * C, 'materialised_matrix_bytes_per_batch': 4 * b * model.n_ent, 'filtered': dictionary is not None}
    # the materialised reference composition on the same inputs, when the (b, N) matrix fits comfortably
    if 4 * b * model.n_ent < (8 << 30):
        ref = tk.EntityInference(model, ents, rels, top_k=k, missing='tails', dictionary=dictionary)
        ref._evaluate_materialised(b_size, verbose=False)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            ref._evaluate_materialised(b_size, verbose=False)
        torch.cuda.synchronize(dev)
        out['materialised_ms_per_evaluate'] = round((time.perf_counter() - t0) / reps * 1e3, 3)
        out['same_predictions'] = bool(torch.equal(ref.predictions, inf.predictions) and torch.equal(ref.scores, inf.scores))
    print(json.dumps(out), flush=True)


def main():
    dev = torch.device('cuda', 0)
    model, tables, kg, kg_test, info = bench.build_workload('transe_fb15k237', dev, weights='xavier')
    idx = kg.filter_index('tails', dev)
    run('cfg2 TransE d=200 FB15k-237 shape', model, kg_test.head_idx, kg_test.relations, 10, 4096, idx)
    run('cfg2 TransE d=200 FB15k-237 shape', model, kg_test.head_idx, kg_test.relations, 10, 4096, None)
    del model
    if '--cfg5' in sys.argv:
        model, kg, kg_test, info = bench.build_cfg5_sample(dev, n_facts=400000, n_test=1024)
        run('cfg5 ComplEx d=512 Wikidata5M shape', model, kg_test.head_idx, kg_test.relations, 10, 1024,
            kg.filter_index('tails', dev), reps=2)


if __name__ == '__main__':
    main()
