#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""How many (query, candidate) pairs of an evaluate() fall inside the error band of a ONE-PRODUCT f16 sweep
(q_hi . e_hi only; certified band  |q.e - q_hi.e_hi| <= ||dq|| ||e|| + ||q_hi|| ||de||), against the band of the
three-product sweep the count kernel runs today?  Decides whether a one-product first level pays (r03 review item 4):
every pair inside the band costs one exact fp32 chain in the recheck (~4 G pairs/s).

    python tools/band_probe.py [--workload transe_fb15k237] [--weights trained|xavier]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='transe_fb15k237')
    ap.add_argument('--weights', default='trained')
    args = ap.parse_args()
    import bench
    dev = torch.device('cuda', 0)
    model, tables, kg, kg_test, info = bench.build_workload(args.workload, dev, weights=args.weights)
    kind = info['kind']
    h, t, r = kg_test.head_idx.to(dev), kg_test.tail_idx.to(dev), kg_test.relations.to(dev)
    with torch.no_grad():
        if kind == 'transe':
            E, R = model.ent_emb.weight.data, model.rel_emb.weight.data
            Q = torch.cat([E[h] + R[r], E[t] - R[r]])
            T = E
            aug = -0.5 * (T * T).sum(1)
        elif kind == 'distmult':
            E, R = model.ent_emb.weight.data, model.rel_emb.weight.data
            Q = torch.cat([E[h] * R[r], R[r] * E[t]])
            T, aug = E, torch.zeros(E.shape[0], device=dev)
        else:
            raise SystemExit('transe / distmult only')
        true = torch.cat([t, h])
        n2, N = Q.shape[0], T.shape[0]
        # f16 hi parts at the kernel's power-of-two scale (no subnormals at these magnitudes)
        sq = 2.0 ** 12
        Qh = (Q * sq).half().float() / sq
        Th = (T * sq).half().float() / sq
        dq, de = (Q - Qh).norm(dim=1), (T - Th).norm(dim=1)
        qn, en, qhn = Q.norm(dim=1), T.norm(dim=1), Qh.norm(dim=1)
        emax, demax = en.max(), de.max()
        W_query = dq * emax + qhn * demax                       # per-query band (max over candidates)
        # three-product band of today's kernel, roughly: 1006 * 2^-24 * ||q|| max||e|| at K = 200
        K = Q.shape[1]
        W3 = (5.0 * K + 6) * 2.0 ** -24 * (qn * emax + 0.5 * emax * emax)
        out = {}
        tot1 = tot1p = tot3 = 0
        per1 = []
        for i0 in range(0, n2, 4096):
            q = Q[i0:i0 + 4096]
            D = q @ T.t() + aug[None, :]
            a = D.gather(1, true[i0:i0 + 4096, None])
            dist_ = (D - a).abs()
            w1 = W_query[i0:i0 + 4096, None]
            c1 = (dist_ <= w1).sum(1)
            wp = dq[i0:i0 + 4096, None] * en[None, :] + qhn[i0:i0 + 4096, None] * de[None, :]    # per-pair band
            c1p = (dist_ <= wp).sum(1)
            c3 = (dist_ <= W3[i0:i0 + 4096, None]).sum(1)
            tot1 += int(c1.sum()); tot1p += int(c1p.sum()); tot3 += int(c3.sum())
            per1.append(c1)
        per1 = torch.cat(per1).float()
        out = {'workload': args.workload, 'weights': args.weights, 'queries': n2, 'candidates': N,
               'rel_band_one_product': float((W_query / (qn * emax)).mean()),
               'pairs_in_band_one_product_per_query_band': tot1, 'per_query_mean': tot1 / n2,
               'per_query_median': float(per1.median()), 'per_query_p90': float(per1.quantile(0.9)),
               'per_query_max': float(per1.max()),
               'queries_with_le_16': float((per1 <= 16).float().mean()), 'queries_with_le_64': float((per1 <= 64).float().mean()),
               'pairs_in_band_one_product_per_pair_band': tot1p, 'per_pair_band_mean': tot1p / n2,
               'pairs_in_band_three_products_approx': tot3, 'three_products_mean': tot3 / n2,
               'recheck_ms_at_4Gpairs_per_s': {'one_product': tot1 / 4e9 * 1e3, 'one_product_pair_band': tot1p / 4e9 * 1e3,
                                               'three_products': tot3 / 4e9 * 1e3}}
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
