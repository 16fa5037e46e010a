import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torchkge_amd as tk
from torchkge_amd import _hip
dev = torch.device('cuda', 0)
model, tables, kg, kg_test, info = bench.build_workload('transh_fb15k237', dev, weights='trained', kg_kind='zipf', train_cfg={'steps': 300})
h, t, r = kg_test.head_idx.to(dev), kg_test.tail_idx.to(dev), kg_test.relations.to(dev)
perm = torch.argsort(r, stable=True)
h, t, r = h[perm], t[perm], r[perm]
true_idx = torch.cat([t, h])
model.split_level = 1
def count(stream, skip_true, tag):
    model.lp_hi_stream = stream
    g = model.lp_guard_begin(dev)
    g.zero_()
    object.__setattr__(model, '_split_level', 1)
    with model.lp_session(), torch.no_grad():
        prob = model.lp_problem(h, t, r, 'both')
        st = prob.pair_scores(true_idx)
        if skip_true:
            prob.split_true = (st, true_idx)
        raw = prob.count_ge(st)
        torch.cuda.synchronize()
        n = int(prob.last_split[0].item())
        lst = prob.last_split[1]['list'][:2 * n].view(n, 2).clone()
        sp = prob.split
        prob.split = None
        exact = prob.count_ge(st)
        prob.split = sp
    model.lp_guard_end()
    d = (raw != exact)
    key = lst[:, 0].long() * 20000 + lst[:, 1].long()
    uniq = torch.unique(key).numel()
    print('%-28s mismatching queries %d, listed %d, distinct listed %d, frag=%s, overflow %g' % (
        tag, int(d.sum()), n, uniq, bool(sp.get('es_frag')), float(g[2])))
    if int(d.sum()):
        q = int(d.nonzero()[0])
        print('   query %d: stream %d exact %d; list entries of it: %s; true %d' % (
            q, int(raw[q]), int(exact[q]), lst[lst[:, 0] == q][:, 1].tolist()[:12], int(true_idx[q])))
    return raw
a = count(False, False, 'old kernel')
b = count(True, False, 'stream qg=auto')
for qg in ('16', '4', '1'):
    os.environ['KGE_HS_QG'] = qg
    count(True, False, 'stream qg=' + qg)
os.environ.pop('KGE_HS_QG')
# zero projection terms: the PM = 1 code path on what is arithmetically a TransE problem
model.norm_vect.weight.data.zero_()
count(False, False, 'old kernel, W = 0')
count(True, False, 'stream, W = 0')
