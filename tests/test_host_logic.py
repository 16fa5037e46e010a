"""CPU-only tests of the host logic: C-ABI exports, KnowledgeGraph / filter
index construction, metrics, sampler probabilities, no-CPU-fallback behaviour,
sharding math, and the multi-process (gloo, world_size 2) exchange paths."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import kge_oracle as orc
from tests.helpers import ROOT, GOLDEN, load_golden, OracleRankEngine

import torchkge_amd as tk
from torchkge_amd import _hip
from torchkge_amd import distributed as kd
from torchkge_amd.exceptions import (NotYetEvaluatedError, WrongArgumentsError, SizeMismatchError,
                                     SanityError)
from torchkge_amd.filter_index import FilterIndex, KEY2_SPAN


def toy_df():
    import pandas as pd
    return pd.DataFrame([[0, 1, 0], [0, 2, 0], [0, 3, 0], [0, 4, 0], [1, 2, 1], [1, 3, 2], [2, 4, 0],
                         [3, 4, 4], [5, 4, 0]], columns=['from', 'to', 'rel'])


def test_library_loads_and_exports_every_declared_symbol():
    lib = _hip.load_library()          # no GPU needed to dlopen and resolve symbols
    hdr = open(os.path.join(ROOT, 'include', 'kge_hip.h')).read()
    declared = set(re.findall(r'\b(kge_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(_hip.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    out = subprocess.check_output(['nm', '-D', '--defined-only', _hip.LIB_PATH], text=True)
    assert declared <= set(re.findall(r' T (kge_[a-z0-9_]+)', out))
    assert lib.kge_abi_version() == _hip.ABI_VERSION and lib.kge_build_arch() == b'gfx950'
    # the descriptor struct mirrors the header field for field
    fields = re.search(r'typedef struct kge_lp_desc \{(.*?)\} kge_lp_desc;', hdr, re.S).group(1)
    names = re.findall(r'\b(\w+)\s*(?:;|,)', re.sub(r'/\*.*?\*/', '', fields, flags=re.S))
    assert names == [f[0] for f in _hip.LpDesc._fields_]


def test_ctypes_signatures_match_the_header_prototypes():
    """Every prototype of include/kge_hip.h against the ctypes argtypes of _hip._SIGNATURES: same number of
    parameters, pointers bound as void*, int / int32_t as c_int, int64_t as c_int64, float as c_float; and
    struct kge_split_args field for field (a drifted binding would corrupt arguments silently)."""
    import ctypes
    hdr = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', 'kge_hip.h')).read(), flags=re.S)
    protos = dict(re.findall(r'\bint\s+(kge_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;', hdr, flags=re.S))

    def kind(param):
        param = param.strip()
        if '*' in param:
            return ctypes.c_void_p
        t = param.split()
        if 'kge_stream_t' in t:
            return ctypes.c_void_p
        if 'int64_t' in t:
            return ctypes.c_int64
        if 'float' in t:
            return ctypes.c_float
        if 'int' in t or 'int32_t' in t:
            return ctypes.c_int
        raise AssertionError('unparsed parameter: %r' % param)
    checked = 0
    for name, args in _hip._SIGNATURES.items():
        assert name in protos, name
        params = [] if protos[name].strip() in ('', 'void') else protos[name].split(',')
        assert len(params) == len(args), (name, len(params), len(args))
        for prm, a in zip(params, args):
            k = kind(prm)
            if k is ctypes.c_void_p:    # pointers are bound as void* or as POINTER(struct)
                assert a is ctypes.c_void_p or issubclass(a, ctypes._Pointer), (name, prm)
            elif k is ctypes.c_int:
                assert a in (ctypes.c_int, ctypes.c_int32), (name, prm)
            else:
                assert a is k, (name, prm)
        checked += 1
    assert checked == len(_hip._SIGNATURES) >= 30
    fields = re.search(r'typedef struct kge_split_args \{(.*?)\} kge_split_args;', hdr, re.S).group(1)
    names = []
    for decl in fields.split(';'):
        decl = decl.strip()
        if decl:
            names += [re.sub(r'[\s\*]', '', x).split(' ')[-1] for x in re.sub(r'^(const\s+)?\w+\s+', '', decl).split(',')]
    assert names == [f[0] for f in _hip.SplitArgs._fields_]


def test_collectives_library_loads_and_exports_its_header():
    """libkge_hip_coll.so (include/kge_hip_coll.h, the RCCL exchange step behind the C-ABI): loads, exports every
    declared entry point with the bound argument counts; libkge_hip.so itself does not depend on librccl."""
    from torchkge_amd import _hip_coll
    lib = _hip_coll.load_library()
    hdr = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', 'kge_hip_coll.h')).read(), flags=re.S)
    protos = dict(re.findall(r'\bint\s+(kge_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;', hdr, flags=re.S))
    assert set(protos) == set(_hip_coll._SIGNATURES)
    for name, args in _hip_coll._SIGNATURES.items():
        assert hasattr(lib, name) and len(protos[name].split(',')) == len(args), name
    out = subprocess.run(['readelf', '-d', _hip.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    assert 'rccl' not in out
    out = subprocess.run(['readelf', '-d', _hip_coll.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    assert 'librccl' in out


def test_column_plan_places_every_query_once_and_groups_by_key():
    """filter_index.ColumnPlan (pure torch ops: runs on CPU tensors): every query of a both-sides batch sits in exactly one
    column slot; the queries of a column share their key (and so their query row); exactly the first query of a column
    writes the split row; single-query columns come first, grouped ones in order of decreasing size; both parts are
    padded to the panel; relation-major order sorts the columns by relation."""
    from torchkge_amd.filter_index import ColumnPlan
    g = torch.Generator().manual_seed(4)
    n_ent, n_rel, B, sets = 50, 4, 700, 4
    h = torch.randint(0, n_ent, (B,), generator=g)
    t = torch.randint(0, n_ent, (B,), generator=g)
    r = torch.randint(0, n_rel, (B,), generator=g)
    t[:300] = 7                                     # a head-side hub key per relation
    pad = lambda n: (n + 191) // 192 * 192
    for rel_major in (False, True):
        cp = ColumnPlan(h, t, r, n_ent, n_rel, sets, pad, relation_major=rel_major)
        key = torch.cat([h * n_rel + r, (t * n_rel + r) + n_ent * n_rel])
        assert cp.n_queries == 2 * B and cp.n_distinct_keys == int(torch.unique(key).numel())
        assert cp.n_single_p % 192 == 0 and cp.n_multi_p % 192 == 0
        assert cp.col_q.shape[0] == max(cp.n_single_p, 1) and cp.members.shape[0] == max(cp.n_multi_p, 1) * sets
        placed = torch.cat([cp.col_q[cp.col_q >= 0], cp.members[cp.members >= 0]]).long()
        assert torch.equal(placed.sort().values, torch.arange(2 * B))
        assert int((cp.col_q >= 0).sum()) == cp.n_single and cp.n_single + cp.n_multi == cp.n_columns
        mem = cp.members.view(-1, sets)[:cp.n_multi]
        sizes = (mem >= 0).sum(1)
        assert int(sizes.min()) >= 2 and (sizes[:-1] >= sizes[1:]).all()          # grouped: >= 2 queries, sizes descending
        assert (mem[:, 0] >= 0).all() and ((mem >= 0).long().diff(dim=1) <= 0).all()   # slots filled from the left
        for row in mem[:50]:                            # the queries of a column share their key
            q = row[row >= 0].long()
            assert (key[q] == key[q[0]]).all()
        # the split row of every column is written by exactly one query: its first
        rows = cp.qs_row[cp.qs_row >= 0].long()
        assert rows.numel() == cp.n_columns and rows.unique().numel() == cp.n_columns
        assert torch.equal(cp.qs_row[cp.col_q[:cp.n_single].long()].long(), torch.arange(cp.n_single))
        assert torch.equal(cp.qs_row[mem[:, 0].long()].long(), cp.n_single_p + torch.arange(cp.n_multi))
        assert (cp.qs_row[mem[:, 1:][mem[:, 1:] >= 0].long()] == -1).all()
        # query -> column and column -> representative are consistent
        assert torch.equal(cp.rep[cp.col_of_q], torch.where(cp.qs_row >= 0, torch.arange(2 * B), cp.rep[cp.col_of_q]))
        assert (key[cp.rep[cp.col_of_q]] == key).all()
        if rel_major:                                   # single-query columns in relation order (per side)
            rq = torch.cat([r, r])[cp.col_q[:cp.n_single].long()]
            side = (cp.col_q[:cp.n_single] >= B)
            assert (rq[~side][:-1] <= rq[~side][1:]).all() and (rq[side][:-1] <= rq[side][1:]).all()


def test_no_cpu_fallback():
    m = tk.TransEModel(8, 10, 3)
    i = torch.tensor([0, 1])
    with pytest.raises(RuntimeError, match='no CPU'):
        m.scoring_function(i, i, i)
    with pytest.raises(RuntimeError):
        m.inference_prepare_candidates(i, i, i)
    with pytest.raises(RuntimeError):
        tk.utils.get_rank(torch.zeros(2, 3), i)
    kg = tk.KnowledgeGraph(df=toy_df())
    with pytest.raises(RuntimeError, match='MI355X'):
        tk.LinkPredictionEvaluator(m, kg).evaluate(4, verbose=False)
    with pytest.raises(RuntimeError):
        tk.BernoulliNegativeSampler(kg).corrupt_batch(kg.head_idx, kg.tail_idx, kg.relations)
    # the oracle is never imported by the product package
    res = subprocess.run(['grep', '-rlE', r'^\s*(from|import)\s+\.*oracle', os.path.join(ROOT, 'torchkge_amd'),
                          '--include=*.py'], stdout=subprocess.PIPE, text=True)
    assert res.stdout.strip() == ''


def test_knowledge_graph_constructor_and_split():
    # reference tests/test_data.py: argument validation
    df = toy_df()
    kg = tk.KnowledgeGraph(df=df)
    assert (kg.n_ent, kg.n_rel, kg.n_facts) == (6, 4, 9)
    with pytest.raises(WrongArgumentsError):
        tk.KnowledgeGraph()
    with pytest.raises(WrongArgumentsError):
        tk.KnowledgeGraph(df=df, kg={'heads': kg.head_idx, 'tails': kg.tail_idx, 'relations': kg.relations})
    with pytest.raises(WrongArgumentsError):
        tk.KnowledgeGraph(kg={'heads': kg.head_idx, 'tails': kg.tail_idx})
    with pytest.raises(WrongArgumentsError):
        tk.KnowledgeGraph(kg={'heads': kg.head_idx, 'tails': kg.tail_idx, 'relations': kg.relations})
    with pytest.raises(SanityError):
        tk.KnowledgeGraph(kg={'heads': kg.head_idx.int(), 'tails': kg.tail_idx, 'relations': kg.relations},
                          ent2ix=kg.ent2ix, rel2ix=kg.rel2ix)
    with pytest.raises(WrongArgumentsError):
        kg.split_kg(sizes=(3, 3))
    with pytest.raises(SizeMismatchError):
        kg.split_kg(sizes=(3, 3, 2, 1))
    tr, te = kg.split_kg(sizes=(6, 3))
    assert (tr.n_facts, te.n_facts) == (6, 3)
    tr, va, te = kg.split_kg(sizes=(5, 2, 2))
    assert torch.equal(te.head_idx, kg.head_idx[7:])
    # children share the FULL graph's filter sets (data_structures.py:236-238)
    dh, dt, dr = orc.build_filter_dicts(kg.head_idx, kg.tail_idx, kg.relations)
    assert dict(te.dict_of_heads) == dict(dh) and dict(te.dict_of_tails) == dict(dt)
    assert dict(te.dict_of_rels) == dict(dr)
    torch.manual_seed(0)
    tr, te = kg.split_kg(share=0.7)
    assert tr.n_facts + te.n_facts == 9
    assert set(torch.cat([tr.head_idx, tr.tail_idx]).tolist()) == set(range(6))   # every entity in train
    tr, va, te = kg.split_kg(share=0.6, validation=True)
    assert tr.n_facts + va.n_facts + te.n_facts == 9
    assert kg[0] == (0, 1, 0) and len(kg) == 9
    assert list(kg.get_df().columns) == ['from', 'to', 'rel']


def test_filter_index_matches_dicts():
    z = np.load(GOLDEN + '/ref_sampler.npz')
    heads, tails, rels = z['heads'], z['tails'], z['rels']
    dh, dt, _ = orc.build_filter_dicts(heads, tails, rels)
    for d, (k1, k2, v) in ((dh, (tails, rels, heads)), (dt, (heads, rels, tails))):
        a = FilterIndex.from_dict(d, 'cpu')
        b = FilterIndex.from_triples(k1, k2, v, 'cpu')
        assert torch.equal(a.keys, b.keys) and torch.equal(a.offsets, b.offsets)
        assert (a.keys[1:] > a.keys[:-1]).all()
        for j in range(0, a.n_keys, 37):
            key = int(a.keys[j])
            want = d[(key // KEY2_SPAN, key % KEY2_SPAN)]
            got_a = set(a.targets[int(a.offsets[j]):int(a.offsets[j + 1])].tolist())
            got_b = set(b.targets[int(b.offsets[j]):int(b.offsets[j + 1])].tolist())
            assert got_a == want and got_b == want
    assert FilterIndex.from_dict({}, 'cpu').n_keys == 0


def test_bernoulli_probabilities_known_answers():
    # reference tests/test_utils.py:77-95
    kg = tk.KnowledgeGraph(df=toy_df())
    t = torch.cat((kg.head_idx.view(-1, 1), kg.tail_idx.view(-1, 1), kg.relations.view(-1, 1)), dim=1)
    assert tk.utils.get_tph(t) == {0: 2., 1: 1., 2: 1., 3: 1.}
    assert tk.utils.get_hpt(t) == {0: 1.5, 1: 1., 2: 1., 3: 1.}
    probs = tk.utils.get_bernoulli_probs(kg)
    for k, v in {0: 0.5714, 1: 0.5, 2: 0.5, 3: 0.5}.items():
        assert abs(probs[k] - v) < 1e-3
    z = np.load(GOLDEN + '/ref_sampler.npz')
    n_ent, n_rel = int(z['n_ent']), int(z['n_rel'])
    kg = tk.KnowledgeGraph(kg={'heads': torch.from_numpy(z['heads']), 'tails': torch.from_numpy(z['tails']),
                               'relations': torch.from_numpy(z['rels'])},
                           ent2ix={i: i for i in range(n_ent)}, rel2ix={i: i for i in range(n_rel)})
    samp = tk.BernoulliNegativeSampler(kg)
    assert np.array_equal(samp.bern_probs.numpy(), z['bern_probs'])
    assert tk.utils.get_mask(10, 1, 3).tolist() == [False, True, True] + [False] * 7


def test_metrics_from_ranks_equal_reference():
    z, _ = load_golden('transe', 2)
    m = tk.TransEModel(4, 5, 2)
    kg = tk.KnowledgeGraph(df=toy_df())
    ev = tk.LinkPredictionEvaluator(m, kg)
    for f in (ev.mrr, ev.mean_rank, ev.hit_at_k, ev.hit_at_k_heads, ev.hit_at_k_tails):
        with pytest.raises(NotYetEvaluatedError):
            f()
    ev.rank_true_heads = torch.from_numpy(z['rank_true_heads'])
    ev.rank_true_tails = torch.from_numpy(z['rank_true_tails'])
    ev.filt_rank_true_heads = torch.from_numpy(z['filt_rank_true_heads'])
    ev.filt_rank_true_tails = torch.from_numpy(z['filt_rank_true_tails'])
    ev.evaluated = True
    assert np.allclose(ev.hit_at_k(10), z['hit10'], atol=0)
    assert np.allclose(ev.mrr(), z['mrr'], atol=0)
    assert np.allclose(ev.mean_rank(), z['mean_rank'], atol=0)
    ev.print_results(k=[1, 3])


def test_model_surface_matches_reference_names():
    m = tk.TransHModel(8, 10, 3)
    assert sorted(m.state_dict()) == ['ent_emb.weight', 'norm_vect.weight', 'rel_emb.weight']
    sd = dict(m.state_dict())
    sd['projected_entities'] = torch.zeros(3, 10, 8)      # present in reference state_dicts
    m.load_state_dict(sd)
    m = tk.TransDModel(8, 6, 10, 3)
    assert sorted(m.state_dict()) == ['ent_emb.weight', 'ent_proj_vect.weight', 'rel_emb.weight',
                                      'rel_proj_vect.weight']
    m = tk.ComplExModel(8, 10, 3)
    assert sorted(m.state_dict()) == ['im_ent_emb.weight', 'im_rel_emb.weight', 're_ent_emb.weight',
                                      're_rel_emb.weight']
    for cls in (tk.TransEModel, tk.DistMultModel):
        m = cls(8, 10, 3)
        assert (m.ent_emb.weight.norm(dim=1) - 1).abs().max() < 1e-5
        assert m.lp_scoring_function.__func__ is not None and m.lp_prep_cands.__func__ is not None


def test_shard_ranges_partition():
    for n, w in [(14541, 8), (10, 3), (7, 8), (4594485, 8), (0, 2)]:
        parts = [kd.shard_range(n, w, r) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        assert all(hi - lo <= kd.shard_size(n, w) for lo, hi in parts)


WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
import torchkge_amd as tk
from oracle import kge_oracle as orc
from tests.helpers import load_golden, OracleRankEngine
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = sys.argv[3]
dist.init_process_group('gloo', rank=rank, world_size=world)
kind = sys.argv[4]
z, tables = load_golden(kind, 2)
n_ent, n_rel = int(z['n_ent']), int(z['n_rel'])
heads, tails, rels = (torch.from_numpy(z[k]) for k in ('heads', 'tails', 'rels'))
kg = tk.KnowledgeGraph(kg={'heads': heads, 'tails': tails, 'relations': rels},
                       ent2ix={i: i for i in range(n_ent)}, rel2ix={i: i for i in range(n_rel)})
nt = int(z['n_test'])
_, kg_test = kg.split_kg(sizes=(len(heads) - nt, nt))
if kind == 'transe':
    m = tk.TransEModel(tables[0].shape[1], n_ent, n_rel)
else:
    m = tk.ComplExModel(tables[0].shape[1], n_ent, n_rel)
ok = True
for shard, exchange, fused in [('entities', 'counts', True), ('entities', 'scores', True),
                               ('entities', 'scores', False), ('queries', 'counts', True)]:
    ev = tk.LinkPredictionEvaluator(m, kg_test, fused=fused, shard=shard, exchange=exchange,
                                    engine=OracleRankEngine(kind, tables))
    ev.evaluate(b_size=13, verbose=False)
    for nm in ('rank_true_heads', 'rank_true_tails', 'filt_rank_true_heads', 'filt_rank_true_tails'):
        same = np.array_equal(getattr(ev, nm).numpy(), z[nm])
        ok = ok and same
        if not same:
            print('MISMATCH', rank, shard, exchange, fused, nm, flush=True)
# the score all-to-all cut into several row tiles per batch (tile = 2 ranks x 4 rows; the last one short)
import torchkge_amd.evaluation as ev_mod
from torchkge_amd import distributed as kd0
ev_mod.SCORE_TILE_BYTES = 4 * kd0.shard_size(n_ent, world) * world * 4
ev = tk.LinkPredictionEvaluator(m, kg_test, shard='entities', exchange='scores', engine=OracleRankEngine(kind, tables))
ev.evaluate(b_size=13, verbose=False)
for nm in ('rank_true_heads', 'rank_true_tails', 'filt_rank_true_heads', 'filt_rank_true_tails'):
    if not np.array_equal(getattr(ev, nm).numpy(), z[nm]):
        ok = False
        print('MISMATCH tiled all-to-all', rank, nm, flush=True)
ev_mod.SCORE_TILE_BYTES = 256 << 20
# ROW-SHARDED entity tables (SURVEY 8e): each rank keeps only its rows of every entity-indexed table,
# query rows are built by the owner rank and summed over the ranks; the ranks must still be the reference's
from torchkge_amd import distributed as kd
from tests.helpers import ShardedOracleEngine
names = ['ent_emb', 'rel_emb'] if kind == 'transe' else ['re_ent_emb', 'im_ent_emb', 're_rel_emb', 'im_rel_emb']
m.load_state_dict({n + '.weight': t.clone() for n, t in zip(names, tables)})
full_bytes = m.entity_table_bytes()
lo, hi = kd.shard_model_(m)
assert (lo, hi) == kd.shard_range(n_ent, world, rank) and m.n_ent == n_ent
assert m.entity_table_bytes() * world <= full_bytes + 4 * world * tables[0].shape[1] * (2 if kind == 'complex' else 1)
assert all(getattr(m, nm).weight.shape[0] == hi - lo for nm in m._ENT_TABLES)
for exchange, fused in [('counts', True), ('scores', True), ('scores', False)]:
    ev = tk.LinkPredictionEvaluator(m, kg_test, fused=fused, shard='entities', exchange=exchange,
                                    engine=ShardedOracleEngine(kind))
    ev.evaluate(b_size=13, verbose=False)
    for nm in ('rank_true_heads', 'rank_true_tails', 'filt_rank_true_heads', 'filt_rank_true_tails'):
        same = np.array_equal(getattr(ev, nm).numpy(), z[nm])
        ok = ok and same
        if not same:
            print('MISMATCH row-sharded', rank, exchange, fused, nm, flush=True)
try:        # a row-sharded model refuses anything that needs the whole tables
    tk.LinkPredictionEvaluator(m, kg_test, engine=ShardedOracleEngine(kind)).evaluate(b_size=13, verbose=False)
    ok = False
except RuntimeError:
    pass
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''


@pytest.mark.parametrize('kind', ['transe', 'complex'])
def test_multiprocess_sharded_evaluation_gloo(kind, tmp_path):
    """world_size 2 on CPU/gloo: entity-sharded (count all-reduce and score
    all-gather) and query-sharded evaluation give the reference's ranks."""
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % {'root': ROOT})
    port = str(29500 + (os.getpid() % 2000) + (0 if kind == 'transe' else 1))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), '2', port, kind]) for r in range(2)]
    codes = [p.wait(timeout=300) for p in procs]
    assert codes == [0, 0]


def test_single_process_engine_hook_equals_reference():
    """world_size 1 through the same evaluator code path (oracle engine)."""
    z, tables = load_golden('distmult', 2)
    n_ent, n_rel = int(z['n_ent']), int(z['n_rel'])
    heads, tails, rels = (torch.from_numpy(z[k]) for k in ('heads', 'tails', 'rels'))
    kg = tk.KnowledgeGraph(kg={'heads': heads, 'tails': tails, 'relations': rels},
                           ent2ix={i: i for i in range(n_ent)}, rel2ix={i: i for i in range(n_rel)})
    nt = int(z['n_test'])
    _, kg_test = kg.split_kg(sizes=(len(heads) - nt, nt))
    m = tk.DistMultModel(tables[0].shape[1], n_ent, n_rel)
    for fused in (True, False):
        ev = tk.LinkPredictionEvaluator(m, kg_test, fused=fused, engine=OracleRankEngine('distmult', tables))
        ev.evaluate(b_size=16, verbose=False)
        for nm in ('rank_true_heads', 'rank_true_tails', 'filt_rank_true_heads', 'filt_rank_true_tails'):
            assert np.array_equal(getattr(ev, nm).numpy(), z[nm])
    assert abs(ev.mrr()[1] - z['mrr'][1]) < 1e-7


def test_filter_index_torch_build_and_disk_cache(tmp_path):
    """Device-style (torch sort / unique) CSR build == numpy build == dict build;
    the on-disk cache round-trips."""
    import numpy as np
    from torchkge_amd.filter_index import FilterIndex
    g = torch.Generator().manual_seed(3)
    n = 5000
    h = torch.randint(0, 60, (n,), generator=g); t = torch.randint(0, 60, (n,), generator=g)
    r = torch.randint(0, 5, (n,), generator=g)
    a = FilterIndex.from_triples(h.numpy(), r.numpy(), t.numpy(), 'cpu')
    b = FilterIndex.from_triples_torch(h, r, t, 'cpu')
    d = {}
    for hh, rr, tt in zip(h.tolist(), r.tolist(), t.tolist()):
        d.setdefault((hh, rr), set()).add(tt)
    c = FilterIndex.from_dict({k: sorted(v) for k, v in d.items()}, 'cpu')
    for x in (b, c):
        assert torch.equal(a.keys, x.keys) and torch.equal(a.offsets, x.offsets)
        assert torch.equal(a.targets[:int(a.offsets[-1])], x.targets[:int(x.offsets[-1])])
    p = a.save(str(tmp_path / 'idx.npz'))
    z = FilterIndex.load(p, 'cpu')
    assert torch.equal(a.keys, z.keys) and torch.equal(a.offsets, z.offsets) and torch.equal(a.targets, z.targets)
    # KnowledgeGraph-level cache: second graph object with the same triples loads the file
    import torchkge_amd as tk
    kw = dict(ent2ix={i: i for i in range(60)}, rel2ix={i: i for i in range(5)})
    kg1 = tk.KnowledgeGraph(kg={'heads': h, 'tails': t, 'relations': r}, **kw)
    kg1.set_filter_cache(str(tmp_path / 'cache'))
    i1 = kg1.filter_index('tails', 'cpu')
    files = sorted((tmp_path / 'cache').iterdir())
    assert len(files) == 1
    kg2 = tk.KnowledgeGraph(kg={'heads': h.clone(), 'tails': t.clone(), 'relations': r.clone()}, **kw)
    kg2.set_filter_cache(str(tmp_path / 'cache'))
    i2 = kg2.filter_index('tails', 'cpu')
    assert torch.equal(i1.keys, i2.keys) and torch.equal(i1.targets, i2.targets) and torch.equal(i1.keys, a.keys)


def test_filter_plan_groups_queries_by_segment():
    """FilterPlan: only the FIRST query of a distinct (non-empty) segment owns its pairs; woff is the
    exclusive prefix sum of the owned lengths; long_q lists the queries with more than 512 entries."""
    from torchkge_amd.filter_index import FilterPlan
    seg_lo = torch.tensor([0, 0, 10, 0, 10, 700, 0, 5])
    seg_hi = torch.tensor([5, 5, 700, 0, 700, 701, 0, 10])      # queries 3 and 6: key absent (empty segment)
    true = torch.arange(8)
    targets = torch.zeros(701, dtype=torch.int32)
    p = FilterPlan(seg_lo, seg_hi, true, targets)
    owned = [5, 0, 690, 0, 0, 1, 0, 5]
    assert p.woff.tolist() == [0] + list(np.cumsum(owned))
    assert p.n_pairs == sum(owned) and p.long_q.tolist() == [2, 4] and p.n_long == 2


def test_internal_batch_of_the_fused_evaluator():
    """LinkPredictionEvaluator._internal_batch: a reference-style small b_size is a lower bound for the batch the fused
    kernels see; bounded by the facts at hand and by the uncertain-pair list of a batch; literal when coalescing is off,
    for the score exchange and for engines other than the HIP one."""
    import torchkge_amd.evaluation as ev

    class _M(object):
        n_ent = 14541

    class _E(ev.LinkPredictionEvaluator):
        def __init__(self, **kw):
            self.coalesce = kw.get('coalesce')
            self.fused, self._generic_model = kw.get('fused', True), False
            self.engine = kw.get('engine', ev.HipRankEngine())
            self.shard, self.exchange, self.model = kw.get('shard'), kw.get('exchange', 'counts'), _M()
    old = ev.COALESCE_BATCH
    try:
        ev.COALESCE_BATCH = 32768
        e = _E()
        assert e._internal_batch(256, 20466) == 20466 and e._internal_batch(256, 10 ** 6) == 32768
        assert e._internal_batch(40000, 10 ** 6) == 40000 and e._internal_batch(7, 0) == 7
        assert _E(coalesce=0)._internal_batch(256, 20466) == 256
        assert _E(fused=False)._internal_batch(256, 20466) == 256
        assert _E(shard='entities', exchange='scores')._internal_batch(256, 20466) == 20466   # row tiles bound the memory
        assert _E(engine=object())._internal_batch(256, 20466) == 256
        _M.n_ent = 4594485          # Wikidata5M: the list of a batch would pass 2 GiB long before 32768 facts
        assert 256 < e._internal_batch(256, 5133) < 5133
        assert e._internal_batch(8192, 5133) == 8192
    finally:
        ev.COALESCE_BATCH = old
        _M.n_ent = 14541


def test_documented_abi_version_is_the_bindings():
    """INTEGRATION.md / DESIGN.md state the ABI version a maintainer binds against: the one the binding checks."""
    import re
    from torchkge_amd import _hip
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    m = re.search(r'kge_abi_version\(\)` \(currently \*\*(\d+)\*\*', open(os.path.join(root, 'INTEGRATION.md')).read())
    assert m and int(m.group(1)) == _hip.ABI_VERSION
    m = re.search(r'`include/kge_hip.h` \(ABI (\d+)\)', open(os.path.join(root, 'DESIGN.md')).read())
    assert m and int(m.group(1)) == _hip.ABI_VERSION


def test_every_profile_file_the_documents_cite_exists():
    """DESIGN.md / README.md / INTEGRATION.md / tools/README.md quote measurements by file: every `profiles/rNN/...` path they
    name is a committed file (or a glob that matches one)."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    missing = []
    for doc in ('DESIGN.md', 'README.md', 'INTEGRATION.md', os.path.join('tools', 'README.md')):
        text = open(os.path.join(root, doc)).read()
        for path in set(re.findall(r'profiles/r0\d/[A-Za-z0-9_.*-]+', text)):
            path = path.rstrip('.,)')
            if not glob.glob(os.path.join(root, path)) and not glob.glob(os.path.join(root, path + '*')):
                missing.append((doc, path))
    assert not missing, missing


def test_names_the_engine_leaves_out_behave_like_absent_attributes():
    """the module-level __getattr__ of the package raises an AttributeError subclass -- hasattr / getattr with
    a default work as for any missing attribute, `from ... import` fails with ImportError, plain access says what to use."""
    import torchkge_amd as tk
    import torchkge_amd.utils as tku
    for mod, name in ((tk, 'RelationInference'), (tk, 'PositionalNegativeSampler'), (tk, 'TripletClassificationEvaluator'),
                      (tku, 'Trainer'), (tku, 'TrainDataLoader')):
        assert not hasattr(mod, name) and getattr(mod, name, None) is None
        with pytest.raises(AttributeError, match='does not provide'):
            getattr(mod, name)
    with pytest.raises(ImportError):
        from torchkge_amd import RelationInference  # noqa: F401
    assert not hasattr(tk, 'no_such_name')


def test_relation_ids_of_a_both_sides_batch_take_the_plans_vector_only_when_it_fits():
    """translation._both_r: the [r | r] vector the evaluator precomputes with a batch's FilterPlan is used as it is -- and
    only for a both-sides batch of exactly that size on the same device; anything else concatenates as before."""
    from torchkge_amd.models.translation import _both_r
    r = torch.tensor([3, 1, 2], dtype=torch.int64)
    hint = torch.cat([r, r])
    assert _both_r(r, _hip.SIDE_BOTH, hint) is hint
    assert torch.equal(_both_r(r, _hip.SIDE_BOTH, None), hint)
    assert torch.equal(_both_r(r, _hip.SIDE_BOTH, torch.cat([r, r, r])), hint)      # a stale vector of another batch size
    assert _both_r(r, _hip.SIDE_TAIL, hint).shape[0] == 3


def test_region_helpers_of_the_c_abi():
    """kge_lp_split_regions: three regions (32 queries each) per panel of 96 padded query rows; kge_lp_dot_table_prep_blocks:
    the sizes of the two scratch arrays of the DOT candidate preparation (host-only helpers: no GPU needed)."""
    lib = _hip.load_library()
    for B in (1, 96, 97, 192, 193, 40932):
        Bp = int(lib.kge_lp_split_rows_padded(B, 1))
        assert Bp % 96 == 0 and int(lib.kge_lp_split_regions(B)) == Bp // 96 * 3
    for rows in (1, 63, 64, 65, 14951, 4594485):
        a, b = int(lib.kge_lp_dot_table_prep_blocks(rows, 0)), int(lib.kge_lp_dot_table_prep_blocks(rows, 1))
        assert 1 <= a <= 2048 and a == min((rows + 63) // 64, 2048)
        assert 1 <= b <= 4096 and b == min(int(lib.kge_lp_split_rows_padded(rows, 0)) // 16, 4096)


def test_training_criteria_equal_the_stock_torch_ones():
    """utils/losses.py writes the three criteria out as elementwise formulas: the same values as the torch.nn criteria with
    reduction='sum' that the reference wraps (utils/losses.py:19-112), and the same gradients."""
    from torchkge_amd.utils import MarginLoss, LogisticLoss, BinaryCrossEntropyLoss
    g = torch.Generator().manual_seed(3)
    pos0, neg0 = 3 * torch.randn(257, generator=g), 3 * torch.randn(257, generator=g)
    ones = torch.ones(257)
    stock = {
        'margin': lambda p, n: torch.nn.MarginRankingLoss(margin=0.5, reduction='sum')(p, n, target=ones),
        'logistic': lambda p, n: torch.nn.SoftMarginLoss(reduction='sum')(p, ones) + torch.nn.SoftMarginLoss(reduction='sum')(n, -ones),
        'bce': lambda p, n: torch.nn.BCELoss(reduction='sum')(torch.sigmoid(torch.cat([p, n])), torch.cat([ones, 0 * ones])),
    }
    ours = {'margin': MarginLoss(0.5), 'logistic': LogisticLoss(), 'bce': BinaryCrossEntropyLoss()}
    for name in stock:
        pa, na = pos0.clone().requires_grad_(True), neg0.clone().requires_grad_(True)
        pb, nb = pos0.clone().requires_grad_(True), neg0.clone().requires_grad_(True)
        la, lb = ours[name](pa, na), stock[name](pb, nb)
        assert torch.allclose(la, lb, rtol=1e-6, atol=0), name
        la.backward(); lb.backward()
        assert torch.allclose(pa.grad, pb.grad, rtol=1e-5, atol=1e-7) and torch.allclose(na.grad, nb.grad, rtol=1e-5, atol=1e-7), name


def test_mfma_kernels_hold_no_lane_crossing_packed_f32_operand():
    """r05 / r06: hipcc's SLP vectoriser packs scalar f32 arithmetic of the count kernels' epilogues into v_pk_*_f32 and, when
    the scalars it wants to splat sit in one register pair, selects them with op_sel -- and the form in which the LOW result
    lane reads the HIGH dword of a VGPR pair (op_sel:[.,1,.]) misreads ~once per 1e3 wave-instructions inside the free-running
    kernel, where another wave's MFMAs co-execute (tools/probe/slp_bisect.sh, profiles/r06/slp_bisect.txt).  The three MFMA
    count kernels are therefore built without the vectoriser; this compiles each of them to gfx950 assembly with the build's
    own flags and checks that NO packed f32 instruction carries an op_sel:[...] with a set bit."""
    from concurrent.futures import ThreadPoolExecutor
    from torchkge_amd.csrc import build as hb
    hipcc = hb._hipcc()
    files = [f for f, fl in hb.EXTRA_FLAGS.items() if '-DKGE_BUILD_NO_SLP=1' in fl]
    assert set(files) >= {'lp_hi_stream.hip', 'lp_hi_chunk.hip', 'lp_split_mfma.hip'}

    def isa(src):
        cmd = [hipcc] + hb.FLAGS + hb.EXTRA_FLAGS[src] + ['-S', '--cuda-device-only', os.path.join(hb.HERE, src), '-o', '-']
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return src, r.stdout
    with ThreadPoolExecutor(max_workers=3) as ex:
        for src, text in ex.map(isa, files):
            assert 'v_mfma_f32_32x32x16_f16' in text, src
            bad = [l.strip() for l in text.split('\n')
                   if re.search(r'\bv_pk_(fma|mul|add)_f32\b', l) and re.search(r'op_sel:\[[01,]*1[01,]*\]', l)]
            assert not bad, (src, bad[:4])
    # ... and without the build's flag the sources refuse to compile at all
    r = subprocess.run([hipcc] + hb.FLAGS + ['-fsyntax-only', '--cuda-device-only', os.path.join(hb.HERE, 'lp_hi_chunk.hip')],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode != 0 and 'KGE_BUILD_NO_SLP' in r.stderr


def test_region_policy_and_lazy_rank_rows_of_the_evaluator():
    """Host logic of r06 (no GPU): (1) the sweep's uncertain pairs go to regions from REGION_MIN_LEVEL0 re-scored pairs per query
    seen on the three-product level -- and, once the one-product level has been measured, only from REGION_MIN_LEVEL1 there;
    never on several ranks, never on level 0.  (2) The four rank vectors are rows of the packed host tensor of the last
    steady-state evaluation, handed out on access; an assigned tensor wins.  (3) The level thresholds scale with the number of
    candidates."""
    from torchkge_amd import evaluation as evm
    m = tk.TransEModel(8, 20, 3, 'L2')
    h = torch.randint(0, 20, (30,)); t = torch.randint(0, 20, (30,)); r = torch.randint(0, 3, (30,))
    kg = tk.KnowledgeGraph(kg={'heads': h, 'tails': t, 'relations': r}, ent2ix={j: j for j in range(20)},
                           rel2ix={j: j for j in range(3)})
    ev = tk.LinkPredictionEvaluator(m, kg, share_state=False)
    assert ev._regions_for(1, False) is False                       # nothing seen yet
    ev._level0_seen = evm.REGION_MIN_LEVEL0 + 0.1
    assert ev._regions_for(1, False) and not ev._regions_for(0, False) and not ev._regions_for(1, True)
    ev._level1_seen = evm.REGION_MIN_LEVEL1 - 0.5                   # measured on the level the regions serve: too few pairs
    assert ev._regions_for(1, False) is False
    ev._level1_seen = evm.REGION_MIN_LEVEL1 + 0.5
    assert ev._regions_for(1, False) is True
    ev._level0_seen = evm.REGION_MIN_LEVEL0 - 0.1
    assert ev._regions_for(1, False) is False
    # (2)
    n = 7
    packed = torch.arange(4 * n + 2, dtype=torch.int64)
    ev.__dict__['_rank_rows'] = packed.as_strided((4, n), (n, 1))
    ev.__dict__['_rank_set'] = {}
    assert torch.equal(ev.rank_true_heads, packed[:n]) and torch.equal(ev.filt_rank_true_tails, packed[3 * n:4 * n])
    assert ev.rank_true_tails is ev.rank_true_tails                 # built once per evaluation
    ev.rank_true_tails = torch.zeros(n, dtype=torch.int64)
    assert int(ev.rank_true_tails.sum()) == 0 and torch.equal(ev.filt_rank_true_heads, packed[2 * n:3 * n])
    # (3)
    e1, l1 = evm.level1_thresholds(14541)
    e5, l5 = evm.level1_thresholds(4594485)
    assert (e1, l1) == (evm.LEVEL1_ENTER, evm.LEVEL1_LEAVE) and e5 == pytest.approx(e1 * 4594485 / 14541) and l5 > e5
    assert evm.level1_thresholds(100) == (e1, l1)                   # never below the FB15k-237 figures


def test_evaluator_state_cache_is_bounded_weak_and_clearable():
    """What evaluators learn about (model, kg, options) is shared at module level (evaluation._EvalState): at most
    MAX_STATES_PER_MODEL states per model (least recently used first out), none once the model is gone, and
    clear_eval_state(model) drops them -- ADVICE r05: the cache used to grow without bound and its graph-segment closures
    kept the model alive through the evaluator."""
    import gc
    import weakref
    from torchkge_amd import evaluation as evm
    m = tk.TransEModel(8, 20, 3, 'L2')
    kgs = []
    for i in range(evm.MAX_STATES_PER_MODEL + 3):
        h = torch.randint(0, 20, (30,)); t = torch.randint(0, 20, (30,)); r = torch.randint(0, 3, (30,))
        kgs.append(tk.KnowledgeGraph(kg={'heads': h, 'tails': t, 'relations': r}, ent2ix={j: j for j in range(20)},
                                     rel2ix={j: j for j in range(3)}))
    evs = [tk.LinkPredictionEvaluator(m, kg) for kg in kgs]
    assert len(evm._STATES[m]) == evm.MAX_STATES_PER_MODEL
    assert tk.LinkPredictionEvaluator(m, kgs[-1])._st is evs[-1]._st          # same (model, kg, options): shared
    assert tk.LinkPredictionEvaluator(m, kgs[0])._st is not evs[0]._st        # the evicted one: a fresh state
    assert tk.LinkPredictionEvaluator(m, kgs[-1], share_state=False)._st is not evs[-1]._st
    # the closures recorded for graph replays hold the STATE, never the evaluator (which holds the model)
    call = evs[-1]._timed(lambda: None)
    assert all(c.cell_contents is not evs[-1] for c in (call.__closure__ or ()))
    tk.clear_eval_state(m)
    assert m not in evm._STATES
    ev = tk.LinkPredictionEvaluator(m, kgs[1])
    assert m in evm._STATES
    ref = weakref.ref(m)
    del m, ev, evs, call
    gc.collect()
    assert ref() is None and len(evm._STATES) == 0 or all(k is not None for k in evm._STATES.keys())
    assert ref() is None


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without torchrun re-executes itself under torch.distributed.run (one rank per GPU; here the
    launch check only: rendezvous on 127.0.0.1 over gloo, one all-reduce, rank 0 prints ONE JSON line, no GPU touched)."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--launch-check'],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.split('\n') if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d['launch_check'] and d['n_gpus'] == 2 and d['world_size'] == 2 and d['sum_of_rank_plus_one'] == 3.0
