"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through
the C-ABI (ctypes binding torchkge_amd/_hip.py), against
  * the C oracle's fmaf-chain contract            -> BIT-EXACT fp32,
  * the integer semantics of the reference         -> BIT-EXACT int64,
  * the golden fixtures generated from the reference (tests/golden) -> 1e-5.
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import kge_oracle as orc
from tests.helpers import load_golden, oracle_clib, fptr, dict_to_csr, GOLDEN

pytestmark = pytest.mark.gpu

TOL = 1e-5     # north-star floating-point tolerance on scores
CASES = [('transe', 2), ('transe', 1), ('transh', 2), ('transd', 2), ('distmult', 2), ('complex', 2)]
i64 = ctypes.c_int64


@pytest.fixture(scope='module')
def hip():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    from torchkge_amd import _hip
    _hip.load_library()
    return _hip


def dev(x):
    return torch.as_tensor(x).cuda()


def build_model(kind, p, tables, n_ent, n_rel):
    import torchkge_amd as tk
    d = tables[0].shape[1]
    if kind == 'transe':
        m = tk.TransEModel(d, n_ent, n_rel, dissimilarity_type='L%d' % p)
        names = ['ent_emb', 'rel_emb']
    elif kind == 'transh':
        m = tk.TransHModel(d, n_ent, n_rel)
        names = ['ent_emb', 'rel_emb', 'norm_vect']
    elif kind == 'transd':
        m = tk.TransDModel(d, tables[1].shape[1], n_ent, n_rel)
        names = ['ent_emb', 'rel_emb', 'ent_proj_vect', 'rel_proj_vect']
    elif kind == 'distmult':
        m = tk.DistMultModel(d, n_ent, n_rel)
        names = ['ent_emb', 'rel_emb']
    else:
        m = tk.ComplExModel(d, n_ent, n_rel)
        names = ['re_ent_emb', 'im_ent_emb', 're_rel_emb', 'im_rel_emb']
    sd = {n + '.weight': t.clone() for n, t in zip(names, tables)}
    m.load_state_dict(sd)
    return m.cuda()


def golden_batch(z):
    B = int(z['b_size']); n = len(z['heads']); nt = int(z['n_test'])
    h = torch.from_numpy(z['heads'][n - nt:][:B]); t = torch.from_numpy(z['tails'][n - nt:][:B])
    r = torch.from_numpy(z['rels'][n - nt:][:B])
    return h, t, r


# ---------------------------------------------------------------------------
# fp32 arithmetic contract: MFMA / VALU tile kernels == CPU fmaf chains, bitwise
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('B,N,K', [(70, 300, 32), (130, 515, 200), (5, 129, 50), (257, 1000, 8), (1, 1, 3)])
def test_mfma_gemm_bit_exact_vs_chain(hip, B, N, K):
    lib = oracle_clib()
    g = torch.Generator().manual_seed(B * 1000 + K)
    A = (torch.rand(B, K, generator=g) * 2 - 1)
    T = (torch.rand(N, K, generator=g) * 2 - 1)
    A1 = (torch.rand(B, K, generator=g) * 2 - 1)
    T1 = (torch.rand(N, K, generator=g) * 2 - 1)
    ref = np.empty((B, N), dtype=np.float32)
    # DOT, one segment
    lib.orc_lp_gemm_chain(fptr(A.numpy()), i64(K), fptr(T.numpy()), i64(K), i64(K), None, i64(0), None, i64(0),
                          i64(0), i64(B), i64(N), 0, None, None, fptr(ref))
    prob = hip.LpProblem(hip.LP_DOT, dev(A), dev(T))
    out = prob.scores().cpu().numpy()
    assert np.array_equal(out, ref)
    # pair kernel == tile kernel
    ci = torch.randint(0, N, (B,), generator=g)
    ps = prob.pair_scores(dev(ci)).cpu().numpy()
    assert np.array_equal(ps, ref[np.arange(B), ci.numpy()])
    # DOT, two segments (ComplEx)
    lib.orc_lp_gemm_chain(fptr(A.numpy()), i64(K), fptr(T.numpy()), i64(K), i64(K), fptr(A1.numpy()), i64(K),
                          fptr(T1.numpy()), i64(K), i64(K), i64(B), i64(N), 0, None, None, fptr(ref))
    prob2 = hip.LpProblem(hip.LP_DOT, dev(A), dev(T), A1=dev(A1), T1=dev(T1))
    assert np.array_equal(prob2.scores().cpu().numpy(), ref)
    # L2 via norm expansion
    qn = np.empty(B, dtype=np.float32); en = np.empty(N, dtype=np.float32)
    lib.orc_row_sqnorm_chain(fptr(A.numpy()), i64(K), i64(B), i64(K), fptr(qn))
    lib.orc_row_sqnorm_chain(fptr(T.numpy()), i64(K), i64(N), i64(K), fptr(en))
    dA, dT = dev(A), dev(T)
    qn_d, en_d = hip.row_sqnorm(dA), hip.row_sqnorm(dT)
    assert np.array_equal(qn_d.cpu().numpy(), qn) and np.array_equal(en_d.cpu().numpy(), en)
    lib.orc_lp_gemm_chain(fptr(A.numpy()), i64(K), fptr(T.numpy()), i64(K), i64(K), None, i64(0), None, i64(0),
                          i64(0), i64(B), i64(N), 1, fptr(qn), fptr(en), fptr(ref))
    prob3 = hip.LpProblem(hip.LP_L2_EXPAND, dA, dT, qn=qn_d, en=en_d)
    assert np.array_equal(prob3.scores().cpu().numpy(), ref)
    # fused count == count on the materialised matrix
    s_true = prob3.pair_scores(dev(ci))
    raw = prob3.count_ge(s_true).cpu().numpy()
    assert np.array_equal(raw, (ref >= ref[np.arange(B), ci.numpy()][:, None]).sum(1))


@pytest.mark.parametrize('B,N,K,p,axpy', [(70, 300, 32, 2, False), (33, 515, 200, 1, False),
                                          (130, 260, 50, 2, False), (65, 300, 24, 2, True),
                                          (20, 131, 200, 1, True), (3, 7, 5, 2, True)])
def test_direct_kernels_bit_exact_vs_chain(hip, B, N, K, p, axpy):
    lib = oracle_clib()
    g = torch.Generator().manual_seed(B + N + K)
    Q = torch.rand(B, K, generator=g) * 2 - 1
    T = torch.rand(N, K + 3, generator=g)[:, :K] * 2 - 1         # ld != K (TransD-style slice)
    Tn = np.ascontiguousarray(T.numpy())
    R = 5
    W = torch.rand(B, K, generator=g) - 0.5
    scal = torch.rand(N, R, generator=g) - 0.5
    ridx = torch.randint(0, R, (B,), generator=g)
    ref = np.empty((B, N), dtype=np.float32)
    lib.orc_lp_direct_chain(fptr(Q.numpy()), i64(K), fptr(Tn), i64(K), i64(K),
                            fptr(W.numpy()) if axpy else None, i64(K),
                            fptr(scal.numpy()) if axpy else None, i64(R),
                            fptr(ridx.numpy()) if axpy else None, i64(B), i64(N), p, fptr(ref))
    Td = dev(torch.rand(N, K + 3))  # device table with ld = K+3
    Td[:, :K] = dev(T)
    mode = hip.LP_L1_DIRECT if p == 1 else hip.LP_L2_DIRECT
    prob = hip.LpProblem(mode, dev(Q), Td, Wq=dev(W) if axpy else None, scal=dev(scal) if axpy else None,
                         r_idx=dev(ridx) if axpy else None, K0=K)
    out = prob.scores().cpu().numpy()
    assert np.array_equal(out, ref)
    ci = torch.randint(0, N, (B,), generator=g)
    st = prob.pair_scores(dev(ci))
    assert np.array_equal(st.cpu().numpy(), ref[np.arange(B), ci.numpy()])
    raw = prob.count_ge(st).cpu().numpy()
    assert np.array_equal(raw, (ref >= ref[np.arange(B), ci.numpy()][:, None]).sum(1))


# ---------------------------------------------------------------------------
# golden fixtures generated from the reference
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('kind,p', CASES)
def test_scoring_function_and_forward_vs_reference(hip, kind, p):
    z, tables = load_golden(kind, p)
    m = build_model(kind, p, tables, int(z['n_ent']), int(z['n_rel']))
    h, t, r = (x.cuda() for x in golden_batch(z))
    with torch.no_grad():
        sf = m.scoring_function(h, t, r)
        pos, neg = m(h, t, r, dev(z['neg_heads']), dev(z['neg_tails']))
    assert np.abs(sf.cpu().numpy() - z['sf']).max() < TOL
    assert np.abs(pos.cpu().numpy() - z['fwd_pos']).max() < TOL
    assert np.abs(neg.cpu().numpy() - z['fwd_neg']).max() < TOL


@pytest.mark.parametrize('kind,p', CASES)
def test_inference_api_vs_reference(hip, kind, p):
    """Drop-in API: inference_prepare_candidates + inference_scoring_function
    (and the lp_* aliases) give the reference's (b, N) score matrices."""
    z, tables = load_golden(kind, p)
    m = build_model(kind, p, tables, int(z['n_ent']), int(z['n_rel']))
    h, t, r = (x.cuda() for x in golden_batch(z))
    h_e, t_e, r_e, cand = m.inference_prepare_candidates(h, t, r, entities=True)
    s_tail = m.inference_scoring_function(h_e, cand, r_e)
    s_head = m.lp_scoring_function(cand, t_e, r_e)
    assert s_tail.shape == z['s_tail'].shape
    assert np.abs(s_tail.cpu().numpy() - z['s_tail']).max() < TOL
    assert np.abs(s_head.cpu().numpy() - z['s_head']).max() < TOL
    if kind == 'transe' and p == 2:
        m.l2_mode = 'direct'     # broadcast-subtract path gives the same matrix within tolerance
        s2 = m.inference_scoring_function(h_e, cand, r_e)
        assert np.abs(s2.cpu().numpy() - z['s_tail']).max() < TOL
        # a materialised (non-broadcast) candidate tensor takes the generic batched kernel
        s3 = m.inference_scoring_function(h_e, cand.contiguous(), r_e)
        assert np.abs(s3.cpu().numpy() - z['s_tail']).max() < TOL


@pytest.mark.parametrize('kind,p', [('transe', 2), ('complex', 2), ('transh', 2)])
def test_rank_and_filter_bit_exact_on_reference_scores(hip, kind, p):
    """get_rank / filter_scores are integer functions of a score matrix: fed the
    reference's own matrix they must reproduce the reference exactly."""
    import torchkge_amd as tk
    from torchkge_amd.utils import get_rank, filter_scores
    z, _ = load_golden(kind, p)
    h, t, r = golden_batch(z)
    heads, tails, rels = (torch.from_numpy(z[k]) for k in ('heads', 'tails', 'rels'))
    _, dict_of_tails, _ = orc.build_filter_dicts(heads, tails, rels)
    s = dev(z['s_tail'])
    rk = get_rank(s, t.cuda())
    assert rk.dtype == torch.int64 and np.array_equal(rk.cpu().numpy(), z['rk_tail'])
    f = filter_scores(s, dict_of_tails, h.cuda(), r.cuda(), t.cuda())
    assert np.array_equal(f.cpu().numpy(), z['f_tail'])
    assert np.array_equal(get_rank(f, t.cuda()).cpu().numpy(), z['frk_tail'])
    # fused rank+filter on the materialised matrix
    from torchkge_amd.filter_index import filter_index_for
    idx = filter_index_for(dict_of_tails, 'cuda')
    lo, hi = idx.lookup(h.cuda(), r.cuda())
    rk2, frk2 = hip.filtered_rank_from_scores(s, t.cuda(), lo, hi, idx.targets)
    assert np.array_equal(rk2.cpu().numpy(), z['rk_tail']) and np.array_equal(frk2.cpu().numpy(), z['frk_tail'])
    # low_values and the reference's known-answer vector (tests/test_utils.py:156-163)
    data = dev(np.array([[1, 2, 3, 4, 0], [1, 2, 1, 3, 0]], dtype=np.float32))
    true = dev(np.array([4, 2]))
    assert get_rank(data, true).tolist() == [5, 4]
    assert get_rank(data, true, low_values=True).tolist() == [1, 3]


def test_filter_edge_cases_bit_exact(hip):
    """missing key, key whose set lacks the true entity (row untouched), -inf
    true score, NaN entries -- against the C oracle."""
    lib = oracle_clib()
    B, N = 6, 40
    g = torch.Generator().manual_seed(3)
    s = torch.randn(B, N, generator=g)
    true = torch.tensor([3, 5, 7, 9, 11, 13])
    s[2, 7] = -float('inf')          # -inf true score: filtered -inf entries still count
    s[3, 1] = float('nan')
    s[4, 11] = float('nan')          # NaN true score: nothing counts
    dictionary = {(0, 0): {3, 4, 5}, (1, 0): {1, 2}, (2, 0): {7, 8, 30}, (3, 0): {9, 1, 2},
                  (4, 0): {11, 0}}
    key1 = torch.arange(B); key2 = torch.zeros(B, dtype=torch.long)
    has, off, tgt = dict_to_csr(dictionary, key1, key2)
    sn = np.ascontiguousarray(s.numpy())
    rk = np.empty(B, dtype=np.int64); frk = np.empty(B, dtype=np.int64)
    lib.orc_filtered_rank(fptr(sn), i64(B), i64(N), fptr(true.numpy()), fptr(has), fptr(off), fptr(tgt),
                          fptr(rk), fptr(frk))
    fs = sn.copy()
    lib.orc_filter_scores(fptr(fs), i64(B), i64(N), fptr(true.numpy()), fptr(has), fptr(off), fptr(tgt))
    # python oracle agrees with the C oracle
    fo = orc.filter_scores(s, dictionary, key1, key2, true)
    assert np.array_equal(fo.numpy(), fs, equal_nan=True)
    assert np.array_equal(orc.get_rank(fo, true).numpy(), frk)
    from torchkge_amd.utils import filter_scores, get_rank
    f = filter_scores(s.cuda(), dictionary, key1.cuda(), key2.cuda(), true.cuda())
    assert np.array_equal(f.cpu().numpy(), fs, equal_nan=True)
    assert np.array_equal(get_rank(s.cuda(), true.cuda()).cpu().numpy(), rk)
    assert np.array_equal(get_rank(f, true.cuda()).cpu().numpy(), frk)
    from torchkge_amd.filter_index import FilterIndex
    idx = FilterIndex.from_dict(dictionary, 'cuda')
    lo, hi = idx.lookup(key1.cuda(), key2.cuda())
    rk2, frk2 = hip.filtered_rank_from_scores(s.cuda(), true.cuda(), lo, hi, idx.targets)
    assert np.array_equal(rk2.cpu().numpy(), rk) and np.array_equal(frk2.cpu().numpy(), frk)


def _rank_bounds(scores, true_idx, tol):
    st = scores.gather(1, true_idx.view(-1, 1))
    return (scores >= st + tol).sum(1), (scores >= st - tol).sum(1)


@pytest.mark.parametrize('coalesce_mode', ['literal', 'default'])
@pytest.mark.parametrize('kind,p', CASES)
def test_evaluator_vs_reference(hip, kind, p, coalesce_mode):
    """LinkPredictionEvaluator: fused == materialised == composed drop-in path
    bit for bit; vs the reference's ranks: equal, except where the reference's
    own scores tie within 2*TOL (then inside the tie interval); metrics 1e-5."""
    import torchkge_amd as tk
    z, tables = load_golden(kind, p)
    n_ent, n_rel = int(z['n_ent']), int(z['n_rel'])
    m = build_model(kind, p, tables, n_ent, n_rel)
    heads, tails, rels = (torch.from_numpy(z[k]) for k in ('heads', 'tails', 'rels'))
    kg = tk.KnowledgeGraph(kg={'heads': heads, 'tails': tails, 'relations': rels},
                           ent2ix={i: i for i in range(n_ent)}, rel2ix={i: i for i in range(n_rel)})
    nt = int(z['n_test']); n = len(heads)
    _, kg_test = kg.split_kg(sizes=(n - nt, nt))
    B = int(z['b_size'])
    ev = tk.LinkPredictionEvaluator(m, kg_test)
    with pytest.raises(tk.NotYetEvaluatedError):
        ev.mrr()
    ev.evaluate(b_size=B, verbose=False)
    ev2 = tk.LinkPredictionEvaluator(m, kg_test, fused=False)
    ev2.evaluate(b_size=7, verbose=False)           # different batch size, short last batch
    names = ['rank_true_heads', 'rank_true_tails', 'filt_rank_true_heads', 'filt_rank_true_tails']
    for nm in names:
        a = getattr(ev, nm)
        assert a.dtype == torch.int64 and a.shape[0] == nt and not a.is_cuda
        assert torch.equal(a, getattr(ev2, nm))
    # composed public API path (what a user-defined Model subclass gets)
    ev3 = tk.LinkPredictionEvaluator(m, kg_test, both_sides=False)
    import types

    def generic_side(self, h, t, r, side, index, lo, hi, sharded):
        key1, true_idx = (h, t) if side == 'tail' else (t, h)
        return self._rank_side_generic(h, t, r, side, index, true_idx, key1)
    ev3._rank_side = types.MethodType(generic_side, ev3)
    ev3.evaluate(b_size=B, verbose=False)
    for nm in names:
        assert torch.equal(getattr(ev, nm), getattr(ev3, nm))
    # two-stream overlap of the short kernels (off by default: the count kernel fills every CU) gives the same ranks
    ev6 = tk.LinkPredictionEvaluator(m, kg_test, overlap=True)
    ev6.evaluate(b_size=B, verbose=False)
    for nm in names:
        assert torch.equal(getattr(ev, nm), getattr(ev6, nm))
    # the default ranks both sides of a batch as ONE problem of 2B queries; side by side gives the same ranks
    ev7 = tk.LinkPredictionEvaluator(m, kg_test, both_sides=False)
    ev7.evaluate(b_size=5, verbose=False)
    for nm in names:
        assert torch.equal(getattr(ev, nm), getattr(ev7, nm))
    # a reference-style small b_size is coalesced into large internal batches (evaluation._internal_batch): same ranks
    ev8 = tk.LinkPredictionEvaluator(m, kg_test, coalesce=32768)
    ev8.evaluate(b_size=3, verbose=False)
    assert ev8._internal_batch(3, nt) == nt                                        # (the shipped default does the same:
    assert ev._internal_batch(3, nt) == (3 if coalesce_mode == 'literal' else nt)  #  conftest switches it off in 'literal' mode)
    for nm in names:
        assert torch.equal(getattr(ev, nm), getattr(ev8, nm))
    # ... and so does the SHIPPED default configuration replayed as a hipGraph (coalesced internal batch: its plan keys,
    # the per-internal-batch relation sort and list capacity differ from the literal-batch runs above)
    ev9 = tk.LinkPredictionEvaluator(m, kg_test, coalesce=32768, graph=True)
    for _ in range(3):
        ev9.evaluate(b_size=3, verbose=False)
        for nm in names:
            assert torch.equal(getattr(ev, nm), getattr(ev9, nm))
    # hipGraph replay of the whole evaluate(): capture call and two replays, tables changed in between
    ev4 = tk.LinkPredictionEvaluator(m, kg_test, graph=True)
    ev4.evaluate(b_size=B, verbose=False)
    for nm in names:
        assert torch.equal(getattr(ev, nm), getattr(ev4, nm))
    prm = next(m.parameters())
    saved = prm.data.clone()
    prm.data.mul_(1.37)
    ev4.evaluate(b_size=B, verbose=False)
    ev5 = tk.LinkPredictionEvaluator(m, kg_test)
    ev5.evaluate(b_size=B, verbose=False)
    for nm in names:
        assert torch.equal(getattr(ev5, nm), getattr(ev4, nm))     # replay sees the new table values
    prm.data.copy_(saved)
    ev4.evaluate(b_size=B, verbose=False)
    for nm in names:
        assert torch.equal(getattr(ev, nm), getattr(ev4, nm))
    # vs the reference
    dh, dt, _ = orc.build_filter_dicts(heads, tails, rels)
    th, tt, tr = heads[n - nt:], tails[n - nt:], rels[n - nt:]
    for side, nm_raw, nm_f, true_idx, dic, k1 in [('tail', 'rank_true_tails', 'filt_rank_true_tails', tt, dt, th),
                                                   ('head', 'rank_true_heads', 'filt_rank_true_heads', th, dh, tt)]:
        s = orc.lp_scores(kind, tables, th, tt, tr, side, p)
        lo, hi = _rank_bounds(s, true_idx, 2 * TOL)
        got = getattr(ev, nm_raw)
        assert ((got >= lo) & (got <= hi)).all()
        f = orc.filter_scores(s, dic, k1, tr, true_idx)
        lo, hi = _rank_bounds(f, true_idx, 2 * TOL)
        got = getattr(ev, nm_f)
        assert ((got >= lo) & (got <= hi)).all()
    for nm in names:   # on these fixtures there are no near-ties: exact equality with the reference
        assert np.array_equal(getattr(ev, nm).numpy(), z[nm])
    assert abs(ev.hit_at_k(10)[1] - z['hit10'][1]) < TOL and abs(ev.hit_at_k(10)[0] - z['hit10'][0]) < TOL
    assert abs(ev.mrr()[1] - z['mrr'][1]) < TOL and abs(ev.mrr()[0] - z['mrr'][0]) < TOL
    assert abs(ev.mean_rank()[1] - z['mean_rank'][1]) < 1e-3
    ev.print_results(k=[1, 10])


# ---------------------------------------------------------------------------
# negative sampling: integer scatter bit-exact
# ---------------------------------------------------------------------------
def test_sampler_same_seed_same_samples(hip):
    import torchkge_amd as tk
    z = np.load(GOLDEN + '/ref_sampler.npz')
    heads, tails, rels = (torch.from_numpy(z[k]) for k in ('heads', 'tails', 'rels'))
    n_ent, n_rel = int(z['n_ent']), int(z['n_rel'])
    kg = tk.KnowledgeGraph(kg={'heads': heads, 'tails': tails, 'relations': rels},
                           ent2ix={i: i for i in range(n_ent)}, rel2ix={i: i for i in range(n_rel)})
    samp = tk.BernoulliNegativeSampler(kg, n_neg=3)
    assert np.array_equal(samp.bern_probs.numpy(), z['bern_probs'])       # host precompute: exact
    h, t, r = heads[:200].cuda(), tails[:200].cuda(), rels[:200].cuda()
    for n_neg in (1, 3):
        torch.manual_seed(99)
        nh, nt = samp.corrupt_batch(h, t, r, n_neg=n_neg)
        torch.manual_seed(99)     # reference op sequence on the same device RNG
        oh, ot, mask, dh, dt = orc.corrupt_batch(h, t, r, n_neg, samp.bern_probs, n_ent)
        assert nh.dtype == torch.int64 and nh.is_cuda
        assert torch.equal(nh, oh) and torch.equal(nt, ot)
    usamp = tk.UniformNegativeSampler(kg, n_neg=2)
    torch.manual_seed(5)
    nh, nt = usamp.corrupt_batch(h, t, r)
    torch.manual_seed(5)
    oh, ot, *_ = orc.corrupt_batch(h, t, r, 2, None, n_ent, uniform=True)
    assert torch.equal(nh, oh) and torch.equal(nt, ot)
    # corrupt_kg driver
    torch.manual_seed(1)
    ch, ct = samp.corrupt_kg(batch_size=128, use_cuda=True)
    assert ch.shape[0] == kg.n_facts and not ch.is_cuda
    # sync-free variant: valid corruption (exactly one side changed per mask)
    samp.sync_free = True
    nh, nt = samp.corrupt_batch(h, t, r, n_neg=2)
    changed_h = nh != h.repeat(2); changed_t = nt != t.repeat(2)
    assert not (changed_h & changed_t).any()
    assert int(nh.min()) >= 0 and int(nh.max()) < n_ent


@pytest.mark.parametrize('B,n_neg,pr', [(1, 1, 0.5), (1000, 1, 0.0), (1000, 2, 1.0), (4097, 3, 0.3),
                                        (32768, 1, 0.57), (300000, 4, 0.5)])
def test_corrupt_scatter_bit_exact(hip, B, n_neg, pr):
    lib = oracle_clib()
    g = torch.Generator().manual_seed(B + n_neg)
    n = B * n_neg
    heads = torch.randint(0, 10000, (B,), generator=g); tails = torch.randint(0, 10000, (B,), generator=g)
    mask = (torch.rand(n, generator=g) < pr).to(torch.uint8)
    k = int(mask.sum())
    dh = torch.randint(1, 10000, (k,), generator=g); dt = torch.randint(1, 10000, (n - k,), generator=g)
    oh = np.empty(n, dtype=np.int64); ot = np.empty(n, dtype=np.int64)
    lib.orc_corrupt_scatter(fptr(heads.numpy()), fptr(tails.numpy()), fptr(mask.numpy()), fptr(dh.numpy()),
                            fptr(dt.numpy()), i64(B), i64(n_neg), fptr(oh), fptr(ot))
    nh, nt = hip.corrupt_scatter(heads.cuda(), tails.cuda(), mask.cuda(), dh.cuda(), dt.cuda(), n_neg)
    assert np.array_equal(nh.cpu().numpy(), oh) and np.array_equal(nt.cpu().numpy(), ot)


# ---------------------------------------------------------------------------
# backward of scoring_function (training path)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('kind,p', CASES)
def test_scoring_function_backward_vs_autograd(hip, kind, p):
    z, tables = load_golden(kind, p)
    m = build_model(kind, p, tables, int(z['n_ent']), int(z['n_rel']))
    g = torch.Generator().manual_seed(11)
    B = 257
    h = torch.randint(0, int(z['n_ent']), (B,), generator=g).cuda()
    t = torch.randint(0, int(z['n_ent']), (B,), generator=g).cuda()
    r = torch.randint(0, int(z['n_rel']), (B,), generator=g).cuda()
    w = torch.rand(B, generator=g).cuda()
    m.zero_grad()
    (m.scoring_function(h, t, r) * w).sum().backward()
    got = [prm.grad.clone() for prm in m._tables()]
    ref_tabs = [x.clone().cuda().requires_grad_(True) for x in tables]
    (orc.score_triples(kind, ref_tabs, h, t, r, p=p) * w).sum().backward()
    for a, b in zip(got, ref_tabs):
        scale = max(1.0, float(b.grad.abs().max()))
        assert (a - b.grad).abs().max().item() < 1e-4 * scale
    # one training step through Model.forward + MarginLoss changes the tables
    import torchkge_amd as tk
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    nh = torch.randint(0, int(z['n_ent']), (B,), generator=g).cuda()
    pos, neg = m(h, t, r, nh, t)
    loss = tk.MarginLoss(0.5)(pos, neg)
    opt.zero_grad(); loss.backward(); opt.step()
    m.normalize_parameters()
    assert torch.isfinite(m._tables()[0]).all()


# ---------------------------------------------------------------------------
# BASELINE sizes: size-independent properties
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('kind,shape,d', [('transe', 'fb15k237', 200), ('complex', 'wn18rr', 200),
                                          ('distmult', 'fb15k', 400), ('transh', 'fb15k237', 100),
                                          ('transd', 'fb15k237', 100)])
def test_full_size_properties(hip, kind, shape, d):
    import torchkge_amd as tk
    n_ent, n_rel = orc.DATASET_SHAPES[shape][:2]
    tables = orc.init_tables(kind, n_ent, n_rel, d, seed=0)
    m = build_model(kind, 2, tables, n_ent, n_rel)
    B = 384
    h, t, r = orc.synthetic_triples(n_ent, n_rel, B, seed=1, device='cuda')
    t[0], h[1], t[2] = 0, n_ent - 1, n_ent - 1            # boundary candidates
    # build a small filter index: every query filters a few random entities + its true one
    g = torch.Generator().manual_seed(2)
    extra = torch.randint(0, n_ent, (B, 5), generator=g)
    dic = {}
    for i in range(B):
        dic.setdefault((int(h[i]), int(r[i])), set()).update([int(t[i])] + extra[i].tolist())
    from torchkge_amd.filter_index import FilterIndex
    idx = FilterIndex.from_dict(dic, 'cuda')
    seg_lo, seg_hi = idx.lookup(h, r)
    prob = m.lp_problem(h, t, r, 'tail')
    scores = prob.scores()
    s_true = prob.pair_scores(t)
    assert torch.equal(s_true, scores.gather(1, t.view(-1, 1)).view(-1))       # pair == tile, bitwise
    raw = prob.count_ge(s_true)
    rk, frk = hip.filtered_rank_from_scores(scores, t, seg_lo, seg_hi, idx.targets)
    assert torch.equal(raw.long(), rk)                                         # fused == materialised
    sub, found = prob.filter_sub(s_true, t, seg_lo, seg_hi, idx.targets)
    rk2, frk2 = hip.rank_finalize(raw, sub, found)
    assert torch.equal(rk2, rk) and torch.equal(frk2, frk)
    assert int(rk.min()) >= 1 and (frk <= rk).all() and (frk >= 1).all()
    # agreement with the drop-in composition get_rank(filter_scores())
    from torchkge_amd.utils import filter_scores, get_rank
    assert torch.equal(get_rank(filter_scores(scores, idx, h, r, t), t), frk)
    # sharding: 3 virtual shards concatenate to the full matrix, counts add up (bitwise)
    parts, cnt = [], torch.zeros(3, B, dtype=torch.int32, device='cuda')
    from torchkge_amd import distributed as kd
    for pidx in range(3):
        lo, hi = kd.shard_range(n_ent, 3, pidx)
        pp = m.lp_problem(h, t, r, 'tail', ent_lo=lo, ent_hi=hi)
        parts.append(pp.scores())
        st = pp.pair_scores(t)
        own = (t >= lo) & (t < hi)
        assert torch.equal(st[own], s_true[own]) and (st[~own] == 0).all()
        cnt[0] += pp.count_ge(s_true)
        s_, f_ = pp.filter_sub(s_true, t, seg_lo, seg_hi, idx.targets)
        cnt[1] += s_; cnt[2] += f_
    assert torch.equal(torch.cat(parts, 1), scores)
    assert torch.equal(cnt[0], raw) and torch.equal(cnt[1], sub) and torch.equal(cnt[2], found)
    if kind in ('distmult', 'complex'):
        # linearity of the bilinear scorer: exact under power-of-two scaling
        Q2 = hip.LpProblem(hip.LP_DOT, prob.keep[0] * 2.0, prob.keep[1],
                           A1=None if prob.keep[2] is None else prob.keep[2] * 2.0, T1=prob.keep[3])
        assert torch.equal(Q2.scores(), scores * 2.0)
    else:
        assert (scores <= 0).all()
        if kind == 'transe':
            # a query equal to a table row scores ~0 against it, and is the row's best candidate
            E = m.ent_emb.weight.data
            pz = m._translational_problem(E[:B].contiguous(), E)
            sz = pz.scores()
            assert (sz[torch.arange(B), torch.arange(B)].abs() < TOL).all()
            assert (sz.argmax(1).cpu() == torch.arange(B)).all()
    # against the oracle on a slice the CPU finishes in seconds
    hs, ts, rs = h[:8].cpu(), t[:8].cpu(), r[:8].cpu()
    so = orc.lp_scores(kind, tables, hs, ts, rs, 'tail', 2)
    assert (scores[:8].cpu() - so).abs().max().item() < TOL
    so = orc.lp_scores(kind, tables, hs, ts, rs, 'head', 2)
    assert (m.lp_problem(h[:8], t[:8], r[:8], 'head').scores().cpu() - so).abs().max().item() < TOL


def test_empty_and_tiny_batches(hip):
    import torchkge_amd as tk
    tables = orc.init_tables('transe', 14, 55, 50, seed=0)       # Nations shape, d = 50 (scalar-load path)
    m = build_model('transe', 2, tables, 14, 55)
    e = torch.zeros(0, dtype=torch.long, device='cuda')
    assert m.scoring_function(e, e, e).shape == (0,)
    prob = m.lp_problem(e, e, e, 'tail')
    assert prob.scores().shape == (0, 14)
    h, t, r = orc.synthetic_triples(14, 55, 5, seed=3, device='cuda')
    s = m.lp_problem(h, t, r, 'tail').scores().cpu()
    so = orc.lp_scores('transe', tables, h.cpu(), t.cpu(), r.cpu(), 'tail', 2)
    assert (s - so).abs().max().item() < TOL
    sf = m.scoring_function(h, t, r).cpu()
    assert (sf - orc.score_triples('transe', tables, h.cpu(), t.cpu(), r.cpu(), p=2)).abs().max().item() < TOL
    with pytest.raises(RuntimeError):
        m.scoring_function(h.cpu(), t.cpu(), r.cpu())            # no CPU fallback


def test_dissimilarities_known_answers(hip):
    from torchkge_amd.utils import l1_dissimilarity, l2_dissimilarity
    a = dev(np.array([[1.4, 2, 3, 4], [5.4, 6, 7, 8]], dtype=np.float32))
    b = dev(np.array([[1.3, 4, 2, 10], [5.9, 8, 6, 7]], dtype=np.float32))
    assert (l1_dissimilarity(a, b).cpu() - torch.tensor([9.1, 4.5])).abs().max() < 1e-5
    assert (l2_dissimilarity(a, b).cpu() - torch.tensor([41.01, 6.25])).abs().max() < 1e-4


# ---------------------------------------------------------------------------
# "next" rows of SURVEY section 8f: relation prediction, top-k inference
# ---------------------------------------------------------------------------
def _relpred_setup(kind):
    import torchkge_amd as tk
    z = np.load(GOLDEN + '/ref_relpred.npz')
    ntab = {'complex': 4, 'transd': 4, 'transh': 3}.get(kind, 2)
    tables = [torch.from_numpy(z['%s_table%d' % (kind, i)]) for i in range(ntab)]
    n_ent, n_rel = int(z['n_ent']), int(z['n_rel'])
    m = build_model(kind, 2, tables, n_ent, n_rel)
    heads, tails, rels = (torch.from_numpy(z[k]) for k in ('heads', 'tails', 'rels'))
    kg = tk.KnowledgeGraph(kg={'heads': heads, 'tails': tails, 'relations': rels},
                           ent2ix={i: i for i in range(n_ent)}, rel2ix={i: i for i in range(n_rel)})
    nt = int(z['n_test'])
    _, kg_test = kg.split_kg(sizes=(len(heads) - nt, nt))
    return z, tables, m, kg, kg_test


@pytest.mark.parametrize('kind', ['transe', 'distmult', 'complex', 'transh', 'transd'])
def test_relation_prediction_vs_reference(hip, kind):
    import torchkge_amd as tk
    z, tables, m, kg, kg_test = _relpred_setup(kind)
    B = int(z['b_size'])
    h, t, r = kg_test.head_idx[:B].cuda(), kg_test.tail_idx[:B].cuda(), kg_test.relations[:B].cuda()
    h_e, t_e, r_e, cand = m.inference_prepare_candidates(h, t, r, entities=False)
    s = m.inference_scoring_function(h_e, t_e, cand)
    assert np.abs(s.cpu().numpy() - z['%s_s_rel' % kind]).max() < TOL
    if kind in ('transh', 'transd'):
        # the handles stand for real (b, n_rel, d) tensors: the generic 3-D path gives the same scores
        ph, pt = h_e.materialize(), t_e.materialize()
        assert tuple(ph.shape) == tuple(h_e.shape) == (B, m.n_rel, cand.shape[2])
        s2 = m.inference_scoring_function(ph, pt, cand.contiguous())
        assert (s2 - s).abs().max().item() < TOL
        P = orc.transh_projected_entities(tables[0], tables[2]) if kind == 'transh' else \
            orc.transd_projected_entities(tables[0], tables[2], tables[3])
        assert (ph.cpu() - P[:, h.cpu()].transpose(0, 1)).abs().max().item() < TOL
    for directed, tag in ((True, 'dir'), (False, 'undir')):
        ev = tk.RelationPredictionEvaluator(m, kg_test, directed=directed)
        with pytest.raises(tk.NotYetEvaluatedError):
            ev.mrr()
        ev.evaluate(b_size=B, verbose=False)
        if directed:
            assert np.array_equal(ev.rank_true_rels.numpy(), z['%s_%s_rank' % (kind, tag)])
            assert np.array_equal(ev.filt_rank_true_rels.numpy(), z['%s_%s_frank' % (kind, tag)])
            assert abs(ev.mrr()[1] - z['%s_%s_mrr' % (kind, tag)][1]) < TOL
            assert abs(ev.hit_at_k(3)[1] - z['%s_%s_hit3' % (kind, tag)][1]) < TOL
        else:
            # (h,?,t) and (t,?,h) scores tie up to rounding (exactly for DistMult, whose
            # score is symmetric): ranks must lie in the reference's own tie interval
            heads, tails, rels = kg_test.head_idx, kg_test.tail_idx, kg_test.relations
            so = torch.cat((orc.rp_scores(kind, tables, heads, tails), orc.rp_scores(kind, tables, heads, tails, swap=True)), 1)
            lo, hi = _rank_bounds(so, rels, 2 * TOL)
            assert ((ev.rank_true_rels >= lo) & (ev.rank_true_rels <= hi)).all()
            fo = torch.cat((orc.filter_scores(so[:, :so.shape[1] // 2], kg.dict_of_rels, heads, tails, rels),
                            orc.filter_scores(so[:, so.shape[1] // 2:], kg.dict_of_rels, heads, tails, rels)), 1)
            lo, hi = _rank_bounds(fo, rels, 2 * TOL)
            assert ((ev.filt_rank_true_rels >= lo) & (ev.filt_rank_true_rels <= hi)).all()
    ev.print_results(k=3)


@pytest.mark.parametrize('B,N,k', [(7, 300, 5), (3, 14541, 10), (2, 5, 5), (4, 1000, 1)])
def test_topk_kernel_exact(hip, B, N, k):
    g = torch.Generator().manual_seed(B * N + k)
    s = torch.randn(B, N, generator=g)
    s[0, min(3, N - 1)] = s[0, 0]                       # a tie: lower index first
    if k < N:
        s[B - 1, N - 1] = float('nan')                  # NaN is never selected
    v, ix = hip.topk(s.cuda(), k)
    sn = torch.where(torch.isnan(s), torch.full_like(s, -float('inf')), s)
    order = np.lexsort((np.arange(N)[None, :].repeat(B, 0), -sn.numpy()), axis=1)[:, :k]
    assert np.array_equal(ix.cpu().numpy(), order)
    assert np.array_equal(v.cpu().numpy(), np.take_along_axis(sn.numpy(), order, 1))
    # exhausted rows (fewer selectable entries than k) yield (-inf, -1)
    s2 = torch.tensor([[1.0, float('nan'), 3.0]])
    v2, i2 = hip.topk(s2.cuda(), 3)
    assert i2.cpu().tolist() == [[2, 0, -1]] and v2.cpu()[0, 2].item() == -float('inf')


@pytest.mark.parametrize('kind,p', [('transe', 2), ('transh', 2), ('complex', 2)])
def test_entity_and_relation_inference(hip, kind, p):
    import torchkge_amd as tk
    z, tables = load_golden(kind, p)
    n_ent, n_rel = int(z['n_ent']), int(z['n_rel'])
    m = build_model(kind, p, tables, n_ent, n_rel)
    h, t, r = golden_batch(z)
    heads, tails, rels = (torch.from_numpy(z[k]) for k in ('heads', 'tails', 'rels'))
    dh, dt, dr = orc.build_filter_dicts(heads, tails, rels)
    K = 5

    def expect(scores, dictionary, k1, k2):
        s = scores.clone()
        if dictionary is not None:
            s = orc.filter_scores(s, dictionary, k1, k2, None)
        v, ix = s.sort(descending=True)
        return v[:, :K], ix[:, :K]

    for missing, dic, side in (('tails', dt, 'tail'), ('heads', dh, 'head'), ('tails', None, 'tail')):
        known = h if missing == 'tails' else t
        inf = tk.EntityInference(m, known, r, top_k=K, missing=missing, dictionary=dic)
        inf.evaluate(b_size=7, verbose=False)
        so = orc.lp_scores(kind, tables, h, t, r, side, p)
        ev, eix = expect(so, dic, known, r)
        assert inf.predictions.shape == (len(known), K) and inf.predictions.dtype == torch.int64
        assert (inf.scores - ev).abs().max().item() < TOL
        gap_ok = (ev[:, :-1] - ev[:, 1:]).min(dim=1).values > 4 * TOL     # unambiguous order only
        assert torch.equal(inf.predictions[gap_ok], eix[gap_ok])
    with pytest.raises(tk.exceptions.WrongArgumentsError):
        tk.EntityInference(m, h, r, missing='both')
    # filter_scores with true_idx=None masks every known target
    from torchkge_amd.utils import filter_scores
    s_ref = torch.from_numpy(z['s_tail'])
    f = filter_scores(s_ref.cuda(), dt, h.cuda(), r.cuda(), None)
    assert np.array_equal(f.cpu().numpy(), orc.filter_scores(s_ref, dt, h, r, None).numpy())


@pytest.mark.parametrize('kind,p', CASES)
def test_tiled_topk_inference_equals_materialised_topk(hip, kind, p):
    """EntityInference processes the candidates tile by tile (kge_topk_chunk: O(b * C) scratch): for every
    tile size -- also tiles smaller than k, ragged last tiles, one tile -- predictions and scores equal the top-k of the
    materialised (b, N) score matrix (kge_lp_scores + kge_filter_scores + kge_topk: score descending, id ascending),
    with and without the known-fact filter; exact ties (duplicated entity rows) come out id-ascending."""
    import torchkge_amd as tk
    n_ent, n_rel, d = 1203, 7, 32
    tables = orc.init_tables(kind, n_ent, n_rel, d, seed=4)
    for tb in tables:
        if tb.shape[0] == n_ent:
            tb[700:710] = tb[100:110]              # exact ties between entity ids 100.. and 700..
    m = build_model(kind, p, tables, n_ent, n_rel)
    h, t, r = orc.synthetic_triples_zipf(n_ent, n_rel, 6000, 21, hubs=((300, 'tail'), (200, 'head')))
    dh, dt, _ = orc.build_filter_dicts(h, t, r)
    q_e, q_r = h[-150:].clone(), r[-150:].clone()
    q_e[:10] = torch.arange(100, 110)
    K = 12
    for missing, dic in (('tails', dt), ('heads', dh), ('tails', None)):
        ref = tk.EntityInference(m, q_e, q_r, top_k=K, missing=missing, dictionary=dic)
        ref._evaluate_materialised(64, verbose=False)
        for tile in (None, 256, 512, 1203, 4096):
            inf = tk.EntityInference(m, q_e, q_r, top_k=K, missing=missing, dictionary=dic, tile=tile)
            inf.evaluate(b_size=64, verbose=False)
            assert torch.equal(inf.predictions, ref.predictions), (kind, missing, tile)
            assert torch.equal(inf.scores, ref.scores)
    # k larger than a tile and than the candidate set: padding (-inf, -1) never displaces a real candidate
    small = build_model(kind, p, orc.init_tables(kind, 300, n_rel, d, seed=5), 300, n_rel)
    qs = torch.arange(0, 40)
    a = tk.EntityInference(small, qs, q_r[:40], top_k=300, tile=256)
    a.evaluate(16, verbose=False)
    b = tk.EntityInference(small, qs, q_r[:40], top_k=300)
    b._evaluate_materialised(16, verbose=False)
    assert torch.equal(a.predictions, b.predictions) and torch.equal(a.scores, b.scores)
    assert (a.predictions.sort(dim=1).values == torch.arange(300)).all()


def test_wikidata5m_scale_properties(hip):
    """BASELINE config 5 shape (ComplEx d=512, 4,594,485 entities, 18.8 GB of
    tables on one 288 GB GPU): the fused count over the full entity range equals
    (i) the sum over 8 virtual shards, (ii) counting on materialised score rows,
    and the pair kernel equals the tile kernel -- exercising 64-bit addressing,
    the 8-bit counter flushes (thousands of tiles per workgroup) and c_base."""
    from torchkge_amd import distributed as kd
    N, d, B = 4594485, 512, 5133
    g = torch.Generator(device='cuda').manual_seed(5)
    Tre = torch.empty(N, d, device='cuda').uniform_(-0.03, 0.03, generator=g)
    Tim = torch.empty(N, d, device='cuda').uniform_(-0.03, 0.03, generator=g)
    A = torch.empty(B, d, device='cuda').uniform_(-0.03, 0.03, generator=g)
    Bq = torch.empty(B, d, device='cuda').uniform_(-0.03, 0.03, generator=g)
    true = torch.randint(0, N, (B,), device='cuda', generator=g)
    true[0], true[1] = 0, N - 1
    prob = hip.LpProblem(hip.LP_DOT, A, Tre, A1=Bq, T1=Tim)
    s_true = prob.pair_scores(true)
    raw = prob.count_ge(s_true)
    assert int(raw.min()) >= 1 and int(raw.max()) <= N
    # (i) virtual shards
    acc = torch.zeros_like(raw)
    st2 = torch.zeros_like(s_true)
    for p in range(8):
        lo, hi = kd.shard_range(N, 8, p)
        pp = hip.LpProblem(hip.LP_DOT, A, Tre[lo:hi], A1=Bq, T1=Tim[lo:hi], c_base=lo)
        st2 += pp.pair_scores(true)
        acc += pp.count_ge(s_true)
    assert torch.equal(st2, s_true) and torch.equal(acc, raw)
    # (ii) materialised rows for a slice of queries (tile kernel in score-writing mode)
    nq = 96
    sub = hip.LpProblem(hip.LP_DOT, A[:nq].contiguous(), Tre, A1=Bq[:nq].contiguous(), T1=Tim)
    S = sub.scores()
    assert torch.equal(S.gather(1, true[:nq].view(-1, 1)).view(-1), s_true[:nq])
    assert torch.equal((S >= s_true[:nq].view(-1, 1)).sum(1).int(), raw[:nq])
    assert torch.equal(hip.get_rank(S, true[:nq]).int(), raw[:nq])
    # against fp64 on a few pairs
    i = torch.arange(0, nq, 7, device='cuda')
    ref = (A[i].double() * Tre[true[i]].double()).sum(1) + (Bq[i].double() * Tim[true[i]].double()).sum(1)
    assert (s_true[i].double() - ref).abs().max().item() < 1e-6
    # (iii) the f16-split prefilter at this scale (19.4 GB split table, K = 1024 + guard column)
    del S, sub
    guard = torch.zeros(8, device='cuda')
    hip.row_sqnorm(Tre, max_io=guard[1:2]); hip.row_sqnorm(Tim, max_io=guard[5:6])
    prob.split = {'Es': hip.split_rows(Tre, X1=Tim, dot=True, nmax0=guard[1:2], nmax1=guard[5:6]),
                  'enmax': guard[1:2], 'enmax1': guard[5:6], 'overflow': guard[2:3]}
    got = prob.count_ge(s_true)
    assert float(guard[2]) == 0.0 and torch.equal(got, raw)


def test_random_shape_sweep_bit_exact(hip):
    """Ragged shapes (B, N, K not multiples of the 128x128x32 tile, K % 4 != 0 ->
    scalar-load path, B or N smaller than one tile): MFMA and VALU tile kernels,
    pair kernel and fused counts vs the C oracle, bit for bit."""
    lib = oracle_clib()
    rng = np.random.RandomState(7)
    shapes = [(1, 2, 1), (2, 129, 7), (127, 127, 9), (129, 257, 33), (300, 40, 64), (31, 1000, 201)]
    shapes += [(int(rng.randint(1, 400)), int(rng.randint(1, 700)), int(rng.randint(1, 260))) for _ in range(10)]
    for B, N, K in shapes:
        g = torch.Generator().manual_seed(B * 7919 + N * 31 + K)
        A = torch.rand(B, K, generator=g) * 2 - 1
        T = torch.rand(N, K, generator=g) * 2 - 1
        ci = torch.randint(0, N, (B,), generator=g)
        ref = np.empty((B, N), dtype=np.float32)
        for mode in ('dot', 'expand', 'l1', 'l2'):
            if mode in ('dot', 'expand'):
                qn = np.empty(B, dtype=np.float32); en = np.empty(N, dtype=np.float32)
                lib.orc_row_sqnorm_chain(fptr(A.numpy()), i64(K), i64(B), i64(K), fptr(qn))
                lib.orc_row_sqnorm_chain(fptr(T.numpy()), i64(K), i64(N), i64(K), fptr(en))
                lib.orc_lp_gemm_chain(fptr(A.numpy()), i64(K), fptr(T.numpy()), i64(K), i64(K), None, i64(0), None,
                                      i64(0), i64(0), i64(B), i64(N), 1 if mode == 'expand' else 0, fptr(qn), fptr(en),
                                      fptr(ref))
                dA, dT = dev(A), dev(T)
                prob = (hip.LpProblem(hip.LP_L2_EXPAND, dA, dT, qn=hip.row_sqnorm(dA), en=hip.row_sqnorm(dT))
                        if mode == 'expand' else hip.LpProblem(hip.LP_DOT, dA, dT))
            else:
                lib.orc_lp_direct_chain(fptr(A.numpy()), i64(K), fptr(T.numpy()), i64(K), i64(K), None, i64(0), None,
                                        i64(0), None, i64(B), i64(N), 1 if mode == 'l1' else 2, fptr(ref))
                prob = hip.LpProblem(hip.LP_L1_DIRECT if mode == 'l1' else hip.LP_L2_DIRECT, dev(A), dev(T))
            tag = (B, N, K, mode)
            assert np.array_equal(prob.scores().cpu().numpy(), ref), tag
            st = prob.pair_scores(dev(ci))
            assert np.array_equal(st.cpu().numpy(), ref[np.arange(B), ci.numpy()]), tag
            assert np.array_equal(prob.count_ge(st).cpu().numpy(),
                                  (ref >= ref[np.arange(B), ci.numpy()][:, None]).sum(1)), tag


def test_l2_auto_mode_guards_norm_expansion(hip):
    """TransE-L2: the MFMA norm expansion is only taken while its cancellation
    error fits the score tolerance; un-normalised tables with large norms fall
    back to the broadcast-subtract kernel and stay within fp32 relative error
    of the reference algorithm (oracle), incl. through evaluate()/hipGraph."""
    import torchkge_amd as tk
    from torchkge_amd import _hip
    g = torch.Generator().manual_seed(11)
    n_ent, n_rel, d, B = 700, 5, 200, 48
    E = torch.randn(n_ent, d, generator=g) * 0.9            # ||e||^2 ~ 160
    R = torch.randn(n_rel, d, generator=g) * 0.9
    h = torch.randint(0, n_ent, (B,), generator=g); t = torch.randint(0, n_ent, (B,), generator=g)
    r = torch.randint(0, n_rel, (B,), generator=g)
    ref = orc.lp_scores('transe', [E, R], h, t, r, 'tail', 2, None)
    m = build_model('transe', 2, [E, R], n_ent, n_rel)
    assert m.l2_mode == 'auto'
    pz = m.lp_problem(h.cuda(), t.cuda(), r.cuda(), 'tail')
    assert pz.desc.mode == _hip.LP_L2_DIRECT
    s = pz.scores().cpu()
    rel_err = ((s - ref).abs() / ref.abs().clamp_min(1.0)).max().item()
    assert rel_err < 2e-6
    m.l2_mode = 'expand'                                     # forced: same problem, visibly worse
    pz = m.lp_problem(h.cuda(), t.cuda(), r.cuda(), 'tail')
    assert pz.desc.mode == _hip.LP_L2_EXPAND
    m.l2_mode = 'auto'
    # small norms -> expansion, decision made once per evaluate() and not kept afterwards
    m2 = build_model('transe', 2, [torch.nn.functional.normalize(E, dim=1), torch.nn.functional.normalize(R, dim=1)],
                     n_ent, n_rel)
    assert m2.lp_problem(h.cuda(), t.cuda(), r.cuda(), 'head').desc.mode == _hip.LP_L2_EXPAND
    kg = tk.KnowledgeGraph(kg={'heads': h, 'tails': t, 'relations': r},
                           ent2ix={i: i for i in range(n_ent)}, rel2ix={i: i for i in range(n_rel)})
    dh, dt, _ = orc.build_filter_dicts(h, t, r)
    for model, tabs in ((m, [E, R]), (m2, [x.weight.data.cpu() for x in (m2.ent_emb, m2.rel_emb)])):
        for graph in (False, True):
            ev = tk.LinkPredictionEvaluator(model, kg, graph=graph)
            ev.evaluate(b_size=32, verbose=False)
            assert model._expand_ok is None
            scale = float(tabs[0].pow(2).sum(1).max()) * 4 + 1.0
            ties = orc.lp_evaluate('transe', tabs, h, t, r, dh, dt, 32, 2, tie_tol=4e-6 * scale)[4]
            got = [ev.rank_true_heads, ev.rank_true_tails, ev.filt_rank_true_heads, ev.filt_rank_true_tails]
            for k in range(4):
                assert bool(((got[k] >= ties[k, :, 0]) & (got[k] <= ties[k, :, 1])).all())


@pytest.mark.parametrize('B,N,K', [(64, 300, 32), (1000, 3000, 200), (193, 257, 17), (5, 2, 1), (700, 1500, 203)])
def test_split_prefilter_counts_equal_exact_counts(hip, B, N, K):
    """kge_lp_split_count + kge_lp_split_recheck (f16 hi/lo-split MFMA prefilter with a
    proven error band, exact re-scoring inside the band) leave exactly the counts of
    kge_lp_count_ge -- also with the band shrunk 16x (the bound is not tight by luck)."""
    g = torch.Generator().manual_seed(B + N)
    E = torch.nn.functional.normalize(torch.randn(N, K, generator=g), dim=1)
    R = torch.nn.functional.normalize(torch.randn(9, K, generator=g), dim=1)
    h = torch.randint(0, N, (B,), generator=g); r = torch.randint(0, 9, (B,), generator=g)
    t = torch.randint(0, N, (B,), generator=g)
    E[N // 2] = E[0]                                   # exact duplicates: ties with the true entity
    dE, dq, dt = E.cuda(), (E[h] + R[r]).cuda().contiguous(), t.cuda()
    guard = torch.zeros(4, device='cuda')
    en = hip.row_sqnorm(dE, max_io=guard[1:2]); qn = hip.row_sqnorm(dq, max_io=guard[0:1])
    prob = hip.LpProblem(hip.LP_L2_EXPAND, dq, dE, qn=qn, en=en)
    st = prob.pair_scores(dt)
    exact = prob.count_ge(st)
    Es, e2 = hip.split_table(dE, aug=en)
    prob.split = {'Es': Es, 'e2pref': e2, 'enmax': guard[1:2], 'overflow': guard[2:3]}   # prefix-norm band
    try:
        for eps in (1.0, 1.0 / 16):
            hip.SPLIT_EPS_SCALE = eps
            got = prob.count_ge(st)
            assert torch.equal(got, exact)
            n_unc = int(prob.last_split[0].item())
            assert B <= n_unc <= 64 * B              # at least the true entity of every query is re-scored
    finally:
        hip.SPLIT_EPS_SCALE = 1.0
    assert float(guard[2]) == 0.0


def test_split_prefilter_overflow_falls_back_to_exact(hip):
    """More near-ties than the uncertain-pair list holds (here: every entity has the
    same embedding) raise the overflow flag; the evaluator then redoes the evaluation
    with the exact fp32 counts."""
    import torchkge_amd as tk
    n_ent, n_rel, d, n = 3000, 3, 32, 40
    g = torch.Generator().manual_seed(5)
    E = torch.nn.functional.normalize(torch.randn(1, d, generator=g), dim=1).repeat(n_ent, 1)
    R = torch.nn.functional.normalize(torch.randn(n_rel, d, generator=g), dim=1)
    h = torch.randint(0, n_ent, (n,), generator=g); t = torch.randint(0, n_ent, (n,), generator=g)
    r = torch.randint(0, n_rel, (n,), generator=g)
    m = build_model('transe', 2, [E, R], n_ent, n_rel)
    kg = tk.KnowledgeGraph(kg={'heads': h, 'tails': t, 'relations': r},
                           ent2ix={i: i for i in range(n_ent)}, rel2ix={i: i for i in range(n_rel)})
    ev = tk.LinkPredictionEvaluator(m, kg)
    ev.evaluate(b_size=64, verbose=False)
    assert m._split_ok is True                         # reset after the evaluation
    m.split_filter = False
    ev2 = tk.LinkPredictionEvaluator(m, kg)
    ev2.evaluate(b_size=64, verbose=False)
    for nm in ('rank_true_heads', 'rank_true_tails', 'filt_rank_true_heads', 'filt_rank_true_tails'):
        assert torch.equal(getattr(ev, nm), getattr(ev2, nm))
    assert int(ev.rank_true_tails.min()) == n_ent      # everything ties: rank = number of entities


@pytest.mark.parametrize('B,N,K,K1,scale', [(300, 1000, 64, 0, 1.0), (257, 700, 40, 40, 30.0), (100, 513, 17, 17, 1e-3),
                                          (64, 300, 200, 200, 1.0)])
def test_split_prefilter_dot_mode_counts_equal_exact_counts(hip, B, N, K, K1, scale):
    """KGE_LP_DOT (DistMult / ComplEx, two K-segments, arbitrary magnitudes -- the
    operands carry their own power-of-two scales): split counts == exact counts,
    also with negative true scores (padding candidates must never count)."""
    g = torch.Generator().manual_seed(B * 7 + K)
    T0 = (torch.randn(N, K, generator=g) * scale).cuda()
    T1 = (torch.randn(N, K1, generator=g) * scale).cuda() if K1 else None
    A0 = (torch.randn(B, K, generator=g) * scale).cuda()
    A1 = (torch.randn(B, K1, generator=g) * scale).cuda() if K1 else None
    A0[0] = 0.0                                         # an all-zero query: every score ties at 0
    if A1 is not None:
        A1[0] = 0.0
    t = torch.randint(0, N, (B,), generator=g).cuda()
    prob = hip.LpProblem(hip.LP_DOT, A0, T0, A1=A1, T1=T1)
    st = prob.pair_scores(t)
    assert bool((st < 0).any())
    exact = prob.count_ge(st)
    guard = torch.zeros(8, device='cuda')
    hip.row_sqnorm(T0, max_io=guard[1:2])
    if T1 is not None:
        hip.row_sqnorm(T1, max_io=guard[5:6])
    Es, e2 = hip.split_table(T0, X1=T1, dot=True, nmax0=guard[1:2], nmax1=guard[5:6] if T1 is not None else None)
    prob.split = {'Es': Es, 'e2pref': e2, 'enmax': guard[1:2], 'enmax1': guard[5:6] if T1 is not None else None,
                  'overflow': guard[2:3]}
    try:
        for eps in (1.0, 1.0 / 16):
            hip.SPLIT_EPS_SCALE = eps
            got = prob.count_ge(st)
            assert torch.equal(got, exact), (eps, int((got != exact).sum()))
    finally:
        hip.SPLIT_EPS_SCALE = 1.0
    if N > 64 * 2:
        assert float(guard[2]) == 0.0 or int(exact[0]) == N   # only the all-tie query may overflow a tile buffer


def test_split_prefilter_random_shape_sweep(hip):
    """Random shapes / magnitudes / alignments for both MFMA modes: split counts == exact counts
    (K not a multiple of 4 takes the scalar staging paths; c_base != 0 is a candidate shard)."""
    g = torch.Generator().manual_seed(2024)
    for trial in range(14):
        B = int(torch.randint(1, 700, (1,), generator=g)); N = int(torch.randint(1, 2500, (1,), generator=g))
        K = int(torch.randint(1, 260, (1,), generator=g))
        dot = trial % 2 == 1
        K1 = int(torch.randint(1, 130, (1,), generator=g)) if (dot and trial % 4 == 3) else 0
        mag = float(10.0 ** torch.empty(1).uniform_(-2, 1.5, generator=g)) if dot else 1.0
        T0 = torch.randn(N, K, generator=g) * mag
        A0 = torch.randn(B, K, generator=g) * mag
        if not dot:                                   # TransE-like: unit entities, q = e + r
            T0 = torch.nn.functional.normalize(T0, dim=1)
            A0 = T0[torch.randint(0, N, (B,), generator=g)] + torch.nn.functional.normalize(A0, dim=1)
        T1 = (torch.randn(N, K1, generator=g) * mag).cuda() if K1 else None
        A1 = (torch.randn(B, K1, generator=g) * mag).cuda() if K1 else None
        T0, A0 = T0.cuda(), A0.cuda().contiguous()
        t = torch.randint(0, N, (B,), generator=g).cuda()
        guard = torch.zeros(8, device='cuda')
        if dot:
            prob = hip.LpProblem(hip.LP_DOT, A0, T0, A1=A1, T1=T1)
            hip.row_sqnorm(T0, max_io=guard[1:2])
            if T1 is not None:
                hip.row_sqnorm(T1, max_io=guard[5:6])
            Es, e2 = hip.split_table(T0, X1=T1, dot=True, nmax0=guard[1:2], nmax1=guard[5:6] if T1 is not None else None)
        else:
            en = hip.row_sqnorm(T0, max_io=guard[1:2]); qn = hip.row_sqnorm(A0, max_io=guard[0:1])
            prob = hip.LpProblem(hip.LP_L2_EXPAND, A0, T0, qn=qn, en=en)
            Es, e2 = hip.split_table(T0, aug=en)
        st = prob.pair_scores(t)
        exact = prob.count_ge(st)
        prob.split = {'Es': Es, 'e2pref': e2 if trial % 3 else None, 'enmax': guard[1:2],   # (every third trial: the plain band)
                      'enmax1': guard[5:6] if T1 is not None else None,
                      'overflow': guard[2:3]}
        got = prob.count_ge(st)
        assert float(guard[2]) == 0.0, (trial, B, N, K, K1)
        assert torch.equal(got, exact), (trial, B, N, K, K1, dot, int((got != exact).sum()))


def test_split_prefilter_nonfinite_embeddings_fall_back(hip):
    """A diverged model (inf in an embedding): the split prefilter raises its flag and the
    evaluator returns the ranks of the exact fp32 path."""
    import torchkge_amd as tk
    g = torch.Generator().manual_seed(8)
    n_ent, n_rel, d, n = 400, 4, 32, 50
    m = tk.DistMultModel(d, n_ent, n_rel).cuda()
    m.ent_emb.weight.data[7, 3] = float('inf')
    h = torch.randint(0, n_ent, (n,), generator=g); t = torch.randint(0, n_ent, (n,), generator=g)
    r = torch.randint(0, n_rel, (n,), generator=g)
    kg = tk.KnowledgeGraph(kg={'heads': h, 'tails': t, 'relations': r},
                           ent2ix={i: i for i in range(n_ent)}, rel2ix={i: i for i in range(n_rel)})
    ev = tk.LinkPredictionEvaluator(m, kg)
    ev.evaluate(b_size=64, verbose=False)
    m.split_filter = False
    ev2 = tk.LinkPredictionEvaluator(m, kg)
    ev2.evaluate(b_size=64, verbose=False)
    for nm in ('rank_true_heads', 'rank_true_tails', 'filt_rank_true_heads', 'filt_rank_true_tails'):
        assert torch.equal(getattr(ev, nm), getattr(ev2, nm))


@pytest.mark.parametrize('mode_name,B,N,K,R', [('H', 130, 300, 64, 7), ('D', 257, 129, 40, 5), ('H', 33, 1000, 203, 11),
                                               ('D', 64, 515, 200, 3)])
def test_projection_modes_bit_exact_vs_chain(hip, mode_name, B, N, K, R):
    """KGE_LP_L2_PROJH / _PROJD (TransH / TransD as an fp32 MFMA GEMM + per-pair gather):
    tile kernel == pair kernel == oracle/kge_oracle.c bit for bit; rank counts consistent."""
    import ctypes as C
    lib = oracle_clib()
    mode = hip.LP_L2_PROJH if mode_name == 'H' else hip.LP_L2_PROJD
    g = torch.Generator().manual_seed(B + N + K)
    A = torch.randn(B, K, generator=g) * 0.3
    T = torch.randn(N, K, generator=g) * 0.3
    X = torch.randn(R, N, generator=g) * 0.2
    yc = torch.randn(N, generator=g) * 0.2
    pz = torch.randn(B, 2, generator=g) * 0.5
    r_idx = torch.randint(0, R, (B,), generator=g)
    dA, dT = A.cuda(), T.cuda()
    qn, en = hip.row_sqnorm(dA), hip.row_sqnorm(dT)
    prob = hip.LpProblem(mode, dA, dT, qn=qn, en=en, Wq=pz.cuda(), scal=X.cuda(), r_idx=r_idx.cuda(),
                         yc=yc.cuda() if mode_name == 'D' else None)
    S = prob.scores().cpu()
    ref = np.empty((B, N), dtype=np.float32)
    i64 = C.c_int64
    lib.orc_lp_proj_chain(C.c_int(4 if mode_name == 'H' else 5), fptr(A.numpy()), i64(K), fptr(T.numpy()), i64(K),
                          i64(K), i64(B), i64(N), fptr(qn.cpu().numpy()), fptr(en.cpu().numpy()), fptr(X.numpy()),
                          i64(N), r_idx.numpy().ctypes.data_as(C.c_void_p), fptr(yc.numpy()), fptr(pz.numpy()),
                          fptr(ref))
    assert np.array_equal(S.numpy(), ref)
    t = torch.randint(0, N, (B,), generator=g).cuda()
    st = prob.pair_scores(t)
    assert torch.equal(st.cpu(), S[torch.arange(B), t.cpu()])
    cnt = prob.count_ge(st).cpu()
    assert torch.equal(cnt, (S >= st.cpu().view(-1, 1)).sum(1).to(torch.int32))


@pytest.mark.parametrize('kind', ['transh', 'transd'])
def test_projection_modes_shards_concatenate_bit_identically(hip, kind):
    """TransH / TransD through the MFMA projection modes: score tiles of 3 entity shards
    (lp_problem(ent_lo, ent_hi)) concatenate bit-identically to the unsharded matrix, partial
    rank counts add up, and the matrix agrees with the broadcast-subtract kernel within 1e-5."""
    z, tables = load_golden(kind, 2)
    n_ent, n_rel = int(z['n_ent']), int(z['n_rel'])
    m = build_model(kind, 2, tables, n_ent, n_rel)
    h, t, r = (x.cuda() for x in golden_batch(z))
    m.l2_mode = 'expand'
    with m.lp_session():
        full = m.lp_problem(h, t, r, 'tail')
        assert full.desc.mode in (hip.LP_L2_PROJH, hip.LP_L2_PROJD)
        S = full.scores()
        st = full.pair_scores(t)
        cnt = full.count_ge(st)
        parts, cnts = [], torch.zeros_like(cnt)
        bounds = [0, n_ent // 3, 2 * n_ent // 3 + 1, n_ent]
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            pr = m.lp_problem(h, t, r, 'tail', ent_lo=lo, ent_hi=hi)
            parts.append(pr.scores())
            cnts += pr.count_ge(st)
        assert torch.equal(torch.cat(parts, dim=1), S)
        assert torch.equal(cnts, cnt)
    m.l2_mode = 'direct'
    with m.lp_session():
        Sd = m.lp_problem(h, t, r, 'tail').scores()
    assert (S - Sd).abs().max().item() < TOL
    assert np.abs(S.cpu().numpy() - z['s_tail']).max() < TOL


@pytest.mark.parametrize('mode_name,B,N,K,R', [('H', 300, 1000, 64, 7), ('D', 257, 700, 40, 5), ('H', 1000, 3000, 200, 37),
                                               ('D', 500, 2049, 200, 11),
                                               # queries in relation order (how the evaluator feeds these modes): a 192-query
                                               # panel then holds <= 8 relation runs and each tile stages their X segments in LDS
                                               ('H', 1000, 3000, 200, -37), ('D', 900, 2049, 200, -11), ('H', 400, 700, 64, -2),
                                               ('D', 1500, 1300, 72, -160),
                                               # (r05: sizes at which the SLP-packed projection epilogue of the free-running
                                               # kernel went wrong for ~1 pair in 3e6 -- profiles/r05/pm_epilogue_slp_bisect.txt)
                                               ('H', 20000, 3000, 200, -37), ('D', 12000, 14541, 200, 237)])
def test_split_prefilter_projection_modes_counts_equal_exact_counts(hip, mode_name, B, N, K, R):
    """TransH / TransD projection modes through the f16-split prefilter (the per-pair term
    x(xz+p) resp. y(yz+2g+p) is added to the approximate accumulator in the epilogue):
    counts == exact fp32 counts, also with the band shrunk 16x."""
    mode = hip.LP_L2_PROJH if mode_name == 'H' else hip.LP_L2_PROJD
    sorted_r, R = R < 0, abs(R)
    g = torch.Generator().manual_seed(B + N + K)
    T = torch.nn.functional.normalize(torch.randn(N, K, generator=g), dim=1)
    W = torch.nn.functional.normalize(torch.randn(R, K, generator=g), dim=1)
    r_idx = torch.randint(0, R, (B,), generator=g)
    if sorted_r:        # two sorted halves, like a both-sides batch (tail-side queries, then head-side ones)
        r_idx = torch.cat([r_idx[:B // 2].sort().values, r_idx[B // 2:].sort().values])
    A = T[torch.randint(0, N, (B,), generator=g)] + 0.7 * torch.nn.functional.normalize(torch.randn(B, K, generator=g), dim=1)
    dT, dA, dW = T.cuda(), A.cuda().contiguous(), W.cuda()
    Np = hip.padded_cols(N)
    Xb = torch.zeros(R, Np, device='cuda')
    X = hip.LpProblem(hip.LP_DOT, dW, dT).scores(Xb[:, :N])
    ycb = torch.zeros(Np, device='cuda')
    ycb[:N] = torch.randn(N, generator=g).cuda() * 0.3
    Wq = dW[r_idx.cuda()]
    if mode_name == 'H':
        pz = torch.stack([hip.row_dot(dA, Wq, scale=2.0), hip.row_sqnorm(Wq) - 2.0], 1).contiguous()
    else:
        pz = torch.stack([hip.row_dot(dA, Wq, scale=-2.0), hip.row_sqnorm(Wq)], 1).contiguous()
    guard = torch.zeros(8, device='cuda')
    en = hip.row_sqnorm(dT, max_io=guard[1:2]); qn = hip.row_sqnorm(dA, max_io=guard[0:1])
    prob = hip.LpProblem(mode, dA, dT, qn=qn, en=en, Wq=pz, scal=X, r_idx=r_idx.cuda(),
                         yc=ycb[:N] if mode_name == 'D' else None)
    t = torch.randint(0, N, (B,), generator=g).cuda()
    st = prob.pair_scores(t)
    exact = prob.count_ge(st)
    hip.absmax(X, guard[3:4]); hip.absmax(ycb, guard[4:5])
    Es, e2 = hip.split_table(dT, aug=en)
    prob.split = {'Es': Es, 'e2pref': e2, 'enmax': guard[1:2], 'overflow': guard[2:3],
                  'xabsmax': guard[3:4], 'yabsmax': guard[4:5] if mode_name == 'D' else None}
    try:
        for eps in (1.0, 1.0 / 16):
            hip.SPLIT_EPS_SCALE = eps
            got = prob.count_ge(st)
            assert torch.equal(got, exact), (eps, int((got != exact).sum()))
    finally:
        hip.SPLIT_EPS_SCALE = 1.0
    assert float(guard[2]) == 0.0
    assert B <= int(prob.last_split[0].item()) <= 64 * B
    # ... and on the ONE-PRODUCT level (planar hi table, thresholds from the measured residuals; same epilogue)
    # ... and on the FREE-RUNNING one-product kernel (fragment-major candidate table, lp_hi_stream.hip; r05)
    for frag in (False, True):
        Eh, de2 = hip.hi_table(dT, aug=en, frag=frag)
        prob.split = {'Es': Eh, 'e2pref': None, 'enmax': guard[1:2], 'overflow': guard[2:3], 'xabsmax': guard[3:4],
                      'yabsmax': guard[4:5] if mode_name == 'D' else None, 'level': 1, 'de2max': de2, 'es_frag': frag}
        got = prob.count_ge(st)
        assert torch.equal(got, exact), (frag, int((got != exact).sum()))
        assert float(guard[2]) == 0.0 and B <= int(prob.last_split[0].item())


@pytest.mark.parametrize('n0,n1,rows', [(1000, 0, 237), (32768, 32768, 14541), (5, 3, 2), (70000, 1, 1 << 20), (1, 0, 1)])
def test_key_sort_is_the_stable_ascending_order(hip, n0, n1, rows):
    """kge_key_sort (radix sort of (id, position) over the id bits): perm == stable argsort of [k0 | k1] -- bit-exact
    integer work, also with heavy-tailed ids (a hub id owning a tenth of the batch)."""
    import ctypes
    g = torch.Generator().manual_seed(n0 + 3 * n1 + rows)
    k0 = torch.randint(0, rows, (n0,), generator=g)
    k0[: n0 // 10] = rows // 2
    k1 = torch.randint(0, rows, (n1,), generator=g) if n1 else None
    keys = k0 if k1 is None else torch.cat([k0, k1])
    want = torch.sort(keys, stable=True).indices
    lib = hip.load_library()
    bits = max(1, int(rows - 1).bit_length())
    nb = int(lib.kge_key_sort_ws_bytes(n0 + n1, bits))
    assert nb > 0
    ws = torch.empty(nb, dtype=torch.uint8, device='cuda')
    perm = torch.empty(n0 + n1, dtype=torch.int64, device='cuda')
    dk0, dk1 = k0.cuda(), (k1.cuda() if k1 is not None else None)
    rc = lib.kge_key_sort(hip._p(dk0), n0, hip._p(dk1), n1, bits, hip._p(perm), hip._p(ws), nb, hip._stream())
    assert rc == 0
    assert torch.equal(perm.cpu(), want)
    # argument errors: short workspace, too many bits
    assert lib.kge_key_sort(hip._p(dk0), n0, hip._p(dk1), n1, bits, hip._p(perm), hip._p(ws), nb - 1, hip._stream()) != 0
    assert lib.kge_key_sort(hip._p(dk0), n0, hip._p(dk1), n1, 33, hip._p(perm), hip._p(ws), nb, hip._stream()) != 0


@pytest.mark.parametrize('kind,p', [('transe', 2), ('transe_l1', 1), ('transh', 2), ('transd', 2), ('distmult', 2), ('complex', 2)])
def test_backward_sorted_reduction_matches_atomic_scatter(hip, kind, p):
    """Large batches reduce per-triple gradient rows in sorted order (kge_segment_sum_rows)
    instead of one atomic per element: same gradients as the atomic scatter (up to fp32
    summation order) for every model family, incl. zero upstream gradients."""
    kind = 'transe' if kind == 'transe_l1' else kind
    z, tables = load_golden(kind, p)
    n_ent, n_rel = int(z['n_ent']), int(z['n_rel'])
    m = build_model(kind, p, tables, n_ent, n_rel)
    g = torch.Generator().manual_seed(77)
    B = 6000
    h = torch.randint(0, n_ent, (B,), generator=g).cuda(); t = torch.randint(0, n_ent, (B,), generator=g).cuda()
    r = torch.randint(0, n_rel, (B,), generator=g).cuda()
    go = torch.randn(B, generator=g).cuda()
    go[::7] = 0.0
    grads = []
    old = hip.BWD_SORTED_MIN_BATCH
    try:
        for thresh in (1, 10 ** 9):
            hip.BWD_SORTED_MIN_BATCH = thresh
            m.zero_grad()
            m.scoring_function(h, t, r).backward(go)
            grads.append([prm.grad.clone() for prm in m.parameters()])
    finally:
        hip.BWD_SORTED_MIN_BATCH = old
    for a, b in zip(*grads):
        assert torch.isfinite(a).all()
        assert (a - b).abs().max().item() <= 1e-4 * max(1.0, b.abs().max().item())


def test_mfma_f16_accumulation_selftest(hip):
    """The tighter error band of the split prefilter (accum_model = 1) is only used when the
    device's f16 MFMA accumulates as measured (tools/probe/mfma_probe.hip): on MI355X it does."""
    assert hip.load_library().kge_mfma_f16_selftest() == 1
    assert hip.split_accum_model() == 1


@pytest.mark.parametrize('B,N,d', [(1, 7, 8), (31, 300, 52), (33, 300, 200), (257, 1000, 256), (1000, 2000, 100)])
@pytest.mark.parametrize('side', ['tail', 'head'])
def test_query_pipeline_equals_separate_kernels(hip, B, N, d, side):
    """kge_lp_query_pipeline (TransE-L2 query side of a batch in one launch) writes bit for bit
    what kge_lp_prep + kge_row_sqnorm + kge_lp_pair_scores + kge_lp_split_rows + the threshold
    pass of kge_lp_split_count write, and the counts that follow are the exact counts."""
    g = torch.Generator().manual_seed(B * 7 + d)
    E = torch.nn.functional.normalize(torch.randn(N, d, generator=g), dim=1).cuda()
    R = (0.3 * torch.randn(5, d, generator=g)).cuda()
    h = torch.randint(0, N, (B,), generator=g).cuda(); t = torch.randint(0, N, (B,), generator=g).cuda()
    r = torch.randint(0, 5, (B,), generator=g).cuda()
    sd = hip.SIDE_TAIL if side == 'tail' else hip.SIDE_HEAD
    true = t if side == 'tail' else h
    guard = torch.zeros(8, device='cuda')
    en = hip.row_sqnorm(E, max_io=guard[1:2])
    Es, e2 = hip.split_table(E, aug=en)

    # the separate kernels
    Q0 = hip.lp_prep(hip.TRANSE_L2, sd, [E, R], d, d, h, t, r)[0]
    g_ref = guard.clone()
    qn = hip.row_sqnorm(Q0, max_io=g_ref[0:1])
    ref = hip.LpProblem(hip.LP_L2_EXPAND, Q0, E, qn=qn, en=en)
    st = ref.pair_scores(true)
    exact = ref.count_ge(st)
    ref.split = {'Es': Es, 'e2pref': e2, 'enmax': g_ref[1:2], 'overflow': g_ref[2:3]}
    prep_ref = ref.split_prepare()
    raw_ref = torch.zeros(B, dtype=torch.int32, device='cuda')
    ref.split_count(prep_ref, st, raw_ref)

    # the fused launch
    pre = hip.lp_query_pipeline(sd, E, R, h, t, r, en, guard[1:2], guard[0:1], e2pref=e2)
    assert torch.equal(pre['Q'], Q0)
    assert torch.equal(pre['qn'].view(torch.int32), qn.view(torch.int32))
    assert torch.equal(pre['s_true'].view(torch.int32), st.view(torch.int32))
    assert torch.equal(pre['Qs'], prep_ref['Qs'])
    Bp = pre['thr'].numel() // 4
    # thresholds: same formula; the magnitude sum of the band comes from the chain's running value here and from
    # cell sums there (equal up to rounding; both inside the band's safety factor)
    thr_a, thr_b = pre['thr'][:2 * Bp].view(-1, 2), prep_ref['thr'][:2 * Bp].view(-1, 2)
    assert torch.equal(torch.isinf(thr_a), torch.isinf(thr_b))
    fin = ~torch.isinf(thr_a)
    assert torch.allclose(thr_a[fin], thr_b[fin], rtol=1e-5, atol=0.0)
    mid_a, mid_b = thr_a[:B].sum(1), thr_b[:B].sum(1)           # band centres agree to rounding
    assert torch.allclose(mid_a, mid_b, rtol=1e-6, atol=1e-3)
    assert float(guard[0]) == float(g_ref[0])                # max ||q||^2 folded into the guard

    pre['true_idx'] = true
    prob = hip.LpProblem(hip.LP_L2_EXPAND, pre['Q'], E, qn=pre['qn'], en=en)
    prob.split = {'Es': Es, 'e2pref': e2, 'enmax': guard[1:2], 'overflow': guard[2:3]}
    prob.pre = pre
    st2 = prob.pair_scores(true)
    assert st2 is pre['s_true']
    assert torch.equal(prob.count_ge(st2), exact)
    assert float(guard[2]) == 0.0


@pytest.mark.parametrize('eps', [1.0, 1.0 / 16])
def test_split_count_on_query_columns_equals_per_query_counts(hip, eps):
    """Queries that share their key share the query row (filter_index.ColumnPlan): the count kernel sweeps one COLUMN per
    distinct row -- single-query columns, then grouped columns with up to kge_lp_split_group_sets() threshold sets --
    and must leave in raw_count exactly what the per-query sweep and the all-fp32 kernel leave there; also with the
    error band shrunk (more listed pairs per grouped column), hub keys with hundreds of queries, keys on both sides."""
    import torchkge_amd as tk
    from torchkge_amd.filter_index import ColumnPlan
    n_ent, n_rel, d = 2500, 6, 64
    tables = orc.init_tables('transe', n_ent, n_rel, d, seed=8)
    m = build_model('transe', 2, tables, n_ent, n_rel)
    g = torch.Generator().manual_seed(12)
    B = 1900
    h = torch.randint(0, n_ent, (B,), generator=g)
    t = torch.randint(0, n_ent, (B,), generator=g)
    r = torch.randint(0, n_rel, (B,), generator=g)
    h[:700] = torch.randint(0, 9, (700,), generator=g)          # tail-side keys (h, r): 54 keys share 700 queries
    t[300:1200] = 17                                             # head-side hub: (17, r) for 900 queries
    r[300:700] = 2
    H, T, R = h.cuda(), t.cuda(), r.cuda()
    cols = ColumnPlan(H, T, R, n_ent, n_rel, hip.split_group_sets(), hip.split_query_rows_padded)
    # every query sits in exactly one column slot; only the first query of a column writes the row
    placed = torch.cat([cols.col_q[cols.col_q >= 0], cols.members[cols.members >= 0]]).long()
    assert torch.equal(placed.sort().values, torch.arange(2 * B, device='cuda'))
    assert cols.n_multi > 0 and cols.n_single > 0 and cols.n_columns < 2 * B
    assert int((cols.qs_row >= 0).sum()) == cols.n_columns
    assert cols.n_single_p % 192 == 0 and cols.n_multi_p % 192 == 0
    old = hip.SPLIT_EPS_SCALE
    hip.SPLIT_EPS_SCALE = eps
    try:
        guard = m.lp_guard_begin(torch.device('cuda', 0))
        true = torch.cat([T, H])
        with m.lp_session():
            raws = []
            for c in (cols, None):
                prob = m.lp_problem(H, T, R, 'both', cols=c)
                assert prob.pre is not None and prob.split is not None
                prob.pre['true_idx'] = true
                s_true = prob.pair_scores(true)
                raws.append(prob.count_ge(s_true))
                n_listed = int(prob.last_split[0].item())
            m.split_filter = False
            pe = m.lp_problem(H, T, R, 'both')
            raws.append(pe.count_ge(pe.pair_scores(true)))
            m.split_filter = True
        assert float(guard[2]) == 0.0           # no overflow
        m.lp_guard_end()
    finally:
        hip.SPLIT_EPS_SCALE = old
    assert torch.equal(raws[0], raws[1]) and torch.equal(raws[0], raws[2])
    assert int(raws[0].min()) >= 1


@pytest.mark.parametrize('coalesce_mode', ['literal', 'default'])
@pytest.mark.parametrize('kind', ['transe', 'transe_l1', 'transe_direct', 'distmult', 'complex', 'transh', 'transd'])
def test_evaluator_dedupes_query_rows_and_keeps_the_ranks(hip, monkeypatch, kind, coalesce_mode):
    """evaluate() on a graph with hub keys (TransE-L2: fused query pipeline writing one split row per column; DistMult /
    ComplEx: rows gathered per column): identical rank vectors with the ColumnPlan path on (default) and off
    (KGE_DEDUPE_QUERIES=0), eager and as hipGraph replays."""
    import torchkge_amd as tk
    import torchkge_amd.evaluation as evm
    n_ent, n_rel, d = 3001, 9, 64
    p_norm = 1 if kind == 'transe_l1' else 2           # (TransE-L1: the SAD prefilter sweeps the columns)
    direct = kind == 'transe_direct'                   # (l2_mode='direct': the packed-FMA broadcast-subtract count does)
    kind = 'transe' if kind in ('transe_l1', 'transe_direct') else kind
    tables = orc.init_tables(kind, n_ent, n_rel, d, seed=6)
    m = build_model(kind, p_norm, tables, n_ent, n_rel)
    if direct:
        m.l2_mode = 'direct'
    h, t, r = orc.synthetic_triples_zipf(n_ent, n_rel, 30000, 3, hubs=((900, 'head'), (400, 'tail')))
    kg = tk.KnowledgeGraph(kg={'heads': h, 'tails': t, 'relations': r}, ent2ix={i: i for i in range(n_ent)},
                           rel2ix={i: i for i in range(n_rel)})
    _, kg_test = kg.split_kg(sizes=(27000, 3000))
    names = ['rank_true_heads', 'rank_true_tails', 'filt_rank_true_heads', 'filt_rank_true_tails']
    monkeypatch.setattr(evm, 'DEDUPE_QUERIES', False)
    ref = tk.LinkPredictionEvaluator(m, kg_test, graph=False)
    ref.evaluate(b_size=1024, verbose=False)
    assert all(getattr(pl, 'cols', None) is None for pl in ref._plans.values())
    monkeypatch.setattr(evm, 'DEDUPE_QUERIES', True)
    for graph in (False, True):
        ev = tk.LinkPredictionEvaluator(m, kg_test, graph=graph)
        for _ in range(3):
            ev.evaluate(b_size=1024, verbose=False)
            for nm in names:
                assert torch.equal(getattr(ev, nm), getattr(ref, nm)), (graph, nm)
        assert any(pl.cols is not None and pl.cols.n_multi > 0 for pl in ev._plans.values())


def test_evaluator_level1_query_columns_on_the_free_running_kernel(hip):
    """r06: the free-running one-product kernel sweeping query COLUMNS (lp_hi_stream_kernel<.., GS = 4>: one matrix sweep per
    distinct query row, the grouped columns' members compared pass by pass from LDS thresholds) -- an opt-in
    (Model.lp_stream_columns / lp_dedupe_level1: measured slower than the per-query sweep at cfg2) whose ranks must equal the
    per-query sweep's, eager and as hipGraph replays, on a graph with hub keys."""
    import torchkge_amd as tk
    n_ent, n_rel, d = 3001, 9, 64
    tables = orc.init_tables('transe', n_ent, n_rel, d, seed=6)
    m = build_model('transe', 2, tables, n_ent, n_rel)
    h, t, r = orc.synthetic_triples_zipf(n_ent, n_rel, 30000, 3, hubs=((900, 'head'), (400, 'tail')))
    kg = tk.KnowledgeGraph(kg={'heads': h, 'tails': t, 'relations': r}, ent2ix={i: i for i in range(n_ent)},
                           rel2ix={i: i for i in range(n_rel)})
    _, kg_test = kg.split_kg(sizes=(27000, 3000))
    names = ['rank_true_heads', 'rank_true_tails', 'filt_rank_true_heads', 'filt_rank_true_tails']
    m.split_level = 1
    m.lp_stream_columns, m.lp_dedupe_level1 = False, False
    ref = tk.LinkPredictionEvaluator(m, kg_test, graph=False, share_state=False)
    ref.evaluate(b_size=1024, verbose=False)
    m.lp_stream_columns, m.lp_dedupe_level1 = True, True
    assert m._level1_stream() and m._level1_stream_columns()
    for graph in (False, True):
        ev = tk.LinkPredictionEvaluator(m, kg_test, graph=graph, share_state=False)
        for _ in range(3):
            ev.evaluate(b_size=1024, verbose=False)
            for nm in names:
                assert torch.equal(getattr(ev, nm), getattr(ref, nm)), (graph, nm)
        assert any(pl.cols is not None and pl.cols.n_multi > 0 for pl in ev._plans.values())


def test_steady_state_fast_replay_follows_every_change(hip, monkeypatch):
    """r06: steady-state evaluate() calls replay the captured hipGraph without the full prologue (LinkPredictionEvaluator.
    _fast_sig / _evaluate_fast).  The shortcut must be invisible: same ranks as the full path, and every change that keys
    the capture -- table VALUES written in place (same addresses: the replay re-reads them), tables REPLACED (new
    addresses), facts edited in place (version bump), another b_size, a switched-off prefilter -- shows in the ranks exactly
    as it does for a fresh evaluator."""
    import torchkge_amd as tk
    import torchkge_amd.evaluation as evm
    n_ent, n_rel, d = 2000, 7, 64
    tables = orc.init_tables('transe', n_ent, n_rel, d, seed=9)
    m = build_model('transe', 2, tables, n_ent, n_rel)
    h, t, r = orc.synthetic_triples_zipf(n_ent, n_rel, 12000, 5, hubs=((300, 'head'),))
    kg = tk.KnowledgeGraph(kg={'heads': h, 'tails': t, 'relations': r}, ent2ix={i: i for i in range(n_ent)},
                           rel2ix={i: i for i in range(n_rel)})
    _, kg_test = kg.split_kg(sizes=(11000, 1000))
    for x in ('head_idx', 'tail_idx', 'relations'):
        setattr(kg_test, x, getattr(kg_test, x).cuda())
    names = ['rank_true_heads', 'rank_true_tails', 'filt_rank_true_heads', 'filt_rank_true_tails']

    def fresh(b=512):
        monkeypatch.setattr(evm, 'FAST_REPLAY', False)
        e = tk.LinkPredictionEvaluator(m, kg_test, graph=False, share_state=False)
        e.evaluate(b_size=b, verbose=False)
        monkeypatch.setattr(evm, 'FAST_REPLAY', True)
        return [getattr(e, nm).clone() for nm in names]

    def same(e, want, what):
        for nm, w in zip(names, want):
            assert torch.equal(getattr(e, nm), w), what

    ev = tk.LinkPredictionEvaluator(m, kg_test, share_state=False)
    want = fresh()
    for i in range(5):
        ev.evaluate(b_size=512, verbose=False)
        same(ev, want, 'steady state %d' % i)
    assert ev._st._fast is not None, 'the fast path is armed after plain replays'
    n_before = ev._n_evaluations
    ev.evaluate(b_size=512, verbose=False)
    assert ev._st._fast is not None and ev._n_evaluations == n_before + 1
    # table values change IN PLACE (an optimiser step): same addresses, the replay reads the new values
    with torch.no_grad():
        m.ent_emb.weight.mul_(0.9).add_(0.01 * torch.randn_like(m.ent_emb.weight))
        m.normalize_parameters()
    want2 = fresh()
    assert any(not torch.equal(a, b) for a, b in zip(want, want2))
    ev.evaluate(b_size=512, verbose=False)
    same(ev, want2, 'values written in place')
    # tables REPLACED (new storage): the capture key changes -> full path, new capture
    m.ent_emb.weight.data = m.ent_emb.weight.data.clone()
    ev.evaluate(b_size=512, verbose=False)
    same(ev, want2, 'tables replaced')
    # facts edited in place
    kg_test.tail_idx[0:10] = (kg_test.tail_idx[0:10] + 1) % n_ent
    want3 = fresh()
    for _ in range(3):
        ev.evaluate(b_size=512, verbose=False)
        same(ev, want3, 'facts edited in place')
    # another b_size, then the prefilter switched off: both key the capture
    ev.evaluate(b_size=300, verbose=False)
    same(ev, want3, 'b_size')
    m.split_filter = False
    ev.evaluate(b_size=300, verbose=False)
    same(ev, want3, 'split_filter off')
    m.split_filter = True


def test_filter_lookup_both_equals_two_lookups(hip):
    """kge_filter_lookup_both = kge_filter_lookup on the tail index and on the head index
    (head segments shifted by the tail index's target count), plus the concatenated true ids."""
    from torchkge_amd.filter_index import FilterIndex, KEY2_SPAN
    g = torch.Generator().manual_seed(5)
    n_ent, n_rel, n = 50, 7, 400
    h = torch.randint(0, n_ent, (n,), generator=g); t = torch.randint(0, n_ent, (n,), generator=g)
    r = torch.randint(0, n_rel, (n,), generator=g)
    d_t, d_h = {}, {}
    for a, b, c in zip(h.tolist(), t.tolist(), r.tolist()):
        d_t.setdefault((a, c), set()).add(b)
        d_h.setdefault((b, c), set()).add(a)
    it, ih = FilterIndex.from_dict(d_t, 'cuda'), FilterIndex.from_dict(d_h, 'cuda')
    qh = torch.randint(0, n_ent + 5, (333,), generator=g).cuda()      # some keys absent
    qt = torch.randint(0, n_ent + 5, (333,), generator=g).cuda()
    qr = torch.randint(0, n_rel, (333,), generator=g).cuda()
    lo, hi, true = hip.filter_lookup_both(it.keys, it.offsets, ih.keys, ih.offsets, it.targets.shape[0], qh, qt, qr,
                                          KEY2_SPAN)
    lo_t, hi_t = it.lookup(qh, qr)
    lo_h, hi_h = ih.lookup(qt, qr)
    base = it.targets.shape[0]
    found_h = hi_h > lo_h
    assert torch.equal(lo[:333], lo_t) and torch.equal(hi[:333], hi_t)
    assert torch.equal(lo[333:], torch.where(found_h, lo_h + base, lo_h))
    assert torch.equal(hi[333:], torch.where(found_h, hi_h + base, hi_h))
    assert torch.equal(true, torch.cat([qt, qh]))
    cat = torch.cat([it.targets, ih.targets])
    for i in (0, 17, 332):
        assert torch.equal(cat[lo[333 + i]:hi[333 + i]], ih.targets[lo_h[i]:hi_h[i]])


# ---------------------------------------------------------------------------
# TransE-L1: certified 16-bit sum-of-absolute-differences prefilter (lp_l1_sad.hip)
# ---------------------------------------------------------------------------
def _sad_problem(hip, Q, E):
    prob = hip.LpProblem(hip.LP_L1_DIRECT, Q, E)
    bounds = torch.zeros(3, dtype=torch.float32, device='cuda')
    hip.absmax(torch.cat([Q.reshape(-1), E.reshape(-1)]), bounds[0:1])     # rmax = 0: the bound is max |x| itself
    prob.sad = {'Ei': hip.sad_rows(E, bounds[0:1], bounds[1:2]), 'emax': bounds[0:1], 'rmax': bounds[1:2],
                'overflow': bounds[2:3]}
    return prob, bounds


@pytest.mark.parametrize('B,N,K', [(64, 300, 32), (1000, 3000, 200), (193, 257, 17), (5, 2, 1), (700, 1500, 203), (129, 129, 8)])
def test_l1_sad_prefilter_counts_equal_exact_counts(hip, B, N, K):
    """kge_lp_sad_count + kge_lp_sad_recheck == kge_lp_count_ge (lp_direct_kernel) for plain L1 problems: random
    operands, exact ties (duplicate candidates, the true candidate itself), thresholds from real pair scores; also
    with the error band halved (the proven bound K is a worst case; the typical error grows like sqrt(K / 6), so
    small K leave little slack) -- and the prefilter really decides most pairs."""
    g = torch.Generator().manual_seed(B * 7 + N + K)
    E = ((torch.rand(N, K, generator=g) * 2 - 1) * 0.3).cuda()
    Q = ((torch.rand(B, K, generator=g) * 2 - 1) * 0.45).cuda()
    if N > 10:
        E[5] = E[3]                      # exact duplicates: ties
        E[N - 1] = E[N // 2]
    ci = torch.randint(0, N, (B,), generator=g).cuda()
    if B > 3:
        Q[1] = E[int(ci[1])]             # distance 0 to its true candidate
    exact = hip.LpProblem(hip.LP_L1_DIRECT, Q, E)
    s_true = exact.pair_scores(ci)
    if B > 4:
        s_true[2] = -float('inf')        # everything counts
        s_true[3] = float('nan')         # nothing counts
    want = exact.count_ge(s_true)
    for eps in (1.0, 0.5):
        old = hip.SPLIT_EPS_SCALE
        hip.SPLIT_EPS_SCALE = eps
        try:
            prob, bounds = _sad_problem(hip, Q, E)
            got = prob.count_ge(s_true)
        finally:
            hip.SPLIT_EPS_SCALE = old
        assert float(bounds[2]) == 0.0
        assert torch.equal(got, want), (eps, int((got != want).sum()))
        n_unc = int(prob.last_split[0])
        assert n_unc <= max(64, 0.05 * B * N), n_unc       # most pairs never reach the exact chain


def test_l1_sad_prefilter_overflow_and_degenerate_tables_fall_back(hip):
    """All candidates identical -> every pair is tied with the true score -> the uncertain list overflows: the flag
    is raised (the evaluator then redoes the count exactly); all-zero tables raise it too."""
    B, N, K = 300, 2000, 64
    E = torch.full((N, K), 0.25, device='cuda')
    Q = torch.full((B, K), 0.5, device='cuda')
    prob, bounds = _sad_problem(hip, Q, E)
    s_true = prob.pair_scores(torch.zeros(B, dtype=torch.long, device='cuda'))
    hip.SPLIT_LIST_PER_QUERY, old = 4, hip.SPLIT_LIST_PER_QUERY
    try:
        prob.count_ge(s_true)
    finally:
        hip.SPLIT_LIST_PER_QUERY = old
    assert float(bounds[2]) == 1.0
    Z = torch.zeros(8, K, device='cuda')
    prob, bounds = _sad_problem(hip, Z, Z.clone())
    prob.count_ge(prob.pair_scores(torch.zeros(8, dtype=torch.long, device='cuda')))
    assert float(bounds[2]) == 1.0


def test_l1_evaluator_uses_the_prefilter_and_matches_exact_counts(hip):
    """TransE-L1 evaluate(): the SAD prefilter is what runs by default; ranks equal the all-exact evaluation."""
    import torchkge_amd as tk
    n_ent, n_rel, d = 3000, 9, 100
    tables = orc.init_tables('transe', n_ent, n_rel, d, seed=5)
    m = build_model('transe', 1, tables, n_ent, n_rel)
    h, t, r = orc.synthetic_triples_zipf(n_ent, n_rel, 20000, 31, hubs=((500, 'head'),))
    kg = tk.KnowledgeGraph(kg={'heads': h, 'tails': t, 'relations': r}, ent2ix={i: i for i in range(n_ent)},
                           rel2ix={i: i for i in range(n_rel)})
    _, kg_test = kg.split_kg(sizes=(18000, 2000))
    res = []
    for split in (True, False):
        m.split_filter = split
        ev = tk.LinkPredictionEvaluator(m, kg_test, graph=True)
        ev.evaluate(b_size=1024, verbose=False)
        ev.evaluate(b_size=1024, verbose=False)
        res.append([ev.rank_true_heads.clone(), ev.rank_true_tails.clone(), ev.filt_rank_true_heads.clone(),
                    ev.filt_rank_true_tails.clone()])
    m.split_filter = True
    for a, b in zip(*res):
        assert torch.equal(a, b)
    prob = None
    guard = m.lp_guard_begin(torch.device('cuda', 0))
    try:
        with m.lp_session():
            prob = m.lp_problem(kg_test.head_idx[:64].cuda(), kg_test.tail_idx[:64].cuda(), kg_test.relations[:64].cuda(), 'both')
            assert prob.sad is not None and guard is not None
    finally:
        m.lp_guard_end()


@pytest.mark.parametrize('n_ent,n_rel,n,seed', [(60, 5, 5000, 3), (14541, 237, 60000, 4), (7, 1, 9, 5), (3000, 11, 1, 6),
                                                 (200000, 50, 300000, 7)])
def test_device_built_index_and_plans_equal_the_torch_builds(hip, n_ent, n_rel, n, seed):
    """The library's own index / plan builders (index_build.hip: kge_filter_index_build, kge_filter_plan_build,
    kge_column_plan_build / _emit -- rocPRIM sorts and scans + flag / scatter kernels) against the ATen compositions they
    replace, run on CPU tensors: every output tensor equal, element for element.  Hub keys (lists longer than 512, keys
    with more queries than a grouped column holds), duplicate facts, a relation-major column order."""
    from torchkge_amd.filter_index import FilterIndex, FilterPlan, ColumnPlan, KEY2_SPAN
    g = torch.Generator().manual_seed(seed)
    h = torch.randint(0, n_ent, (n,), generator=g)
    t = torch.randint(0, n_ent, (n,), generator=g)
    r = torch.randint(0, n_rel, (n,), generator=g)
    if n >= 5000:       # hub keys: one (h, r) with 900 tails, one (t, r) with 700 heads, and exact duplicates
        h[:900], r[:900] = 1, 0
        t[1000:1700], r[1000:1700] = 2, n_rel - 1
        h[2000:2050], t[2000:2050], r[2000:2050] = h[0], t[0], r[0]
    for k1, k2, v in ((h, r, t), (t, r, h)):
        ref = FilterIndex.from_triples(k1.numpy(), k2.numpy(), v.numpy(), 'cpu')
        for bounds in ((n_ent, n_rel, n_ent), (None, None, None)):
            got = FilterIndex.from_triples_device(k1, k2, v, 'cuda', *bounds)
            assert torch.equal(got.keys.cpu(), ref.keys) and torch.equal(got.offsets.cpu(), ref.offsets)
            assert torch.equal(got.targets.cpu()[:int(ref.offsets[-1])], ref.targets[:int(ref.offsets[-1])])
    # plans of a batch of test facts: the last facts of the graph (they hit the hub keys when n is large)
    idx_t = FilterIndex.from_triples_device(h, r, t, 'cuda', n_ent, n_rel, n_ent)
    idx_h = FilterIndex.from_triples_device(t, r, h, 'cuda', n_ent, n_rel, n_ent)
    B = min(n, 3000)
    sel = torch.cat([torch.arange(0, min(n, 1200)), torch.arange(max(0, n - (B - min(n, 1200))), n)])[:B]
    hb, tb, rb = h[sel].cuda(), t[sel].cuda(), r[sel].cuda()
    lo, hi, true = hip.filter_lookup_both(idx_t.keys, idx_t.offsets, idx_h.keys, idx_h.offsets, idx_t.targets.shape[0],
                                          hb, tb, rb, KEY2_SPAN)
    targets = torch.cat([idx_t.targets, idx_h.targets])
    pd = FilterPlan(lo, hi, true, targets)
    pc = FilterPlan(lo.cpu(), hi.cpu(), true.cpu(), targets.cpu())
    assert pd.n_pairs == pc.n_pairs and pd.n_long == pc.n_long
    assert torch.equal(pd.woff.cpu(), pc.woff) and torch.equal(pd.long_q.cpu(), pc.long_q)
    sets = hip.split_group_sets()
    for rel_major in (False, True):
        cd = ColumnPlan(hb, tb, rb, n_ent, n_rel, sets, hip.split_query_rows_padded, relation_major=rel_major)
        cc = ColumnPlan(hb.cpu(), tb.cpu(), rb.cpu(), n_ent, n_rel, sets, hip.split_query_rows_padded, relation_major=rel_major)
        for nm in ('n_queries', 'n_single', 'n_multi', 'n_single_p', 'n_multi_p', 'sets', 'n_columns', 'n_distinct_keys'):
            assert getattr(cd, nm) == getattr(cc, nm), nm
        for nm in ('col_q', 'members', 'qs_row', 'col_of_q', 'rep'):
            assert torch.equal(getattr(cd, nm).cpu(), getattr(cc, nm)), nm
    # the relation order of a batch (TransH / TransD): the library's radix sort == a stable argsort
    keys = (torch.arange(B) // 1000) * n_rel + rb.cpu()
    assert torch.equal(hip.sort_perm(keys.cuda(), (B // 1000 + 1) * n_rel).cpu(), torch.argsort(keys, stable=True))


@pytest.mark.parametrize('frag', [False, True])
@pytest.mark.parametrize('B,N,K,cols_on', [(64, 300, 32, False), (1000, 3000, 200, False), (193, 257, 17, False), (5, 2, 1, False),
                                           (700, 1500, 203, False), (1000, 3000, 200, True), (400, 5000, 64, True),
                                           (3000, 20000, 200, False), (2500, 9000, 400, False), (300, 1100, 500, False)])
def test_split_one_product_level_counts_equal_exact_counts(hip, B, N, K, cols_on, frag):
    """The ONE-PRODUCT level of the split prefilter (kge_split_args.level = 1: planar hi operands, one MFMA product per
    k16 unit, thresholds from the operands' measured f16 residuals) + exact recheck leave exactly the counts of the
    fp32 kernel -- TransE-L2 shaped problems, per query and over query columns (hub keys), also with the band shrunk."""
    from torchkge_amd.filter_index import ColumnPlan
    g = torch.Generator().manual_seed(B * 11 + K)
    E = torch.nn.functional.normalize(torch.randn(N, K, generator=g), dim=1)
    R = torch.randn(7, K, generator=g) * (0.6 / K ** 0.5)
    if cols_on:     # a both-sides batch of B // 2 facts with repeated (h, r) / (t, r) keys
        nb = B // 2
        h = torch.randint(0, max(N // 20, 1), (nb,), generator=g); t = torch.randint(0, N, (nb,), generator=g)
        r = torch.randint(0, 3, (nb,), generator=g)
        h[:40], r[:40] = 1, 0                     # one key with 40 queries: ten grouped columns
        q = torch.cat([E[h] + R[r], E[t] - R[r]])
        true = torch.cat([t, h])
        cols = ColumnPlan(h.cuda(), t.cuda(), r.cuda(), N, 7, hip.split_group_sets(), hip.split_query_rows_padded)
        B = 2 * nb
    else:
        h = torch.randint(0, N, (B,), generator=g); r = torch.randint(0, 7, (B,), generator=g)
        q = E[h] + R[r]
        true = torch.randint(0, N, (B,), generator=g)
        cols = None
    dE, dq, dt = E.cuda(), q.cuda().contiguous(), true.cuda()
    guard = torch.zeros(8, device='cuda')
    en = hip.row_sqnorm(dE, max_io=guard[1:2]); qn = hip.row_sqnorm(dq, max_io=guard[0:1])
    prob = hip.LpProblem(hip.LP_L2_EXPAND, dq, dE, qn=qn, en=en)
    st = prob.pair_scores(dt)
    exact = prob.count_ge(st)
    # frag: the FREE-RUNNING kernel (lp_hi_stream.hip, r05) on the fragment-major table -- it sweeps per query (grouped
    # columns are dropped by split_prepare), single-query columns pass through col_q
    Eh, de2 = hip.hi_table(dE, aug=en, frag=frag)
    assert 0.0 <= float(de2) < 1e-6 * float(guard[1])         # ||e - hi(e)|| ~ 2^-12 ||e|| (0 when every value is an f16)
    prob.split = {'Es': Eh, 'e2pref': None, 'enmax': guard[1:2], 'overflow': guard[2:3], 'level': 1, 'de2max': de2,
                  'list_stat': guard[6:7], 'es_frag': frag}
    prob.cols = cols
    try:
        # (the band's residual term bounds ACTUAL f16 rounding errors by Cauchy-Schwarz: tight at small K, so only a
        # mild shrink at K >= 200 -- tests/test_split_band_model.py)
        for eps in ((1.0, 0.5) if K >= 200 else (1.0,)):
            hip.SPLIT_EPS_SCALE = eps
            guard[6] = 0
            got = prob.count_ge(st)
            assert torch.equal(got, exact), (eps, int((got != exact).sum()))
            n_unc = int(prob.last_split[0].item())
            assert B <= n_unc                        # at least the true entity of every query is re-scored
            assert float(guard[6]) == n_unc          # the re-scored pairs are reported (the evaluator's level policy)
    finally:
        hip.SPLIT_EPS_SCALE = 1.0
    assert float(guard[2]) == 0.0


@pytest.mark.parametrize('frag', [False, True])
@pytest.mark.parametrize('B,N,K,K1,scale', [(300, 1000, 64, 0, 1.0), (257, 700, 40, 40, 30.0), (100, 513, 17, 17, 1e-3),
                                          (64, 300, 200, 200, 1.0), (1200, 4100, 400, 0, 1.0)])
def test_split_one_product_level_dot_mode(hip, B, N, K, K1, scale, frag):
    """The one-product level on KGE_LP_DOT problems (DistMult / ComplEx: two K-segments, operands with their own
    power-of-two scales, negative true scores, an all-zero query)."""
    g = torch.Generator().manual_seed(B * 7 + K + 1)
    T0 = (torch.randn(N, K, generator=g) * scale).cuda()
    T1 = (torch.randn(N, K1, generator=g) * scale).cuda() if K1 else None
    A0 = (torch.randn(B, K, generator=g) * scale).cuda()
    A1 = (torch.randn(B, K1, generator=g) * scale).cuda() if K1 else None
    A0[0] = 0.0
    if A1 is not None:
        A1[0] = 0.0
    t = torch.randint(0, N, (B,), generator=g).cuda()
    prob = hip.LpProblem(hip.LP_DOT, A0, T0, A1=A1, T1=T1)
    st = prob.pair_scores(t)
    exact = prob.count_ge(st)
    guard = torch.zeros(8, device='cuda')
    hip.row_sqnorm(T0, max_io=guard[1:2])
    nm1 = None
    if T1 is not None:
        hip.row_sqnorm(T1, max_io=guard[5:6])
        nm1 = guard[5:6]
    Eh, de2 = hip.hi_table(T0, X1=T1, dot=True, nmax0=guard[1:2], nmax1=nm1, frag=frag)
    prob.split = {'Es': Eh, 'e2pref': None, 'enmax': guard[1:2], 'enmax1': nm1, 'overflow': guard[2:3], 'level': 1,
                  'de2max': de2, 'list_stat': guard[6:7], 'es_frag': frag}
    try:
        for eps in ((1.0, 0.5) if K + K1 >= 200 else (1.0,)):
            hip.SPLIT_EPS_SCALE = eps
            got = prob.count_ge(st)
            assert torch.equal(got, exact), (eps, int((got != exact).sum()))
    finally:
        hip.SPLIT_EPS_SCALE = 1.0


@pytest.mark.parametrize('nt', [3, 4])
@pytest.mark.parametrize('mode,B,N,K,K1,scale', [('l2', 1500, 6000, 1024, 0, 1.0), ('l2', 700, 3000, 512, 0, 1.0),
                                               ('l2', 130, 70001, 1024, 0, 1.0), ('dot', 500, 4000, 512, 512, 1.0),
                                               ('dot', 300, 2000, 256, 256, 30.0), ('dot', 1100, 2500, 512, 0, 1e-2),
                                               ('dot', 64, 513, 1024, 0, 1.0)])
def test_split_one_product_level_long_rows_chunked_panel(hip, monkeypatch, mode, B, N, K, K1, scale, nt):
    """Rows too long for a resident query panel (33 / 65 k16 units: K = 512 / 1024, ComplEx d = 512 of BASELINE cfg5) take the
    CHUNKED-panel kernel (lp_hi_chunk.hip, r06: the panel streamed through a two-slot LDS ring, 96 or 128 queries per panel):
    counts after the exact recheck == the fp32 kernel's; batch sizes that leave partial panels, candidate counts that leave
    partial tiles.  At K = 1024 the one-product band holds ~1 % of the candidates around a threshold in the BULK of the score
    distribution -- more than the list's N / 100 entries per query, which is what sends an unfitted model back to three
    products -- so most true entities here sit where a fitted model's do (around rank 0.2 % of N: a handful of listed pairs
    per query), one query in eight at rank 5 % (hundreds), four in the bulk."""
    monkeypatch.setenv('KGE_HC_NT', str(nt))
    assert hip.hi_stream_ok(K + K1)
    g = torch.Generator().manual_seed(B * 13 + K + nt)
    guard = torch.zeros(8, device='cuda')
    if mode == 'l2':
        E = torch.nn.functional.normalize(torch.randn(N, K, generator=g), dim=1)
        R = torch.randn(7, K, generator=g) * (0.6 / K ** 0.5)
        h = torch.randint(0, N, (B,), generator=g); r = torch.randint(0, 7, (B,), generator=g)
        dE, dq = E.cuda(), (E[h] + R[r]).cuda().contiguous()
        en = hip.row_sqnorm(dE, max_io=guard[1:2]); qn = hip.row_sqnorm(dq, max_io=guard[0:1])
        prob = hip.LpProblem(hip.LP_L2_EXPAND, dq, dE, qn=qn, en=en)
        Eh, de2 = hip.hi_table(dE, aug=en, frag=True)
        split = {'Es': Eh, 'e2pref': None, 'enmax': guard[1:2], 'overflow': guard[2:3], 'level': 1, 'de2max': de2,
                 'list_stat': guard[6:7], 'es_frag': True}
    else:
        T0 = (torch.randn(N, K, generator=g) * scale).cuda()
        T1 = (torch.randn(N, K1, generator=g) * scale).cuda() if K1 else None
        A0 = (torch.randn(B, K, generator=g) * scale).cuda()
        A1 = (torch.randn(B, K1, generator=g) * scale).cuda() if K1 else None
        A0[0] = 0.0
        if A1 is not None:
            A1[0] = 0.0
        prob = hip.LpProblem(hip.LP_DOT, A0, T0, A1=A1, T1=T1)
        hip.row_sqnorm(T0, max_io=guard[1:2])
        nm1 = None
        if T1 is not None:
            hip.row_sqnorm(T1, max_io=guard[5:6])
            nm1 = guard[5:6]
        Eh, de2 = hip.hi_table(T0, X1=T1, dot=True, nmax0=guard[1:2], nmax1=nm1, frag=True)
        split = {'Es': Eh, 'e2pref': None, 'enmax': guard[1:2], 'enmax1': nm1, 'overflow': guard[2:3], 'level': 1,
                 'de2max': de2, 'list_stat': guard[6:7], 'es_frag': True}
    # true entity of query i = the candidate at a chosen rank of its exact score row
    rank_of = torch.full((B,), max(1, N // 500), dtype=torch.long)
    rank_of[::8] = max(1, N // 20)
    rank_of[torch.randperm(B, generator=g)[:4]] = N // 2
    t = torch.empty(B, dtype=torch.long, device='cuda')
    for lo in range(0, B, 256):
        sc = prob.scores_rows(lo, min(lo + 256, B), torch.empty(min(256, B - lo), N, device='cuda'))
        order = torch.argsort(sc, dim=1, descending=True)
        t[lo:lo + 256] = order.gather(1, rank_of[lo:lo + 256].cuda().view(-1, 1)).view(-1)
        del sc, order
    st = prob.pair_scores(t)
    exact = prob.count_ge(st)
    assert int(exact.max()) >= N // 2 and int(exact.min()) <= N // 500 + 64
    prob.split = split
    try:
        for eps in (1.0, 0.5):
            hip.SPLIT_EPS_SCALE = eps
            guard[6] = 0
            got = prob.count_ge(st)
            assert float(guard[2]) == 0.0, 'list overflow'
            assert torch.equal(got, exact), (eps, int((got != exact).sum()))
            assert int(prob.last_split[0].item()) >= B
    finally:
        hip.SPLIT_EPS_SCALE = 1.0


@pytest.mark.parametrize('kind,d', [('complex', 200), ('distmult', 400), ('complex', 136), ('distmult', 200)])
@pytest.mark.parametrize('seg_bytes', [None, 8192])
def test_region_recheck_of_long_rows_in_segments(hip, kind, d, seg_bytes, monkeypatch):
    """r06: kge_lp_split_recheck_regions for rows longer than one LDS segment (DistMult / ComplEx d = 400: K = 400 / 2 x 200;
    KGE_REGION_MAX_BYTES=8192: segments of 32 columns, every shape in several pieces and ComplEx's [Re | Im] boundary inside
    one) -- the chains of a batch of pairs rest in registers between segments: same counts and as many listed pairs as the
    global list + kge_lp_split_recheck, equal to the exact fp32 counts."""
    if seg_bytes is not None:
        monkeypatch.setenv('KGE_REGION_MAX_BYTES', str(seg_bytes))
    cplx = kind == 'complex'
    n_ent, n_rel, B = 1800, 6, 900
    tables = orc.init_tables(kind, n_ent, n_rel, d, seed=13)
    m = build_model(kind, 2, tables, n_ent, n_rel)
    tabs = [hip.f32c(x.data) for x in m._tables()]
    T0, T1 = tabs[0], (tabs[1] if cplx else None)
    rel = tabs[2:] if cplx else tabs[1:]
    gen = torch.Generator().manual_seed(d)
    h = torch.randint(0, n_ent, (B,), generator=gen).cuda(); t = torch.randint(0, n_ent, (B,), generator=gen).cuda()
    h[: B // 2] = h[0]          # a hub: a region with many pairs (several batches of the segmented recheck)
    r = torch.randint(0, n_rel, (B,), generator=gen).cuda()
    true = torch.cat([t, h])
    got, listed = {}, {}
    for regions in (False, True):
        g = torch.zeros(8, device='cuda')
        nm1 = g[5:6] if cplx else None
        Eh, dnb, _ws = hip.dot_table_prep(T0, T1, g[1:2], nm1, True)
        pre = hip.lp_dot_query_pipeline(hip.SIDE_BOTH, T0, T1, rel[0], rel[1] if cplx else None, h, t, r, g[1:2], nm1, g[7:8],
                                        g[0:1], g[2:3], zero_counts=True, dn_bmax=dnb, regions=regions)
        assert (pre.get('region_count') is not None) == regions
        pre['true_idx'] = true
        prob = hip.LpProblem(hip.LP_DOT, pre['Q'], T0, A1=pre['Q1'], T1=T1)
        prob.split = {'Es': Eh, 'e2pref': None, 'enmax': g[1:2], 'enmax1': nm1, 'overflow': g[2:3], 'level': 1, 'de2max': g[7:8],
                      'list_stat': g[6:7], 'es_frag': True}
        prob.pre = pre
        st = prob.pair_scores(true)
        got[regions] = prob.count_ge(st).clone()
        assert float(g[2]) == 0.0
        listed[regions] = (int(prob.last_split[0]), float(g[6]))
        if regions:
            assert int(pre['region_count'].sum()) == listed[True][0]
            ref = hip.LpProblem(hip.LP_DOT, pre['Q'], T0, A1=pre['Q1'], T1=T1)
            assert torch.equal(got[True], ref.count_ge(st))
    assert torch.equal(got[False], got[True])
    assert listed[False] == listed[True] and listed[True][0] > 0


@pytest.mark.parametrize('kind,d,n_ent', [('distmult', 64, 1500), ('complex', 40, 777), ('complex', 512, 300), ('distmult', 400, 2049)])
def test_dot_candidate_table_in_one_pass(hip, kind, d, n_ent):
    """r06: kge_lp_dot_table_prep_fused -- the fragment-major hi table of a DOT candidate table in ONE pass, scaled by the
    squared-norm maxima a previous evaluation left: the same table bit for bit as the two-launch preparation when those
    maxima still hold, the same residual bound, this pass's norm maxima per block; the query pipeline folds them, stores
    them for the next call and raises the overflow flag when the table has left the binade its scale was chosen for."""
    cplx = kind == 'complex'
    tables = orc.init_tables(kind, n_ent, 7, d, seed=3)
    m = build_model(kind, 2, tables, n_ent, 7)
    tabs = [hip.f32c(x.data) for x in m._tables()]
    T0, T1 = tabs[0], (tabs[1] if cplx else None)
    rel = tabs[2:] if cplx else tabs[1:]
    with torch.no_grad():
        T0[::5] *= 2.5
    g = torch.zeros(8, device='cuda')
    nm1 = g[5:6] if cplx else None
    Eh, dnb, _ws = hip.dot_table_prep(T0, T1, g[1:2], nm1, True)
    prev = torch.stack([g[1], g[5]]).contiguous()
    g2 = torch.zeros(8, device='cuda')
    Eh2, dnb2, nmb = hip.dot_table_prep(T0, T1, g2[1:2], g2[5:6] if cplx else None, True, prev_nmax=prev)
    assert torch.equal(Eh2, Eh)
    assert float(dnb2.max()) == pytest.approx(float(dnb.max()), rel=1e-5)
    nb = nmb.shape[0] // 2
    n0, n1 = float(nmb[:nb].max()), float(nmb[nb:].max())
    assert n0 == pytest.approx(float(g[1]), rel=2e-4) and n0 >= float((T0.double() ** 2).sum(1).max()) * (1 - 1e-6)
    if cplx:
        assert n1 == pytest.approx(float(g[5]), rel=2e-4) and n1 >= float((T1.double() ** 2).sum(1).max()) * (1 - 1e-6)
    else:
        assert n1 == 0.0
    B = 300
    gen = torch.Generator().manual_seed(d)
    h = torch.randint(0, n_ent, (B,), generator=gen).cuda(); t = torch.randint(0, n_ent, (B,), generator=gen).cuda()
    r = torch.randint(0, 7, (B,), generator=gen).cuda()

    def pipe(gv, prev_):
        return hip.lp_dot_query_pipeline(hip.SIDE_BOTH, T0, T1, rel[0], rel[1] if cplx else None, h, t, r, gv[1:2],
                                         gv[5:6] if cplx else None, gv[7:8], gv[0:1], gv[2:3], zero_counts=True, dn_bmax=dnb2,
                                         nm_bmax=nmb, prev_nmax=prev_)
    pre = pipe(g2, prev)
    assert float(g2[2]) == 0.0 and float(g2[1]) == n0 and float(g2[5]) == n1      # folded, no flag, stored for the next call
    assert float(prev[0]) == n0 and float(prev[1]) == n1
    # counts through the sweep on the one-pass operands == exact counts
    true = torch.cat([t, h])
    ref = hip.LpProblem(hip.LP_DOT, pre['Q'], T0, A1=pre['Q1'], T1=T1)
    st = ref.pair_scores(true)
    exact = ref.count_ge(st)
    prob = hip.LpProblem(hip.LP_DOT, pre['Q'], T0, A1=pre['Q1'], T1=T1)
    pre['true_idx'] = true
    prob.split = {'Es': Eh2, 'e2pref': None, 'enmax': g2[1:2], 'enmax1': g2[5:6] if cplx else None, 'overflow': g2[2:3], 'level': 1,
                  'de2max': g2[7:8], 'list_stat': g2[6:7], 'es_frag': True}
    prob.pre = pre
    got = prob.count_ge(prob.pair_scores(true))
    assert float(g2[2]) == 0.0 and torch.equal(got, exact)
    # a table scaled under maxima of ANOTHER binade (norms 16 x smaller / larger then): the pipeline says so and leaves the
    # true maxima for the next call
    for f in (1.0 / 16, 16.0):
        stale = torch.tensor([n0 * f, n1 * f], device='cuda')
        g3 = torch.zeros(8, device='cuda')
        _E, dnb3, nmb3 = hip.dot_table_prep(T0, T1, g3[1:2], g3[5:6] if cplx else None, True, prev_nmax=stale)
        hip.lp_dot_query_pipeline(hip.SIDE_BOTH, T0, T1, rel[0], rel[1] if cplx else None, h, t, r, g3[1:2],
                                  g3[5:6] if cplx else None, g3[7:8], g3[0:1], g3[2:3], zero_counts=True, dn_bmax=dnb3,
                                  nm_bmax=nmb3, prev_nmax=stale)
        assert float(g3[2]) == 2.0 and float(stale[0]) == n0 and float(stale[1]) == n1


@pytest.mark.parametrize('kind', ['transe', 'transh', 'complex'])
def test_ranks_written_straight_to_pinned_host_memory(hip, kind, monkeypatch):
    """r06: the finalize launches of a single-GPU both-sides evaluation write ranks and flags into a pinned host buffer whose
    address they read from a mailbox (kge_rank_finalize_both out_indirect; TransH: the packed result by one coalesced pass,
    kge_copy_i64_indirect) -- same ranks as through the device-to-host copy, eager and as graph replays, and every call hands
    out its OWN buffer: the tensors of an earlier call survive later calls on changed tables."""
    import torchkge_amd as tk
    import torchkge_amd.evaluation as evm
    n_ent, n_rel, d = 2500, 7, 64
    tables = orc.init_tables(kind, n_ent, n_rel, d, seed=9)
    m = build_model(kind, 2, tables, n_ent, n_rel)
    h, t, r = orc.synthetic_triples_zipf(n_ent, n_rel, 16000, 3, hubs=((700, 'tail'),))
    kg = tk.KnowledgeGraph(kg={'heads': h, 'tails': t, 'relations': r}, ent2ix={i: i for i in range(n_ent)},
                           rel2ix={i: i for i in range(n_rel)})
    _, kg_test = kg.split_kg(sizes=(14800, 1200))

    def ranks(ev):
        return [ev.rank_true_heads, ev.rank_true_tails, ev.filt_rank_true_heads, ev.filt_rank_true_tails]
    monkeypatch.setattr(evm, 'DIRECT_HOST_RANKS', False)
    ev0 = tk.LinkPredictionEvaluator(m, kg_test, graph=False, share_state=False)
    ev0.evaluate(256, verbose=False)
    want = [x.clone() for x in ranks(ev0)]
    monkeypatch.setattr(evm, 'DIRECT_HOST_RANKS', True)
    assert hip.host_device_pointer(torch.zeros(4, dtype=torch.int64, pin_memory=True)) is not None
    for graph in (False, None):
        ev = tk.LinkPredictionEvaluator(m, kg_test, graph=graph, share_state=False)
        kept = []
        for _ in range(4):
            ev.evaluate(256, verbose=False)
            got = ranks(ev)
            assert all(x.is_pinned() for x in got) and all(torch.equal(a, b) for a, b in zip(want, got))
            kept.append(got)
        assert ev._st.__dict__.get('_mailbox') is not None and ev._st.__dict__['_mailbox'][1] is not None
        assert len({x[0].data_ptr() for x in kept}) == len(kept)        # a fresh buffer per call
        # other tables: new ranks in a new buffer, the old tensors untouched
        ent = m.ent_emb if kind != 'complex' else m.re_ent_emb
        with torch.no_grad():
            ent.weight.data[::3] *= -1.0
        ev.evaluate(256, verbose=False)
        assert not all(torch.equal(a, b) for a, b in zip(want, ranks(ev)))
        assert all(torch.equal(a, b) for got in kept for a, b in zip(want, got))
        with torch.no_grad():
            ent.weight.data[::3] *= -1.0


@pytest.mark.parametrize('kind', ['distmult', 'complex'])
def test_evaluator_one_pass_table_follows_growing_tables(hip, kind):
    """The evaluator on the one-product level takes the DOT candidate table's scale from the previous evaluation's maxima
    (one pass over the table): ranks stay those of the exact fp32 counts while the tables change between evaluations --
    slowly (same binade: no redo) and by a factor of 4 (the scale is stale: that evaluation is redone, the next one is
    back on one pass) -- as hipGraph replays."""
    import torchkge_amd as tk
    n_ent, n_rel, d = 3000, 9, 64
    tables = orc.init_tables(kind, n_ent, n_rel, d, seed=7)
    m = build_model(kind, 2, tables, n_ent, n_rel)
    h, t, r = orc.synthetic_triples_zipf(n_ent, n_rel, 20000, 5, hubs=((900, 'head'),))
    kg = tk.KnowledgeGraph(kg={'heads': h, 'tails': t, 'relations': r}, ent2ix={i: i for i in range(n_ent)},
                           rel2ix={i: i for i in range(n_rel)})
    _, kg_test = kg.split_kg(sizes=(18500, 1500))

    def ranks(ev):
        return [ev.rank_true_heads.clone(), ev.rank_true_tails.clone(), ev.filt_rank_true_heads.clone(),
                ev.filt_rank_true_tails.clone()]

    def exact():
        m.split_filter = False
        try:
            e = tk.LinkPredictionEvaluator(m, kg_test, graph=False, share_state=False)
            e.evaluate(512, verbose=False)
            return ranks(e)
        finally:
            m.split_filter = True
    m.split_level = 1
    ev = tk.LinkPredictionEvaluator(m, kg_test)
    ev._level = 1
    ent_tabs = [m.ent_emb] if kind == 'distmult' else [m.re_ent_emb, m.im_ent_emb]
    redone = []
    for step, f in enumerate((1.0, 1.0, 1.01, 1.02, 4.0, 1.0, 1.03, 0.2, 1.0)):
        with torch.no_grad():
            for e_ in ent_tabs:
                e_.weight.data.mul_(f)
        ev.evaluate(512, verbose=False)
        for a, b in zip(exact(), ranks(ev)):
            assert torch.equal(a, b), 'step %d (tables x %g)' % (step, f)
        redone.append(bool(getattr(ev, '_last_redo', False)))
    assert m.__dict__['_lp_dot_prev'][1] is not None
    # (which evaluations had to be redone is the evaluator's business; the scale changes by 16 and 25 must have been noticed)
    if hasattr(ev, '_last_redo'):
        assert redone[4] and redone[7] and not redone[2] and not redone[3] and not redone[6]


@pytest.mark.parametrize('kind', ['transe', 'distmult', 'complex', 'transh', 'transd'])
def test_evaluator_level_policy_and_identical_ranks(hip, kind):
    """LinkPredictionEvaluator with the one-product level forced on (model.split_level = 1), forced off (0) and on
    'auto' (first evaluation three products, the next ones follow the re-scored pair count): identical rank vectors,
    eager and as hipGraph replays; an evaluation whose one-product list overflows is redone on three products."""
    import torchkge_amd as tk
    import torchkge_amd.evaluation as evm
    n_ent, n_rel, d = 4000, 9, 64
    tables = orc.init_tables(kind, n_ent, n_rel, d, seed=5, d_rel=(48 if kind == 'transd' else None))
    m = build_model(kind, 2, tables, n_ent, n_rel)
    h, t, r = orc.synthetic_triples_zipf(n_ent, n_rel, 30000, 31, hubs=((1500, 'head'), (600, 'tail')))
    kg = tk.KnowledgeGraph(kg={'heads': h, 'tails': t, 'relations': r}, ent2ix={i: i for i in range(n_ent)},
                           rel2ix={i: i for i in range(n_rel)})
    _, kg_test = kg.split_kg(sizes=(28000, 2000))

    def ranks(ev):
        return [ev.rank_true_heads.clone(), ev.rank_true_tails.clone(), ev.filt_rank_true_heads.clone(),
                ev.filt_rank_true_tails.clone()]
    m.split_level = 0
    ev0 = tk.LinkPredictionEvaluator(m, kg_test, graph=False)
    ev0.evaluate(512, verbose=False)
    want = ranks(ev0)
    assert ev0._level == 0 and ev0.last_rescored_per_query >= 1.0
    m.split_level = 1
    for graph in (False, True):
        ev1 = tk.LinkPredictionEvaluator(m, kg_test, graph=graph)
        ev1._level = 1
        for _ in range(3):
            ev1.evaluate(512, verbose=False)
            for a, b in zip(want, ranks(ev1)):
                assert torch.equal(a, b)
        assert ev1.last_rescored_per_query > ev0.last_rescored_per_query     # the wider band re-scores more pairs
    if kind == 'transe':
        # the fused query pipeline on level 1: a second split_count on the same operands recomputes the thresholds from the
        # per-query residuals the pipeline left (thr_ready = 0) -- same counts, equal to the exact kernel's
        hb, tb, rb = kg_test.head_idx[:700].cuda(), kg_test.tail_idx[:700].cuda(), kg_test.relations[:700].cuda()
        m.lp_guard_begin(torch.device('cuda'))
        try:
            with m.lp_session():
                prob = m.lp_problem(hb, tb, rb, 'both')
                assert prob.pre is not None and int(prob.split['level']) == 1
                true = torch.cat([tb, hb])
                prob.pre['true_idx'] = true
                st = prob.pair_scores(true)
                prep = prob.split_prepare()
                raws = []
                for _ in range(2):
                    raw = torch.zeros(prob.B, dtype=torch.int32, device='cuda')
                    prob.split_count(prep, st, raw)
                    prob.split_recheck(prep, st, raw)
                    raws.append(raw)
                prob.split = None
                exact = prob.count_ge(st)
                assert torch.equal(raws[0], exact) and torch.equal(raws[1], exact)
        finally:
            m.lp_guard_end()
    m.split_level = 'auto'
    old = evm.LEVEL1_ENTER, evm.LEVEL1_LEAVE
    try:
        evm.LEVEL1_ENTER, evm.LEVEL1_LEAVE = 1e9, 1e9        # always enter, never leave
        ev2 = tk.LinkPredictionEvaluator(m, kg_test)
        seen = []
        for _ in range(4):
            ev2.evaluate(512, verbose=False)
            seen.append(ev2._level)
            for a, b in zip(want, ranks(ev2)):
                assert torch.equal(a, b)
        assert seen == [1, 1, 1, 1]                          # (the level the NEXT evaluation will use)
        evm.LEVEL1_LEAVE = 0.0                               # ... and leave again at once
        ev2.evaluate(512, verbose=False)
        assert ev2._level == 0
        for a, b in zip(want, ranks(ev2)):
            assert torch.equal(a, b)
    finally:
        evm.LEVEL1_ENTER, evm.LEVEL1_LEAVE = old
        m.split_level = 'auto'


@pytest.mark.parametrize('rows,K', [(1, 4), (6268, 200), (40943, 200), (300, 17), (129, 513), (65536, 400)])
def test_bound_only_row_norms_match_the_chain_to_rounding(hip, rows, K):
    """kge_row_sqnorm_any_order (16 lanes per row, any summation order: bounds / scales of the DOT-mode prefilter) against the
    sequential chain: equal to a few ulps of a K-term sum, maximum folded into the device scalar."""
    g = torch.Generator().manual_seed(rows + K)
    X = (torch.randn(rows, K, generator=g) * 2).cuda()
    m = torch.zeros(2, device='cuda')
    a = hip.row_sqnorm(X, max_io=m[0:1])
    b = hip.row_sqnorm(X, max_io=m[1:2], bound_only=True)
    ref = (X.double() ** 2).sum(1)
    assert float(((b.double() - ref).abs() / ref.clamp_min(1e-30)).max()) < 1e-6
    assert float(((a - b).abs() / a.clamp_min(1e-30)).max()) < 1e-5
    assert abs(float(m[0]) - float(m[1])) <= 1e-5 * float(m[0]) and float(m[1]) == float(b.max())
    Xs = X[:, :K - 1] if K > 1 else X          # a strided view (ld != K): the scalar path
    if K > 1:
        c = hip.row_sqnorm(Xs, K=K - 1, bound_only=True)
        refs = (Xs.double() ** 2).sum(1)
        assert float(((c.double() - refs).abs() / refs.clamp_min(1e-30)).max()) < 1e-6


@pytest.mark.parametrize('N,K', [(14541, 200), (257, 64), (1000, 12), (5, 4), (4097, 400)])
def test_table_prep_l2_equals_the_separate_kernels(hip, N, K):
    """kge_lp_table_prep_l2 (r05: norms + fragment-major hi table + residual maximum in one pass) == kge_row_sqnorm (bit for
    bit: the scores contain en) + kge_lp_hi_rows_frag (byte for byte) + its residual maximum (a bound: equal up to the
    summation order)."""
    g = torch.Generator().manual_seed(N + K)
    E = (torch.nn.functional.normalize(torch.randn(N, K, generator=g), dim=1) * 1.3).cuda()
    guard = torch.zeros(8, device='cuda')
    en_ref = hip.row_sqnorm(E, max_io=guard[1:2])
    Eh_ref, de2_ref = hip.hi_table(E, aug=en_ref, frag=True)
    g2 = torch.zeros(8, device='cuda')
    got = hip.table_prep_l2(E, g2[1:2], g2[7:8])
    assert got is not None
    en, Eh = got
    assert torch.equal(en, en_ref) and float(g2[1]) == float(guard[1])
    assert torch.equal(Eh, Eh_ref)
    assert abs(float(g2[7]) - float(de2_ref)) <= 3e-4 * float(de2_ref) + 1e-30
    assert float(g2[7]) >= float(de2_ref) * (1 - 1e-5)
    # deferred maxima: per-block values instead of the atomics; their maxima are the two scalars
    g3 = torch.zeros(8, device='cuda')
    en3, Eh3, bm = hip.table_prep_l2(E, g3[1:2], g3[7:8], deferred_max=True)
    assert torch.equal(en3, en_ref) and torch.equal(Eh3, Eh_ref) and float(g3[1]) == 0.0 and float(g3[7]) == 0.0
    nb = bm.shape[0] // 2
    assert float(bm[:nb].max()) == float(guard[1]) and float(bm[nb:].max()) == float(g2[7])


def test_fresh_evaluators_share_what_was_learned_about_model_and_graph(hip):
    """The reference idiom builds LinkPredictionEvaluator(model, kg) anew per validation (evaluation.py:252-262): plans,
    level and the captured hipGraph are kept per (model, kg, options) at module level, so the THIRD fresh evaluator
    replays; ranks identical throughout; share_state=False (or other options) keeps a private state; the state dies with
    the model."""
    import gc
    import torchkge_amd as tk
    from torchkge_amd import evaluation as evm
    n_ent, n_rel, d = 4000, 11, 64
    tables = orc.init_tables('transe', n_ent, n_rel, d, seed=5)
    m = build_model('transe', 2, tables, n_ent, n_rel)
    h, t, r = orc.synthetic_triples(n_ent, n_rel, 30000, seed=9)
    kg = tk.KnowledgeGraph(kg={'heads': h, 'tails': t, 'relations': r}, ent2ix={i: i for i in range(n_ent)},
                           rel2ix={i: i for i in range(n_rel)})
    _, kg_test = kg.split_kg(sizes=(28000, 2000))
    ref = None
    states = []
    for i in range(4):
        ev = tk.LinkPredictionEvaluator(m, kg_test)
        ev.evaluate(512, verbose=False)
        ranks = [ev.rank_true_heads, ev.rank_true_tails, ev.filt_rank_true_heads, ev.filt_rank_true_tails]
        if ref is None:
            ref = ranks
        for a, b in zip(ref, ranks):
            assert torch.equal(a, b)
        states.append(ev._st)
        assert ev._n_evaluations == i + 1
    assert all(s_ is states[0] for s_ in states)
    assert states[0]._graph is not None and states[0]._aux_stream is not None      # captured by the second, replayed since
    priv = tk.LinkPredictionEvaluator(m, kg_test, share_state=False)
    assert priv._st is not states[0] and priv._n_evaluations == 0
    other = tk.LinkPredictionEvaluator(m, kg_test, graph=False)                   # other options: another state
    assert other._st is not states[0]
    # a guard vector dirtied outside an evaluation (a guarded section opened by hand) is zeroed again before the replay
    g = m.lp_guard_begin(torch.device('cuda', 0))
    g.fill_(3.0)
    m.lp_guard_end()
    ev = tk.LinkPredictionEvaluator(m, kg_test)
    ev.evaluate(512, verbose=False)
    for a, b in zip(ref, [ev.rank_true_heads, ev.rank_true_tails, ev.filt_rank_true_heads, ev.filt_rank_true_tails]):
        assert torch.equal(a, b)
    n_before = len(evm._STATES)
    del ev, priv, other, states, m
    gc.collect()
    assert len(evm._STATES) < n_before


@pytest.mark.parametrize('kind,B,d_e,d_r', [('transh', 1000, 64, 64), ('transd', 777, 72, 48), ('transh', 5, 200, 200)])
def test_projection_query_side_in_two_launches_equals_the_separate_kernels(hip, kind, B, d_e, d_r):
    """r05: kge_lp_prep_hi (query rows + their planar f16 hi operand + residuals in one launch) and kge_proj_query_stats
    (||q||^2, p, z in one launch) against the launches they replace: kge_lp_prep + kge_lp_hi_rows, and kge_row_sqnorm +
    kge_row_dot (+ the add / stack glue) -- bit for bit (the residuals: a bound, equal up to the summation order)."""
    n_ent, n_rel = 900, 13
    tables = orc.init_tables(kind, n_ent, n_rel, d_e, seed=8, d_rel=(d_r if kind == 'transd' else None))
    m = build_model(kind, 2, tables, n_ent, n_rel)
    g = torch.Generator().manual_seed(B)
    h = torch.randint(0, n_ent, (B,), generator=g).cuda(); t = torch.randint(0, n_ent, (B,), generator=g).cuda()
    r = torch.randint(0, n_rel, (B,), generator=g).cuda()
    tabs = [x.data for x in m._tables()]
    k = hip.TRANSH if kind == 'transh' else hip.TRANSD
    Q0, _, _, Wq = hip.lp_prep(k, hip.SIDE_BOTH, tabs, d_e, d_r, h, t, r, want_w=True)
    Q0b, _, _, _, Qh, dn2 = hip.lp_prep(k, hip.SIDE_BOTH, tabs, d_e, d_r, h, t, r, want_hi=True)
    assert torch.equal(Q0, Q0b)
    Qh_ref, dn2_ref = hip.hi_rows(Q0, is_query=True, want_dn2=True)
    assert torch.equal(Qh, Qh_ref)
    assert torch.allclose(dn2, dn2_ref, rtol=3e-4, atol=0) and bool((dn2 >= dn2_ref * (1 - 1e-5)).all())
    Wt = tabs[2] if kind == 'transh' else tabs[3]
    rb = torch.cat([r, r])
    scale, z_add = (2.0, -2.0) if kind == 'transh' else (-2.0, 0.0)
    gq = torch.zeros(1, device='cuda')
    qn, pz = hip.proj_query_stats(Q0, Wt, rb, scale, z_add, qmax_io=gq)
    gq2 = torch.zeros(1, device='cuda')
    qn_ref = hip.row_sqnorm(Q0, max_io=gq2)
    pz_ref = torch.stack([hip.row_dot(Q0, Wq, scale=scale), hip.row_sqnorm(Wq) + z_add], dim=1)
    assert torch.equal(qn, qn_ref) and torch.equal(pz, pz_ref) and float(gq) == float(gq2)


@pytest.mark.parametrize('kind,B,d,frag', [('distmult', 1000, 64, True), ('complex', 777, 40, True), ('complex', 333, 200, False),
                                           ('distmult', 5, 400, True)])
@pytest.mark.parametrize('side,qpw', [('both', 4), ('both', 16), ('tail', 16), ('head', 4)])
def test_dot_query_side_in_one_launch_equals_the_separate_kernels(hip, kind, B, d, frag, side, qpw, monkeypatch):
    """r05: kge_lp_dot_query_pipeline (DistMult / ComplEx, one-product level: q, exact true scores, planar hi operand with
    PER-QUERY scales, residuals, thresholds, zeroed counters in one launch) against kge_lp_prep + kge_lp_pair_scores (bit
    for bit) and, through kge_lp_split_count + recheck, against the exact fp32 counts -- with the thresholds of the launch
    (thr_ready) and with thresholds recomputed for other true scores (kge_split_args.q_scale_per_query)."""
    monkeypatch.setenv('KGE_DQPIPE_QPW', str(qpw))      # queries per wavefront (the library picks 4 for small batches)
    n_ent, n_rel = 1500, 11
    tables = orc.init_tables(kind, n_ent, n_rel, d, seed=5)
    m = build_model(kind, 2, tables, n_ent, n_rel)
    g = torch.Generator().manual_seed(B + d)
    h = torch.randint(0, n_ent, (B,), generator=g).cuda(); t = torch.randint(0, n_ent, (B,), generator=g).cuda()
    r = torch.randint(0, n_rel, (B,), generator=g).cuda()
    tabs = [hip.f32c(x.data) for x in m._tables()]
    # rows of different magnitude, so that the per-query scales differ (moderately: the band of the one-product level is
    # relative to ||q|| max||e||, and an untrained table with a few huge rows would put every small candidate inside it)
    with torch.no_grad():
        tabs[0][::7] *= 3.0
        tabs[0][1::7] *= 0.2
    cplx = kind == 'complex'
    ent, rel = (tabs[:2], tabs[2:]) if cplx else (tabs[:1], tabs[1:])
    sd = hip.side_code(side)
    k = hip.COMPLEX if cplx else hip.DISTMULT
    Q0, Q1, _, _ = hip.lp_prep(k, sd, tabs, d, d, h, t, r, want_q1=cplx) if cplx else hip.lp_prep(k, sd, tabs, d, d, h, t, r)
    T0, T1 = ent[0], (ent[1] if cplx else None)
    guard = torch.zeros(8, device='cuda')
    hip.row_sqnorm(T0, max_io=guard[1:2], bound_only=True)
    if cplx:
        hip.row_sqnorm(T1, max_io=guard[5:6], bound_only=True)
    nm1 = guard[5:6] if cplx else None
    Eh, de2 = hip.hi_table(T0, X1=T1, dot=True, nmax0=guard[1:2], nmax1=nm1, frag=frag)
    # the candidate side in two launches (norm maxima per block, hi table + residual maxima per block): same table, same
    # scalars -- the residual maximum once the query pipeline has folded the block maxima into its slot
    g2 = torch.zeros(8, device='cuda')
    Eh2, dnb, _ws = hip.dot_table_prep(T0, T1, g2[1:2], g2[5:6] if cplx else None, frag)
    assert torch.equal(Eh2, Eh) and float(g2[1]) == float(guard[1]) and float(g2[5]) == float(guard[5])
    # (the fragment-major table comes from the coalesced kernel, r06: the same exact residuals summed in another order)
    assert float(dnb.max()) == pytest.approx(float(de2), rel=1e-5)
    split = {'Es': Eh2, 'e2pref': None, 'enmax': g2[1:2], 'enmax1': g2[5:6] if cplx else None, 'overflow': guard[2:3], 'level': 1,
             'de2max': g2[7:8], 'list_stat': guard[6:7], 'es_frag': frag}
    pre = hip.lp_dot_query_pipeline(sd, T0, T1, rel[0], rel[1] if cplx else None, h, t, r, g2[1:2], g2[5:6] if cplx else None,
                                    g2[7:8], guard[0:1], guard[2:3], zero_counts=True, dn_bmax=dnb)
    assert float(g2[7]) == float(dnb.max())
    assert torch.equal(pre['Q'], Q0) and (not cplx or torch.equal(pre['Q1'], Q1))
    true = torch.cat([t, h]) if side == 'both' else (t if side == 'tail' else h)
    ref = hip.LpProblem(hip.LP_DOT, Q0, T0, A1=Q1 if cplx else None, T1=T1)
    st_ref = ref.pair_scores(true)
    assert torch.equal(pre['s_true'].view(torch.int32), st_ref.view(torch.int32))
    qn_ref = (Q0.double() ** 2).sum(1) + ((Q1.double() ** 2).sum(1) if cplx else 0)
    assert torch.allclose(pre['qn'].double(), qn_ref, rtol=1e-5, atol=0)
    assert int(pre['counts'].abs().sum()) == 0 and float(guard[0]) == float(pre['qn'].max())
    exact = ref.count_ge(st_ref)
    pre['true_idx'] = true
    prob = hip.LpProblem(hip.LP_DOT, pre['Q'], T0, A1=pre['Q1'], T1=T1)
    prob.split, prob.pre = split, pre
    st = prob.pair_scores(true)
    assert st is pre['s_true']
    got = prob.count_ge(st)
    assert float(guard[2]) == 0.0, 'uncertain-pair list overflowed (%d pairs listed)' % int(prob.last_split[0])
    assert torch.equal(got, exact), '%d of %d counts differ (max %d)' % (
        int((got != exact).sum()), got.numel(), int((got - exact).abs().max()))
    # other thresholds on the same prepared operands: the threshold kernel must use the per-query scales too
    st2 = (st_ref * 0.5).contiguous()
    got2 = prob.count_ge(st2)
    assert float(guard[2]) == 0.0
    assert torch.equal(got2, ref.count_ge(st2))


@pytest.mark.parametrize('B,N,d', [(50, 300, 64), (700, 2100, 200), (1500, 900, 104), (700, 2100, 400), (300, 1200, 500), (2000, 700, 288)])
@pytest.mark.parametrize('waves', [2, 4])
def test_region_recheck_equals_the_global_list(hip, B, N, d, waves, monkeypatch):
    """r05: the free-running sweep leaves its uncertain pairs in REGIONS of the list (one per 32 consecutive queries,
    kge_split_args.region_count) and kge_lp_split_recheck_regions re-scores a region with its query rows resident in LDS --
    the same counts, the same number of listed pairs as the one global list + kge_lp_split_recheck, and the exact counts;
    a second sweep on the same operands (other thresholds) starts from zeroed region counters."""
    monkeypatch.setenv('KGE_RECHECK_REGION_WAVES', str(waves))
    g = torch.Generator().manual_seed(B + d)
    E = torch.nn.functional.normalize(torch.randn(N, d, generator=g), dim=1).cuda()
    R = (0.3 * torch.randn(5, d, generator=g)).cuda()
    h = torch.randint(0, N, (B,), generator=g).cuda(); t = torch.randint(0, N, (B,), generator=g).cuda()
    h[: B // 3] = h[0]                  # a hub: many queries share their row (and their uncertain candidates)
    r = torch.randint(0, 5, (B,), generator=g).cuda()
    true = torch.cat([t, h])
    got, listed = {}, {}
    for regions in (False, True):
        guard = torch.zeros(8, device='cuda')
        en, Ef, tpb = hip.table_prep_l2(E, guard[1:2], guard[7:8], deferred_max=True)
        pre = hip.lp_query_pipeline(hip.SIDE_BOTH, E, R, h, t, r, en, guard[1:2], guard[0:1], level=1, de2max=guard[7:8],
                                    tp_bmax=tpb, zero_counts=True, regions=regions)
        assert (pre.get('region_count') is not None) == regions
        pre['true_idx'] = true
        prob = hip.LpProblem(hip.LP_L2_EXPAND, pre['Q'], E, qn=pre['qn'], en=en)
        prob.split = {'Es': Ef, 'e2pref': None, 'enmax': guard[1:2], 'overflow': guard[2:3], 'level': 1, 'de2max': guard[7:8],
                      'list_stat': guard[6:7], 'es_frag': True}
        prob.pre = pre
        st = prob.pair_scores(true)
        got[regions] = prob.count_ge(st).clone()
        assert float(guard[2]) == 0.0
        listed[regions] = (int(prob.last_split[0]), float(guard[6]))
        if regions:
            assert int(pre['region_count'].sum()) == listed[True][0]
            st2 = (st - 0.05).contiguous()
            ref = hip.LpProblem(hip.LP_L2_EXPAND, pre['Q'], E, qn=pre['qn'], en=en)
            assert torch.equal(prob.count_ge(st2), ref.count_ge(st2))
            assert torch.equal(got[True], ref.count_ge(st))
    assert torch.equal(got[False], got[True])
    assert listed[False] == listed[True] and listed[True][0] > 0
