import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchkge_amd import _hip as hip
hip.load_library()
def run(B, N, K, R, same_rel=False, sorted_r=False, zero_x=False, tag=''):
    g = torch.Generator().manual_seed(B + N + K)
    T = torch.nn.functional.normalize(torch.randn(N, K, generator=g), dim=1)
    W = torch.nn.functional.normalize(torch.randn(R, K, generator=g), dim=1)
    r_idx = torch.randint(0, R, (B,), generator=g)
    if same_rel:
        r_idx[:] = 3
    if sorted_r:
        r_idx = r_idx.sort().values
    A = T[torch.randint(0, N, (B,), generator=g)] + 0.7 * torch.nn.functional.normalize(torch.randn(B, K, generator=g), dim=1)
    dT, dA, dW = T.cuda(), A.cuda().contiguous(), W.cuda()
    Np = hip.padded_cols(N)
    Xb = torch.zeros(R, Np, device='cuda')
    X = hip.LpProblem(hip.LP_DOT, dW, dT).scores(Xb[:, :N])
    if zero_x:
        Xb.zero_()
    Wq = dW[r_idx.cuda()]
    pz = torch.stack([hip.row_dot(dA, Wq, scale=2.0), hip.row_sqnorm(Wq) - 2.0], 1).contiguous()
    guard = torch.zeros(8, device='cuda')
    en = hip.row_sqnorm(dT, max_io=guard[1:2]); qn = hip.row_sqnorm(dA, max_io=guard[0:1])
    prob = hip.LpProblem(hip.LP_L2_PROJH, dA, dT, qn=qn, en=en, Wq=pz, scal=X, r_idx=r_idx.cuda())
    t = torch.randint(0, N, (B,), generator=g).cuda()
    st = prob.pair_scores(t)
    exact = prob.count_ge(st)
    hip.absmax(X, guard[3:4])
    out = []
    for frag in (False, True):
        Eh, de2 = hip.hi_table(dT, aug=en, frag=frag)
        prob.split = {'Es': Eh, 'e2pref': None, 'enmax': guard[1:2], 'overflow': guard[2:3], 'xabsmax': guard[3:4],
                      'yabsmax': None, 'level': 1, 'de2max': de2, 'es_frag': frag}
        got = prob.count_ge(st)
        out.append(int((got != exact).sum()))
    print('%-40s B=%d N=%d K=%d R=%d: mismatching queries old %d, stream %d' % (tag, B, N, K, R, out[0], out[1]))
run(4096, 14541, 200, 237, tag='random relations')
run(40932, 14541, 200, 237, same_rel=True, tag='one relation, full size')
run(20000, 3000, 200, 37, tag='many queries, small N')
