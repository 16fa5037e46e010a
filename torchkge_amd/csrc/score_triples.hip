// K1: fused embedding gather + on-the-fly L2 normalisation + score, one
// wavefront per triple (gfx950).  HBM-bound: every table row is read exactly
// once (16 B per lane, coalesced within the row), nothing but the fp32 score is
// written.  Replaces the 6-10 ATen kernels of Model.scoring_function:
//   TransE   models/translation.py:69-81      -|| h^ + r - t^ ||_p (p=1) or squared L2
//   TransH   models/translation.py:183-206    -|| p_w(h^) + r - p_w(t^) ||^2
//   TransD   models/translation.py:538-568    -|| (h^.hp^) rp^ + h^[:dr] + r^ - ... ||^2
//   DistMult models/bilinear.py:188-199       sum h^ r t^
//   ComplEx  models/bilinear.py:460-473       Re<h, r, conj t>   (no normalisation)
// (x^ = x / max(||x||_2, 1e-12), torch.nn.functional.normalize).
// The backward kernel recomputes the forward intermediates from the same
// gathered rows and scatters gradients with fp32 atomics.
#include <type_traits>
#include "kge_common.h"

namespace {

constexpr int WAVES_PER_BLOCK = 4;

template <bool VEC4, int NE>
__device__ __forceinline__ void load_row(const float *__restrict__ p, int d, int lane, float (&x)[NE])
{
    if (VEC4) {
#pragma unroll
        for (int c = 0; c < NE / 4; ++c) {
            const int k = (c * 64 + lane) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < d) v = *reinterpret_cast<const float4 *>(p + k);
            x[4 * c] = v.x; x[4 * c + 1] = v.y; x[4 * c + 2] = v.z; x[4 * c + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int k = e * 64 + lane;
            x[e] = (k < d) ? p[k] : 0.f;
        }
    }
}

template <bool VEC4, int NE>
__device__ __forceinline__ int elem_index(int e, int lane)
{
    return VEC4 ? ((e >> 2) * 64 + lane) * 4 + (e & 3) : e * 64 + lane;
}

// Wavefront sum on the DPP cross-lane path of the VALU (quad permutes, row mirrors, row broadcasts;
// the total lands in lane 63 and is read back as a wave-uniform scalar): ~7 dependent full-rate VALU
// ops instead of 6 ds_bpermute round trips through the LDS crossbar (__shfl_xor).  A triple's score is a
// chain of up to six such reductions (norms, projections, the final sum), so this kernel is bound by
// their latency, not by bytes.  (Summation order differs from wave_sum(): this file only.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_dpp(float v)
{
    v += dpp_mov<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141, 0xf>(v);   // row_half_mirror
    v += dpp_mov<0x140, 0xf>(v);   // row_mirror: every lane holds its row's 16-lane sum
    v += dpp_mov<0x142, 0xa>(v);   // row_bcast15 into rows 1 and 3
    v += dpp_mov<0x143, 0xc>(v);   // row_bcast31 into rows 2 and 3: lane 63 holds the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

template <int NE>
__device__ __forceinline__ float sumsq(const float (&x)[NE])
{
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < NE; ++e) s = fmaf(x[e], x[e], s);
    return wave_sum_dpp(s);
}
template <int NE>
__device__ __forceinline__ float dotp(const float (&x)[NE], const float (&y)[NE])
{
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < NE; ++e) s = fmaf(x[e], y[e], s);
    return wave_sum_dpp(s);
}
// F.normalize: x / max(||x||, eps).  EXACT = true keeps the IEEE division per element (the backward, whose two
// reduction modes are compared bit for bit); the forward multiplies by the reciprocal (one division per row,
// <= 1 ulp per element: far inside the 1e-5 score tolerance, ~40 VALU ops per row cheaper).
template <int NE, bool EXACT = true>
__device__ __forceinline__ float normalize_inplace(float (&x)[NE])
{
    const float n = fmaxf(sqrtf(sumsq<NE>(x)), 1e-12f);
    if (EXACT) {
#pragma unroll
        for (int e = 0; e < NE; ++e) x[e] = x[e] / n;
    } else {
        const float rn = 1.0f / n;
#pragma unroll
        for (int e = 0; e < NE; ++e) x[e] = x[e] * rn;
    }
    return n;
}

struct ScoreParams {
    int kind;
    const float *t0, *t1, *t2, *t3;
    int d_ent, d_rel;
    const int64_t *h, *t, *r;
    int64_t B;
    float *out;
};

// Rows a triple gathers, per model kind: table and index of row j (0: h, 1: t, 2: r)
//   TransE / DistMult: E[h] E[t] R[r]            TransH: E[h] E[t] R[r] W[r]
//   ComplEx: Re[h] Im[h] Re[t] Im[t] Rre[r] Rim[r]   TransD: E[h] E[t] Ep[h] Ep[t] R[r] Rp[r]
template <int KIND> struct RowSet;
template <> struct RowSet<KGE_TRANSE_L1> { static constexpr int NR = 3; };
template <> struct RowSet<KGE_TRANSE_L2> { static constexpr int NR = 3; };
template <> struct RowSet<KGE_DISTMULT> { static constexpr int NR = 3; };
template <> struct RowSet<KGE_TRANSH> { static constexpr int NR = 4; };
template <> struct RowSet<KGE_COMPLEX> { static constexpr int NR = 6; };
template <> struct RowSet<KGE_TRANSD> { static constexpr int NR = 6; };

template <int KIND, bool VEC4, int NE>
__device__ __forceinline__ void gather_rows(const ScoreParams &p, int64_t hi, int64_t ti, int64_t ri, int lane,
                                            float (&x)[RowSet<KIND>::NR][NE])
{
    const int de = p.d_ent, dr = p.d_rel;
    if (KIND == KGE_COMPLEX) {
        load_row<VEC4, NE>(p.t0 + hi * de, de, lane, x[0]);
        load_row<VEC4, NE>(p.t1 + hi * de, de, lane, x[1]);
        load_row<VEC4, NE>(p.t0 + ti * de, de, lane, x[2]);
        load_row<VEC4, NE>(p.t1 + ti * de, de, lane, x[3]);
        load_row<VEC4, NE>(p.t2 + ri * dr, dr, lane, x[4]);
        load_row<VEC4, NE>(p.t3 + ri * dr, dr, lane, x[5]);
    } else if (KIND == KGE_TRANSD) {
        load_row<VEC4, NE>(p.t0 + hi * de, de, lane, x[0]);
        load_row<VEC4, NE>(p.t0 + ti * de, de, lane, x[1]);
        load_row<VEC4, NE>(p.t2 + hi * de, de, lane, x[2]);
        load_row<VEC4, NE>(p.t2 + ti * de, de, lane, x[3]);
        load_row<VEC4, NE>(p.t1 + ri * dr, dr, lane, x[4]);
        load_row<VEC4, NE>(p.t3 + ri * dr, dr, lane, x[5]);
    } else {
        load_row<VEC4, NE>(p.t0 + hi * de, de, lane, x[0]);
        load_row<VEC4, NE>(p.t0 + ti * de, de, lane, x[1]);
        load_row<VEC4, NE>(p.t1 + ri * dr, dr, lane, x[2]);
        if (KIND == KGE_TRANSH) load_row<VEC4, NE>(p.t2 + ri * dr, dr, lane, x[RowSet<KIND>::NR - 1]);
    }
}

template <int KIND, bool VEC4, int NE>
__device__ __forceinline__ float score_rows(const ScoreParams &p, int lane, float (&x)[RowSet<KIND>::NR][NE])
{
    if (KIND == KGE_TRANSE_L1 || KIND == KGE_TRANSE_L2) {
        float (&h)[NE] = x[0], (&t)[NE] = x[1], (&r)[NE] = x[2];
        normalize_inplace<NE, false>(h);
        normalize_inplace<NE, false>(t);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const float diff = (h[e] + r[e]) - t[e];
            s = (KIND == KGE_TRANSE_L1) ? s + fabsf(diff) : fmaf(diff, diff, s);
        }
        s = wave_sum_dpp(s);
        if (KIND == KGE_TRANSE_L2) { const float n = sqrtf(s); s = n * n; } // norm(p=2)**2
        return -s;
    } else if (KIND == KGE_DISTMULT) {
        float (&h)[NE] = x[0], (&t)[NE] = x[1], (&r)[NE] = x[2];
        normalize_inplace<NE, false>(h);
        normalize_inplace<NE, false>(t);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < NE; ++e) s += (h[e] * r[e]) * t[e];
        return wave_sum_dpp(s);
    } else if (KIND == KGE_COMPLEX) {
        float (&reh)[NE] = x[0], (&imh)[NE] = x[1], (&ret)[NE] = x[2], (&imt)[NE] = x[3], (&rer)[NE] = x[4], (&imr)[NE] = x[5];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < NE; ++e)
            s += reh[e] * (rer[e] * ret[e] + imr[e] * imt[e]) + imh[e] * (rer[e] * imt[e] - imr[e] * ret[e]);
        return wave_sum_dpp(s);
    } else if (KIND == KGE_TRANSH) {
        float (&h)[NE] = x[0], (&t)[NE] = x[1], (&r)[NE] = x[2], (&w)[NE] = x[3];
        normalize_inplace<NE, false>(h);
        normalize_inplace<NE, false>(t);
        normalize_inplace<NE, false>(w);
        const float hw = dotp<NE>(h, w), tw = dotp<NE>(t, w);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const float ph = h[e] - hw * w[e];
            const float pt = t[e] - tw * w[e];
            const float diff = (ph + r[e]) - pt;
            s = fmaf(diff, diff, s);
        }
        s = wave_sum_dpp(s);
        const float n = sqrtf(s);
        return -(n * n);
    } else { // KGE_TRANSD
        float (&h)[NE] = x[0], (&t)[NE] = x[1], (&hp)[NE] = x[2], (&tp)[NE] = x[3], (&r)[NE] = x[4], (&rp)[NE] = x[5];
        normalize_inplace<NE, false>(h);
        normalize_inplace<NE, false>(t);
        normalize_inplace<NE, false>(hp);
        normalize_inplace<NE, false>(tp);
        normalize_inplace<NE, false>(r);
        normalize_inplace<NE, false>(rp);
        const float sh = dotp<NE>(h, hp), st = dotp<NE>(t, tp);
        const int dr = p.d_rel;
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const bool in = elem_index<VEC4, NE>(e, lane) < dr; // ent[:, :rel_emb_dim]
            const float ph = rp[e] * sh + (in ? h[e] : 0.f);
            const float pt = rp[e] * st + (in ? t[e] : 0.f);
            const float diff = (ph + r[e]) - pt;
            s = fmaf(diff, diff, s);
        }
        s = wave_sum_dpp(s);
        const float n = sqrtf(s);
        return -(n * n);
    }
}

// One wavefront per triple, specialised per model kind, software-pipelined over the wavefront's triples
// (PF): the rows of triple i + nwaves are in flight while triple i is normalised / reduced -- a triple is a
// dependent chain  index load -> row gathers -> up to eight wave reductions, and with a handful of triples
// per wavefront nothing else hides that chain.
template <int KIND, bool VEC4, int NE, bool PF>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64) void score_fwd_kernel(const ScoreParams p)
{
    constexpr int NR = RowSet<KIND>::NR;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * WAVES_PER_BLOCK;
    if (wave >= p.B) return;
    if (!PF) {
        for (int64_t i = wave; i < p.B; i += nwaves) {
            float x[NR][NE];
            gather_rows<KIND, VEC4, NE>(p, p.h[i], p.t[i], p.r[i], lane, x);
            const float score = score_rows<KIND, VEC4, NE>(p, lane, x);
            if (lane == 0) p.out[i] = score;
        }
        return;
    }
    float a[NR][NE], b[NR][NE];
    gather_rows<KIND, VEC4, NE>(p, p.h[wave], p.t[wave], p.r[wave], lane, a);
    for (int64_t i = wave; i < p.B; i += 2 * nwaves) {
        const int64_t i1 = i + nwaves, i2 = i1 + nwaves;
        if (i1 < p.B) gather_rows<KIND, VEC4, NE>(p, p.h[i1], p.t[i1], p.r[i1], lane, b);
        const float sa = score_rows<KIND, VEC4, NE>(p, lane, a);
        if (lane == 0) p.out[i] = sa;
        if (i1 >= p.B) break;
        if (i2 < p.B) gather_rows<KIND, VEC4, NE>(p, p.h[i2], p.t[i2], p.r[i2], lane, a);
        const float sb = score_rows<KIND, VEC4, NE>(p, lane, b);
        if (lane == 0) p.out[i1] = sb;
    }
}

// ---------------------------------------------------------------------------
// backward: d(sum_i go_i * score_i) / d tables, scattered with fp32 atomics.
// With x^ = x/n, n = max(||x||, eps): d/dx = (g - x^ (x^.g)) / n   (n > eps)
// ---------------------------------------------------------------------------
struct BwdParams {
    ScoreParams f;
    const float *go;
    float *g0, *g1, *g2, *g3;
    // row mode (rows != NULL): no atomics -- every triple stores its gradient rows at
    // rows[(stream * B + i) * rows_ld ...]; kge_segment_sum_rows then reduces them by target
    // row in sorted order.  Streams (target table, index):
    //   TransE / DistMult: 0 (g0,h) 1 (g0,t) 2 (g1,r)          TransH: 0 (g0,h) 1 (g0,t) 2 (g1,r) 3 (g2,r)
    //   ComplEx: 0 (g0,h) 1 (g0,t) 2 (g1,h) 3 (g1,t) 4 (g2,r) 5 (g3,r)
    //   TransD:  0 (g0,h) 1 (g0,t) 2 (g2,h) 3 (g2,t) 4 (g1,r) 5 (g3,r)
    float *rows;
    int64_t rows_ld;
};

template <bool VEC4, int NE>
__device__ __forceinline__ void store_row(float *__restrict__ g, int d, int lane, const float (&x)[NE])
{
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int k = elem_index<VEC4, NE>(e, lane);
        if (k < d) g[k] = x[e];
    }
}

template <bool VEC4, int NE>
__device__ __forceinline__ void scatter_row(float *__restrict__ g, int d, int lane, const float (&x)[NE])
{
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int k = elem_index<VEC4, NE>(e, lane);
        if (k < d && x[e] != 0.f) atomicAdd(g + k, x[e]);
    }
}

// gradient wrt the un-normalised row given gradient wrt the normalised row
template <int NE>
__device__ __forceinline__ void normalize_bwd(const float (&xn)[NE], float n, float (&g)[NE])
{
    const float xg = dotp<NE>(xn, g);
    const bool clamped = n <= 1e-12f; // x/eps: plain scaling
#pragma unroll
    for (int e = 0; e < NE; ++e) g[e] = clamped ? g[e] / n : (g[e] - xn[e] * xg) / n;
}

template <bool VEC4, int NE>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64) void score_bwd_kernel(const BwdParams q)
{
    const ScoreParams &p = q.f;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * WAVES_PER_BLOCK;
    const int de = p.d_ent, dr = p.d_rel;

    for (int64_t i = wave; i < p.B; i += nwaves) {
        const int64_t hi = p.h[i], ti = p.t[i], ri = p.r[i];
        const float go = q.go[i];
#define KGE_EMIT(STREAM, G, IDX, D, X)                                                              \
    do {                                                                                            \
        if (q.rows) store_row<VEC4, NE>(q.rows + ((int64_t)(STREAM) * p.B + i) * q.rows_ld, D, lane, X); \
        else scatter_row<VEC4, NE>((G) + (IDX) * (D), D, lane, X);                                  \
    } while (0)
        if (go == 0.f) {
            if (q.rows) {   // this triple's rows still have to exist (as zeros) for the reduction
                const int ns = (p.kind == KGE_TRANSD || p.kind == KGE_COMPLEX) ? 6 : (p.kind == KGE_TRANSH ? 4 : 3);
                for (int st = 0; st < ns; ++st)
                    for (int k = lane; k < de; k += 64) q.rows[((int64_t)st * p.B + i) * q.rows_ld + k] = 0.f;
            }
            continue;
        }
        if (p.kind == KGE_TRANSE_L1 || p.kind == KGE_TRANSE_L2) {
            float h[NE], t[NE], r[NE], gd[NE];
            load_row<VEC4, NE>(p.t0 + hi * de, de, lane, h);
            load_row<VEC4, NE>(p.t0 + ti * de, de, lane, t);
            load_row<VEC4, NE>(p.t1 + ri * dr, dr, lane, r);
            const float nh = normalize_inplace<NE>(h), nt = normalize_inplace<NE>(t);
            // score = -sum f(diff); dscore/ddiff = -sign(diff) (L1) or -2 diff (L2)
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const float diff = (h[e] + r[e]) - t[e];
                const float sg = (diff > 0.f) ? 1.f : ((diff < 0.f) ? -1.f : 0.f);
                gd[e] = go * ((p.kind == KGE_TRANSE_L1) ? -sg : -2.f * diff);
            }
            KGE_EMIT(2, q.g1, ri, dr, gd);       // d/dr = gd
            float gh[NE], gt[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) { gh[e] = gd[e]; gt[e] = -gd[e]; }
            normalize_bwd<NE>(h, nh, gh);
            normalize_bwd<NE>(t, nt, gt);
            KGE_EMIT(0, q.g0, hi, de, gh);
            KGE_EMIT(1, q.g0, ti, de, gt);
        } else if (p.kind == KGE_DISTMULT) {
            float h[NE], t[NE], r[NE], gh[NE], gt[NE], gr[NE];
            load_row<VEC4, NE>(p.t0 + hi * de, de, lane, h);
            load_row<VEC4, NE>(p.t0 + ti * de, de, lane, t);
            load_row<VEC4, NE>(p.t1 + ri * dr, dr, lane, r);
            const float nh = normalize_inplace<NE>(h), nt = normalize_inplace<NE>(t);
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                gh[e] = go * r[e] * t[e];
                gt[e] = go * h[e] * r[e];
                gr[e] = go * h[e] * t[e];
            }
            normalize_bwd<NE>(h, nh, gh);
            normalize_bwd<NE>(t, nt, gt);
            KGE_EMIT(0, q.g0, hi, de, gh);
            KGE_EMIT(1, q.g0, ti, de, gt);
            KGE_EMIT(2, q.g1, ri, dr, gr);
        } else if (p.kind == KGE_COMPLEX) {
            float reh[NE], imh[NE], ret[NE], imt[NE], rer[NE], imr[NE], g[NE];
            load_row<VEC4, NE>(p.t0 + hi * de, de, lane, reh);
            load_row<VEC4, NE>(p.t1 + hi * de, de, lane, imh);
            load_row<VEC4, NE>(p.t0 + ti * de, de, lane, ret);
            load_row<VEC4, NE>(p.t1 + ti * de, de, lane, imt);
            load_row<VEC4, NE>(p.t2 + ri * dr, dr, lane, rer);
            load_row<VEC4, NE>(p.t3 + ri * dr, dr, lane, imr);
            // s = reh(rer ret + imr imt) + imh(rer imt - imr ret)
#pragma unroll
            for (int e = 0; e < NE; ++e) g[e] = go * (rer[e] * ret[e] + imr[e] * imt[e]);
            KGE_EMIT(0, q.g0, hi, de, g); // d/d reh
#pragma unroll
            for (int e = 0; e < NE; ++e) g[e] = go * (rer[e] * imt[e] - imr[e] * ret[e]);
            KGE_EMIT(2, q.g1, hi, de, g); // d/d imh
#pragma unroll
            for (int e = 0; e < NE; ++e) g[e] = go * (reh[e] * rer[e] - imh[e] * imr[e]);
            KGE_EMIT(1, q.g0, ti, de, g); // d/d ret
#pragma unroll
            for (int e = 0; e < NE; ++e) g[e] = go * (reh[e] * imr[e] + imh[e] * rer[e]);
            KGE_EMIT(3, q.g1, ti, de, g); // d/d imt
#pragma unroll
            for (int e = 0; e < NE; ++e) g[e] = go * (reh[e] * ret[e] + imh[e] * imt[e]);
            KGE_EMIT(4, q.g2, ri, dr, g); // d/d rer
#pragma unroll
            for (int e = 0; e < NE; ++e) g[e] = go * (reh[e] * imt[e] - imh[e] * ret[e]);
            KGE_EMIT(5, q.g3, ri, dr, g); // d/d imr
        } else if (p.kind == KGE_TRANSH) {
            float h[NE], t[NE], r[NE], w[NE], gd[NE];
            load_row<VEC4, NE>(p.t0 + hi * de, de, lane, h);
            load_row<VEC4, NE>(p.t0 + ti * de, de, lane, t);
            load_row<VEC4, NE>(p.t1 + ri * dr, dr, lane, r);
            load_row<VEC4, NE>(p.t2 + ri * dr, dr, lane, w);
            const float nh = normalize_inplace<NE>(h), nt = normalize_inplace<NE>(t);
            const float nw = normalize_inplace<NE>(w);
            const float hw = dotp<NE>(h, w), tw = dotp<NE>(t, w);
            // diff = (h - t) - (hw - tw) w + r ; score = -|diff|^2
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const float diff = ((h[e] - hw * w[e]) + r[e]) - (t[e] - tw * w[e]);
                gd[e] = go * (-2.f * diff);
            }
            KGE_EMIT(2, q.g1, ri, dr, gd); // d/dr
            const float gdw = dotp<NE>(gd, w);
            float gh[NE], gt[NE], gw[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                gh[e] = gd[e] - gdw * w[e];         // d/dh^ = gd - (gd.w) w
                gt[e] = -gh[e];
                // d/dw^ = -(hw - tw) gd - (gd.w)(h - t)
                gw[e] = -(hw - tw) * gd[e] - gdw * (h[e] - t[e]);
            }
            normalize_bwd<NE>(h, nh, gh);
            normalize_bwd<NE>(t, nt, gt);
            normalize_bwd<NE>(w, nw, gw);
            KGE_EMIT(0, q.g0, hi, de, gh);
            KGE_EMIT(1, q.g0, ti, de, gt);
            KGE_EMIT(3, q.g2, ri, dr, gw);
        } else { // KGE_TRANSD
            float h[NE], t[NE], hp[NE], tp[NE], r[NE], rp[NE], gd[NE];
            load_row<VEC4, NE>(p.t0 + hi * de, de, lane, h);
            load_row<VEC4, NE>(p.t0 + ti * de, de, lane, t);
            load_row<VEC4, NE>(p.t2 + hi * de, de, lane, hp);
            load_row<VEC4, NE>(p.t2 + ti * de, de, lane, tp);
            load_row<VEC4, NE>(p.t1 + ri * dr, dr, lane, r);
            load_row<VEC4, NE>(p.t3 + ri * dr, dr, lane, rp);
            const float nh = normalize_inplace<NE>(h), nt = normalize_inplace<NE>(t);
            const float nhp = normalize_inplace<NE>(hp), ntp = normalize_inplace<NE>(tp);
            const float nr = normalize_inplace<NE>(r), nrp = normalize_inplace<NE>(rp);
            const float sh = dotp<NE>(h, hp), st = dotp<NE>(t, tp);
            // diff_k = (sh - st) rp_k + [k<dr](h_k - t_k) + r_k   (k < dr; zero beyond)
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const bool in = elem_index<VEC4, NE>(e, lane) < dr;
                const float diff = ((rp[e] * sh + (in ? h[e] : 0.f)) + r[e]) - (rp[e] * st + (in ? t[e] : 0.f));
                gd[e] = in ? go * (-2.f * diff) : 0.f;
            }
            const float gdrp = dotp<NE>(gd, rp);
            float gr[NE], grp[NE], gh[NE], gt[NE], ghp[NE], gtp[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                gr[e] = gd[e];
                grp[e] = (sh - st) * gd[e];
                gh[e] = gd[e] + gdrp * hp[e];   // via [:dr] slice and via sh = h.hp
                gt[e] = -gd[e] - gdrp * tp[e];
                ghp[e] = gdrp * h[e];
                gtp[e] = -gdrp * t[e];
            }
            normalize_bwd<NE>(r, nr, gr);
            normalize_bwd<NE>(rp, nrp, grp);
            normalize_bwd<NE>(h, nh, gh);
            normalize_bwd<NE>(t, nt, gt);
            normalize_bwd<NE>(hp, nhp, ghp);
            normalize_bwd<NE>(tp, ntp, gtp);
            KGE_EMIT(4, q.g1, ri, dr, gr);
            KGE_EMIT(5, q.g3, ri, dr, grp);
            KGE_EMIT(0, q.g0, hi, de, gh);
            KGE_EMIT(1, q.g0, ti, de, gt);
            KGE_EMIT(2, q.g2, hi, de, ghp);
            KGE_EMIT(3, q.g2, ti, de, gtp);
        }
    }
}

#undef KGE_EMIT

// Reduction of per-triple gradient rows by target row, in SORTED key order: a wavefront walks 32
// consecutive sorted entries, sums runs of equal keys in registers and flushes each run with one
// atomic row-add -- a handful of atomics per chunk instead of one per (triple, dimension), and the
// heavily shared relation rows (n_rel << B) stop serialising on the same addresses.
template <int NE>
__global__ __launch_bounds__(256) void segment_sum_kernel(const float *__restrict__ rows, int64_t ld, int d,
                                                          const int64_t *__restrict__ k0, int64_t n0,
                                                          const int64_t *__restrict__ k1, int64_t n1,
                                                          const int64_t *__restrict__ perm,
                                                          float *__restrict__ out, int64_t out_ld)
{
    constexpr int CH = 32, UN = 4;
    const int64_t M = n0 + n1;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t j0 = wave * CH; j0 < M; j0 += nwaves * CH) {
        const int n = (int)min((int64_t)CH, M - j0);
        // the chunk's (row, key) pairs: one coalesced load, then broadcast by shuffle
        const int64_t jl = j0 + min(lane, n - 1);
        const int64_t my_row = perm[jl], my_key = my_row < n0 ? k0[my_row] : k1[my_row - n0];
        float acc[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) acc[e] = 0.f;
        int64_t cur = __shfl(my_key, 0, 64);
        for (int j = 0; j < n; j += UN) {
            float v[UN][NE];
            int64_t kk[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {      // UN independent row loads in flight
                const int ju = min(j + u, n - 1);
                kk[u] = __shfl(my_key, ju, 64);
                const float *row = rows + __shfl(my_row, ju, 64) * ld;
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const int k = lane + 64 * e;
                    v[u][e] = (k < d && j + u < n) ? row[k] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                if (j + u < n && kk[u] != cur) {    // wave-uniform
#pragma unroll
                    for (int e = 0; e < NE; ++e) {
                        const int k = lane + 64 * e;
                        if (k < d && acc[e] != 0.f) atomicAdd(out + cur * out_ld + k, acc[e]);
                        acc[e] = 0.f;
                    }
                    cur = kk[u];
                }
#pragma unroll
                for (int e = 0; e < NE; ++e) acc[e] += v[u][e];
            }
        }
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int k = lane + 64 * e;
            if (k < d && acc[e] != 0.f) atomicAdd(out + cur * out_ld + k, acc[e]);
        }
    }
}

// counting sort of small-integer keys (entity / relation ids): histogram, (host: cumsum), scatter.
// Atomics are aggregated per wavefront: one atomic per DISTINCT key among a wavefront's 64 keys.  Real
// graphs are heavy-tailed (a hub entity / relation owns 10% of a batch) and same-address atomics
// serialise at ~12 ns each past the L2s: 100 us per launch at B = 32768 on a Zipf batch before this.
__device__ __forceinline__ int64_t readlane64(int64_t v, int src)
{
    const unsigned lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), src);
    const unsigned hi = __builtin_amdgcn_readlane((int)((uint64_t)v >> 32), src);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
__global__ void key_hist_kernel(const int64_t *__restrict__ k0, int64_t n0, const int64_t *__restrict__ k1, int64_t n1,
                                int32_t *hist)
{
    const int lane = threadIdx.x & 63;
    const int64_t n = n0 + n1, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t jb = (int64_t)blockIdx.x * blockDim.x + threadIdx.x - lane; jb < n; jb += stride) {   // wave-uniform trip count
        const int64_t j = jb + lane;
        const bool act = j < n;
        const int64_t key = act ? (j < n0 ? k0[j] : k1[j - n0]) : -1;
        unsigned long long todo = __ballot(act);
        while (todo) {
            const int lead = __ffsll((long long)todo) - 1;
            const int64_t kl = readlane64(key, lead);
            const unsigned long long same = __ballot(act && key == kl);
            if (lane == lead) atomicAdd(&hist[kl], (int)__popcll(same));
            todo &= ~same;
        }
    }
}
__global__ void key_scatter_kernel(const int64_t *__restrict__ k0, int64_t n0, const int64_t *__restrict__ k1,
                                   int64_t n1, const int64_t *__restrict__ offsets, int32_t *cursor, int64_t *perm)
{
    const int lane = threadIdx.x & 63;
    const int64_t n = n0 + n1, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t jb = (int64_t)blockIdx.x * blockDim.x + threadIdx.x - lane; jb < n; jb += stride) {
        const int64_t j = jb + lane;
        const bool act = j < n;
        const int64_t key = act ? (j < n0 ? k0[j] : k1[j - n0]) : -1;
        unsigned long long todo = __ballot(act);
        while (todo) {
            const int lead = __ffsll((long long)todo) - 1;
            const int64_t kl = readlane64(key, lead);
            const unsigned long long same = __ballot(act && key == kl);
            int base = 0;
            if (lane == lead) base = atomicAdd(&cursor[kl], (int)__popcll(same));
            base = __builtin_amdgcn_readlane(base, lead);
            if (act && key == kl) perm[offsets[kl] + base + (int)__popcll(same & ((1ull << lane) - 1ull))] = j;
            todo &= ~same;
        }
    }
}

inline int grid_for(int64_t B)
{
    int64_t blocks = (B + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    static const int64_t cap = kge_env_int("KGE_K1_BLOCKS", 256 * 32); // one triple per wavefront up to B = 32768 (measured 7-15 % faster than 4 per wave), grid-stride beyond
    return (int)(blocks < cap ? (blocks > 0 ? blocks : 1) : cap);
}

template <typename P, typename KernelSel>
int dispatch_ne(const P &p, int dmax, bool vec4, int64_t B, hipStream_t s, KernelSel sel)
{
    const int per_lane = vec4 ? ((dmax + 255) / 256) * 4 : (dmax + 63) / 64;
    if (per_lane <= 4) return sel(p, std::integral_constant<int, 4>{}, vec4, B, s);
    if (per_lane <= 8) return sel(p, std::integral_constant<int, 8>{}, vec4, B, s);
    if (per_lane <= 16) return sel(p, std::integral_constant<int, 16>{}, vec4, B, s);
    return KGE_EINVAL; // d > 1024 (vec4) / 1024 (scalar) not supported by the register-resident kernel
}

int check_common(int kind, const float *t0, const float *t1, const float *t2, const float *t3,
                 int d_ent, int d_rel, const int64_t *h, const int64_t *t, const int64_t *r, int64_t B)
{
    if (kind < KGE_TRANSE_L1 || kind > KGE_COMPLEX) return KGE_EINVAL;
    if (!t0 || !t1 || d_ent <= 0 || d_rel <= 0 || B < 0) return KGE_EINVAL;
    if (B > 0 && (!h || !t || !r)) return KGE_EINVAL;
    if ((kind == KGE_TRANSH || kind == KGE_TRANSD || kind == KGE_COMPLEX) && !t2) return KGE_EINVAL;
    if ((kind == KGE_TRANSD || kind == KGE_COMPLEX) && !t3) return KGE_EINVAL;
    if (kind != KGE_TRANSD && d_ent != d_rel) return KGE_EINVAL;
    if (kind == KGE_TRANSD && d_ent < d_rel) return KGE_EINVAL;
    return 0;
}

bool tables_vec4(const float *t0, const float *t1, const float *t2, const float *t3, int d_ent, int d_rel)
{
    bool v = (d_ent % 4 == 0) && (d_rel % 4 == 0) && kge_aligned16(t0) && kge_aligned16(t1);
    if (t2) v = v && kge_aligned16(t2);
    if (t3) v = v && kge_aligned16(t3);
    return v;
}

} // namespace

extern "C" int kge_score_triples(int kind, const float *t0, const float *t1, const float *t2,
                                 const float *t3, int d_ent, int d_rel, const int64_t *h,
                                 const int64_t *t, const int64_t *r, int64_t B, float *out,
                                 kge_stream_t stream)
{
    int rc = check_common(kind, t0, t1, t2, t3, d_ent, d_rel, h, t, r, B);
    if (rc) return rc;
    if (B == 0) return 0;
    if (!out) return KGE_EINVAL;
    ScoreParams p{kind, t0, t1, t2, t3, d_ent, d_rel, h, t, r, B, out};
    const bool vec4 = tables_vec4(t0, t1, t2, t3, d_ent, d_rel);
    auto sel = [](const ScoreParams &pp, auto ne, bool v4, int64_t b, hipStream_t s) -> int {
        constexpr int NE = decltype(ne)::value;
        constexpr bool PF = NE <= 8;    // two row sets in registers (6 rows x 16 floats x 2 would not fit)
        const dim3 grid(grid_for(b)), block(WAVES_PER_BLOCK * 64);
#define KGE_FWD_CASE(K)                                                                                   \
    case K:                                                                                               \
        if (v4) hipLaunchKernelGGL((score_fwd_kernel<K, true, NE, PF>), grid, block, 0, s, pp);           \
        else hipLaunchKernelGGL((score_fwd_kernel<K, false, NE, PF>), grid, block, 0, s, pp);             \
        break;
        switch (pp.kind) {
            KGE_FWD_CASE(KGE_TRANSE_L1)
            KGE_FWD_CASE(KGE_TRANSE_L2)
            KGE_FWD_CASE(KGE_TRANSH)
            KGE_FWD_CASE(KGE_TRANSD)
            KGE_FWD_CASE(KGE_DISTMULT)
            KGE_FWD_CASE(KGE_COMPLEX)
        default: return KGE_EINVAL;
        }
#undef KGE_FWD_CASE
        KGE_CHECK_LAUNCH();
        return 0;
    };
    return dispatch_ne(p, d_ent, vec4, B, kge_s(stream), sel);
}

extern "C" int kge_score_triples_bwd(int kind, const float *t0, const float *t1, const float *t2,
                                     const float *t3, int d_ent, int d_rel, const int64_t *h,
                                     const int64_t *t, const int64_t *r, int64_t B, const float *go,
                                     float *g0, float *g1, float *g2, float *g3, float *rows, int64_t rows_ld,
                                     kge_stream_t stream)
{
    int rc = check_common(kind, t0, t1, t2, t3, d_ent, d_rel, h, t, r, B);
    if (rc) return rc;
    if (B == 0) return 0;
    if (!go || (!rows && (!g0 || !g1))) return KGE_EINVAL;
    if (rows && rows_ld < d_ent) return KGE_EINVAL;
    if (!rows && (kind == KGE_TRANSH || kind == KGE_TRANSD || kind == KGE_COMPLEX) && !g2) return KGE_EINVAL;
    if (!rows && (kind == KGE_TRANSD || kind == KGE_COMPLEX) && !g3) return KGE_EINVAL;
    BwdParams q{{kind, t0, t1, t2, t3, d_ent, d_rel, h, t, r, B, nullptr}, go, g0, g1, g2, g3, rows, rows_ld};
    const bool vec4 = tables_vec4(t0, t1, t2, t3, d_ent, d_rel);
    auto sel = [](const BwdParams &qq, auto ne, bool v4, int64_t b, hipStream_t s) -> int {
        constexpr int NE = decltype(ne)::value;
        if (v4) hipLaunchKernelGGL((score_bwd_kernel<true, NE>), dim3(grid_for(b)), dim3(WAVES_PER_BLOCK * 64), 0, s, qq);
        else hipLaunchKernelGGL((score_bwd_kernel<false, NE>), dim3(grid_for(b)), dim3(WAVES_PER_BLOCK * 64), 0, s, qq);
        KGE_CHECK_LAUNCH();
        return 0;
    };
    return dispatch_ne(q, d_ent, vec4, B, kge_s(stream), sel);
}

extern "C" int kge_segment_sum_rows(const float *rows, int64_t ld, int d, const int64_t *k0, int64_t n0,
                                    const int64_t *k1, int64_t n1, const int64_t *perm, float *out, int64_t out_ld,
                                    kge_stream_t stream)
{
    const int64_t M = n0 + n1;
    if (n0 < 0 || n1 < 0 || d <= 0 || d > 1024 || ld < d || out_ld < d) return KGE_EINVAL;
    if (M == 0) return 0;
    if (!rows || (n0 > 0 && !k0) || (n1 > 0 && !k1) || !perm || !out) return KGE_EINVAL;
    const int64_t chunks = (M + 31) / 32;
    const int grid = (int)((chunks + 3) / 4 < 256 * 8 ? (chunks + 3) / 4 : 256 * 8);
    hipStream_t s = kge_s(stream);
    if (d <= 256) hipLaunchKernelGGL(segment_sum_kernel<4>, dim3(grid), dim3(256), 0, s, rows, ld, d, k0, n0, k1, n1, perm, out, out_ld);
    else if (d <= 512) hipLaunchKernelGGL(segment_sum_kernel<8>, dim3(grid), dim3(256), 0, s, rows, ld, d, k0, n0, k1, n1, perm, out, out_ld);
    else hipLaunchKernelGGL(segment_sum_kernel<16>, dim3(grid), dim3(256), 0, s, rows, ld, d, k0, n0, k1, n1, perm, out, out_ld);
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_key_hist(const int64_t *k0, int64_t n0, const int64_t *k1, int64_t n1, int32_t *hist,
                            kge_stream_t stream)
{
    if (n0 < 0 || n1 < 0 || !hist || (n0 > 0 && !k0) || (n1 > 0 && !k1)) return KGE_EINVAL;
    if (n0 + n1 == 0) return 0;
    const int64_t n = n0 + n1;
    hipLaunchKernelGGL(key_hist_kernel, dim3((int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0,
                       kge_s(stream), k0, n0, k1, n1, hist);
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_key_scatter(const int64_t *k0, int64_t n0, const int64_t *k1, int64_t n1, const int64_t *offsets,
                               int32_t *cursor, int64_t *perm, kge_stream_t stream)
{
    if (n0 < 0 || n1 < 0 || !offsets || !cursor || !perm || (n0 > 0 && !k0) || (n1 > 0 && !k1)) return KGE_EINVAL;
    if (n0 + n1 == 0) return 0;
    const int64_t n = n0 + n1;
    hipLaunchKernelGGL(key_scatter_kernel, dim3((int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0,
                       kge_s(stream), k0, n0, k1, n1, offsets, cursor, perm);
    KGE_CHECK_LAUNCH();
    return 0;
}
