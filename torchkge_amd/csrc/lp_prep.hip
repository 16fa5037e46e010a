// Query-side preparation for all-candidates link prediction (gfx950):
// the device equivalent of Model.inference_prepare_candidates
//   TransE   models/translation.py:105-125       DistMult models/bilinear.py:247-267
//   TransH   models/translation.py:234-258       ComplEx  models/bilinear.py:530-556
//   TransD   models/translation.py:603-627
// plus the per-entity reductions that stand in for evaluate_projections
// (translation.py:260-284, :629-652): instead of caching P[r,e,:] (R x N x d
// floats) the engine keeps one scalar per (entity[,relation]).
// One wavefront per row; all HBM-bound and tiny next to the scoring kernels.
#include "kge_common.h"

namespace {

constexpr int WPB = 4;

inline int grid_rows(int64_t rows)
{
    int64_t b = (rows + WPB - 1) / WPB;
    const int64_t cap = 256 * 8;
    return (int)(b < cap ? (b > 0 ? b : 1) : cap);
}
inline int grid_threads(int64_t n, int bs)
{
    int64_t b = (n + bs - 1) / bs;
    const int64_t cap = 256 * 8;
    return (int)(b < cap ? (b > 0 ? b : 1) : cap);
}

struct PrepParams {
    int kind, side;
    const float *t0, *t1, *t2, *t3;
    int d_ent, d_rel;
    const int64_t *h, *t, *r;
    int64_t B;
    float *Q0, *Q1, *Wq;
    int64_t ent_lo, ent_n;  // row-sharded entity tables: this rank holds rows [ent_lo, ent_lo + ent_n); ent_n < 0: whole tables
    // optional (TransH / TransD, one-product level of the split prefilter, r05): the PLANAR f16 hi operand of the query rows
    // ([Bp][hi_units_p][32 B], scale 2^12, two augmentation columns 1, 1 -- what kge_lp_hi_rows(is_query, aug_mode 2) builds
    // from Q0 in a launch of its own) and ||q - hi(q)||^2 per query; rows [nq, Bp) are written as zeros
    _Float16 *Qh;
    int hi_units_p;
    int64_t Bp;
    float *q_dn2;
    int inflight;           // TransH / TransD rows of <= 256 columns: all loads of a query at once (KGE_PREP_INFLIGHT=0: the loops)
};

__global__ __launch_bounds__(WPB * 64) void lp_prep_kernel(const PrepParams p)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * WPB;
    const int de = p.d_ent, dr = p.d_rel;
    const bool both = p.side == KGE_SIDE_BOTH;   // 2B queries: [0,B) tail side, [B,2B) head side
    const bool proj = p.side == KGE_SIDE_PROJ_H || p.side == KGE_SIDE_PROJ_T; // projection only, no relation term
    const int64_t nq = both ? 2 * p.B : p.B;
    const int hk = p.hi_units_p * 16;                       // f16 elements per row of Qh
    const float hscale = 4096.0f;                           // (the L2 modes' fixed operand scale, SPLIT_SCALE_LOG2 = 12)
    float dn_acc = 0.f;
    // one hi element: k < dr data, k == dr / dr + 1 the augmentation columns (1, 1), zeros behind
    auto emit_hi = [&](_Float16 *qh, int k, float val) __attribute__((always_inline)) {
        const float xs = val * hscale;
        const _Float16 hh = (_Float16)xs;
        const float dd = xs - (float)hh;
        dn_acc = fmaf(dd, dd, dn_acc);
        qh[k] = hh;
    };
    const int64_t n_rows = p.Qh ? p.Bp : nq;
    for (int64_t i = wave; i < n_rows; i += nwaves) {
        _Float16 *qh = p.Qh ? p.Qh + i * hk : nullptr;
        if (i >= nq) {      // padding rows of the hi operand
            for (int k = lane; k < hk; k += 64) qh[k] = (_Float16)0.f;
            continue;
        }
        if (qh) {
            for (int k = dr + lane; k < hk; k += 64) qh[k] = (_Float16)((k == dr || k == dr + 1) ? hscale : 0.f);
            dn_acc = 0.f;
        }
        const bool tail = both ? i < p.B : p.side == KGE_SIDE_TAIL;
        const bool use_h = tail || p.side == KGE_SIDE_PROJ_H;
        const int64_t f = (both && i >= p.B) ? i - p.B : i;   // the fact this query belongs to
        int64_t ei = use_h ? p.h[f] : p.t[f]; // the entity that stays in the query
        const int64_t ri = p.r[f];
        float *q0 = p.Q0 + i * dr;
        if (p.ent_n >= 0) {
            // row-sharded tables (one shard per GPU): only the OWNER of entity ei can build this query row; every
            // other rank writes zeros, and the sum over ranks (x + 0 is exact) hands the row to all of them.
            // Relation-side outputs (Wq) come from replicated tables and are complete on every rank.
            const bool owned = ei >= p.ent_lo && ei < p.ent_lo + p.ent_n;
            ei -= p.ent_lo;
            if (!owned) {
                for (int k = lane; k < dr; k += 64) q0[k] = 0.f;
                if (p.kind == KGE_COMPLEX) {
                    float *q1 = p.Q1 + i * dr;
                    for (int k = lane; k < dr; k += 64) q1[k] = 0.f;
                } else if (p.kind == KGE_TRANSH || p.kind == KGE_TRANSD) {
                    const float *w = (p.kind == KGE_TRANSH ? p.t2 : p.t3) + ri * dr;
                    if (p.Wq) {
                        float *wq = p.Wq + i * dr;
                        for (int k = lane; k < dr; k += 64) wq[k] = w[k];
                    }
                }
                continue;
            }
        }
        switch (p.kind) {
        case KGE_TRANSE_L1:
        case KGE_TRANSE_L2: {
            const float *e = p.t0 + ei * de, *r = p.t1 + ri * dr;
            for (int k = lane; k < dr; k += 64) q0[k] = proj ? e[k] : (tail ? e[k] + r[k] : e[k] - r[k]);
            break;
        }
        case KGE_DISTMULT: {
            const float *e = p.t0 + ei * de, *r = p.t1 + ri * dr;
            for (int k = lane; k < dr; k += 64) q0[k] = proj ? e[k] : (tail ? e[k] * r[k] : r[k] * e[k]);
            break;
        }
        case KGE_COMPLEX: {
            const float *re = p.t0 + ei * de, *im = p.t1 + ei * de;
            const float *rr = p.t2 + ri * dr, *ir = p.t3 + ri * dr;
            float *q1 = p.Q1 + i * dr;
            for (int k = lane; k < dr; k += 64) {
                if (proj) {
                    q0[k] = re[k];
                    q1[k] = im[k];
                } else if (tail) { // bilinear.py:514-515
                    q0[k] = re[k] * rr[k] - im[k] * ir[k];
                    q1[k] = re[k] * ir[k] + im[k] * rr[k];
                } else {    // bilinear.py:521-522 (re = re_t, im = im_t)
                    q0[k] = rr[k] * re[k] + ir[k] * im[k];
                    q1[k] = rr[k] * im[k] - ir[k] * re[k];
                }
            }
            break;
        }
        case KGE_TRANSH: {
            const float *e = p.t0 + ei * de, *r = p.t1 + ri * dr, *w = p.t2 + ri * dr;
            float *wq = p.Wq ? p.Wq + i * dr : nullptr;
            if (dr <= 256 && p.inflight) {
                // (r06) rows of up to 256 columns: ALL loads of the query in flight at once -- the kernel is a chain of load
                // latencies per query (dot -> reduction -> second pass: 49-59 us for 40.9 k queries); same operations in
                // the same order as the loops below: same bits
                float ev[4], wv[4], rv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = lane + 64 * j;
                    const bool in = k < dr;
                    ev[j] = in ? e[k] : 0.f; wv[j] = in ? w[k] : 0.f; rv[j] = in ? r[k] : 0.f;
                }
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (lane + 64 * j < dr) a = fmaf(ev[j], wv[j], a);
                a = wave_sum(a);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = lane + 64 * j;
                    if (k < dr) {
                        const float pe = ev[j] - a * wv[j]; // translation.py:281
                        const float val = proj ? pe : (tail ? pe + rv[j] : pe - rv[j]);
                        q0[k] = val;
                        if (wq) wq[k] = wv[j];
                        if (qh) emit_hi(qh, k, val);
                    }
                }
                break;
            }
            float a = 0.f;
            for (int k = lane; k < dr; k += 64) a = fmaf(e[k], w[k], a);
            a = wave_sum(a);
            for (int k = lane; k < dr; k += 64) {
                const float pe = e[k] - a * w[k]; // translation.py:281
                const float val = proj ? pe : (tail ? pe + r[k] : pe - r[k]);
                q0[k] = val;
                if (wq) wq[k] = w[k];
                if (qh) emit_hi(qh, k, val);
            }
            break;
        }
        case KGE_TRANSD: {
            const float *e = p.t0 + ei * de, *ep = p.t2 + ei * de;
            const float *r = p.t1 + ri * dr, *rp = p.t3 + ri * dr;
            float *wq = p.Wq ? p.Wq + i * dr : nullptr;
            if (de <= 256 && dr <= de && p.inflight) {
                // (r06: as TransH -- every load of the query in flight at once, same operations in the same order)
                float ev[4], pv[4], rv[4], rpv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = lane + 64 * j;
                    ev[j] = k < de ? e[k] : 0.f; pv[j] = k < de ? ep[k] : 0.f;
                    rv[j] = k < dr ? r[k] : 0.f; rpv[j] = k < dr ? rp[k] : 0.f;
                }
                float sc = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (lane + 64 * j < de) sc = fmaf(pv[j], ev[j], sc);
                sc = wave_sum(sc);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = lane + 64 * j;
                    if (k < dr) {
                        const float pe = sc * rpv[j] + ev[j]; // translation.py:646
                        const float val = proj ? pe : (tail ? pe + rv[j] : pe - rv[j]);
                        q0[k] = val;
                        if (wq) wq[k] = rpv[j];
                        if (qh) emit_hi(qh, k, val);
                    }
                }
                break;
            }
            float sc = 0.f;
            for (int k = lane; k < de; k += 64) sc = fmaf(ep[k], e[k], sc);
            sc = wave_sum(sc);
            for (int k = lane; k < dr; k += 64) {
                const float pe = sc * rp[k] + e[k]; // translation.py:646
                const float val = proj ? pe : (tail ? pe + r[k] : pe - r[k]);
                q0[k] = val;
                if (wq) wq[k] = rp[k];
                if (qh) emit_hi(qh, k, val);
            }
            break;
        }
        }
        if (qh && p.q_dn2) {    // ||q - hi(q)||^2, unscaled (a bound of the error band: any summation order, 1.0001 for it)
            const float dn = wave_sum(dn_acc) * (1.0f / (hscale * hscale)) * 1.0001f;
            if (lane == 0) p.q_dn2[i] = dn;
        }
    }
}

// Relation candidates of the projection models (relation prediction, `entities=False`):
//   out[i, rho] = -|| p_rho(h_i) + R[rho] - p_rho(t_i) ||^2   for EVERY relation rho
//   TransH  p_rho(e) = E[e] - (E[e].W[rho]) W[rho]                    translation.py:252-256, :270-282
//   TransD  p_rho(e) = (Ep[e].E[e]) Rp[rho] + E[e, :d_r]             translation.py:621-626, :639-650
// scored as interfaces.py:261-272 does (-dissimilarity(proj_h + r, proj_t)).  The reference gathers
// two (b, n_rel, d) slices of its (n_rel, n_ent, d) projection cache; here one wavefront per
// (fact, relation) pair forms both projections in registers: the rank-1 form, no cache.
// Block = 4 waves = 4 consecutive relations of one fact (h, t rows shared through the L1/L2).
__global__ __launch_bounds__(WPB * 64) void rel_proj_scores_kernel(int kind, const float *__restrict__ E,
                                                                   const float *__restrict__ R,
                                                                   const float *__restrict__ Wt, /* W or Rp */
                                                                   const float *__restrict__ Ep, int de, int dr,
                                                                   const int64_t *__restrict__ h,
                                                                   const int64_t *__restrict__ t, int64_t B,
                                                                   int64_t n_rel, float *out, int64_t ldo)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * WPB, total = B * n_rel;
    for (int64_t pi = wave; pi < total; pi += nwaves) {
        const int64_t i = pi / n_rel, rho = pi - i * n_rel;
        const float *eh = E + h[i] * de, *et = E + t[i] * de;
        const float *r = R + rho * dr, *w = Wt + rho * dr;
        float ah = 0.f, at = 0.f;
        if (kind == KGE_TRANSH) {
            for (int k = lane; k < dr; k += 64) { ah = fmaf(eh[k], w[k], ah); at = fmaf(et[k], w[k], at); }
        } else {
            const float *ph = Ep + h[i] * de, *pt = Ep + t[i] * de;
            for (int k = lane; k < de; k += 64) { ah = fmaf(ph[k], eh[k], ah); at = fmaf(pt[k], et[k], at); }
        }
        ah = wave_sum(ah);
        at = wave_sum(at);
        float acc = 0.f;
        for (int k = lane; k < dr; k += 64) {
            float pjh, pjt;
            if (kind == KGE_TRANSH) { pjh = eh[k] - ah * w[k]; pjt = et[k] - at * w[k]; }
            else { pjh = ah * w[k] + eh[k]; pjt = at * w[k] + et[k]; }
            const float diff = (pjh + r[k]) - pjt;
            acc = fmaf(diff, diff, acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) out[i * ldo + rho] = -acc;
    }
}

// serial single-accumulator chains (one thread per row): these feed
// L2_EXPAND's qn / en and must match oracle orc_row_sqnorm_chain bit for bit.
__global__ void row_sqnorm_kernel(const float *__restrict__ X, int64_t ld, int64_t rows, int K, float *out,
                                  float *max_io)
{
    float big = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (int64_t)gridDim.x * blockDim.x) {
        const float *x = X + i * ld;
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(x[k], x[k], acc);
        out[i] = acc;
        // squared norms are >= 0 (or NaN, whose bit pattern is above +inf): their
        // order as floats is their order as unsigned bit patterns
        big = __uint_as_float(max(__float_as_uint(big), __float_as_uint(acc)));
    }
    if (max_io) {
        unsigned m = __float_as_uint(big);
        for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
        if ((threadIdx.x & 63) == 0) kge_atomic_max_u32(reinterpret_cast<unsigned *>(max_io), m);
    }
}
// The same chains with the rows fetched cooperatively: a wavefront owns RPW rows (64, or 16 for tables of a few
// ten thousand rows: the chains are latency bound, 16 rows per wavefront put four times as many wavefronts in flight --
// 14,541 rows are 228 wavefronts of 64, less than one per CU), loads them KGE_PS_KC k at a time as 16-byte pieces (all
// loads of a chunk in flight) and hands lane r its row through LDS (conflict-free b128).  Same chain, same bits.
template <int RPW, int NW>      // NW independent wavefronts per block, each on its own LDS slice: they share the final atomic
__global__ __launch_bounds__(64 * NW) void row_sqnorm_staged_kernel(const float *__restrict__ X, int64_t ld, int64_t rows,
                                                                    int K, float *out, float *max_io)
{
    __shared__ __attribute__((aligned(16))) float xs_all[NW * RPW * KGE_PS_LD];
    __shared__ unsigned wmax[NW];
    constexpr int NP = KGE_PS_KC / 4, ITS = RPW * NP / 64;      // pieces per row chunk; load passes per full chunk
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float *xs = xs_all + wv * RPW * KGE_PS_LD;
    float big = 0.f;
    const int64_t ngroups = (rows + RPW - 1) / RPW;
    for (int64_t grp = (int64_t)blockIdx.x * NW + wv; grp < ngroups; grp += (int64_t)gridDim.x * NW) {
        const int64_t row0 = grp * RPW;
        float acc = 0.f;
        for (int k0 = 0; k0 < K; k0 += KGE_PS_KC) {   // K % 4 == 0, ld % 4 == 0, X 16-byte aligned (checked by the host)
            const int kc = min(KGE_PS_KC, K - k0);
            const int pieces = kc >> 2;
            if (kc == KGE_PS_KC) {
                float4 v[ITS];
#pragma unroll
                for (int it = 0; it < ITS; ++it) {
                    const int idx = lane + 64 * it, rr = idx / NP, pc = idx % NP;
                    const int64_t r = min(row0 + rr, rows - 1);
                    v[it] = *reinterpret_cast<const float4 *>(X + r * ld + k0 + pc * 4);
                }
#pragma unroll
                for (int it = 0; it < ITS; ++it) {
                    const int idx = lane + 64 * it, rr = idx / NP, pc = idx % NP;
                    *reinterpret_cast<float4 *>(xs + rr * KGE_PS_LD + pc * 4) = v[it];
                }
            } else {
                for (int idx = lane; idx < RPW * pieces; idx += 64) {
                    const int rr = idx / pieces, pc = idx - rr * pieces;
                    const int64_t r = min(row0 + rr, rows - 1);
                    *reinterpret_cast<float4 *>(xs + rr * KGE_PS_LD + pc * 4) =
                        *reinterpret_cast<const float4 *>(X + r * ld + k0 + pc * 4);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (lane < RPW) {
                const float *x = xs + lane * KGE_PS_LD;
                for (int k = 0; k < kc; k += 4) {
                    const float4 t = *reinterpret_cast<const float4 *>(x + k);
                    acc = fmaf(t.x, t.x, acc);
                    acc = fmaf(t.y, t.y, acc);
                    acc = fmaf(t.z, t.z, acc);
                    acc = fmaf(t.w, t.w, acc);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
        if (lane < RPW && row0 + lane < rows) {
            out[row0 + lane] = acc;
            big = __uint_as_float(max(__float_as_uint(big), __float_as_uint(acc)));
        }
    }
    if (max_io) {       // one atomic per block (same-address atomics serialise in the L2)
        unsigned m = __float_as_uint(big);
        for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
        if (NW == 1) {
            if (lane == 0) kge_atomic_max_u32(reinterpret_cast<unsigned *>(max_io), m);
        } else {
            if (lane == 0) wmax[wv] = m;
            __syncthreads();
            if (threadIdx.x == 0) {
                unsigned mm = wmax[0];
#pragma unroll
                for (int w = 1; w < NW; ++w) mm = max(mm, wmax[w]);
                kge_atomic_max_u32(reinterpret_cast<unsigned *>(max_io), mm);
            }
        }
    }
}
// Squared row norms in ANY summation order (bounds and scales only -- the DOT modes of the split prefilter; no score
// contains them, so the sequential chain of the reference is not needed): 16 lanes per row, float4 loads, four rows of a
// lane group in flight, a shuffle tree; the maximum leaves the block as ONE atomic (same-address atomics serialise in
// the L2: one per wavefront of a 40 k-row table cost more than the sweep itself).
__global__ __launch_bounds__(256) void row_sqnorm_any_kernel(const float *__restrict__ X, int64_t ld, int64_t rows, int K,
                                                             float *out, float *max_io)
{
    __shared__ unsigned wmax[4];
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const bool vec = (K % 4 == 0) && (ld % 4 == 0) && ((size_t)X & 15) == 0;
    float big = 0.f;
    for (int64_t r0 = (int64_t)blockIdx.x * 64; r0 < rows; r0 += (int64_t)gridDim.x * 64) {    // 64 rows per block and step
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (vec) {
            for (int k = sub * 4; k < K; k += 64) {
                float4 t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t r = min(r0 + grp * 4 + j, rows - 1);
                    t[j] = *reinterpret_cast<const float4 *>(X + r * ld + k);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j] = fmaf(t[j].x, t[j].x, acc[j]); acc[j] = fmaf(t[j].y, t[j].y, acc[j]);
                    acc[j] = fmaf(t[j].z, t[j].z, acc[j]); acc[j] = fmaf(t[j].w, t[j].w, acc[j]);
                }
            }
        } else {
            for (int k = sub; k < K; k += 16)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x = X[min(r0 + grp * 4 + j, rows - 1) * ld + k];
                    acc[j] = fmaf(x, x, acc[j]);
                }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = acc[j];
            a += __shfl_xor(a, 8, 64); a += __shfl_xor(a, 4, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 1, 64);
            const int64_t r = r0 + grp * 4 + j;
            if (r < rows) {
                if (sub == 0) out[r] = a;
                big = __uint_as_float(max(__float_as_uint(big), __float_as_uint(a)));
            }
        }
    }
    if (max_io) {
        unsigned m = __float_as_uint(big);
        for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0)
            kge_atomic_max_u32(reinterpret_cast<unsigned *>(max_io), max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])));
    }
}

__global__ void row_dot_kernel(const float *__restrict__ X, const float *__restrict__ Y, int64_t ld,
                               int64_t rows, int K, float scale, float *out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (int64_t)gridDim.x * blockDim.x) {
        const float *x = X + i * ld, *y = Y + i * ld;
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(x[k], y[k], acc);
        out[i] = scale * acc;
    }
}

// the same chain with both rows staged cooperatively (see row_sqnorm_staged_kernel)
template <int RPW>
__global__ __launch_bounds__(64) void row_dot_staged_kernel(const float *__restrict__ X, const float *__restrict__ Y,
                                                            int64_t ld, int64_t rows, int K, float scale, float *out)
{
    __shared__ __attribute__((aligned(16))) float xs[RPW * KGE_PS_LD];
    __shared__ __attribute__((aligned(16))) float ys[RPW * KGE_PS_LD];
    const int lane = threadIdx.x;
    const int64_t ngroups = (rows + RPW - 1) / RPW;
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int64_t row0 = grp * RPW;
        float acc = 0.f;
        for (int k0 = 0; k0 < K; k0 += KGE_PS_KC) {   // K % 4 == 0, ld % 4 == 0, 16-byte aligned (checked by the host)
            const int kc = min(KGE_PS_KC, K - k0);
            const int pieces = kc >> 2;
            for (int idx = lane; idx < RPW * pieces; idx += 64) {
                const int rr = idx / pieces, pc = idx - rr * pieces;
                const int64_t r = min(row0 + rr, rows - 1);
                *reinterpret_cast<float4 *>(xs + rr * KGE_PS_LD + pc * 4) =
                    *reinterpret_cast<const float4 *>(X + r * ld + k0 + pc * 4);
                *reinterpret_cast<float4 *>(ys + rr * KGE_PS_LD + pc * 4) =
                    *reinterpret_cast<const float4 *>(Y + r * ld + k0 + pc * 4);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (lane < RPW) {
                const float *x = xs + lane * KGE_PS_LD, *y = ys + lane * KGE_PS_LD;
                for (int k = 0; k < kc; k += 4) {
                    const float4 a = *reinterpret_cast<const float4 *>(x + k), b = *reinterpret_cast<const float4 *>(y + k);
                    acc = fmaf(a.x, b.x, acc);
                    acc = fmaf(a.y, b.y, acc);
                    acc = fmaf(a.z, b.z, acc);
                    acc = fmaf(a.w, b.w, acc);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
        if (lane < RPW && row0 + lane < rows) out[row0 + lane] = scale * acc;
    }
}

// Per-query scalars of the projection modes (TransH / TransD) in ONE pass over the query rows: what kge_row_sqnorm(Q),
// kge_row_dot(Q, W[r]) * scale, kge_row_sqnorm(W[r]) + z_add and the glue between them (a gather of the W rows, an add, a
// stack) produce in six launches -- the SAME sequential chains (k ascending, one accumulator each), so the same bits:
//   qn[i] = ||q_i||^2          pz[i] = (scale * (q_i . w_{r_i}),  ||w_{r_i}||^2 + z_add)
// A wavefront owns 16 rows: rows of Q and the gathered rows of W staged through LDS, lanes 0..15 / 16..31 / 32..47 run the
// three chains of row (lane & 15) side by side.  NW wavefronts per block share the final max ||q||^2 atomic.
template <int NW>
__global__ __launch_bounds__(64 * NW) void proj_query_stats_kernel(const float *__restrict__ Q, int64_t ldq,
                                                                   const float *__restrict__ W, int64_t ldw,
                                                                   const int64_t *__restrict__ r_idx, int64_t rows, int K,
                                                                   float scale, float z_add, float *qn, float *pz,
                                                                   float *qmax_io, int32_t *zero_i32, int64_t zero_n)
{
    // (r06) the batch's rank counters zeroed by this launch (as the fused query pipelines do): no fill node of their own
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < zero_n; j += (int64_t)gridDim.x * blockDim.x) zero_i32[j] = 0;
    constexpr int RPW = 16;
    __shared__ __attribute__((aligned(16))) float xs_all[NW * RPW * KGE_PS_LD];
    __shared__ __attribute__((aligned(16))) float ys_all[NW * RPW * KGE_PS_LD];
    __shared__ unsigned wmax[NW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float *xs = xs_all + wv * RPW * KGE_PS_LD, *ys = ys_all + wv * RPW * KGE_PS_LD;
    const int row_l = lane & 15, chain = lane >> 4;        // chain 0: q.q, 1: q.w, 2: w.w (3: idle)
    float big = 0.f;
    const int64_t ngroups = (rows + RPW - 1) / RPW;
    for (int64_t grp = (int64_t)blockIdx.x * NW + wv; grp < ngroups; grp += (int64_t)gridDim.x * NW) {
        const int64_t row0 = grp * RPW;
        float acc = 0.f;
        // (r06) the next chunk's row pieces are in flight while this chunk's chains run (RPW x 8 pieces of 16 bytes per
        // operand: two per lane) -- the kernel was one load latency per 32 columns
        static_assert(RPW * (KGE_PS_KC / 4) <= 128, "two pieces per lane and operand");
        // (NAMED registers: as arrays carried around the chunk loop hipcc leaves them in scratch memory)
        float4 px0 = make_float4(0.f, 0.f, 0.f, 0.f), px1 = px0, py0 = px0, py1 = px0;
#define KGE_QS_FETCH(IT, KK)                                                                                  \
    {                                                                                                         \
        const int pieces_ = min(KGE_PS_KC, K - (KK)) >> 2, idx_ = lane + 64 * IT;                             \
        if (idx_ < RPW * pieces_) {                                                                           \
            const int rr_ = idx_ / pieces_, pc_ = idx_ - rr_ * pieces_;                                       \
            const int64_t r_ = min(row0 + rr_, rows - 1);                                                     \
            px##IT = *reinterpret_cast<const float4 *>(Q + r_ * ldq + (KK) + pc_ * 4);                        \
            py##IT = *reinterpret_cast<const float4 *>(W + r_idx[r_] * ldw + (KK) + pc_ * 4);                 \
        }                                                                                                     \
    }
#define KGE_QS_STORE(IT, PIECES)                                                                              \
    {                                                                                                         \
        const int idx_ = lane + 64 * IT;                                                                      \
        if (idx_ < RPW * (PIECES)) {                                                                          \
            const int rr_ = idx_ / (PIECES), pc_ = idx_ - rr_ * (PIECES);                                     \
            *reinterpret_cast<float4 *>(xs + rr_ * KGE_PS_LD + pc_ * 4) = px##IT;                             \
            *reinterpret_cast<float4 *>(ys + rr_ * KGE_PS_LD + pc_ * 4) = py##IT;                             \
        }                                                                                                     \
    }
        KGE_QS_FETCH(0, 0) KGE_QS_FETCH(1, 0)
        for (int k0 = 0; k0 < K; k0 += KGE_PS_KC) {   // K % 4 == 0, leading dimensions % 4 == 0, 16-byte aligned (host-checked)
            const int kc = min(KGE_PS_KC, K - k0);
            const int pieces = kc >> 2;
            KGE_QS_STORE(0, pieces) KGE_QS_STORE(1, pieces)
            if (k0 + KGE_PS_KC < K) { KGE_QS_FETCH(0, k0 + KGE_PS_KC) KGE_QS_FETCH(1, k0 + KGE_PS_KC) }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (chain < 3) {
                const float *x = (chain == 2 ? ys : xs) + row_l * KGE_PS_LD;
                const float *y = (chain == 0 ? xs : ys) + row_l * KGE_PS_LD;
                for (int k = 0; k < kc; k += 4) {
                    const float4 a = *reinterpret_cast<const float4 *>(x + k), b = *reinterpret_cast<const float4 *>(y + k);
                    acc = fmaf(a.x, b.x, acc);
                    acc = fmaf(a.y, b.y, acc);
                    acc = fmaf(a.z, b.z, acc);
                    acc = fmaf(a.w, b.w, acc);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
        const float uw = __shfl(acc, row_l + 16, 64), ww = __shfl(acc, row_l + 32, 64);
        if (lane < RPW && row0 + lane < rows) {
            qn[row0 + lane] = acc;
            pz[2 * (row0 + lane)] = scale * uw;
            pz[2 * (row0 + lane) + 1] = ww + z_add;
            big = __uint_as_float(max(__float_as_uint(big), __float_as_uint(acc)));
        }
    }
#undef KGE_QS_STORE
#undef KGE_QS_FETCH
    if (qmax_io) {      // one atomic per block (same-address atomics serialise in the L2)
        unsigned m = __float_as_uint(big);
        for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
        if (lane == 0) wmax[wv] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned mm = wmax[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) mm = max(mm, wmax[w]);
            kge_atomic_max_u32(reinterpret_cast<unsigned *>(qmax_io), mm);
        }
    }
}

__global__ void ewise_kernel(int op, const float *__restrict__ a, const float *__restrict__ b,
                             const float *__restrict__ c, const float *__restrict__ d, int64_t n, float *out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v;
        switch (op) {
        case KGE_EW_ADD: v = a[i] + b[i]; break;
        case KGE_EW_SUB: v = a[i] - b[i]; break;
        case KGE_EW_MUL: v = a[i] * b[i]; break;
        case KGE_EW_MULSUB: v = a[i] * b[i] - c[i] * d[i]; break;
        default: v = a[i] * b[i] + c[i] * d[i]; break;
        }
        out[i] = v;
    }
}

__global__ __launch_bounds__(WPB * 64) void gather_rows_kernel(const float *__restrict__ X, int64_t ld,
                                                               const int64_t *__restrict__ idx,
                                                               int64_t rows, int K, float *out)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    for (int64_t i = wave; i < rows; i += (int64_t)gridDim.x * WPB) {
        const float *x = X + idx[i] * ld;
        float *o = out + i * K;
        for (int k = lane; k < K; k += 64) o[k] = x[k];
    }
}

__global__ __launch_bounds__(WPB * 64) void normalize_rows_kernel(float *X, int64_t ld, int64_t rows, int K)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    for (int64_t i = wave; i < rows; i += (int64_t)gridDim.x * WPB) {
        float *x = X + i * ld;
        float ss = 0.f;
        for (int k = lane; k < K; k += 64) ss = fmaf(x[k], x[k], ss);
        ss = wave_sum(ss);
        const float n = fmaxf(sqrtf(ss), 1e-12f);
        for (int k = lane; k < K; k += 64) x[k] = x[k] / n;
    }
}

} // namespace

// rows up to which the staged chain kernels run 16 rows per wavefront (131,072 rows = 8,192 wavefronts: 32 per CU)
static const int64_t KGE_ROWS_SMALL = 131072;

extern "C" int kge_row_sqnorm(const float *X, int64_t ld, int64_t rows, int K, float *out, float *max_io,
                              kge_stream_t stream)
{
    if (rows < 0 || K <= 0 || ld < K) return KGE_EINVAL;
    if (rows == 0) return 0;
    if (!X || !out) return KGE_EINVAL;
    if (K % 4 == 0 && ld % 4 == 0 && kge_aligned16(X)) {
        // (latency-bound chains: 16 rows per wavefront until that fills the chip several times over)
        if (rows <= KGE_ROWS_SMALL) {
            const int64_t blocks = (rows + 63) / 64;        // 4 wavefronts x 16 rows
            auto k = row_sqnorm_staged_kernel<16, 4>;
            hipLaunchKernelGGL(k, dim3((int)blocks), dim3(256), 0, kge_s(stream), X, ld, rows, K, out, max_io);
        } else {
            const int64_t groups = (rows + 63) / 64;
            auto k = row_sqnorm_staged_kernel<64, 1>;
            hipLaunchKernelGGL(k, dim3((int)(groups < 256 * 14 ? groups : 256 * 14)), dim3(64), 0,
                               kge_s(stream), X, ld, rows, K, out, max_io);
        }
    } else {
        hipLaunchKernelGGL(row_sqnorm_kernel, dim3(grid_threads(rows, 64)), dim3(64), 0, kge_s(stream), X, ld, rows, K,
                           out, max_io);
    }
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_row_sqnorm_any_order(const float *X, int64_t ld, int64_t rows, int K, float *out, float *max_io,
                                        kge_stream_t stream)
{
    if (rows < 0 || K <= 0 || ld < K) return KGE_EINVAL;
    if (rows == 0) return 0;
    if (!X || !out) return KGE_EINVAL;
    const int64_t blocks = (rows + 63) / 64;
    hipLaunchKernelGGL(row_sqnorm_any_kernel, dim3((int)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, kge_s(stream),
                       X, ld, rows, K, out, max_io);
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_row_dot(const float *X, const float *Y, int64_t ld, int64_t rows, int K, float scale,
                           float *out, kge_stream_t stream)
{
    if (rows < 0 || K <= 0 || ld < K) return KGE_EINVAL;
    if (rows == 0) return 0;
    if (!X || !Y || !out) return KGE_EINVAL;
    if (K % 4 == 0 && ld % 4 == 0 && kge_aligned16(X) && kge_aligned16(Y)) {
        if (rows <= KGE_ROWS_SMALL) {
            const int64_t groups = (rows + 15) / 16;
            hipLaunchKernelGGL(row_dot_staged_kernel<16>, dim3((int)groups), dim3(64), 0, kge_s(stream), X, Y, ld, rows, K, scale, out);
        } else {
            const int64_t groups = (rows + 63) / 64;
            hipLaunchKernelGGL(row_dot_staged_kernel<64>, dim3((int)(groups < 256 * 14 ? groups : 256 * 14)), dim3(64), 0,
                               kge_s(stream), X, Y, ld, rows, K, scale, out);
        }
    } else {
        hipLaunchKernelGGL(row_dot_kernel, dim3(grid_threads(rows, 64)), dim3(64), 0, kge_s(stream), X, Y, ld, rows, K, scale, out);
    }
    KGE_CHECK_LAUNCH();
    return 0;
}

/* Per-query scalars of the projection modes in one launch: qn[i] = ||Q[i]||^2, pz[i] = (scale * (Q[i] . W[r_idx[i]]),
 * ||W[r_idx[i]]||^2 + z_add) -- the chains of kge_row_sqnorm / kge_row_dot (same bits), *qmax_io = max(., max qn).
 * KGE_EUNSUPPORTED unless K % 4 == 0, ldq % 4 == 0, ldw % 4 == 0 and both matrices are 16-byte aligned. */
extern "C" int kge_proj_query_stats(const float *Q, int64_t ldq, const float *W, int64_t ldw, const int64_t *r_idx,
                                    int64_t rows, int K, float scale, float z_add, float *qn, float *pz, float *qmax_io,
                                    int32_t *zero_i32, int64_t zero_n, kge_stream_t stream)
{
    if (rows < 0 || K <= 0 || ldq < K || ldw < K || zero_n < 0 || (zero_n > 0 && !zero_i32)) return KGE_EINVAL;
    if (rows == 0) return 0;
    if (!Q || !W || !r_idx || !qn || !pz) return KGE_EINVAL;
    if (K % 4 || ldq % 4 || ldw % 4 || !kge_aligned16(Q) || !kge_aligned16(W)) return KGE_EUNSUPPORTED;
    const int64_t blocks = (rows + 63) / 64;        // 4 wavefronts x 16 rows
    hipLaunchKernelGGL(proj_query_stats_kernel<4>, dim3((int)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, kge_s(stream), Q, ldq,
                       W, ldw, r_idx, rows, K, scale, z_add, qn, pz, qmax_io, zero_i32, zero_n);
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_ewise(int op, const float *a, const float *b, const float *c, const float *d, int64_t n,
                         float *out, kge_stream_t stream)
{
    if (op < KGE_EW_ADD || op > KGE_EW_MULADD || n < 0) return KGE_EINVAL;
    if (n == 0) return 0;
    if (!a || !b || !out) return KGE_EINVAL;
    if (op >= KGE_EW_MULSUB && (!c || !d)) return KGE_EINVAL;
    hipLaunchKernelGGL(ewise_kernel, dim3(grid_threads(n, 256)), dim3(256), 0, kge_s(stream), op, a, b, c, d, n, out);
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_gather_rows(const float *X, int64_t ld, const int64_t *idx, int64_t rows, int K,
                               float *out, kge_stream_t stream)
{
    if (rows < 0 || K <= 0 || ld < K) return KGE_EINVAL;
    if (rows == 0) return 0;
    if (!X || !idx || !out) return KGE_EINVAL;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_rows(rows)), dim3(WPB * 64), 0, kge_s(stream), X, ld, idx, rows, K, out);
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_normalize_rows(float *X, int64_t ld, int64_t rows, int K, kge_stream_t stream)
{
    if (rows < 0 || K <= 0 || ld < K) return KGE_EINVAL;
    if (rows == 0) return 0;
    if (!X) return KGE_EINVAL;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3(grid_rows(rows)), dim3(WPB * 64), 0, kge_s(stream), X, ld, rows, K);
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_lp_prep(int kind, int side, const float *t0, const float *t1, const float *t2,
                           const float *t3, int d_ent, int d_rel, const int64_t *h, const int64_t *t,
                           const int64_t *r, int64_t B, float *Q0, float *Q1, float *qn, float *Wq,
                           kge_stream_t stream)
{
    return kge_lp_prep_sharded(kind, side, t0, t1, t2, t3, d_ent, d_rel, h, t, r, B, 0, -1, Q0, Q1, qn, Wq, stream);
}

extern "C" int kge_lp_prep_sharded(int kind, int side, const float *t0, const float *t1, const float *t2,
                                   const float *t3, int d_ent, int d_rel, const int64_t *h, const int64_t *t,
                                   const int64_t *r, int64_t B, int64_t ent_lo, int64_t ent_n, float *Q0, float *Q1,
                                   float *qn, float *Wq, kge_stream_t stream)
{
    return kge_lp_prep_hi(kind, side, t0, t1, t2, t3, d_ent, d_rel, h, t, r, B, ent_lo, ent_n, Q0, Q1, qn, Wq, nullptr, 0, 0,
                          nullptr, stream);
}

extern "C" int kge_lp_prep_hi(int kind, int side, const float *t0, const float *t1, const float *t2, const float *t3,
                              int d_ent, int d_rel, const int64_t *h, const int64_t *t, const int64_t *r, int64_t B,
                              int64_t ent_lo, int64_t ent_n, float *Q0, float *Q1, float *qn, float *Wq, void *Qh,
                              int hi_units_p, int64_t Bp, float *q_dn2, kge_stream_t stream)
{
    if (kind < KGE_TRANSE_L1 || kind > KGE_COMPLEX) return KGE_EINVAL;
    if (ent_n >= 0 && ent_lo < 0) return KGE_EINVAL;
    if (side < KGE_SIDE_TAIL || side > KGE_SIDE_BOTH) return KGE_EINVAL;
    if (!t0 || !t1 || d_ent <= 0 || d_rel <= 0 || B < 0) return KGE_EINVAL;
    if (B == 0) return 0;
    if (!h || !t || !r || !Q0) return KGE_EINVAL;
    if (kind == KGE_COMPLEX && (!t2 || !t3 || !Q1)) return KGE_EINVAL;
    if (kind == KGE_TRANSH && !t2) return KGE_EINVAL;       // (Wq optional: NULL = the gathered rows are not needed)
    if (kind == KGE_TRANSD && (!t2 || !t3 || d_ent < d_rel)) return KGE_EINVAL;
    if (kind != KGE_TRANSD && d_ent != d_rel) return KGE_EINVAL;
    const int64_t nq = side == KGE_SIDE_BOTH ? 2 * B : B;
    if (Qh) {   // the hi operand rides the projection models' unsharded launch only
        if ((kind != KGE_TRANSH && kind != KGE_TRANSD) || ent_n >= 0 || hi_units_p * 16 < d_rel + 2 || Bp < nq) return KGE_EINVAL;
    }
    PrepParams p{kind, side, t0, t1, t2, t3, d_ent, d_rel, h, t, r, B, Q0, Q1, Wq, ent_lo, ent_n,
                 reinterpret_cast<_Float16 *>(Qh), hi_units_p, Bp, q_dn2, kge_env_int("KGE_PREP_INFLIGHT", 1)};
    hipLaunchKernelGGL(lp_prep_kernel, dim3(grid_rows(Qh ? Bp : nq)), dim3(WPB * 64), 0, kge_s(stream), p);
    KGE_CHECK_LAUNCH();
    if (qn) return kge_row_sqnorm(Q0, d_rel, nq, d_rel, qn, nullptr, stream);
    return 0;
}

extern "C" int kge_relation_scores_proj(int kind, const float *E, const float *R, const float *Wt, const float *Ep,
                                        int d_ent, int d_rel, const int64_t *h, const int64_t *t, int64_t B,
                                        int64_t n_rel, float *out, int64_t ldo, kge_stream_t stream)
{
    if (kind != KGE_TRANSH && kind != KGE_TRANSD) return KGE_EINVAL;
    if (!E || !R || !Wt || d_ent <= 0 || d_rel <= 0 || B < 0 || n_rel < 0 || ldo < n_rel) return KGE_EINVAL;
    if (kind == KGE_TRANSD && (!Ep || d_ent < d_rel)) return KGE_EINVAL;
    if (kind == KGE_TRANSH && d_ent != d_rel) return KGE_EINVAL;
    if (B == 0 || n_rel == 0) return 0;
    if (!h || !t || !out) return KGE_EINVAL;
    hipLaunchKernelGGL(rel_proj_scores_kernel, dim3(grid_rows(B * n_rel)), dim3(WPB * 64), 0, kge_s(stream), kind, E, R,
                       Wt, Ep, d_ent, d_rel, h, t, B, n_rel, out, ldo);
    KGE_CHECK_LAUNCH();
    return 0;
}
