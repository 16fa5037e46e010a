// Certified integer prefilter for the fused rank count of TransE-L1 (gfx950).
//
// The rank of a test fact needs  #{c : s[i,c] >= s_true[i]}  with  s[i,c] = -sum_k |q_i[k] - e_c[k]|
// (TranslationModel.inference_scoring_function with l1_dissimilarity: interfaces.py:253-260,
// utils/dissimilarities.py:11-16; get_rank, utils/operations.py:37-61).  The exact kernel (lp_direct.hip) is
// bound by VALU issue: L1 is not bilinear -- no matrix-core form -- and costs one subtract and one add-|.| per
// (pair, k): 2 issue slots per element.  gfx950 has a sum-of-absolute-differences instruction,
//   v_sad_u16  D = |S0.lo16 - S1.lo16| + |S0.hi16 - S1.hi16| + S2         (half rate: 1 slot per element)
// so the left side is evaluated on 16-bit fixed-point copies of the operands with a proven error band, and only
// the pairs inside the band are re-scored by the exact fp32 chain:
//
//   X = rint(x * s) + 32768  (u16),  s = 32700 / xmax,  xmax >= every |x| of both operands
//   D_ic = sum_k |Q_ik - E_ck|                    exact integer arithmetic (u32 accumulators)
//   |D - s L| <= 1.004 K   (each operand is within 0.5 of x s after rounding, + 0.002 for the product's own rounding)
//   the exact kernel's value F = fl(L) is within (K + 1) 2^-24 L of L  (one rounding per subtract / add, ascending k)
//   count  <=>  F_c <= U_i,  U_i = -s_true_i.      D <= T_lo_i: certainly counted;  D >= T_hi_i: certainly not;
//   T_lo < D < T_hi: UNCERTAIN -> listed, re-scored exactly (lp_pair_score_staged_direct == lp_pair_score ==
//   lp_direct_kernel, bit for bit).
// raw_count first receives #{D < T_hi}; kge_lp_sad_recheck takes 1 off for every listed pair whose exact score is
// below s_true: the counts are EXACTLY those of kge_lp_count_ge.  tests/test_gpu_parity.py checks that.
//
// Kernel: the register tiling of lp_direct.hip (256 threads as 16 x 16, 8 x 8 pairs per thread, operands staged
// through double-buffered LDS 32 k at a time: 64 B per row, row stride 80 B -> conflict-free b128), 4 v_sad_u16
// per (pair, 8 k).  Uncertain pairs are buffered in LDS per tile and handed to the global list by one atomic.
#include "kge_common.h"

#ifndef KGE_SAD_UNROLL
#define KGE_SAD_UNROLL 1
#endif

namespace {

constexpr int SK = 32;                 // k per stage
constexpr int SROW = 80;               // LDS row stride in bytes (64 data + 16 pad: 5 x 16 B, odd -> conflict-free)
constexpr int NT = 256, TM = 8, TN = 8, BM = 16 * TM, BN = 16 * TN;
constexpr int STAGE_BYTES = (BM + BN) * SROW;
constexpr int UNC_CAP = 2048;
constexpr int GSETS = 4;               // queries per grouped column (= kge_lp_split_group_sets())
constexpr int SMEM_BYTES = 2 * STAGE_BYTES + GSETS * (BM * 4 /* rc */ + BM * 8 /* thresholds */) + 16 + UNC_CAP * 4;

__device__ __forceinline__ float sad_scale(float emax, float rmax)
{
    const float m = emax + rmax;
    return (m > 0.f && m < INFINITY) ? 32700.0f / m : 1.0f;
}

struct SadRowsParams {
    const float *X;
    int64_t ld, rows;
    int K, Kp;
    const float *emax, *rmax;          // device scalars: max |x| of the entity table / of the relation table
    uint16_t *out;                     // (rows, Kp)
    const int64_t *row_index;          // optional: output row r <- source row row_index[r] (query columns)
};

__global__ void sad_rows_kernel(const SadRowsParams p)
{
    const float s = sad_scale(*p.emax, *p.rmax);
    const int64_t total = p.rows * (p.Kp / 8);
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = idx / (p.Kp / 8);
        const int k0 = (int)(idx % (p.Kp / 8)) * 8;
        union { uint16_t h[8]; uint4 v; } o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e;
            float x = k < p.K ? p.X[(p.row_index ? p.row_index[row] : row) * p.ld + k] : 0.f;
            // (k >= K: both operands hold 32768 there -> |difference| = 0)
            float r = rintf(x * s);
            r = fminf(fmaxf(r, -32767.0f), 32767.0f);      // never active for |x| <= xmax (s leaves 0.2 % of slack)
            o.h[e] = (uint16_t)((int)r + 32768);
        }
        *reinterpret_cast<uint4 *>(p.out + row * p.Kp + k0) = o.v;
    }
}

struct SadThrParams {
    const float *s_true;
    const float *emax, *rmax;
    int64_t B;
    int K;
    float eps_scale;
    int2 *thr;                  // (T_lo, T_hi) per query
    int32_t *list_count;
    float *overflow;
};

__global__ void sad_thr_kernel(const SadThrParams p)
{
    const float em = *p.emax, rm = *p.rmax;
    const float s = sad_scale(em, rm);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *p.list_count = 0;
        if (!(em + rm < INFINITY) || !(em + rm > 0.f)) *p.overflow = 1.0f;     // NaN / inf / all-zero tables: exact path
    }
    const float beta = 1.004f * (float)p.K;                           // |D - s L|
    const float gamma2 = 2.0f * 1.001f * (float)(p.K + 1) * 5.9604645e-8f;   // 2 x the exact chain's relative error
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.B; i += (int64_t)gridDim.x * blockDim.x) {
        const float U = -p.s_true[i];                                 // = fl(L_true) >= 0 (NaN: nothing counts)
        int lo = -1, hi = 0;
        if (U >= 0.f && U < INFINITY) {
            // certainly counted:  (D + beta) / s * (1 + gamma) <= U ;  certainly not:  (D - beta) / s * (1 - gamma) > U.
            // s U <= 65400 K < 2^24 for K <= 256; beyond, fp32 spacing of s U is covered by the relative term
            const float su = s * U;
            const float wid = (beta + su * gamma2 + su * 2.4e-7f + 3.0f) * p.eps_scale;
            const float flo = floorf(su - wid), fhi = ceilf(su + wid);
            lo = flo < -1.0f ? -1 : (flo > 2.0e9f ? 2000000000 : (int)flo);
            hi = fhi < 0.0f ? 0 : (fhi > 2.0e9f ? 2000000000 : (int)fhi);
        } else if (U == INFINITY) {      // s_true = -inf: every finite score counts
            lo = 2000000000; hi = 2000000000;
        }
        p.thr[i] = make_int2(lo, hi);
    }
}

struct SadParams {
    const uint16_t *Q, *E;
    int64_t ldq, lde;           // in u16 elements (multiples of 8)
    int64_t B, N;
    int Kp;
    const int2 *thr;
    int32_t *raw_count;
    int32_t *list;
    int32_t cap;
    int32_t *list_count;
    float *overflow;
    int row_panels, col_tiles, tiles_per_block;
    // COLUMNS instead of queries (kge_sad_args.col_q / members, as for the f16-split count): the operand rows are the
    // distinct query rows of the batch; qmap[row * GS + set] = the query that compares row's distances with ITS
    // thresholds (< 0: none); NULL: row == query
    const int32_t *qmap;
};

// GS = 1: one query per row (optionally through qmap); GS = GSETS: grouped columns -- the distances of a row are compared
// with the thresholds of each of its queries (the SAD sweep runs once per distinct query row)
template <int GS>
__global__ __launch_bounds__(NT, 2) void lp_l1_sad_count_kernel(const SadParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *rc = reinterpret_cast<int *>(smem + 2 * STAGE_BYTES);                                   // [GS][BM]
    int2 *thr_s = reinterpret_cast<int2 *>(smem + 2 * STAGE_BYTES + GSETS * BM * 4);              // [GS][BM]
    int *unc_cnt = reinterpret_cast<int *>(smem + 2 * STAGE_BYTES + GSETS * BM * 12);
    unsigned *unc_list = reinterpret_cast<unsigned *>(smem + 2 * STAGE_BYTES + GSETS * BM * 12 + 16);
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;

    // same XCD-aware (row panel, column chunk) assignment as lp_direct_kernel
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int xq = nblk >> 3, xr = nblk & 7, xcd = bid & 7, loc = bid >> 3;
    const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + loc;
    const int rp = lid % p.row_panels, cc = lid / p.row_panels;
    const int64_t row0 = (int64_t)rp * BM;
    const int tile_begin = cc * p.tiles_per_block;
    const int tile_end = min(tile_begin + p.tiles_per_block, p.col_tiles);
    const int ntiles = tile_end - tile_begin;
    if (ntiles <= 0) return;

    const int Kp = p.Kp, S = (Kp + SK - 1) / SK, G = ntiles * S;
    const int srow = tid >> 2, skc = tid & 3;       // staging: 4 threads x 16 B cover one 64-byte row segment
    uint4 stQ[2], stT[2];
    auto prefetch = [&](int g) {
        const int ti = g / S, s = g - ti * S;
        const int k = s * SK + skc * 8;
        const int kc = k < Kp ? k : 0;              // clamped, unconditional loads (zeroed when staged)
        const int64_t col0 = (int64_t)(tile_begin + ti) * BN;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            stQ[j] = *reinterpret_cast<const uint4 *>(p.Q + min(row0 + srow + 64 * j, p.B - 1) * p.ldq + kc);
            stT[j] = *reinterpret_cast<const uint4 *>(p.E + min(col0 + srow + 64 * j, p.N - 1) * p.lde + kc);
        }
    };
    auto stage_store = [&](int buf, int g) {
        const int ti = g / S, s = g - ti * S;
        const bool k_ok = s * SK + skc * 8 < Kp;
        const int64_t col0 = (int64_t)(tile_begin + ti) * BN;
        char *Qs = smem + buf * STAGE_BYTES, *Ts = Qs + BM * SROW;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool qok = k_ok && row0 + srow + 64 * j < p.B, tok = k_ok && col0 + srow + 64 * j < p.N;
            // (component-wise selects: a select between two uint4 VALUES made hipcc index a scratch copy of them)
            *reinterpret_cast<uint4 *>(Qs + (srow + 64 * j) * SROW + skc * 16) =
                make_uint4(qok ? stQ[j].x : 0u, qok ? stQ[j].y : 0u, qok ? stQ[j].z : 0u, qok ? stQ[j].w : 0u);
            *reinterpret_cast<uint4 *>(Ts + (srow + 64 * j) * SROW + skc * 16) =
                make_uint4(tok ? stT[j].x : 0u, tok ? stT[j].y : 0u, tok ? stT[j].z : 0u, tok ? stT[j].w : 0u);
        }
    };

    unsigned acc[TM][TN];
    int cnt[TM];        // GS == 1: per-row counters in registers over the whole sweep; grouped columns: one LDS atomic per
                        // (row, set) and tile instead (32 more live registers would spill next to the 64 accumulators)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        cnt[i] = 0;
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = 0u;
    }
    for (int idx = tid; idx < BM * GS; idx += NT) {
        const int lrow = idx / GS, gs = idx - lrow * GS;
        const int64_t row = row0 + lrow;
        int64_t q = -1;
        if (row < p.B) q = p.qmap ? (int64_t)p.qmap[row * GS + gs] : row;
        rc[gs * BM + lrow] = 0;
        thr_s[gs * BM + lrow] = q >= 0 ? p.thr[q] : make_int2(-1, 0);      // (D < 0 never holds: such a set counts nothing)
    }
    if (tid == 0) *unc_cnt = 0;

    prefetch(0);
    stage_store(0, 0);
    __syncthreads();

    for (int g = 0; g < G; ++g) {
        const int buf = g & 1;
        if (g + 1 < G) prefetch(g + 1);
        const int ti = g / S, s = g - ti * S;
        const int nk8 = min(SK / 8, (Kp - s * SK + 7) >> 3);
        const char *Qb = smem + buf * STAGE_BYTES + ty * SROW;
        const char *Tb = smem + buf * STAGE_BYTES + BM * SROW + tx * SROW;
#if KGE_SAD_UNROLL > 1
#pragma unroll KGE_SAD_UNROLL
#endif
        for (int k8 = 0; k8 < nk8; ++k8) {
            uint4 q[TM], t[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) q[i] = *reinterpret_cast<const uint4 *>(Qb + 16 * i * SROW + k8 * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j) t[j] = *reinterpret_cast<const uint4 *>(Tb + 16 * j * SROW + k8 * 16);
            // component-outer: 64 independent SADs between two updates of the same accumulator (a pair's four SADs
            // written back to back form a dependent chain that the two waves of a SIMD cannot hide)
#define KGE_SAD_COMP(C)                                                                          \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                               \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                           \
            acc[i][j] = __builtin_amdgcn_sad_u16(q[i].C, t[j].C, acc[i][j]);
            KGE_SAD_COMP(x)
            __builtin_amdgcn_sched_barrier(0);
            KGE_SAD_COMP(y)
            __builtin_amdgcn_sched_barrier(0);
            KGE_SAD_COMP(z)
            __builtin_amdgcn_sched_barrier(0);
            KGE_SAD_COMP(w)
#undef KGE_SAD_COMP
        }

        const bool tile_done = s == S - 1;
        const int64_t col0 = (int64_t)(tile_begin + ti) * BN;
        if (tile_done) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int lrow = ty + 16 * i;
#pragma unroll
                for (int gs = 0; gs < GS; ++gs) {
                    const int2 th = thr_s[gs * BM + lrow];
                    int c = 0;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int lcol = tx + 16 * j;
                        const int D = (int)acc[i][j];           // <= 65534 * Kp < 2^31 (host check)
                        const bool in = col0 + lcol < p.N;
                        if (in && D < th.y) {
                            ++c;
                            if (D > th.x) {                     // inside the band: list it ([lcol:7][set:2][lrow:7])
                                const int idx = atomicAdd(unc_cnt, 1);
                                if (idx < UNC_CAP) unc_list[idx] = ((unsigned)lcol << 9) | ((unsigned)gs << 7) | (unsigned)lrow;
                            }
                        }
                    }
                    if (GS == 1) cnt[i] += c;
                    else if (c) atomicAdd(&rc[gs * BM + lrow], c);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = 0u;
            }
        }
        if (g + 1 < G) stage_store(buf ^ 1, g + 1);
        __syncthreads();
        if (tile_done) {        // block-uniform: wave 0 hands this tile's uncertain pairs to the global list
            if (tid < 64) {
                const int n = *unc_cnt, nc = min(n, UNC_CAP);
                if (n > 0) {
                    int base = 0;
                    if (tid == 0) base = atomicAdd(p.list_count, nc);
                    base = __shfl(base, 0, 64);
                    if (n > UNC_CAP && tid == 0) *p.overflow = 1.0f;
                    for (int i = tid; i < nc; i += 64) {
                        const unsigned e = unc_list[i];
                        const int pos = base + i;
                        if ((unsigned)pos < (unsigned)p.cap) {
                            const int64_t rw = row0 + (e & 127u);
                            p.list[2 * pos] = p.qmap ? p.qmap[rw * GS + ((e >> 7) & 3u)] : (int32_t)rw;
                            p.list[2 * pos + 1] = (int32_t)(col0 + (e >> 9));
                        } else {
                            *p.overflow = 1.0f;
                        }
                    }
                    if (tid == 0) *unc_cnt = 0;     // (every lane of this wave has read n above)
                }
            }
            __syncthreads();
        }
    }

    if (GS == 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
            if (cnt[i]) atomicAdd(&rc[ty + 16 * i], cnt[i]);
    }
    __syncthreads();
    for (int idx = tid; idx < BM * GS; idx += NT) {
        const int lrow = idx / GS, gs = idx - lrow * GS;
        const int64_t row = row0 + lrow;
        const int v = rc[gs * BM + lrow];
        if (row < p.B && v) {
            const int64_t q = p.qmap ? (int64_t)p.qmap[row * GS + gs] : row;
            if (q >= 0) atomicAdd(&p.raw_count[q], v);
        }
    }
}

// exact re-scoring of the listed pairs (plain direct modes): one lane per pair, rows staged cooperatively
template <bool VEC4, bool L1>
__global__ __launch_bounds__(64, 2) void direct_recheck_kernel(const kge_lp_desc d, const float *__restrict__ s_true,
                                                               const int32_t *__restrict__ list, int32_t cap,
                                                               const int32_t *__restrict__ list_count, int32_t *raw_count)
{
    __shared__ __attribute__((aligned(16))) float qs[64 * KGE_PS_LD];
    __shared__ __attribute__((aligned(16))) float es[64 * KGE_PS_LD];
    const int lane = threadIdx.x;
    const int n = (int)min((unsigned)*list_count, (unsigned)cap);
    const int ngroups = (n + 63) >> 6;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int pi = grp * 64 + lane;
        const bool valid = pi < n;
        const int pj = valid ? pi : grp * 64;
        const int qi = list[2 * pj], ci = list[2 * pj + 1];
        const float sc = lp_pair_score_staged_direct<VEC4, L1>(d, qi, ci, qs, es);
        if (valid && !(sc >= s_true[qi])) atomicSub(&raw_count[qi], 1);
    }
}

} // namespace

extern "C" int64_t kge_lp_sad_cols_padded(int K) { return ((int64_t)K + 7) / 8 * 8; }

extern "C" int kge_lp_sad_rows(const float *X, int64_t ld, int64_t rows, int K, const float *emax, const float *rmax,
                               void *out, const int64_t *row_index, kge_stream_t stream)
{
    if (rows < 0 || K <= 0 || ld < K) return KGE_EINVAL;
    if (rows == 0) return 0;
    if (!X || !emax || !rmax || !out) return KGE_EINVAL;
    SadRowsParams p{X, ld, rows, K, (int)kge_lp_sad_cols_padded(K), emax, rmax, reinterpret_cast<uint16_t *>(out), row_index};
    const int64_t total = rows * (p.Kp / 8);
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(sad_rows_kernel, dim3(grid), dim3(256), 0, kge_s(stream), p);
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_lp_sad_count(const kge_lp_desc *d, const kge_sad_args *a, const float *s_true, int32_t *raw_count,
                                kge_stream_t stream)
{
    int rc = kge_lp_desc_check(d);
    if (rc) return rc;
    if (d->mode != KGE_LP_L1_DIRECT || d->Wq) return KGE_EINVAL;
    if (d->B == 0 || d->N == 0) return 0;
    if (!a || !a->Qi || !a->Ei || !a->emax || !a->rmax || !a->thr || !a->list || a->cap <= 0 || !a->list_count ||
        !a->overflow || !s_true || !raw_count)
        return KGE_EINVAL;
    if (d->B > INT32_MAX || d->N > INT32_MAX || d->K0 > 32000) return KGE_EINVAL;      // D <= 65534 K < 2^31
    hipStream_t st = kge_s(stream);
    SadThrParams t{s_true, a->emax, a->rmax, d->B, d->K0, a->eps_scale, reinterpret_cast<int2 *>(a->thr), a->list_count,
                   a->overflow};
    hipLaunchKernelGGL(sad_thr_kernel, dim3((int)((d->B + 255) / 256 < 2048 ? (d->B + 255) / 256 : 2048)), dim3(256), 0, st, t);
    SadParams p;
    p.Q = reinterpret_cast<const uint16_t *>(a->Qi);
    p.E = reinterpret_cast<const uint16_t *>(a->Ei);
    p.Kp = (int)kge_lp_sad_cols_padded(d->K0);
    p.ldq = p.lde = p.Kp;
    p.B = d->B; p.N = d->N;
    p.thr = reinterpret_cast<const int2 *>(a->thr);
    p.raw_count = raw_count;
    p.list = a->list; p.cap = a->cap; p.list_count = a->list_count; p.overflow = a->overflow;
    p.col_tiles = (int)((d->N + BN - 1) / BN);
    p.qmap = nullptr;
    auto launch = [&](auto kern, int64_t rows) -> int {
        p.B = rows;
        p.row_panels = (int)((rows + BM - 1) / BM);
        const int target_blocks = kge_env_int("KGE_LP_TARGET_BLOCKS", 16384);
        int chunks = (target_blocks + p.row_panels - 1) / p.row_panels;
        if (chunks > p.col_tiles) chunks = p.col_tiles;
        if (chunks < 1) chunks = 1;
        p.tiles_per_block = (p.col_tiles + chunks - 1) / chunks;
        const int col_chunks = (p.col_tiles + p.tiles_per_block - 1) / p.tiles_per_block;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, dim3(p.row_panels * col_chunks), dim3(NT), SMEM_BYTES, st, p);
        return 0;
    };
    if (a->col_q || a->members) {
        // columns: Qi holds n_single_p rows with one query each (col_q), then n_multi_p rows with up to GSETS queries
        if (a->n_single_p < 0 || a->n_multi_p < 0 || (a->n_single_p > 0 && !a->col_q) || (a->n_multi_p > 0 && !a->members))
            return KGE_EINVAL;
        if (a->n_single_p > 0) {
            p.qmap = a->col_q;
            rc = launch(lp_l1_sad_count_kernel<1>, a->n_single_p);
            if (rc) return rc;
        }
        if (a->n_multi_p > 0) {
            p.qmap = a->members;
            p.Q = reinterpret_cast<const uint16_t *>(a->Qi) + a->n_single_p * p.ldq;
            rc = launch(lp_l1_sad_count_kernel<GSETS>, a->n_multi_p);
            if (rc) return rc;
        }
    } else {
        rc = launch(lp_l1_sad_count_kernel<1>, d->B);
        if (rc) return rc;
    }
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_lp_sad_recheck(const kge_lp_desc *d, const float *s_true, const int32_t *list, int32_t cap,
                                  const int32_t *list_count, int32_t *raw_count, kge_stream_t stream)
{
    int rc = kge_lp_desc_check(d);
    if (rc) return rc;
    if ((d->mode != KGE_LP_L1_DIRECT && d->mode != KGE_LP_L2_DIRECT) || d->Wq) return KGE_EINVAL;
    if (d->B == 0 || d->N == 0) return 0;
    if (!s_true || !list || cap <= 0 || !list_count || !raw_count) return KGE_EINVAL;
    const int grid = cap / 64 + 1 < 256 * 14 ? cap / 64 + 1 : 256 * 14;
    hipStream_t st = kge_s(stream);
    const bool v4 = kge_lp_vec4(*d), l1 = d->mode == KGE_LP_L1_DIRECT;
    if (v4 && l1) hipLaunchKernelGGL((direct_recheck_kernel<true, true>), dim3(grid), dim3(64), 0, st, *d, s_true, list, cap, list_count, raw_count);
    else if (v4) hipLaunchKernelGGL((direct_recheck_kernel<true, false>), dim3(grid), dim3(64), 0, st, *d, s_true, list, cap, list_count, raw_count);
    else if (l1) hipLaunchKernelGGL((direct_recheck_kernel<false, true>), dim3(grid), dim3(64), 0, st, *d, s_true, list, cap, list_count, raw_count);
    else hipLaunchKernelGGL((direct_recheck_kernel<false, false>), dim3(grid), dim3(64), 0, st, *d, s_true, list, cap, list_count, raw_count);
    KGE_CHECK_LAUNCH();
    return 0;
}
