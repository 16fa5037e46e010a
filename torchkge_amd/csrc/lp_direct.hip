// Translational all-candidates scores by broadcast-subtract + L_p reduction
// (fp32 VALU), gfx950.
//
//   S[i,c] = - sum_k | q_i[k] - e_c[k] (+ a(i,c) * w_i[k]) |^p ,  p in {1,2}
//
// Replaces TranslationModel.inference_scoring_function (interfaces.py:240-260)
// for TransE-L1/L2 (translation.py:105-125) and, with the rank-1 term, for
// TransH (cand = E[c] - a[c,r_i] W[r_i], translation.py:234-284) and TransD
// (cand = s_c Rp[r_i] + E[c,:dr], translation.py:603-652) WITHOUT the reference's
// (R,N,d) projection cache or its (b,N,d) intermediates.
//
// Register-tiled: 256 threads as 16(query) x 16(candidate); each thread owns a
// TM x 8 micro-tile with one fp32 accumulator per pair (the contract of
// lp_pair_score: p = 2 one fmaf per k in ascending order; p = 1 one add per
// aligned group of four k, the group's |diff| summed as (|d0|+|d1|)+(|d2|+|d3|),
// absent k count as 0 -- the same instruction count as a plain chain, a quarter
// of its roundings at full magnitude), q / e / w tiles staged through
// double-buffered LDS in BK = 32 slices (row stride 36: conflict-free b128).
#include "kge_common.h"

namespace {

constexpr int BK = 32, LDS_LD = BK + 4, NTHREADS = 256, TN = 8, BN = 16 * TN;

struct DirectParams {
    kge_lp_desc d;
    float *out;
    int64_t ldo;
    const float *s_true;
    int *raw_count;
    int row_panels, col_tiles, tiles_per_block, col_chunks;
    int out_vec4;       // out rows are 16-byte aligned (ldo % 4 == 0, aligned base): float4 stores
    // COUNT over query COLUMNS (packed L2 kernel, no rank-1 term): row r of the problem is the query row rep[r];
    // qmap[r * GS + set] = the query that compares row r's scores with ITS true score (< 0: none); NULL: row == query
    const int64_t *rep;
    const int32_t *qmap;
};

template <bool VEC4>
__device__ __forceinline__ void g_load8(const float *__restrict__ base, int64_t ld, int64_t row,
                                        bool row_ok, int k, int K, float (&v)[8])
{
    if (VEC4) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (row_ok) {
            const float *p = base + row * ld + k;
            if (k + 4 <= K) a = *reinterpret_cast<const float4 *>(p);
            if (k + 8 <= K) b = *reinterpret_cast<const float4 *>(p + 4);
        }
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (row_ok && k + j < K) ? base[row * ld + k + j] : 0.f;
    }
}

__device__ __forceinline__ void lds_store8(float *dst, const float (&v)[8])
{
    *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4 *>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

template <bool L1>
__device__ __forceinline__ float acc_step(float acc, float diff)
{
    return L1 ? acc + fabsf(diff) : fmaf(diff, diff, acc);
}

// TM = 8 (plain) or 4 (AXPY: the per-pair scalar a(i,c) also lives in registers)
template <bool VEC4, bool L1, bool AXPY, bool COUNT, int TM>
__global__ __launch_bounds__(NTHREADS, 2) void lp_direct_kernel(const DirectParams p)
{
    constexpr int BM = 16 * TM;
    constexpr int QCH = BM * 4 / NTHREADS; // 8-float chunks of the q tile per thread (2 or 1)
    constexpr int Q_FLOATS = BM * LDS_LD, T_FLOATS = BN * LDS_LD;
    constexpr int BUF_FLOATS = Q_FLOATS * (AXPY ? 2 : 1) + T_FLOATS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const kge_lp_desc &d = p.d;
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;

    const int nblk_grid = gridDim.x, bid = blockIdx.x;
    const int xq = nblk_grid >> 3, xr = nblk_grid & 7, xcd = bid & 7, loc = bid >> 3;
    const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + loc;
    const int rp = lid % p.row_panels, cc = lid / p.row_panels;
    const int64_t row0 = (int64_t)rp * BM;
    const int tile_begin = cc * p.tiles_per_block;
    const int tile_end = min(tile_begin + p.tiles_per_block, p.col_tiles);
    const int ntiles = tile_end - tile_begin;
    if (ntiles <= 0) return;

    const int K = d.K0;
    const int S = (K + BK - 1) / BK;
    const int G = ntiles * S;

    float stQ[QCH][8], stW[AXPY ? QCH : 1][8], stT[2][8];
    const int srow = tid >> 2, skc = tid & 3;

    // unconditional loads from clamped addresses (all in flight together);
    // out-of-range rows / k are zeroed when staged into LDS (see lp_gemm_mfma.hip)
    auto ld8 = [&](const float *base, int64_t ld, int64_t r, int64_t rmax, int k, float (&v)[8]) {
        if (VEC4) {
            const int k0 = (k < K) ? k : 0, k1 = (k + 4 < K) ? k + 4 : 0;
            const float *pr = base + min(r, rmax) * ld;
            const float4 a = *reinterpret_cast<const float4 *>(pr + k0);
            const float4 b = *reinterpret_cast<const float4 *>(pr + k1);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
            v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
            g_load8<false>(base, ld, r, r <= rmax, k, K, v);
        }
    };
    auto zero8 = [&](bool r_ok, int k, float (&v)[8]) {
        if (VEC4) {
            const bool k0_ok = k < K, k1_ok = k + 4 < K;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (r_ok && (e < 4 ? k0_ok : k1_ok)) ? v[e] : 0.f;
        }
    };
    auto prefetch = [&](int g) {
        const int ti = g / S, s = g - ti * S;
        const int k = s * BK + skc * 8;
        const int64_t col0 = (int64_t)(tile_begin + ti) * BN;
#pragma unroll
        for (int j = 0; j < QCH; ++j) {
            const int64_t r = row0 + srow + 64 * j;
            ld8(d.A0, d.lda0, r, d.B - 1, k, stQ[j]);
            if (AXPY) ld8(d.Wq, d.ldw, r, d.B - 1, k, stW[j]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) ld8(d.T0, d.ldt0, col0 + srow + 64 * j, d.N - 1, k, stT[j]);
    };
    auto stage_store = [&](int buf, int g) {
        float *Qs = smem + buf * BUF_FLOATS;
        float *Ts = Qs + Q_FLOATS;
        float *Ws = Ts + T_FLOATS;
        {
            const int ti = g / S, s = g - ti * S;
            const int k = s * BK + skc * 8;
            const int64_t col0 = (int64_t)(tile_begin + ti) * BN;
#pragma unroll
            for (int j = 0; j < QCH; ++j) {
                const bool ok = row0 + srow + 64 * j < d.B;
                zero8(ok, k, stQ[j]);
                if (AXPY) zero8(ok, k, stW[j]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) zero8(col0 + srow + 64 * j < d.N, k, stT[j]);
        }
#pragma unroll
        for (int j = 0; j < QCH; ++j) {
            lds_store8(Qs + (srow + 64 * j) * LDS_LD + skc * 8, stQ[j]);
            if (AXPY) lds_store8(Ws + (srow + 64 * j) * LDS_LD + skc * 8, stW[j]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) lds_store8(Ts + (srow + 64 * j) * LDS_LD + skc * 8, stT[j]);
    };

    float acc[TM][TN];
    float av[AXPY ? TM : 1][AXPY ? TN : 1];
    int cnt[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        cnt[i] = 0;
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = 0.f;
    }

    int *rc = reinterpret_cast<int *>(smem + 2 * BUF_FLOATS);
    float *st_s = smem + 2 * BUF_FLOATS + BM;
    if (tid < BM) {
        const int64_t row = row0 + tid;
        rc[tid] = 0;
        st_s[tid] = (COUNT && row < d.B) ? p.s_true[row] : 0.f;
    }

    auto load_av = [&](int ti) {
        if (AXPY) {
            const int64_t col0 = (int64_t)(tile_begin + ti) * BN;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int64_t row = row0 + ty + 16 * i;
                const int64_t rsel = (d.scal_ld > 1) ? d.r_idx[min(row, d.B - 1)] : 0;
#pragma unroll
                for (int j = 0; j < TN; ++j) { // clamped, unconditional (values of padded rows/cols are unused)
                    const int64_t col = min(col0 + tx + 16 * j, d.N - 1);
                    av[AXPY ? i : 0][AXPY ? j : 0] = d.scal[col * d.scal_ld + rsel];
                }
            }
        }
    };

    prefetch(0);
    stage_store(0, 0);
    load_av(0);
    __syncthreads();

    for (int g = 0; g < G; ++g) {
        const int buf = g & 1;
        if (g + 1 < G) prefetch(g + 1);
        const int ti = g / S, s = g - ti * S;
        const int nk4 = min(BK / 4, (K - s * BK + 3) >> 2);

        const float *Qb = smem + buf * BUF_FLOATS + ty * LDS_LD;
        const float *Tb = smem + buf * BUF_FLOATS + Q_FLOATS + tx * LDS_LD;
        const float *Wb = smem + buf * BUF_FLOATS + Q_FLOATS + T_FLOATS + ty * LDS_LD;
#pragma unroll 2
        for (int k4 = 0; k4 < nk4; ++k4) {
            float4 q[TM], t[TN], w[AXPY ? TM : 1];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                q[i] = *reinterpret_cast<const float4 *>(Qb + 16 * i * LDS_LD + k4 * 4);
                if (AXPY) w[AXPY ? i : 0] = *reinterpret_cast<const float4 *>(Wb + 16 * i * LDS_LD + k4 * 4);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) t[j] = *reinterpret_cast<const float4 *>(Tb + 16 * j * LDS_LD + k4 * 4);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float dx = q[i].x - t[j].x, dy = q[i].y - t[j].y, dz = q[i].z - t[j].z, dw = q[i].w - t[j].w;
                    if (AXPY) {
                        const float a = av[AXPY ? i : 0][AXPY ? j : 0];
                        const float4 wv = w[AXPY ? i : 0];
                        dx = fmaf(a, wv.x, dx); dy = fmaf(a, wv.y, dy);
                        dz = fmaf(a, wv.z, dz); dw = fmaf(a, wv.w, dw);
                    }
                    float v = acc[i][j];
                    if (L1) {   // the L1 contract: one add per aligned 4-group, the group summed as a tree
                        v = v + ((fabsf(dx) + fabsf(dy)) + (fabsf(dz) + fabsf(dw)));
                    } else {
                        v = acc_step<L1>(v, dx); v = acc_step<L1>(v, dy);
                        v = acc_step<L1>(v, dz); v = acc_step<L1>(v, dw);
                    }
                    acc[i][j] = v;
                }
        }

        if (s == S - 1) {
            const int64_t col0 = (int64_t)(tile_begin + ti) * BN;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int lrow = ty + 16 * i;
                const int64_t row = row0 + lrow;
                const float stv = COUNT ? st_s[lrow] : 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int64_t col = col0 + tx + 16 * j;
                    const float sc = -acc[i][j];
                    if (COUNT) cnt[i] += (sc >= stv) ? (col < d.N ? 1 : 0) : 0;
                    else if (col < d.N && row < d.B) p.out[row * p.ldo + col] = sc;
                    acc[i][j] = 0.f;
                }
            }
            if (ti + 1 < ntiles) load_av(ti + 1);
        }

        if (g + 1 < G) stage_store(buf ^ 1, g + 1);
        __syncthreads();
    }

    if (COUNT) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
            if (cnt[i]) atomicAdd(&rc[ty + 16 * i], cnt[i]);
        __syncthreads();
        if (tid < BM) {
            const int64_t row = row0 + tid;
            const int v = rc[tid];
            if (row < d.B && v) atomicAdd(&p.raw_count[row], v);
        }
    }
}

// ---- packed-FMA variant of the L2 kernels (16-byte aligned operands) ---------------------------------------------
// gfx950 issues v_fma_f32 at ~2/3 and v_pk_fma_f32 / v_pk_add_f32 (two lanes' worth of work per instruction) at
// ~0.55 of the v_add_f32 rate (tools/probe/valu_rate_probe.hip: 42.6 / 35.9 / 34.8 vs 63.9 T lane-instr/s), so
// sub + fma per (pair, k) costs 0.039 units while a packed sub + a packed fma over TWO pairs cost 0.028 per pair.
// The two halves of a packed instruction are two ADJACENT CANDIDATES at the same k (the per-pair chains stay
// "ascending k, one accumulator": bit-identical to lp_pair_score and to the scalar kernel above), the query value is
// broadcast to both halves with op_sel.  That needs the candidate tile TRANSPOSED in LDS ([k][candidate], row stride
// 132: 8 consecutive candidates of one k are two ds_read_b128) and a thread's 8 candidates contiguous (tx*8 + j).
typedef float f32x2 __attribute__((ext_vector_type(2)));

// d = q.{lo|hi} - t  (both halves)      QH: 0 -> broadcast q.x, 1 -> broadcast q.y
template <int QH>
__device__ __forceinline__ f32x2 pk_sub_bcast(f32x2 q, f32x2 t)
{
    f32x2 d;
    if (QH == 0) asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(q), "v"(t));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(q), "v"(t));
    return d;
}
// d = a * w.{lo|hi} + d
template <int WH>
__device__ __forceinline__ f32x2 pk_fma_bcast(f32x2 a, f32x2 w, f32x2 d)
{
    if (WH == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(d) : "v"(a), "v"(w));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(d) : "v"(a), "v"(w));
    return d;
}
__device__ __forceinline__ f32x2 pk_sq_acc(f32x2 d, f32x2 acc)
{
    asm("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(acc) : "v"(d));
    return acc;
}

constexpr int LDT_PK = BN + 4;      // row stride (floats) of the transposed candidate tile

template <bool AXPY, bool COUNT, int TM, int GS = 1>
__global__ __launch_bounds__(NTHREADS, 2) void lp_direct_pk_kernel(const DirectParams p)
{
    static_assert(GS == 1 || (COUNT && !AXPY), "grouped columns: plain L2 counts only");
    constexpr int BM = 16 * TM;
    constexpr int QCH = BM * 4 / NTHREADS;
    constexpr int Q_FLOATS = BM * LDS_LD, T_FLOATS = BK * LDT_PK;
    constexpr int BUF_FLOATS = Q_FLOATS * (AXPY ? 2 : 1) + T_FLOATS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const kge_lp_desc &d = p.d;
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;

    const int nblk_grid = gridDim.x, bid = blockIdx.x;
    const int xq = nblk_grid >> 3, xr = nblk_grid & 7, xcd = bid & 7, loc = bid >> 3;
    const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + loc;
    const int rp = lid % p.row_panels, cc = lid / p.row_panels;
    const int64_t row0 = (int64_t)rp * BM;
    const int tile_begin = cc * p.tiles_per_block;
    const int tile_end = min(tile_begin + p.tiles_per_block, p.col_tiles);
    const int ntiles = tile_end - tile_begin;
    if (ntiles <= 0) return;

    const int K = d.K0;                 // K % 4 == 0 (the dispatcher checks)
    const int S = (K + BK - 1) / BK;
    const int G = ntiles * S;

    float4 stQ[QCH][2], stW[AXPY ? QCH : 1][2], stT[2][2];
    const int srow = tid >> 2, skc = tid & 3;

    // unconditional loads from clamped addresses; out-of-range rows / k are zeroed when staged into LDS
    auto ld4 = [&](const float *base, int64_t ld, int64_t r, int64_t rmax, int k) {
        return *reinterpret_cast<const float4 *>(base + min(r, rmax) * ld + (k < K ? k : 0));
    };
    // (component-wise selects: a select between two float4 VALUES made hipcc index a scratch copy of them)
    auto sel4 = [](bool c, const float4 v) { return make_float4(c ? v.x : 0.f, c ? v.y : 0.f, c ? v.z : 0.f, c ? v.w : 0.f); };
    auto prefetch = [&](int g) {
        const int ti = g / S, s = g - ti * S;
        const int kq = s * BK + skc * 8;                    // query rows: 8 consecutive k per thread
        const int kt = s * BK + skc * 4;                    // candidate rows: k = 4 skc + {0..3} and 16 + 4 skc + {0..3}
        const int64_t col0 = (int64_t)(tile_begin + ti) * BN;
#pragma unroll
        for (int j = 0; j < QCH; ++j) {
            int64_t r = row0 + srow + 64 * j;
            if (p.rep) r = p.rep[min(r, d.B - 1)];          // (column -> the query row that provides it)
            stQ[j][0] = ld4(d.A0, d.lda0, r, p.rep ? INT64_MAX : d.B - 1, kq);
            stQ[j][1] = ld4(d.A0, d.lda0, r, p.rep ? INT64_MAX : d.B - 1, kq + 4);
            if (AXPY) {
                stW[j][0] = ld4(d.Wq, d.ldw, r, d.B - 1, kq);
                stW[j][1] = ld4(d.Wq, d.ldw, r, d.B - 1, kq + 4);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            stT[j][0] = ld4(d.T0, d.ldt0, col0 + srow + 64 * j, d.N - 1, kt);
            stT[j][1] = ld4(d.T0, d.ldt0, col0 + srow + 64 * j, d.N - 1, kt + 16);
        }
    };
    auto stage_store = [&](int buf, int g) {
        float *Qs = smem + buf * BUF_FLOATS;
        float *Ts = Qs + Q_FLOATS;
        float *Ws = Ts + T_FLOATS;
        const int ti = g / S, s = g - ti * S;
        const int kq = s * BK + skc * 8, kt = s * BK + skc * 4;
        const int64_t col0 = (int64_t)(tile_begin + ti) * BN;
#pragma unroll
        for (int j = 0; j < QCH; ++j) {
            const bool ok = row0 + srow + 64 * j < d.B;
            float *qd = Qs + (srow + 64 * j) * LDS_LD + skc * 8;
            *reinterpret_cast<float4 *>(qd) = sel4(ok && kq < K, stQ[j][0]);
            *reinterpret_cast<float4 *>(qd + 4) = sel4(ok && kq + 4 < K, stQ[j][1]);
            if (AXPY) {
                float *wd = Ws + (srow + 64 * j) * LDS_LD + skc * 8;
                *reinterpret_cast<float4 *>(wd) = sel4(ok && kq < K, stW[j][0]);
                *reinterpret_cast<float4 *>(wd + 4) = sel4(ok && kq + 4 < K, stW[j][1]);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {      // transposed: Ts[k][candidate]
            const int cl = srow + 64 * j;
            const bool ok = col0 + cl < d.N;
            const float4 a = sel4(ok && kt < K, stT[j][0]), b = sel4(ok && kt + 16 < K, stT[j][1]);
            float *td = Ts + (skc * 4) * LDT_PK + cl;
            td[0] = a.x; td[LDT_PK] = a.y; td[2 * LDT_PK] = a.z; td[3 * LDT_PK] = a.w;
            td += 16 * LDT_PK;
            td[0] = b.x; td[LDT_PK] = b.y; td[2 * LDT_PK] = b.z; td[3 * LDT_PK] = b.w;
        }
    };

    f32x2 acc[TM][TN / 2];
    f32x2 av[AXPY ? TM : 1][AXPY ? TN / 2 : 1];
    int cnt[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        cnt[i] = 0;
#pragma unroll
        for (int j = 0; j < TN / 2; ++j) acc[i][j] = (f32x2){0.f, 0.f};
    }

    int *rc = reinterpret_cast<int *>(smem + 2 * BUF_FLOATS);        // [GS][BM]
    float *st_s = smem + 2 * BUF_FLOATS + GS * BM;                    // [GS][BM]
    for (int idx = tid; idx < BM * GS; idx += NTHREADS) {
        const int lrow = idx / GS, gs = idx - lrow * GS;
        const int64_t row = row0 + lrow;
        int64_t q = -1;
        if (COUNT && row < d.B) q = p.qmap ? (int64_t)p.qmap[row * GS + gs] : row;
        rc[gs * BM + lrow] = 0;
        st_s[gs * BM + lrow] = q >= 0 ? p.s_true[q] : INFINITY;      // (+inf: "sc >= st" never holds -> counts nothing)
    }

    auto load_av = [&](int ti) {
        if (AXPY) {
            const int64_t col0 = (int64_t)(tile_begin + ti) * BN;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int64_t row = row0 + ty + 16 * i;
                const int64_t rsel = (d.scal_ld > 1) ? d.r_idx[min(row, d.B - 1)] : 0;
#pragma unroll
                for (int j = 0; j < TN; ++j) { // clamped, unconditional (values of padded rows/cols are unused)
                    const int64_t col = min(col0 + tx * TN + j, d.N - 1);
                    const float a = d.scal[col * d.scal_ld + rsel];
                    if (j & 1) av[AXPY ? i : 0][AXPY ? j / 2 : 0].y = a;
                    else av[AXPY ? i : 0][AXPY ? j / 2 : 0].x = a;
                }
            }
        }
    };

    prefetch(0);
    stage_store(0, 0);
    load_av(0);
    __syncthreads();

    for (int g = 0; g < G; ++g) {
        const int buf = g & 1;
        if (g + 1 < G) prefetch(g + 1);
        const int ti = g / S, s = g - ti * S;
        const int nk4 = min(BK / 4, (K - s * BK + 3) >> 2);

        const float *Qb = smem + buf * BUF_FLOATS + ty * LDS_LD;
        const float *Tb = smem + buf * BUF_FLOATS + Q_FLOATS + tx * TN;
        const float *Wb = smem + buf * BUF_FLOATS + Q_FLOATS + T_FLOATS + ty * LDS_LD;
#pragma unroll 2
        for (int k4 = 0; k4 < nk4; ++k4) {
            float4 q[TM], w[AXPY ? TM : 1];
            f32x2 t[4][TN / 2];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                q[i] = *reinterpret_cast<const float4 *>(Qb + 16 * i * LDS_LD + k4 * 4);
                if (AXPY) w[AXPY ? i : 0] = *reinterpret_cast<const float4 *>(Wb + 16 * i * LDS_LD + k4 * 4);
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const float4 a = *reinterpret_cast<const float4 *>(Tb + (k4 * 4 + kk) * LDT_PK);
                const float4 b = *reinterpret_cast<const float4 *>(Tb + (k4 * 4 + kk) * LDT_PK + 4);
                t[kk][0] = (f32x2){a.x, a.y}; t[kk][1] = (f32x2){a.z, a.w};
                t[kk][2] = (f32x2){b.x, b.y}; t[kk][3] = (f32x2){b.z, b.w};
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const f32x2 q01 = {q[i].x, q[i].y}, q23 = {q[i].z, q[i].w};
                f32x2 w01 = q01, w23 = q23;
                if (AXPY) { w01 = (f32x2){w[AXPY ? i : 0].x, w[AXPY ? i : 0].y}; w23 = (f32x2){w[AXPY ? i : 0].z, w[AXPY ? i : 0].w}; }
#pragma unroll
                for (int jj = 0; jj < TN / 2; ++jj) {
                    f32x2 v = acc[i][jj];
                    f32x2 d0 = pk_sub_bcast<0>(q01, t[0][jj]), d1 = pk_sub_bcast<1>(q01, t[1][jj]);
                    f32x2 d2 = pk_sub_bcast<0>(q23, t[2][jj]), d3 = pk_sub_bcast<1>(q23, t[3][jj]);
                    if (AXPY) {
                        const f32x2 a2 = av[AXPY ? i : 0][AXPY ? jj : 0];
                        d0 = pk_fma_bcast<0>(a2, w01, d0); d1 = pk_fma_bcast<1>(a2, w01, d1);
                        d2 = pk_fma_bcast<0>(a2, w23, d2); d3 = pk_fma_bcast<1>(a2, w23, d3);
                    }
                    v = pk_sq_acc(d0, v); v = pk_sq_acc(d1, v); v = pk_sq_acc(d2, v); v = pk_sq_acc(d3, v);
                    acc[i][jj] = v;
                }
            }
        }

        if (s == S - 1) {
            const int64_t col0 = (int64_t)(tile_begin + ti) * BN + tx * TN;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int lrow = ty + 16 * i;
                const int64_t row = row0 + lrow;
                float sc[TN];
#pragma unroll
                for (int jj = 0; jj < TN / 2; ++jj) {
                    sc[2 * jj] = -acc[i][jj].x; sc[2 * jj + 1] = -acc[i][jj].y;
                    acc[i][jj] = (f32x2){0.f, 0.f};
                }
                if (COUNT) {
#pragma unroll
                    for (int gs = 0; gs < GS; ++gs) {       // a column's scores against the true score of each of its queries
                        const float stv = st_s[gs * BM + lrow];
                        int c = 0;
#pragma unroll
                        for (int j = 0; j < TN; ++j) c += (sc[j] >= stv) ? (col0 + j < d.N ? 1 : 0) : 0;
                        if (GS == 1) cnt[i] += c;
                        else if (c) atomicAdd(&rc[gs * BM + lrow], c);
                    }
                } else if (row < d.B) {
                    float *o = p.out + row * p.ldo + col0;
                    if (col0 + TN <= d.N && p.out_vec4) {
                        *reinterpret_cast<float4 *>(o) = make_float4(sc[0], sc[1], sc[2], sc[3]);
                        *reinterpret_cast<float4 *>(o + 4) = make_float4(sc[4], sc[5], sc[6], sc[7]);
                    } else {
#pragma unroll
                        for (int j = 0; j < TN; ++j) if (col0 + j < d.N) o[j] = sc[j];
                    }
                }
            }
            if (ti + 1 < ntiles) load_av(ti + 1);
        }

        if (g + 1 < G) stage_store(buf ^ 1, g + 1);
        __syncthreads();
    }

    if (COUNT) {
        if (GS == 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                if (cnt[i]) atomicAdd(&rc[ty + 16 * i], cnt[i]);
        }
        __syncthreads();
        for (int idx = tid; idx < BM * GS; idx += NTHREADS) {
            const int lrow = idx / GS, gs = idx - lrow * GS;
            const int64_t row = row0 + lrow;
            const int v = rc[gs * BM + lrow];
            if (row < d.B && v) {
                const int64_t q = p.qmap ? (int64_t)p.qmap[row * GS + gs] : row;
                if (q >= 0) atomicAdd(&p.raw_count[q], v);
            }
        }
    }
}

template <bool AXPY, bool COUNT, int GS = 1>
int launch_pk(DirectParams &p, hipStream_t s)
{
    constexpr int TM = AXPY ? 4 : 8;
    constexpr int BM = 16 * TM;
    constexpr int BUF_FLOATS = BM * LDS_LD * (AXPY ? 2 : 1) + BK * LDT_PK;
    constexpr int SMEM_BYTES = 2 * BUF_FLOATS * 4 + 2 * GS * BM * 4;
    const kge_lp_desc &d = p.d;
    p.row_panels = (int)((d.B + BM - 1) / BM);
    p.col_tiles = (int)((d.N + BN - 1) / BN);
    const int target_blocks = kge_env_int("KGE_LP_TARGET_BLOCKS", 16384);
    int chunks = (target_blocks + p.row_panels - 1) / p.row_panels;
    if (chunks > p.col_tiles) chunks = p.col_tiles;
    if (chunks < 1) chunks = 1;
    p.tiles_per_block = (p.col_tiles + chunks - 1) / chunks;
    p.col_chunks = (p.col_tiles + p.tiles_per_block - 1) / p.tiles_per_block;
    const int grid = p.row_panels * p.col_chunks;
    auto k = lp_direct_pk_kernel<AXPY, COUNT, TM, GS>;
    static int attr_dev[16];    // per instantiation, per device
    if (int e = kge_ensure_dyn_smem(reinterpret_cast<const void *>(k), SMEM_BYTES, attr_dev)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(NTHREADS), SMEM_BYTES, s, p);
    KGE_CHECK_LAUNCH();
    return 0;
}

template <bool VEC4, bool L1, bool AXPY, bool COUNT>
int launch(DirectParams &p, hipStream_t s)
{
    constexpr int TM = AXPY ? 4 : 8;
    constexpr int BM = 16 * TM;
    constexpr int BUF_FLOATS = BM * LDS_LD * (AXPY ? 2 : 1) + BN * LDS_LD;
    constexpr int SMEM_BYTES = 2 * BUF_FLOATS * 4 + 2 * BM * 4;
    const kge_lp_desc &d = p.d;
    p.row_panels = (int)((d.B + BM - 1) / BM);
    p.col_tiles = (int)((d.N + BN - 1) / BN);
    const int target_blocks = kge_env_int("KGE_LP_TARGET_BLOCKS", 16384);
    int chunks = (target_blocks + p.row_panels - 1) / p.row_panels;
    if (chunks > p.col_tiles) chunks = p.col_tiles;
    if (chunks < 1) chunks = 1;
    p.tiles_per_block = (p.col_tiles + chunks - 1) / chunks;
    p.col_chunks = (p.col_tiles + p.tiles_per_block - 1) / p.tiles_per_block;
    const int grid = p.row_panels * p.col_chunks;

    auto k = lp_direct_kernel<VEC4, L1, AXPY, COUNT, TM>;
    static int attr_dev[16];    // per instantiation, per device
    if (int e = kge_ensure_dyn_smem(reinterpret_cast<const void *>(k), SMEM_BYTES, attr_dev)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(NTHREADS), SMEM_BYTES, s, p);
    KGE_CHECK_LAUNCH();
    return 0;
}

template <bool VEC4, bool L1>
int dispatch2(DirectParams &p, bool axpy, bool count, hipStream_t s)
{
    if (axpy) return count ? launch<VEC4, L1, true, true>(p, s) : launch<VEC4, L1, true, false>(p, s);
    return count ? launch<VEC4, L1, false, true>(p, s) : launch<VEC4, L1, false, false>(p, s);
}

} // namespace

// Rank counts of a plain KGE_LP_L2_DIRECT problem over query COLUMNS (the packed-FMA kernel): rows [0, n_single_p) carry
// one query each (col_q), rows [n_single_p, n_single_p + n_multi_p) up to 4 (members); row r is the query row rep[r].
int kge_lp_direct_count_cols(const kge_lp_desc *d, const float *s_true, int32_t *raw_count, const int64_t *rep,
                             const int32_t *col_q, int64_t n_single_p, const int32_t *members, int64_t n_multi_p,
                             hipStream_t s)
{
    if (d->mode != KGE_LP_L2_DIRECT || d->Wq || !rep || !s_true || !raw_count) return KGE_EINVAL;
    if (!((d->K0 % 4 == 0) && (d->lda0 % 4 == 0) && (d->ldt0 % 4 == 0) && kge_aligned16(d->A0) && kge_aligned16(d->T0)))
        return KGE_EINVAL;
    if (n_single_p < 0 || n_multi_p < 0 || (n_single_p > 0 && !col_q) || (n_multi_p > 0 && !members)) return KGE_EINVAL;
    if (d->N == 0) return 0;
    DirectParams p;
    p.d = *d;
    p.out = nullptr; p.ldo = 0; p.s_true = s_true; p.raw_count = raw_count; p.out_vec4 = 0;
    int rc = 0;
    if (n_single_p > 0) {
        p.d.B = n_single_p; p.rep = rep; p.qmap = col_q;
        rc = launch_pk<false, true, 1>(p, s);
        if (rc) return rc;
    }
    if (n_multi_p > 0) {
        p.d.B = n_multi_p; p.rep = rep + n_single_p; p.qmap = members;
        rc = launch_pk<false, true, 4>(p, s);
    }
    return rc;
}

int kge_lp_direct_run(const kge_lp_desc *d, float *out, int64_t ldo, const float *s_true,
                      int32_t *raw_count, hipStream_t s)
{
    if (d->B == 0 || d->N == 0) return 0;
    if ((out != nullptr) == (raw_count != nullptr)) return KGE_EINVAL;
    DirectParams p;
    p.rep = nullptr; p.qmap = nullptr;
    p.d = *d;
    p.out = out;
    p.ldo = ldo;
    p.s_true = s_true;
    p.raw_count = raw_count;
    p.out_vec4 = (out != nullptr && (ldo % 4 == 0) && kge_aligned16(out)) ? 1 : 0;
    const bool axpy = d->Wq != nullptr;
    bool vec4 = (d->K0 % 4 == 0) && (d->lda0 % 4 == 0) && (d->ldt0 % 4 == 0) &&
                kge_aligned16(d->A0) && kge_aligned16(d->T0);
    if (axpy) vec4 = vec4 && (d->ldw % 4 == 0) && kge_aligned16(d->Wq);
    const bool l1 = d->mode == KGE_LP_L1_DIRECT;
    const bool count = raw_count != nullptr;
    if (vec4 && !l1 && !kge_env_int("KGE_DIRECT_SCALAR", 0)) {      // L2 on aligned operands: the packed-FMA kernel
        if (axpy) return count ? launch_pk<true, true>(p, s) : launch_pk<true, false>(p, s);
        return count ? launch_pk<false, true>(p, s) : launch_pk<false, false>(p, s);
    }
    if (vec4) return l1 ? dispatch2<true, true>(p, axpy, count, s) : dispatch2<true, false>(p, axpy, count, s);
    return l1 ? dispatch2<false, true>(p, axpy, count, s) : dispatch2<false, false>(p, axpy, count, s);
}
