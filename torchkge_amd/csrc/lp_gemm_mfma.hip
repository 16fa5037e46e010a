// All-candidates link-prediction scores as an fp32 MFMA GEMM (gfx950).
//
//   S[i,c] = epilogue( sum_k A0[i,k]*T0[c,k]  (+ sum_k A1[i,k]*T1[c,k]) )
//
// Replaces the (b, N, d) broadcast products of the reference's
// inference_scoring_function (bilinear.py:234-240 DistMult, :514-522 ComplEx)
// and, through ||q-e||^2 = ||q||^2 + ||e||^2 - 2 q.e, the TransE-L2 case of
// interfaces.py:253-260.  Exact fp32: v_mfma_f32_32x32x2_f32 is a k-ordered
// fmaf chain, so the scalar pair kernel (kge_common.h: lp_pair_score)
// reproduces every score bit for bit -- which is what lets the fused rank path
// count `>=` without ever writing the (B,N) matrix.
//
// Structure
//  * block tile 128 queries x 128 candidates, BK = 32 per step, operands staged
//    global -> VGPR -> LDS (double buffered; row stride 36 floats makes the
//    ds_read_b128 / ds_write_b128 patterns bank-conflict free).  Lane-half h
//    of a wave reads the 4 consecutive k = 8b+4h.. of 8-block b and feeds
//    element j to MFMA j, so within every 8-block the products are accumulated
//    in the order k = 0,4,1,5,2,6,3,7 -- the fixed order lp_chain_dot restates;
//  * NWM x NWN waves per block, each owning (128/NWM) x (128/NWN) outputs as
//    MT x NT tiles of 32x32 (8 waves: 1x2 tiles, 32 accumulator VGPRs, 4
//    waves/SIMD resident; 4 waves: 2x2 tiles);
//  * the LDS write of step g+1 is issued in the MIDDLE of step g's MFMA stream
//    (its latency hides behind the remaining MFMAs), one barrier per step;
//  * persistent-style work split: the (row panel, candidate tile) space is cut
//    into gridDim.x contiguous, equal ranges (<= 2 resident blocks per CU), so
//    there is no tail round; blocks of one XCD get adjacent ranges (shared L2);
//  * rank counts live in registers across a range and are flushed with one
//    LDS reduction + one global atomic per row when the row panel changes.
#include "kge_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = 128, BN = 128, BK = 32, LDS_LD = BK + 4;
constexpr int TILE_FLOATS = BM * LDS_LD;                     // one operand tile
constexpr int SMEM_BYTES = 4 * TILE_FLOATS * 4 + 6 * BM * 4; // 2 bufs x (A,B) + row counters, s_true, qn, (p, z, X row) of the PROJ modes

struct GemmParams {
    kge_lp_desc d;
    float *out;
    int64_t ldo;
    const float *s_true;
    int *raw_count;
    int row_panels, col_tiles;
    int64_t n_items;   // row_panels * col_tiles
    int dbg;           // env KGE_DBG: 1 = every block loads tile (0,0) (cache-ceiling probe, wrong results)
};

template <bool VEC4, bool COUNT, int MODE, int NWM, int NWN>
__global__ __launch_bounds__(64 * NWM * NWN, (NWM * NWN) / 2) void lp_gemm_kernel(const GemmParams p)
{
    constexpr int NTHREADS = 64 * NWM * NWN;
    constexpr int MT = BM / (32 * NWM), NT = BN / (32 * NWN); // 32x32 tiles per wave
    constexpr bool WRITE = !COUNT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const kge_lp_desc &d = p.d;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wr = wid / NWN, wc = wid % NWN;
    const int l31 = lane & 31, half = lane >> 5;

    // contiguous, balanced range of (row panel, tile) items for this block;
    // blocks resident on one XCD (bid % 8) get adjacent ranges
    const int nb = gridDim.x, bid = blockIdx.x;
    const int xq = nb >> 3, xr = nb & 7, xcd = bid & 7, loc = bid >> 3;
    const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + loc;
    const int64_t item_begin = p.n_items * lid / nb, item_end = p.n_items * (lid + 1) / nb;
    const int nitems = (int)(item_end - item_begin);
    if (nitems <= 0) return;

    const int nb0 = (d.K0 + 7) >> 3, nb1 = (d.K1 + 7) >> 3;
    const int steps0 = (nb0 + 3) >> 2, steps1 = (nb1 + 3) >> 2;
    const int S = steps0 + steps1;
    const int G = nitems * S;

    // staging: a 16-byte piece (4 consecutive k) per lane; 8 consecutive lanes
    // cover one full 128-byte row segment, so a wave-wide dwordx4 load touches
    // 8 whole lines instead of 16 rows x scattered 16-byte pieces
    constexpr int PIECES = 1024 / NTHREADS;       // pieces per thread per operand
    constexpr int ROWS_PER_PASS = NTHREADS / 8;
    const int srow0 = tid >> 3, spc = tid & 7;
    float4 stA[PIECES], stB[PIECES];
    bool st_kok[4] = {true, true, true, true};    // k validity of the staged piece's 4 elements

    // Prefetch stream: walks (item, step) in order with incrementally updated
    // row pointers (item decode / 64-bit address math once per tile, not per
    // step).  Loads are UNCONDITIONAL from clamped, always-valid addresses so
    // all of a thread's loads are in flight together (a guarded load makes
    // hipcc branch around it and wait vmcnt(0) on the spot).  Rows beyond B / N
    // are clamped duplicates whose results the epilogue never uses; only k >= K
    // (last step of a segment) must be zeroed when staged into LDS.
    int pf_it = 0, pf_s = 0, pf_k = spc * 4, pf_K = d.K0;
    const float *pfA[PIECES], *pfB[PIECES];       // row bases of the current item / segment
    int64_t pf_ra[PIECES], pf_rb[PIECES];
    auto pf_new_item = [&]() {
        const int item = (int)item_begin + pf_it;
        const int64_t row0 = (int64_t)(item / p.col_tiles) * BM, col0 = (int64_t)(item % p.col_tiles) * BN;
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            pf_ra[j] = min(row0 + srow0 + ROWS_PER_PASS * j, d.B - 1);
            pf_rb[j] = min(col0 + srow0 + ROWS_PER_PASS * j, d.N - 1);
            if (p.dbg & 1) { pf_ra[j] = srow0; pf_rb[j] = srow0; }
            pfA[j] = d.A0 + pf_ra[j] * d.lda0;
            pfB[j] = d.T0 + pf_rb[j] * d.ldt0;
        }
        pf_k = spc * 4;
        pf_K = d.K0;
    };
    auto prefetch = [&]() { // loads step (pf_it, pf_s), then advances the stream
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            if (VEC4) {
                const int kc = (pf_k < pf_K) ? pf_k : 0;
                stA[j] = *reinterpret_cast<const float4 *>(pfA[j] + kc);
                stB[j] = *reinterpret_cast<const float4 *>(pfB[j] + kc);
            } else { // K or leading dims not multiples of 4: scalar loads, still unconditional
                const int k0 = (pf_k < pf_K) ? pf_k : 0, k1 = (pf_k + 1 < pf_K) ? pf_k + 1 : 0,
                          k2 = (pf_k + 2 < pf_K) ? pf_k + 2 : 0, k3 = (pf_k + 3 < pf_K) ? pf_k + 3 : 0;
                stA[j] = make_float4(pfA[j][k0], pfA[j][k1], pfA[j][k2], pfA[j][k3]);
                stB[j] = make_float4(pfB[j][k0], pfB[j][k1], pfB[j][k2], pfB[j][k3]);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) st_kok[e] = pf_k + e < pf_K;
        pf_k += BK;
        if (++pf_s == steps0 && steps1 > 0) { // switch to segment 1 of the same item
#pragma unroll
            for (int j = 0; j < PIECES; ++j) {
                pfA[j] = d.A1 + pf_ra[j] * d.lda1;
                pfB[j] = d.T1 + pf_rb[j] * d.ldt1;
            }
            pf_k = spc * 4;
            pf_K = d.K1;
        }
        if (pf_s == S) {
            pf_s = 0;
            if (++pf_it < nitems) pf_new_item();
        }
    };
    auto stage_store = [&](int buf) {
        float *As = smem + buf * 2 * TILE_FLOATS;
        float *Bs = As + TILE_FLOATS;
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            float4 a = stA[j], b = stB[j];
            if (!st_kok[3]) { // partial / absent piece: zero the k >= K elements (rare, wave-divergent at most)
                a.x = st_kok[0] ? a.x : 0.f; a.y = st_kok[1] ? a.y : 0.f; a.z = st_kok[2] ? a.z : 0.f; a.w = 0.f;
                b.x = st_kok[0] ? b.x : 0.f; b.y = st_kok[1] ? b.y : 0.f; b.z = st_kok[2] ? b.z : 0.f; b.w = 0.f;
            }
            *reinterpret_cast<float4 *>(As + (srow0 + ROWS_PER_PASS * j) * LDS_LD + spc * 4) = a;
            *reinterpret_cast<float4 *>(Bs + (srow0 + ROWS_PER_PASS * j) * LDS_LD + spc * 4) = b;
        }
    };

    f32x16 acc[MT][NT];
    unsigned cnt[MT][4]; // 4 x 8-bit counters per register (row r -> byte r & 3 of cnt[mt][r >> 2])
    int tiles_since_flush = 0;
#pragma unroll
    for (int a = 0; a < MT; ++a) {
#pragma unroll
        for (int r = 0; r < 4; ++r) cnt[a][r] = 0u;
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    }

    // per-row vectors of the current row panel (LDS, read as b128 in the epilogue)
    int *rc = reinterpret_cast<int *>(smem + 4 * TILE_FLOATS);
    float *st_s = smem + 4 * TILE_FLOATS + BM;
    float *qn_s = st_s + BM;
    float *p_s = qn_s + BM, *z_s = p_s + BM;                 // PROJ modes: per-query (p_i, z_i)
    int *xo_s = reinterpret_cast<int *>(z_s + BM);           //             and their row r_i of X
    constexpr bool EXPANDED = MODE != KGE_LP_DOT;            // qn / en enter the epilogue
    constexpr bool PROJ = MODE >= KGE_LP_L2_PROJH;
    auto load_panel = [&](int64_t row0) {
        if (tid < BM) {
            const int64_t row = row0 + tid;
            rc[tid] = 0;
            st_s[tid] = (COUNT && row < d.B) ? p.s_true[row] : 0.f;
            qn_s[tid] = (EXPANDED && row < d.B) ? d.qn[row] : 0.f;
            if (PROJ) {
                const int64_t rc = row < d.B ? row : d.B - 1;
                p_s[tid] = d.Wq[rc * d.ldw];
                z_s[tid] = d.Wq[rc * d.ldw + 1];
                xo_s[tid] = (int)d.r_idx[rc];
            }
        }
    };
    auto flush_counts = [&](int64_t row0) { // block-uniform call sites only
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int v = (int)((cnt[mt][r >> 2] >> (8 * (r & 3))) & 0xffu);
                if (v) atomicAdd(&rc[(wr * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half], v);
            }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) cnt[mt][q] = 0u;
        tiles_since_flush = 0;
        __syncthreads();
        if (tid < BM) {
            const int64_t row = row0 + tid;
            const int v = rc[tid];
            rc[tid] = 0;
            if (row < d.B && v) atomicAdd(&p.raw_count[row], v);
        }
        __syncthreads();
    };

    int64_t cur_row0 = (item_begin / p.col_tiles) * BM;
    load_panel(cur_row0);
    pf_new_item();
    prefetch();
    stage_store(0);
    __syncthreads();

// fragment loads (LDS -> VGPR) and the 4*MT*NT MFMAs of one 8-block, kept apart so
// the loads of block b+1 can be issued BEFORE the MFMAs of block b (software
// pipelining: LDS latency hides behind 1024+ cycles of matrix work)
#define KGE_LOAD(AF, BF, BLK)                                                                       \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                               \
        AF[mt] = *reinterpret_cast<const float4 *>(Ab + mt * 32 * LDS_LD + (BLK) * 8);              \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                               \
        BF[nt] = *reinterpret_cast<const float4 *>(Bb + nt * 32 * LDS_LD + (BLK) * 8);
#define KGE_MMA(AF, BF)                                                                             \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                               \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                           \
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(AF[mt].x, BF[nt].x, acc[mt][nt], 0, 0, 0); \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                               \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                           \
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(AF[mt].y, BF[nt].y, acc[mt][nt], 0, 0, 0); \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                               \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                           \
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(AF[mt].z, BF[nt].z, acc[mt][nt], 0, 0, 0); \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                               \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                           \
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(AF[mt].w, BF[nt].w, acc[mt][nt], 0, 0, 0);
#define KGE_BLOCK(BLK)                                                                              \
    {                                                                                               \
        float4 af[MT], bf[NT];                                                                      \
        KGE_LOAD(af, bf, BLK)                                                                       \
        KGE_MMA(af, bf)                                                                             \
    }

    float en_pref[NT], y_pref[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { en_pref[nt] = 0.f; y_pref[nt] = 0.f; }
    int it = 0, s = 0;
    for (int g = 0; g < G; ++g) {
        const int buf = g & 1;
        const bool more = g + 1 < G;
        if (more && !(p.dbg & 2)) prefetch();

        const bool seg1 = s >= steps0;
        const int kb0 = (seg1 ? s - steps0 : s) << 2;
        const int nblk = min(4, (seg1 ? nb1 : nb0) - kb0);
        if (EXPANDED && s == S - 1) { // ||e_c||^2 of this tile's columns: issued a whole
            const int item = (int)item_begin + it;    // step ahead of the epilogue that consumes them
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int64_t col = (int64_t)(item % p.col_tiles) * BN + wc * NT * 32 + nt * 32 + l31;
                en_pref[nt] = d.en[min(col, d.N - 1)]; // clamped, unconditional
                if (MODE == KGE_LP_L2_PROJD) y_pref[nt] = d.yc[min(col, d.N - 1)];
            }
        }

        const float *Ab = smem + buf * 2 * TILE_FLOATS + (wr * MT * 32 + l31) * LDS_LD + half * 4;
        const float *Bb = smem + buf * 2 * TILE_FLOATS + TILE_FLOATS + (wc * NT * 32 + l31) * LDS_LD + half * 4;
        if (nblk == 4) { // full step; the LDS write of the next step hides behind the last MFMA block
            float4 af0[MT], bf0[NT], af1[MT], bf1[NT];
            KGE_LOAD(af0, bf0, 0)
            KGE_LOAD(af1, bf1, 1)
            __builtin_amdgcn_sched_barrier(0);
            KGE_MMA(af0, bf0)
            __builtin_amdgcn_sched_barrier(0);
            KGE_LOAD(af0, bf0, 2)
            __builtin_amdgcn_sched_barrier(0);
            KGE_MMA(af1, bf1)
            __builtin_amdgcn_sched_barrier(0);
            KGE_LOAD(af1, bf1, 3)
            __builtin_amdgcn_sched_barrier(0);
            KGE_MMA(af0, bf0)
            __builtin_amdgcn_sched_barrier(0);
            if (more && !(p.dbg & 4)) stage_store(buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            KGE_MMA(af1, bf1)
        } else {         // K tail (wave-uniform)
            if (more) stage_store(buf ^ 1);
            KGE_BLOCK(0)
            if (nblk > 1) KGE_BLOCK(1)
            if (nblk > 2) KGE_BLOCK(2)
        }

        if (s == S - 1) { // tile finished: branch-free epilogue
            const int item = (int)item_begin + it;
            const int64_t colb = (int64_t)(item % p.col_tiles) * BN + wc * NT * 32;
            float enr[NT];
            int cmask[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int64_t col = colb + nt * 32 + l31;
                cmask[nt] = col < d.N ? 1 : 0;
                enr[nt] = EXPANDED ? en_pref[nt] : 0.f;
            }
            if constexpr (PROJ) {
                // projection modes: per (tile row group of 4 rows) fetch the rows' scalars once, issue the
                // 4 x NT gathers X[r_i, c] together (each a 128-byte segment per lane half), then combine
                int64_t colc[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) colc[nt] = min(colb + nt * 32 + l31, d.N - 1);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int lrow0 = (wr * MT + mt) * 32 + 8 * gq + 4 * half;
                        const float4 stv = COUNT ? *reinterpret_cast<const float4 *>(st_s + lrow0) : make_float4(0.f, 0.f, 0.f, 0.f);
                        const float4 qnv = *reinterpret_cast<const float4 *>(qn_s + lrow0);
                        const float4 pv = *reinterpret_cast<const float4 *>(p_s + lrow0);
                        const float4 zv = *reinterpret_cast<const float4 *>(z_s + lrow0);
                        const int4 xov = *reinterpret_cast<const int4 *>(xo_s + lrow0);
                        const float *xr[4] = {d.scal + (int64_t)xov.x * d.scal_ld, d.scal + (int64_t)xov.y * d.scal_ld,
                                              d.scal + (int64_t)xov.z * d.scal_ld, d.scal + (int64_t)xov.w * d.scal_ld};
                        const float qn4[4] = {qnv.x, qnv.y, qnv.z, qnv.w}, st4[4] = {stv.x, stv.y, stv.z, stv.w};
                        const float p4[4] = {pv.x, pv.y, pv.z, pv.w}, z4[4] = {zv.x, zv.y, zv.z, zv.w};
                        float xv[NT][4];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int e = 0; e < 4; ++e) xv[nt][e] = xr[e][colc[nt]];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int r = gq * 4 + e;
                                const float v0 = fmaf(-2.0f, acc[mt][nt][r], qn4[e] + enr[nt]);
                                const float sc = lp_epilogue_proj(MODE, v0, xv[nt][e], y_pref[nt], p4[e], z4[e]);
                                if (WRITE) {
                                    const int64_t row = cur_row0 + lrow0 + e;
                                    if (cmask[nt] && row < d.B) p.out[row * p.ldo + colb + nt * 32 + l31] = sc;
                                }
                                if (COUNT) cnt[mt][gq] += (sc >= st4[e]) ? (unsigned)cmask[nt] << (8 * e) : 0u;
                                acc[mt][nt][r] = 0.f;
                            }
                        }
                    }
                }
            } else {
    #pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    float4 stv[4], qnv[4]; // rows 8*gq + 4*half + {0..3} of this 32-row tile
    #pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int lrow0 = (wr * MT + mt) * 32 + 8 * gq + 4 * half;
                        stv[gq] = COUNT ? *reinterpret_cast<const float4 *>(st_s + lrow0) : make_float4(0.f, 0.f, 0.f, 0.f);
                        qnv[gq] = EXPANDED ? *reinterpret_cast<const float4 *>(qn_s + lrow0)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
    #pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
    #pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int gq = r >> 2, e = r & 3;
                            const float qn = e == 0 ? qnv[gq].x : (e == 1 ? qnv[gq].y : (e == 2 ? qnv[gq].z : qnv[gq].w));
                            const float st = e == 0 ? stv[gq].x : (e == 1 ? stv[gq].y : (e == 2 ? stv[gq].z : stv[gq].w));
                            const float sc = lp_epilogue(MODE, acc[mt][nt][r], qn, enr[nt]);
                            if (WRITE) {
                                const int64_t row = cur_row0 + (wr * MT + mt) * 32 + e + 8 * gq + 4 * half;
                                if (cmask[nt] && row < d.B) p.out[row * p.ldo + colb + nt * 32 + l31] = sc;
                            }
                            if (COUNT) cnt[mt][gq] += (sc >= st) ? (unsigned)cmask[nt] << (8 * e) : 0u;
                            acc[mt][nt][r] = 0.f;
                        }
                    }
                }
            }
            // row panel change (block-uniform): flush counts, load the next panel's vectors
            if (more) {
                const int64_t next_row0 = (int64_t)((item + 1) / p.col_tiles) * BM;
                if (next_row0 != cur_row0) {
                    if (COUNT) flush_counts(cur_row0);
                    else __syncthreads();
                    cur_row0 = next_row0;
                    load_panel(cur_row0);
                } else if (COUNT && ++tiles_since_flush >= 255 / NT) { // 8-bit counters: +NT at most per tile
                    flush_counts(cur_row0);
                }
            }
        }
        if (++s == S) { s = 0; ++it; }
        if (!(p.dbg & 8)) __syncthreads();
    }
#undef KGE_BLOCK
#undef KGE_MMA
#undef KGE_LOAD

    if (COUNT) flush_counts(cur_row0);
}

template <bool VEC4, bool COUNT, int MODE, int NWM, int NWN>
int launch(const GemmParams &p, int grid, hipStream_t s)
{
    auto k = lp_gemm_kernel<VEC4, COUNT, MODE, NWM, NWN>;
    static int attr_dev[16];    // per instantiation, per device
    if (int e = kge_ensure_dyn_smem(reinterpret_cast<const void *>(k), SMEM_BYTES, attr_dev)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * NWM * NWN), SMEM_BYTES, s, p);
    KGE_CHECK_LAUNCH();
    return 0;
}

template <bool COUNT, int NWM, int NWN>
int dispatch(const GemmParams &p, bool vec4, int grid, hipStream_t s)
{
    if (p.d.mode == KGE_LP_DOT)
        return vec4 ? launch<true, COUNT, KGE_LP_DOT, NWM, NWN>(p, grid, s)
                    : launch<false, COUNT, KGE_LP_DOT, NWM, NWN>(p, grid, s);
    if (p.d.mode == KGE_LP_L2_PROJH)
        return vec4 ? launch<true, COUNT, KGE_LP_L2_PROJH, NWM, NWN>(p, grid, s)
                    : launch<false, COUNT, KGE_LP_L2_PROJH, NWM, NWN>(p, grid, s);
    if (p.d.mode == KGE_LP_L2_PROJD)
        return vec4 ? launch<true, COUNT, KGE_LP_L2_PROJD, NWM, NWN>(p, grid, s)
                    : launch<false, COUNT, KGE_LP_L2_PROJD, NWM, NWN>(p, grid, s);
    return vec4 ? launch<true, COUNT, KGE_LP_L2_EXPAND, NWM, NWN>(p, grid, s)
                : launch<false, COUNT, KGE_LP_L2_EXPAND, NWM, NWN>(p, grid, s);
}

int num_cus()
{
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

} // namespace

// Internal entry shared by kge_lp_scores / kge_lp_count_ge for the MFMA modes.
int kge_lp_gemm_run(const kge_lp_desc *d, float *out, int64_t ldo, const float *s_true,
                    int32_t *raw_count, hipStream_t s)
{
    if (d->B == 0 || d->N == 0) return 0;
    if ((out != nullptr) == (raw_count != nullptr)) return KGE_EINVAL;
    GemmParams p;
    p.d = *d;
    p.out = out;
    p.ldo = ldo;
    p.s_true = s_true;
    p.raw_count = raw_count;
    p.row_panels = (int)((d->B + BM - 1) / BM);
    p.col_tiles = (int)((d->N + BN - 1) / BN);
    p.n_items = (int64_t)p.row_panels * p.col_tiles;
    p.dbg = kge_env_int("KGE_DBG", 0);
    // one balanced range per resident block slot: 2 blocks (75 KB LDS each) per CU
    const int slots = num_cus() * 2 * kge_env_int("KGE_LP_ROUNDS", 1);
    const int grid = (int)(p.n_items < slots ? p.n_items : slots);

    bool vec4 = (d->K0 % 4 == 0) && (d->lda0 % 4 == 0) && (d->ldt0 % 4 == 0) &&
                kge_aligned16(d->A0) && kge_aligned16(d->T0);
    if (d->K1 > 0)
        vec4 = vec4 && (d->K1 % 4 == 0) && (d->lda1 % 4 == 0) && (d->ldt1 % 4 == 0) &&
               kge_aligned16(d->A1) && kge_aligned16(d->T1);

    const bool count = raw_count != nullptr;
    // 4 waves (2x2 MFMA tiles each) measured ~1% ahead of 8 waves (1x2) on MI355X and has no spills
    if (kge_env_int("KGE_LP_WAVES", 4) == 4)
        return count ? dispatch<true, 2, 2>(p, vec4, grid, s) : dispatch<false, 2, 2>(p, vec4, grid, s);
    return count ? dispatch<true, 4, 2>(p, vec4, grid, s) : dispatch<false, 4, 2>(p, vec4, grid, s);
}
