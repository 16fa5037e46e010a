// The one-product level for LONG rows (K > 512: ComplEx d = 512, BASELINE cfg5) -- the free-running kernel of
// lp_hi_stream.hip with the query panel STREAMED through LDS instead of resident there (r06).
//
// lp_hi_stream_kernel keeps a 96-query panel resident in LDS for a whole sweep of the candidate tiles: 96 rows x
// (32 B x units + 16) bytes, which stops fitting the 160 KB of a CU at 52 k16 units (K = 816).  Here the panel's K
// extent is cut into CHUNKS of CU units that pass through a two-slot LDS ring:
//
//   * a wave still owns 64 CANDIDATE rows of the tile and reads their MFMA A-fragments straight from the
//     fragment-major table into a register ring (one coalesced 1-KiB global_load_dwordx4 per (32 rows, unit), RING - 1
//     units ahead) -- the streamed operand never touches LDS, exactly as in lp_hi_stream.hip;
//   * the QUERY operand of chunk c + 1 is copied global -> registers -> LDS slot (c + 1) & 1 by all waves while they
//     compute on chunk c out of slot c & 1: one 16-byte piece per thread behind each of the chunk's first units, written
//     two units later; ONE block-wide barrier per chunk (in front of the chunk's last unit) hands the slots over.  A block
//     keeps ITS panel for a whole candidate sweep, so the chunk stream simply cycles 0 .. NCH-1, 0 .. through the panel
//     (the last chunk of an item prefetches chunk 0 of the next item's panel, the same or a new one);
//   * 78 (NT = 3) or 104 (NT = 4) MFMAs per wave between two barriers; inside a chunk the waves run free as before
//     (waits placed by the compiler: there is no LDS-DMA in the kernel);
//   * with the panel streamed, its height is no longer bound by the LDS: NT = 4 (128 queries, 128 accumulator VGPRs)
//     reuses every candidate fragment four times instead of three -- candidate bytes from the L2 per MFMA 256 instead of
//     341 (+ 64 / 85 of query chunk per MFMA), the co-limit of this shape: 4.6 M candidate rows x 2 KB are streamed from
//     HBM once per group of query panels and read from the L2 by every block of the XCD.
//
// Epilogue, uncertain-pair lists, work order: lp_hi_stream.hip's (PM = 0, one global list).
#include "kge_common.h"
#include <type_traits>
#ifndef KGE_BUILD_NO_SLP
#error "build with -fno-slp-vectorize -DKGE_BUILD_NO_SLP=1 (torchkge_amd/csrc/build.py): SLP-packed v_pk_fma_f32 with a lane-crossing op_sel misreads beside co-executing MFMAs (profiles/r06/slp_bisect.txt)"
#endif

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int HC_MT = 2;
constexpr int HC_WROWS = 64;                    // candidate rows per wave
constexpr int HC_WLIST = 384;                   // uncertain pairs buffered per wave (int2 entries), NT sub-lists

#define HC_SGB(mask) __builtin_amdgcn_sched_group_barrier(mask, 1, 0)

// BDB: 2 = the query fragments of unit g + 1 fetched into a second register set during unit g; 1 = ONE set, the fragment of
// sub-tile nt re-fetched right behind the two MFMAs that read it (NT = 4: 16 registers fewer, and the 128 accumulators fit)
template <int NW, int NT, int UNITS, int CU, int RING, int BDB>
__global__ __launch_bounds__(64 * NW, 2) void lp_hi_chunk_kernel(const kge_hi_stream_params p)
{
    constexpr int NTHREADS = 64 * NW;
    constexpr int TQ = NT * 32;
    constexpr int PF = RING - 1;
    constexpr int NCH = (UNITS + CU - 1) / CU;
    constexpr int CUL = UNITS - (NCH - 1) * CU;     // units of the last chunk
    constexpr int RSC = CU * 32 + 16;               // slot row stride (bytes): an odd number of 16-byte pieces
    constexpr int BUFB = TQ * RSC;                  // bytes of one slot
    constexpr int RPP = NTHREADS / 32;              // panel rows per staging pass (32 lanes x 16 B cover a row's chunk)
    constexpr int S = TQ / RPP;                     // staging pieces per thread per chunk
    constexpr int SUBLIST = HC_WLIST / NT;
    static_assert(TQ % RPP == 0, "staging passes");
    static_assert(2 * CU <= 32, "a row's chunk is staged by 32 lanes");
    static_assert(CUL >= S + 3 && CU >= S + 3, "staging (loads at units 0..S-1, stores two units later) ends before the barrier");
    static_assert(UNITS > PF, "prefetch ring");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    int2 *wlist = reinterpret_cast<int2 *>(smem + 2 * BUFB) + wid * HC_WLIST;

    // ---- work order: lp_hi_stream_kernel's (QG panels interleaved under one sweep of the candidate tiles)
    const int QG = p.qg;
    const int nb = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3;
    const int nbx = (nb - xcd + 7) >> 3;
    const int nx = nb < 8 ? nb : 8;
    const int64_t x_begin = p.n_items * xcd / nx, x_end = p.n_items * (xcd + 1) / nx;
    const int64_t item_begin = x_begin + loc;
    const int nitems = item_begin < x_end ? (int)((x_end - item_begin + nbx - 1) / nbx) : 0;
    if (nitems <= 0) return;
    const int full_panels = (p.q_panels / QG) * QG;
    const int64_t full_items = (int64_t)full_panels * p.c_tiles;
    auto item_qp_ct = [&](int i, int &qp, int &ct) __attribute__((always_inline)) {
        int64_t idx = item_begin + (int64_t)i * nbx;
        int base = 0, gsz = QG;
        if (idx < full_items) {
            const int64_t per = (int64_t)QG * p.c_tiles;
            const int grp = (int)(idx / per);
            idx -= grp * per;
            base = grp * QG;
        } else {
            idx -= full_items;
            base = full_panels;
            const int rem = p.q_panels - full_panels;
            gsz = 1;
            for (int sz = QG >> 1; sz >= 1; sz >>= 1) {
                if (rem & sz) {
                    if (idx < (int64_t)sz * p.c_tiles) { gsz = sz; break; }
                    idx -= (int64_t)sz * p.c_tiles;
                    base += sz;
                }
            }
        }
        ct = (int)(idx / gsz);
        qp = base + (int)(idx - (int64_t)ct * gsz);
    };

    const int n_groups32 = (int)(p.rows_p >> 5);
    const int64_t gstride = (int64_t)p.units_p << 10;               // bytes per 32-row group of the fragment-major table
    auto tile_ptr = [&](int ct, bool &active) __attribute__((always_inline)) -> const char * {
        int g = ct * (NW * 2) + wid * 2;
        active = g + 1 < n_groups32;
        g = min(g, n_groups32 - 2);                                 // (past the table: valid rows, results dropped)
        return p.Ef + g * gstride;
    };
    const unsigned lane16 = lane * 16;

    f32x16 acc[HC_MT][NT];
    f16x8 A[RING][HC_MT], Bf[BDB][NT];
    uint4 stg[3];
    int cnt[NT], nl[NT];
    float alo[NT], ahi[NT];
    int qid[NT], tru[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { cnt[nt] = 0; nl[nt] = 0; }

    auto load_thresholds = [&](int64_t q0) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int64_t col = q0 + nt * 32 + l31;
            int64_t q = -1;
            if (col < p.q_rows) q = p.col_q ? (int64_t)p.col_q[col] : col;
            if (q >= p.B) q = -1;
            float2 t = make_float2(INFINITY, INFINITY);
            if (q >= 0) t = p.thr[q];
            alo[nt] = t.x; ahi[nt] = t.y; qid[nt] = (int)q;
            tru[nt] = (p.true_idx && q >= 0) ? (int)(p.true_idx[q] - p.c_base) : -1;
        }
    };
    auto flush_counts = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int v = cnt[nt] + __shfl_xor(cnt[nt], 32, 64);
            if (half == 0 && v != 0 && qid[nt] >= 0) atomicAdd(&p.raw_count[qid[nt]], v);
            cnt[nt] = 0;
        }
    };
    auto flush_all = [&]() __attribute__((always_inline)) {
        int total = 0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) total += nl[nt];
        if (total > 0) {
            int base = 0;
            if (lane == 0) base = atomicAdd(p.list_count, total);
            base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                for (int i = lane; i < nl[nt]; i += 64) {
                    const int pos = base + i;
                    if ((unsigned)pos < (unsigned)p.cap) reinterpret_cast<int2 *>(p.list)[pos] = wlist[nt * SUBLIST + i];
                    else *p.overflow = 1.0f;
                }
                base += nl[nt];
                nl[nt] = 0;
            }
        }
    };

    // ---- the chunk ring.  Readers: fragment (row nt * 32 + l31, unit u of the chunk, k-half `half`) of slot `cur`;
    // writers: thread (srow, spiece) copies 16-byte piece spiece of rows srow + RPP * s, s = 0 .. S-1, into slot cur ^ 1
    const unsigned b_lane = (unsigned)(l31 * RSC + half * 16);
    unsigned b_cur = b_lane;                                        // slot 0
    const int srow = tid >> 5, spiece = tid & 31;
    const unsigned st_lane = (unsigned)(srow * RSC + spiece * 16);
    unsigned st_nxt = st_lane + BUFB;                               // slot 1
    auto load_B1 = [&](f16x8 &dst, int nt, int u) __attribute__((always_inline)) {
        dst = *reinterpret_cast<const f16x8 *>(smem + b_cur + nt * 32 * RSC + u * 32);
    };
    auto load_B = [&](f16x8 (&dst)[NT], int u) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) load_B1(dst[nt], nt, u);
    };
    auto load_A = [&](f16x8 (&dst)[HC_MT], const char *tp, int u) __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < HC_MT; ++mt)
            dst[mt] = *reinterpret_cast<const f16x8 *>(tp + mt * gstride + (u << 10) + lane16);
    };
    // piece s of chunk `ch` (`cuc` units) of the panel that starts at query row q0 (rows past the operand: its last row --
    // their thresholds are +inf)
    // (every lane loads -- lanes past the chunk's pieces re-read its last piece, the same cache line -- and only the store is
    // predicated: a conditionally assigned staging register made hipcc keep the three of them in scratch memory)
    auto stage_load = [&](int64_t q0, int ch, int cuc, int s) __attribute__((always_inline)) -> uint4 {
        const int64_t row = min(q0 + srow + RPP * s, p.q_rows - 1);
        return *reinterpret_cast<const uint4 *>(p.Qh + row * p.q_row_bytes + ch * (CU * 32) + min(spiece, 2 * cuc - 1) * 16);
    };
    auto stage_store = [&](const uint4 &src, unsigned st_base, int cuc, int s) __attribute__((always_inline)) {
        if (spiece < 2 * cuc) *reinterpret_cast<uint4 *>(smem + st_base + s * (RPP * RSC)) = src;
    };

    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;

    int qp_cur, ct_cur;
    item_qp_ct(0, qp_cur, ct_cur);
    int64_t cur_q0 = (int64_t)qp_cur * TQ;
    bool act_cur;
    const char *tp_cur = tile_ptr(ct_cur, act_cur);
#pragma unroll
    for (int u = 0; u < PF; ++u) load_A(A[u], tp_cur, u);
    {   // chunk 0 of the first panel into slot 0
        constexpr int cu0 = NCH == 1 ? CUL : CU;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            stage_store(stage_load(cur_q0, 0, cu0, s), st_lane, cu0, s);
        }
    }
    load_thresholds(cur_q0);
    __syncthreads();
    load_B(Bf[0], 0);

    for (int it = 0; it < nitems; ++it) {
        // the next item is known in front of the sweep: its panel's chunk 0 is staged under this item's last chunk
        const bool more = it + 1 < nitems;
        int qp_next = qp_cur, ct_next = ct_cur;
        if (more) item_qp_ct(it + 1, qp_next, ct_next);
        const int64_t next_q0 = (int64_t)qp_next * TQ;

        // ---- the K sweep of one wave tile, fully unrolled: per unit NT x 2 MFMAs, NT ds_read_b128 (queries of the next
        // unit), 2 global loads (candidates RING - 1 units ahead), at most one staging load and one staging store
        // RESTAGE = false (rows of ONE chunk only, NCH == 1: the panel is RESIDENT in its slot): the next item sweeps the same
        // panel -- no staging, no barrier, no slot change; the waves of the block run free across items as in lp_hi_stream.hip
        auto sweep = [&](auto rs_tag) __attribute__((always_inline)) {
        constexpr bool RESTAGE = decltype(rs_tag)::value;
#pragma unroll
        for (int g = 0; g < UNITS; ++g) {
            const int c = g / CU, uc = g - c * CU;
            const int cuc = c == NCH - 1 ? CUL : CU;
            const bool last_in_chunk = uc == cuc - 1;
            const int cn = c + 1 < NCH ? c + 1 : 0;                 // the chunk being staged
            const int cucn = cn == NCH - 1 ? CUL : CU;
            if (last_in_chunk && RESTAGE) {
                // slot cur ^ 1 is complete (every thread's stores), slot cur has been read for the last time (the fragments
                // of this unit were fetched one unit ago): hand over
                __syncthreads();
                b_cur = (2 * b_lane + BUFB) - b_cur;
                st_nxt = (2 * st_lane + BUFB) - st_nxt;
            }
            const bool has_B = g + 1 < UNITS, has_A = g + PF < UNITS;
            const bool st_w = RESTAGE && !last_in_chunk && uc >= 2 && uc - 2 < S, st_r = RESTAGE && !last_in_chunk && uc < S;
            const int un = last_in_chunk ? 0 : uc + 1;              // unit g + 1 inside ITS chunk (slot `cur` after the hand-over)
            if (BDB == 2) {
                if (has_B) load_B(Bf[(g + 1) & 1], un);
                if (has_A) load_A(A[(g + PF) % RING], tp_cur, g + PF);
                if (st_w) stage_store(stg[(uc - 2) % 3], st_nxt, cucn, uc - 2);
                if (st_r) stg[uc % 3] = stage_load(c + 1 < NCH ? cur_q0 : next_q0, cn, cucn, uc);
#pragma unroll
                for (int mt = 0; mt < HC_MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[g % RING][mt], Bf[g & 1][nt],
                                                                            g == 0 ? zero16 : acc[mt][nt], 0, 0, 0);
                // interleave: one memory operation behind each MFMA
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    HC_SGB(0x008);
                    if (has_B) HC_SGB(0x100);
                }
#pragma unroll
                for (int mt = 0; mt < HC_MT; ++mt) {
                    HC_SGB(0x008);
                    if (has_A) HC_SGB(0x020);
                }
                HC_SGB(0x008);
                if (st_r) HC_SGB(0x020);
                if (NT * HC_MT - NT - HC_MT > 1) {
#pragma unroll
                    for (int i = 0; i < NT * HC_MT - NT - HC_MT - 1; ++i) HC_SGB(0x008);
                }
                if (st_w) HC_SGB(0x200);
            } else {
                // sub-tile by sub-tile: the two MFMAs that read fragment nt, then its re-fetch for unit g + 1 (needed 2 NT - 2
                // MFMAs later); candidate and staging traffic behind the MFMA pairs
                if (has_A) load_A(A[(g + PF) % RING], tp_cur, g + PF);
                if (st_w) stage_store(stg[(uc - 2) % 3], st_nxt, cucn, uc - 2);
                if (st_r) stg[uc % 3] = stage_load(c + 1 < NCH ? cur_q0 : next_q0, cn, cucn, uc);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                    for (int mt = 0; mt < HC_MT; ++mt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[g % RING][mt], Bf[0][nt],
                                                                            g == 0 ? zero16 : acc[mt][nt], 0, 0, 0);
                    if (has_B) load_B1(Bf[0][nt], nt, un);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    HC_SGB(0x008);
                    if (nt < HC_MT && has_A) HC_SGB(0x020);
                    if (nt == HC_MT && st_r) HC_SGB(0x020);
                    if (nt == HC_MT + 1 && st_w) HC_SGB(0x200);
                    HC_SGB(0x008);
                    if (has_B) HC_SGB(0x100);
                }
            }
        }
        };
        if (NCH > 1 || qp_next != qp_cur) sweep(std::true_type{});
        else sweep(std::false_type{});

        // ---- the next item: its first candidate fragments fly under this tile's epilogue; its chunk 0 sits in slot cur
        bool act_next;
        const char *tp_next = tile_ptr(ct_next, act_next);
        const bool switching = qp_next != qp_cur;
#pragma unroll
        for (int u = 0; u < PF; ++u) load_A(A[u], tp_next, u);
        load_B(Bf[0], 0);

        // ---- compare epilogue (lp_hi_stream_kernel's: w = v - a_lo, sign bits -> popcount, band test on the bit patterns)
        if (act_cur) {
            const int64_t c0 = (int64_t)ct_cur * (NW * HC_WROWS) + wid * HC_WROWS;
            int cl_base = 4 * half;
            asm volatile("" : "+v"(cl_base));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float lo_n = alo[nt];
                const float hwf = ahi[nt] - lo_n;
                const unsigned hwb = hwf >= 0.f ? __float_as_uint(hwf) : 0u;
                unsigned smask = 0u;
#pragma unroll
                for (int mt = 0; mt < HC_MT; ++mt) {
#pragma unroll
                    for (int gh = 0; gh < 2; ++gh) {
                        unsigned bq[2][4];
#pragma unroll
                        for (int qq = 0; qq < 2; ++qq) {
                            const int g4 = 2 * gh + qq;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const unsigned b = __float_as_uint(acc[mt][nt][g4 * 4 + e] - lo_n);
                                smask = __builtin_amdgcn_alignbit(smask, b, 31);
                                bq[qq][e] = b;
                            }
                        }
                        const unsigned mq = min(min(min(bq[0][0], bq[0][1]), min(bq[0][2], bq[0][3])),
                                                min(min(bq[1][0], bq[1][1]), min(bq[1][2], bq[1][3])));
                        if (__ballot(mq <= hwb)) {      // some lane holds an uncertain pair among these 8 rows
#pragma unroll
                            for (int qq = 0; qq < 2; ++qq) {
                                const int g4 = 2 * gh + qq;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const int cand = (int)c0 + cl_base + mt * 32 + e + 8 * g4;
                                    const bool unc = bq[qq][e] <= hwb && cand != tru[nt];
                                    const unsigned long long m = __ballot(unc);
                                    if (m) {
                                        const int pos = nl[nt] + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                                                           __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                                        if (unc && pos < SUBLIST) wlist[nt * SUBLIST + pos] = make_int2(qid[nt], cand);
                                        nl[nt] += __popcll(m);
                                    }
                                }
                            }
                        }
                    }
                }
                cnt[nt] += 32 - __popc(smask);
            }
        }
        // the list buffer: a tile that outran it raises the overflow flag (the caller redoes the count one level down)
        int nl_max = 0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (nl[nt] > SUBLIST) {
                if (lane == 0) *p.overflow = 1.0f;
                nl[nt] = SUBLIST;
            }
            nl_max = max(nl_max, nl[nt]);
        }
        if (nl_max >= SUBLIST / 3) flush_all();

        if (switching) {        // new panel: per-lane thresholds only -- its chunks arrive through the ring
            flush_counts();
            load_thresholds(next_q0);
        }
        qp_cur = qp_next; ct_cur = ct_next; tp_cur = tp_next; act_cur = act_next; cur_q0 = next_q0;
    }
    flush_counts();
    flush_all();
}

template <int NW, int NT, int UNITS, int CU, int RING, int BDB>
int hc_launch(const kge_hi_stream_params &p, int grid, hipStream_t s)
{
    auto k = lp_hi_chunk_kernel<NW, NT, UNITS, CU, RING, BDB>;
    constexpr int smem = 2 * (NT * 32) * (CU * 32 + 16) + NW * HC_WLIST * 8;
    static_assert(smem <= 160 * 1024, "LDS");
    static int attr_dev[16];    // per instantiation, per device
    if (int e = kge_ensure_dyn_smem(reinterpret_cast<const void *>(k), smem, attr_dev)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * NW), smem, s, p);
    KGE_CHECK_LAUNCH();
    return 0;
}

} // namespace

// k16 units the chunked kernel is instantiated for (beyond kge_hi_stream_max_units())
int kge_hi_chunk_supported(int units) { return units == 65 || units == 33 || units == 26 || units == 13; }

int kge_hi_chunk_query_rows(int units)
{
    (void)units;
    return kge_env_int("KGE_HC_NT", 4) == 3 ? 96 : 128;
}

// p.units, p.units_p, p.rows_p, p.q_rows, p.B and the pointers filled by the caller (kge_lp_split_count); PM = 0 only
int kge_hi_chunk_launch(kge_hi_stream_params p, int num_cus, hipStream_t s)
{
    if (!kge_hi_chunk_supported(p.units) || p.rows_p % 64 != 0 || p.rows_p < 64 || p.region_count) return KGE_EUNSUPPORTED;
    const int nt = kge_env_int("KGE_HC_NT", 4) == 3 ? 3 : 4;
    const int tq = nt * 32;
    constexpr int NW = 8;
    const int tile_rows = NW * HC_WROWS;
    p.q_panels = (int)((p.q_rows + tq - 1) / tq);
    p.c_tiles = (int)((p.rows_p + tile_rows - 1) / tile_rows);
    p.n_items = (int64_t)p.q_panels * p.c_tiles;
    p.panel_bytes = 0;
    if (p.n_items == 0) return 0;
    const int64_t slots = num_cus;
    int grid = (int)(p.n_items < slots ? p.n_items : slots);
    p.qg = 1;
    if (grid >= 8 && p.n_items >= slots) {
        grid -= grid % 8;
        const int nbx = grid / 8;
        while (p.qg < 64 && nbx % (p.qg * 2) == 0) p.qg *= 2;
        // panels interleaved under one candidate sweep: 8, not "as many as divide the blocks of an XCD" -- a block re-reads ITS
        // panel (TQ x 2 KB) once per candidate tile, and an XCD's 4 MiB L2 cannot hold 32 of them beside the tiles: measured at
        // cfg5's shape (profiles/r06/hi_chunk_work_order_sweep.txt, L2-miss bytes per launch / ms): 32 panels 224 GB / 83.9,
        // 16: 173 / 83.7, 8: 139 / 84.7, 4: 215 / 93.4.  (Model: G panels x 278 KB + 32/G tiles x 1.1 MB per 32 items.)
        const int cap_qg = kge_env_int("KGE_HS_QG", 8);
        while (p.qg > cap_qg && p.qg > 1) p.qg /= 2;
    }
    if (p.units == 65) {
        if (nt == 3) return hc_launch<8, 3, 65, 13, 6, 2>(p, grid, s);
        return hc_launch<8, 4, 65, 13, 4, 1>(p, grid, s);
    }
    if (p.units == 13) {        // (K <= 206: ONE chunk = a RESIDENT 96 / 128-query panel; an experiment against lp_hi_stream.hip's
                                //  two workgroups of four waves per CU, KGE_HC_FORCE=1)
        if (nt == 3) return hc_launch<8, 3, 13, 13, 6, 2>(p, grid, s);
        return hc_launch<8, 4, 13, 13, 4, 1>(p, grid, s);
    }
    if (p.units == 26) {        // (K = 400: an experiment against the resident-panel kernel, KGE_HC_FORCE=1)
        if (nt == 3) return hc_launch<8, 3, 26, 13, 6, 2>(p, grid, s);
        return hc_launch<8, 4, 26, 13, 4, 1>(p, grid, s);
    }
    if (nt == 3) return hc_launch<8, 3, 33, 11, 6, 2>(p, grid, s);
    return hc_launch<8, 4, 33, 11, 4, 1>(p, grid, s);
}
