// The one-product level of the split prefilter as FREE-RUNNING wavefronts (r05).
//
// Same arithmetic as lp_split_count_kernel<.., LV = 1> (lp_split_mfma.hip): acc = sum over the k16 units of
// hi(q) . hi(e) on v_mfma_f32_32x32x16_f16, compared with the two per-query thresholds (a_lo, a_hi); raw_count +=
// #{acc >= a_lo}, pairs inside the band go to the list that kge_lp_split_recheck re-scores exactly.  What is different
// is who waits for whom.  The r04 kernel stages both operands through a double-buffered LDS stage that all eight waves
// of the block fill and read, one barrier per stage: the two waves of a SIMD run the same phase at the same time (MFMA
// group, LDS-DMA issue, fragment wait, compare epilogue) and the matrix pipe idles through every phase but one
// (profiles/r04/lv1_stall_counters.txt: 28-34 % busy, VALU and MFMA co-executing 3.7 % of the time).  Here
//
//   * the QUERY panel (96 queries x all k16 units, 32 B per unit + 16 B row padding: conflict-free ds_read_b128) is
//     RESIDENT in LDS for a whole sweep of the candidate tiles -- loaded once per panel, read-only in between;
//   * every wave owns 64 CANDIDATE rows of the tile and reads their MFMA fragments straight from global memory into
//     registers: the candidate table is laid out FRAGMENT-MAJOR ([32-row group][k16 unit][lane][16 B], kge_lp_hi_rows
//     with frag = 1), so one global_load_dwordx4 per (32 rows, unit) is a fully coalesced 1-KiB read that lands in the
//     exact register layout of the A operand -- no LDS staging, no ds_read, no LDS-DMA for the streamed operand;
//   * hence NO barrier in the tile loop: a wave's only dependencies are its own loads (vmcnt / lgkmcnt, placed by the
//     compiler -- without LDS-DMA in the kernel hipcc's wait insertion is exact).  The waves drift apart, and while one
//     wave of a SIMD runs its compare epilogue (VALU) or waits for fragments, its partner's MFMAs have the pipe;
//   * uncertain pairs go to a per-WAVE list in LDS (ballot + mbcnt positions: no LDS atomics), flushed by the wave
//     itself; the only block-wide synchronisation is the change of the query panel.
//
// Wave tile 64 candidates x 96 queries (2 x 3 MFMA tiles, 96 accumulator VGPRs) as before; a workgroup is NW waves x 64
// rows of ONE panel: NW = 4 (256 threads, TWO workgroups per CU, panel <= 22 units) or NW = 8 (512 threads, one per CU).
#include "kge_common.h"
#ifndef KGE_BUILD_NO_SLP
#error "build with -fno-slp-vectorize -DKGE_BUILD_NO_SLP=1 (torchkge_amd/csrc/build.py): SLP-packed v_pk_fma_f32 with a lane-crossing op_sel misreads beside co-executing MFMAs (profiles/r06/slp_bisect.txt)"
#endif

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int HS_TQ = 96;                       // queries per panel (MFMA columns = lanes, 3 tiles of 32)
constexpr int HS_NT = 3, HS_MT = 2;
constexpr int HS_WROWS = 64;                    // candidate rows per wave
constexpr int HS_WLIST = 384;                   // uncertain pairs buffered per wave (int2 entries) ...
constexpr int HS_SUBLIST = HS_WLIST / 3;        // ... as one sub-list per 32-query sub-tile of the panel
constexpr int HS_PF = 3, HS_RING = 4;           // candidate fragments: units in flight / ring slots
#ifndef HS_SCALAR_SUB
#define HS_SCALAR_SUB 1                         /* epilogue: w = v - a_lo as four v_sub_f32 instead of two v_pk_add_f32 (-3 %) */
#endif
#ifndef HS_TEST8
#define HS_TEST8 1                              /* epilogue: the uncertain-pair test per 8 elements instead of per 4 (-1 %) */
#endif

// PROBE (timing probes, wrong results; env KGE_HS_PROBE, instantiated for <4, 13, 0> only): 1 no compare epilogue, 2 every
// wave streams the table's first rows (cache-hot candidates), 4 no query-fragment reads in the K sweep, 8 no candidate loads
// in the K sweep, 16 no MFMAs; 32 / 64 / 96: VALID results, epilogue variants (scalar subtracts / test per 8 elements / both)
// GS > 0 (r06; PM = 0 only): the panel's 96 rows are GROUPED columns -- one query row shared by up to GS queries of the same key
// (p.members[column * GS + s], < 0: unused), which differ only in their thresholds: the matrix sweep runs once per column,
// the compare epilogue once per member (thresholds, per-member counters and the sub-tiles' pass counts live in LDS).
// NT (r06): 32-query sub-tiles per panel -- 3 (96 queries, 96 accumulator VGPRs) or 4 (128 queries, 128 accumulators: every
// candidate fragment feeds four MFMAs instead of three; PM = 0, GS = 0 only)
template <int NW, int UNITS /* 0: runtime (<= 32) */, int PM, int PROBE = 0, int GS = 0, int NT = 3>
__global__ __launch_bounds__(64 * NW, 2) void lp_hi_stream_kernel(const kge_hi_stream_params p)
{
    static_assert(GS == 0 || PM == 0, "grouped columns: plain thresholds only");
    static_assert(NT == 3 || (NT == 4 && GS == 0), "128-query panels: one query per column");
    constexpr int TQn = 32 * NT, SUBn = HS_WLIST / NT;
    constexpr int NTHREADS = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int units = UNITS ? UNITS : p.units;
    const int RS = units * 32 + 16;                                 // panel row stride (bytes): (2 units + 1) chunks, odd
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    char *panel = smem;
    int2 *wlist = reinterpret_cast<int2 *>(smem + p.panel_bytes) + wid * HS_WLIST;
    float4 *pthr = reinterpret_cast<float4 *>(smem + p.panel_bytes + NW * HS_WLIST * 8);     // PM: per query of the panel
    int *prow = reinterpret_cast<int *>(pthr + TQn);
    // GS: per (column, member) (a_lo, a_hi, query id, its true candidate) and the member's count of this panel's sweep
    [[maybe_unused]] float4 *mthr = pthr;
    [[maybe_unused]] int *mcnt = reinterpret_cast<int *>(mthr + TQn * (GS > 0 ? GS : 1));
    [[maybe_unused]] int npass[NT] = {};                                   // GS: compare passes of a sub-tile = its fullest column's members

    // work order: as lp_split_count_kernel -- QG panels interleaved under a sweep of the candidate tiles, XCD x owns an
    // eighth of the item list, its blocks take stride-nbx positions (nbx a multiple of QG: a block keeps its panel)
    const int QG = p.qg;
    const int nb = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3;
    const int nbx = (nb - xcd + 7) >> 3;
    const int nx = nb < 8 ? nb : 8;
    const int64_t x_begin = p.n_items * xcd / nx, x_end = p.n_items * (xcd + 1) / nx;
    const int64_t item_begin = x_begin + loc;
    const int nitems = item_begin < x_end ? (int)((x_end - item_begin + nbx - 1) / nbx) : 0;
    if (nitems <= 0) return;
    // item -> (query panel, candidate tile).  Panels are grouped: floor(P / QG) groups of QG panels, then one group per set
    // bit of the remainder (sizes QG/2 .. 1); inside a group the items run (panel 0, tile 0) (panel 1, tile 0) .. so
    // that a block stepping by nbx -- a multiple of every group size -- keeps ITS panel while the blocks of the XCD sweep
    // the candidate tiles together (every tile enters the L2 once per group).
    const int full_panels = (p.q_panels / QG) * QG;
    const int full_items = full_panels * p.c_tiles;
    auto item_qp_ct = [&](int i, int &qp, int &ct) __attribute__((always_inline)) {
        int idx = (int)item_begin + i * nbx;
        int base = 0, gsz = QG;
        if (idx < full_items) {
            const int per = QG * p.c_tiles, grp = idx / per;
            idx -= grp * per;
            base = grp * QG;
        } else {
            idx -= full_items;
            base = full_panels;
            const int rem = p.q_panels - full_panels;
            gsz = 1;
            for (int sz = QG >> 1; sz >= 1; sz >>= 1) {
                if (rem & sz) {
                    if (idx < sz * p.c_tiles) { gsz = sz; break; }
                    idx -= sz * p.c_tiles;
                    base += sz;
                }
            }
        }
        ct = idx / gsz;
        qp = base + (idx - ct * gsz);
    };

    // candidate fragments of this wave: 32-row groups g, g + 1 of the fragment-major table
    const int n_groups32 = (int)(p.rows_p >> 5);
    const int64_t gstride = (int64_t)p.units_p << 10;               // bytes per 32-row group
    auto tile_ptr = [&](int ct, bool &active) __attribute__((always_inline)) -> const char * {
        int g = ct * (NW * 2) + wid * 2;
        active = g + 1 < n_groups32;
        g = min(g, n_groups32 - 2);                                 // (past the table: valid rows, results dropped)
        if (PROBE & 2) g = wid * 2;
        return p.Ef + g * gstride;
    };
    const unsigned lane16 = lane * 16;

    f32x16 acc[HS_MT][NT];
    f16x8 A[HS_RING][HS_MT], Bf[2][NT];
    int cnt[NT] = {};
    float alo[NT], ahi[NT];
    int qid[NT], tru[NT];                                     // query id / its true candidate (local index; -1: none)
    // This wave's LDS list: one sub-list per 32-query sub-tile of the panel (wave-uniform fill counts).  Flushed into the
    // global list -- or, with p.region_count, into the REGION of (panel, sub-tile): the exact recheck then takes a region
    // at a time with the sub-tile's 32 query rows resident in LDS (kge_lp_split_recheck_regions: half the row fetches).
    int nl[NT] = {};

    auto flush_sub = [&](int nt, int qp) __attribute__((always_inline)) {
        if (nl[nt] > 0) {
            int32_t *ctr = p.list_count;
            int2 *dst = reinterpret_cast<int2 *>(p.list);
            unsigned lim = (unsigned)p.cap;
            if (p.region_count) {
                const int reg = qp * NT + nt;
                ctr = p.region_count + reg;
                dst += (int64_t)reg * p.region_cap;
                lim = (unsigned)p.region_cap;
            }
            int base = 0;
            if (lane == 0) base = atomicAdd(ctr, nl[nt]);
            base = __builtin_amdgcn_readfirstlane(base);
            for (int i = lane; i < nl[nt]; i += 64) {
                const int pos = base + i;
                if ((unsigned)pos < lim) dst[pos] = wlist[nt * SUBn + i];
                else *p.overflow = 1.0f;
            }
            nl[nt] = 0;
        }
    };
    [[maybe_unused]] bool first_panel = true;
    auto load_panel = [&](int64_t q0) __attribute__((always_inline)) {
        // rows of the planar query operand -> LDS rows of stride RS
        const int cpr = 2 * units;                                  // 16-byte chunks per row
        const int total = TQn * cpr;
        for (int n = tid; n < total; n += NTHREADS) {
            const int row = n / cpr, c = n - row * cpr;
            // (a 128-query panel may reach past the operand's rows, padded to 96s: its last row again -- thresholds +inf)
            const uint4 v = *reinterpret_cast<const uint4 *>(p.Qh + min(q0 + row, p.q_rows - 1) * p.q_row_bytes + c * 16);
            *reinterpret_cast<uint4 *>(panel + row * RS + c * 16) = v;
        }
        if (PM) {
            if (tid < TQn) {
                int64_t q = -1;
                if (q0 + tid < p.q_rows) q = p.col_q ? (int64_t)p.col_q[q0 + tid] : q0 + tid;
                // (p_i, z_i) pre-multiplied by -2^23, the accumulators' scale (exact): the epilogue's projection term is then two
                // FMAs per element, v + x (x z' + p'), instead of fma, mul, fma (r06; one rounding fewer than before -- inside the
                // band's 9 x 2^-22 allowance for this term either way)
                float4 t4 = q >= 0 ? p.thr4[q] : make_float4(INFINITY, INFINITY, 0.f, 0.f);
                t4.z *= -8388608.0f; t4.w *= -8388608.0f;
                pthr[tid] = t4;
                prow[tid] = (int)p.r_idx[min(max(q, (int64_t)0), p.B - 1)];
            }
        }
        if constexpr (GS > 0) {
            // (called between two block barriers: thread idx owns entry idx of mthr / mcnt -- the previous panel's counts
            // leave through it before the entry is rewritten)
            for (int idx = tid; idx < TQn * GS; idx += NTHREADS) {
                if (!first_panel) {
                    const int c = mcnt[idx], oq = __float_as_int(mthr[idx].z);
                    if (c != 0 && oq >= 0) atomicAdd(&p.raw_count[oq], c);
                }
                const int64_t col = q0 + idx / GS;
                int64_t q = col < p.q_rows ? (int64_t)p.members[col * GS + (idx % GS)] : -1;
                if (q >= p.B) q = -1;
                float2 t = make_float2(INFINITY, INFINITY);
                int tr = -1;
                if (q >= 0) {
                    t = p.thr[q];
                    if (p.true_idx) tr = (int)(p.true_idx[q] - p.c_base);
                }
                mthr[idx] = make_float4(t.x, t.y, __int_as_float((int)q), __int_as_float(tr));
                mcnt[idx] = 0;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {    // members of this lane's column -> the sub-tile's maximum (wave-uniform)
                const int64_t col = q0 + nt * 32 + l31;
                int m = 0;
                if (col < p.q_rows) {
#pragma unroll
                    for (int s = 0; s < GS; ++s) {
                        const int q = p.members[col * GS + s];
                        m += (q >= 0 && q < p.B) ? 1 : 0;
                    }
                }
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
                npass[nt] = __builtin_amdgcn_readfirstlane(m);
            }
            first_panel = false;
            return;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int64_t col = q0 + nt * 32 + l31;
            int64_t q = -1;
            if (col < p.q_rows) q = p.col_q ? (int64_t)p.col_q[col] : col;
            if (q >= p.B) q = -1;
            float2 t = make_float2(INFINITY, INFINITY);
            if (q >= 0) {
                if (PM) { const float4 t4 = p.thr4[q]; t = make_float2(t4.x, t4.y); }
                else t = p.thr[q];
            }
            alo[nt] = t.x; ahi[nt] = t.y; qid[nt] = (int)q;
            // the pair (query, its true entity) scores s_true exactly: it is counted (acc >= a_lo) and a re-score could
            // never take it back -- it need not be listed (a tenth of a fitted model's list)
            tru[nt] = (p.true_idx && q >= 0) ? (int)(p.true_idx[q] - p.c_base) : -1;
        }
    };
    auto flush_counts = [&]() __attribute__((always_inline)) {
        if constexpr (GS > 0) return;       // (grouped columns count in LDS: mcnt, flushed by load_panel / at the end)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int v = cnt[nt] + __shfl_xor(cnt[nt], 32, 64);
            if (half == 0 && v != 0 && qid[nt] >= 0) atomicAdd(&p.raw_count[qid[nt]], v);
            cnt[nt] = 0;
        }
    };

    // fragment addresses inside the panel: row nt * 32 + l31, unit u, k-half `half`
    const unsigned b_lane = (unsigned)(l31 * RS + half * 16);
    auto load_B = [&](f16x8 (&dst)[NT], int u) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            dst[nt] = *reinterpret_cast<const f16x8 *>(panel + b_lane + nt * 32 * RS + u * 32);
    };
    auto load_A = [&](f16x8 (&dst)[HS_MT], const char *tp, int u) __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < HS_MT; ++mt)
            dst[mt] = *reinterpret_cast<const f16x8 *>(tp + mt * gstride + (u << 10) + lane16);
    };

    // the one global list: the three sub-lists behind ONE atomic (a returning same-address atomic per sub-list tripled the
    // waves' stalls: 0.50 -> 0.55 ms per evaluate, profiles/r05/region_recheck_ab.txt)
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
                    if ((unsigned)pos < (unsigned)p.cap) reinterpret_cast<int2 *>(p.list)[pos] = wlist[nt * SUBn + i];
                    else *p.overflow = 1.0f;
                }
                base += nl[nt];
                nl[nt] = 0;
            }
        }
    };

    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;

    int qp_cur, ct_cur;
    item_qp_ct(0, qp_cur, ct_cur);
    int64_t cur_q0 = (int64_t)qp_cur * TQn;
    load_panel(cur_q0);
    bool act_cur;
    const char *tp_cur = tile_ptr(ct_cur, act_cur);
#pragma unroll
    for (int u = 0; u < HS_PF; ++u)
        if (u < units) load_A(A[u], tp_cur, u);
    __syncthreads();
    load_B(Bf[0], 0);

    for (int it = 0; it < nitems; ++it) {
        // ---- the K sweep of one wave tile: per unit 6 MFMAs, 3 ds_read_b128 (queries of the next unit), 2 global loads
        // (candidates three units ahead) -- all register-to-register dependencies, waits placed by the compiler
        if constexpr (UNITS != 0) {
#pragma unroll
            for (int u = 0; u < UNITS; ++u) {
                if (u + 1 < UNITS && !(PROBE & 4)) load_B(Bf[(u + 1) & 1], u + 1);
                if (u + HS_PF < UNITS && !(PROBE & 8)) load_A(A[(u + HS_PF) % HS_RING], tp_cur, u + HS_PF);
                if (PROBE & 16) {
                    if (u == 0) {
#pragma unroll
                        for (int mt = 0; mt < HS_MT; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = zero16;
                    }
                    // (keep the loads alive)
#pragma unroll
                    for (int mt = 0; mt < HS_MT; ++mt) acc[mt][0][0] += (float)A[u % HS_RING][mt][0];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[0][nt][1] += (float)Bf[u & 1][nt][0];
                    continue;
                }
#pragma unroll
                for (int mt = 0; mt < HS_MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[u % HS_RING][mt], Bf[u & 1][nt],
                                                                            u == 0 ? zero16 : acc[mt][nt], 0, 0, 0);
                // interleave: one load behind each of the first NT + 2 MFMAs of the unit (NT query fragments, 2 candidate loads)
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#pragma unroll
                for (int i = 0; i < NT - 2; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
        } else {
            // any number of units: a runtime loop over groups of four (the ring's period), guards on the tail
#pragma unroll
            for (int mt = 0; mt < HS_MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = zero16;
            for (int u0 = 0; u0 < units; u0 += HS_RING) {
#pragma unroll
                for (int j = 0; j < HS_RING; ++j) {
                    const int u = u0 + j;
                    if (u < units) {
                        if (u + 1 < units) load_B(Bf[(j + 1) & 1], u + 1);
                        if (u + HS_PF < units) load_A(A[(j + HS_PF) % HS_RING], tp_cur, u + HS_PF);
#pragma unroll
                        for (int mt = 0; mt < HS_MT; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[j][mt], Bf[j & 1][nt], acc[mt][nt], 0, 0, 0);
                    }
                }
            }
        }

        // ---- the next item: its first candidate fragments fly under this tile's epilogue
        const bool more = it + 1 < nitems;
        int qp_next = qp_cur, ct_next = ct_cur;
        if (more) item_qp_ct(it + 1, qp_next, ct_next);
        bool act_next;
        const char *tp_next = tile_ptr(ct_next, act_next);
        const bool switching = qp_next != qp_cur;
#pragma unroll
        for (int u = 0; u < HS_PF; ++u)
            if (u < units) load_A(A[u], tp_next, u);
        if (!switching) load_B(Bf[0], 0);

        // ---- grouped columns: the same compare once per MEMBER of the column (thresholds from LDS, runtime pass count)
        if constexpr (GS > 0) {
            if (act_cur) {
                const int64_t c0 = (int64_t)ct_cur * (NW * HS_WROWS) + wid * HS_WROWS;
                int cl_base = 4 * half;
                asm volatile("" : "+v"(cl_base));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    for (int s = 0; s < npass[nt]; ++s) {
                        const float4 t4 = mthr[(nt * 32 + l31) * GS + s];
                        const float lo_n = t4.x;
                        const int qid_s = __float_as_int(t4.z), tru_s = __float_as_int(t4.w);
                        const float hwf = t4.y - lo_n;
                        const unsigned hwb = hwf >= 0.f ? __float_as_uint(hwf) : 0u;
                        unsigned smask = 0u;
#pragma unroll
                        for (int mt = 0; mt < HS_MT; ++mt) {
#pragma unroll
                            for (int gh = 0; gh < 2; ++gh) {
                                unsigned bq[2][4];
#pragma unroll
                                for (int qq = 0; qq < 2; ++qq) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        const unsigned b = __float_as_uint(acc[mt][nt][(2 * gh + qq) * 4 + e] - lo_n);
                                        smask = __builtin_amdgcn_alignbit(smask, b, 31);
                                        bq[qq][e] = b;
                                    }
                                }
                                const unsigned mq = min(min(min(bq[0][0], bq[0][1]), min(bq[0][2], bq[0][3])),
                                                        min(min(bq[1][0], bq[1][1]), min(bq[1][2], bq[1][3])));
                                if (__ballot(mq <= hwb)) {
#pragma unroll
                                    for (int qq = 0; qq < 2; ++qq) {
#pragma unroll
                                        for (int e = 0; e < 4; ++e) {
                                            const int cand = (int)c0 + cl_base + mt * 32 + e + 8 * (2 * gh + qq);
                                            const bool unc = bq[qq][e] <= hwb && cand != tru_s;
                                            const unsigned long long m = __ballot(unc);
                                            if (m) {
                                                const int pos = nl[nt] + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                                                                   __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                                                if (unc && pos < SUBn) wlist[nt * SUBn + pos] = make_int2(qid_s, cand);
                                                nl[nt] += __popcll(m);
                                            }
                                        }
                                    }
                                }
                            }
                        }
                        const int c = 32 - __popc(smask);
                        if (c != 0 && qid_s >= 0) atomicAdd(&mcnt[(nt * 32 + l31) * GS + s], c);
                        // (a sub-list that fills up inside the member loop leaves at once: up to GS passes append to it per tile)
                        if (nl[nt] >= SUBn / 2) {
                            if (nl[nt] > SUBn) {
                                if (lane == 0) *p.overflow = 1.0f;
                                nl[nt] = SUBn;
                            }
                            flush_all();
                        }
                    }
                }
            }
        } else
        // ---- compare epilogue (as lp_split_count_kernel: w = v - a_lo, sign bits -> popcount, band test on the bits)
        if (act_cur && !(PROBE & 1)) {
            const int64_t c0 = (int64_t)ct_cur * (NW * HS_WROWS) + wid * HS_WROWS;
            int cl_base = 4 * half;
            asm volatile("" : "+v"(cl_base));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float lo_n = alo[nt], hi_n = ahi[nt], p_n = 0.f, z_n = 0.f;
                const float *xrow = nullptr;
                if (PM) {
                    const float4 t4 = pthr[nt * 32 + l31];
                    p_n = t4.z; z_n = t4.w;
                    xrow = p.X + (int64_t)prow[nt * 32 + l31] * p.ldx + c0 + 4 * half;
                }
                const f32x2 nlo2 = {-lo_n, -lo_n};
                const float hwf = hi_n - lo_n;
                const unsigned hwb = hwf >= 0.f ? __float_as_uint(hwf) : 0u;
                unsigned smask = 0u;
#pragma unroll
                for (int mt = 0; mt < HS_MT; ++mt) {
#pragma unroll
                    for (int gh = 0; gh < 2; ++gh) {       // (the projection gathers two quads at a time: register budget)
                    float4 x4[4], y4[4];
                    if (PM) {
#pragma unroll
                        for (int g4 = 2 * gh; g4 < 2 * gh + 2; ++g4) {
                            x4[g4] = *reinterpret_cast<const float4 *>(xrow + mt * 32 + 8 * g4);
                            if (PM == 2) y4[g4] = *reinterpret_cast<const float4 *>(p.yc + c0 + 4 * half + mt * 32 + 8 * g4);
                        }
                    }
                    unsigned bq[2][4];          // bit patterns of w = v - a_lo of the two quads of this half tile
#pragma unroll
                    for (int g4 = 2 * gh; g4 < 2 * gh + 2; ++g4) {
                        float vq[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            vq[e] = acc[mt][nt][g4 * 4 + e];
                            if (PM) {
                                const float xe = e == 0 ? x4[g4].x : (e == 1 ? x4[g4].y : (e == 2 ? x4[g4].z : x4[g4].w));
                                if (PM == 1) {
                                    vq[e] = fmaf(xe, fmaf(xe, z_n, p_n), vq[e]);
                                } else {
                                    const float ye = e == 0 ? y4[g4].x : (e == 1 ? y4[g4].y : (e == 2 ? y4[g4].z : y4[g4].w));
                                    vq[e] = fmaf(ye, fmaf(ye, z_n, fmaf(-16777216.0f, xe, p_n)), vq[e]);
                                }
                            }
                        }
                        unsigned b0, b1, b2, b3;
                        if (HS_SCALAR_SUB || (PROBE & 32)) {        // four v_sub_f32 (same values: one rounding each)
                            b0 = __float_as_uint(vq[0] - lo_n); b1 = __float_as_uint(vq[1] - lo_n);
                            b2 = __float_as_uint(vq[2] - lo_n); b3 = __float_as_uint(vq[3] - lo_n);
                        } else {
                            const f32x2 w01 = (f32x2){vq[0], vq[1]} + nlo2, w23 = (f32x2){vq[2], vq[3]} + nlo2;
                            b0 = __float_as_uint(w01.x); b1 = __float_as_uint(w01.y);
                            b2 = __float_as_uint(w23.x); b3 = __float_as_uint(w23.y);
                        }
                        smask = __builtin_amdgcn_alignbit(smask, b0, 31);
                        smask = __builtin_amdgcn_alignbit(smask, b1, 31);
                        smask = __builtin_amdgcn_alignbit(smask, b2, 31);
                        smask = __builtin_amdgcn_alignbit(smask, b3, 31);
                        bq[g4 & 1][0] = b0; bq[g4 & 1][1] = b1; bq[g4 & 1][2] = b2; bq[g4 & 1][3] = b3;
                    }
                    // uncertain pairs (0 <= w <= band width, as unsigned bit patterns): tested per quad, or per PAIR of quads
                    // (HS_TEST8: half the ballots and branches; the listing below then walks 8 elements)
                    const unsigned mq0 = min(min(min(bq[0][0], bq[0][1]), bq[0][2]), bq[0][3]);
                    const unsigned mq1 = min(min(min(bq[1][0], bq[1][1]), bq[1][2]), bq[1][3]);
                    const bool test8 = HS_TEST8 || (PROBE & 64);
#pragma unroll
                    for (int hq = 0; hq < 2; ++hq) {
                        if (test8 && hq == 1) break;
                        const unsigned mq = test8 ? min(mq0, mq1) : (hq == 0 ? mq0 : mq1);
                        if (__ballot(mq <= hwb)) {      // some lane holds an uncertain pair among these rows
                            // (kept SMALL: unrolled 12 / 24 times; capacity is checked once per tile below -- an entry
                            // past the buffer raises the overflow flag, like UNC_CAP of lp_split_count_kernel)
#pragma unroll
                            for (int qq = 0; qq < 2; ++qq) {
                                if (!test8 && qq != hq) continue;
                                const int g4 = 2 * gh + qq;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const int cand = (int)c0 + cl_base + mt * 32 + e + 8 * g4;
                                    const bool unc = bq[qq][e] <= hwb && cand != tru[nt];
                                    const unsigned long long m = __ballot(unc);
                                    if (m) {
                                        const int pos = nl[nt] + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                                                           __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                                        if (unc && pos < SUBn) wlist[nt * SUBn + pos] = make_int2(qid[nt], cand);
                                        nl[nt] += __popcll(m);
                                    }
                                }
                            }
                        }
                    }
                    }
                }
                cnt[nt] += 32 - __popc(smask);
            }
        }

        if (PROBE & 1) {        // (probe: the accumulators stay alive through one add each)
#pragma unroll
            for (int mt = 0; mt < HS_MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) cnt[nt] += __float_as_int(acc[mt][nt][5]) & 1;
        }
        // the list buffer: a tile that outran it raises the overflow flag (the caller redoes the count on the next level down);
        // flushed while >= 2/3 of it is free for the next tile (a density of 4 % of the tile's pairs: UNC_CAP's)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (nl[nt] > SUBn) {
                if (lane == 0) *p.overflow = 1.0f;
                nl[nt] = SUBn;
            }
        }
        if (p.region_count) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                if (nl[nt] >= SUBn / 3 || switching) flush_sub(nt, qp_cur);   // (a region belongs to ONE panel)
        } else {
            int nl_max = 0;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) nl_max = max(nl_max, nl[nt]);
            if (nl_max >= SUBn / 3) flush_all();
        }

        // ---- query panel change (block-uniform): the only block-wide synchronisation of the sweep
        if (switching) {
            flush_counts();
            __syncthreads();                    // every wave is done reading the old panel
            cur_q0 = (int64_t)qp_next * TQn;
            load_panel(cur_q0);
            __syncthreads();
            load_B(Bf[0], 0);
        }
        qp_cur = qp_next; ct_cur = ct_next; tp_cur = tp_next; act_cur = act_next;
    }
    flush_counts();
    if constexpr (GS > 0) {
        __syncthreads();        // every wave's LDS counts of the last panel are in
        for (int idx = tid; idx < TQn * GS; idx += NTHREADS) {
            const int c = mcnt[idx], oq = __float_as_int(mthr[idx].z);
            if (c != 0 && oq >= 0) atomicAdd(&p.raw_count[oq], c);
        }
    }
    if (p.region_count) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) flush_sub(nt, qp_cur);
    } else {
        flush_all();
    }
}

constexpr int HS_NT_DEFAULT = 4;
constexpr int HS_GS = 4;        // queries per grouped column (= kge_lp_split_group_sets(), the layout of kge_split_args.members)

template <int NW, int UNITS, int PM, int PROBE = 0, int GS = 0, int NT = 3>
int hs_launch(const kge_hi_stream_params &p, int grid, int smem, hipStream_t s)
{
    auto k = lp_hi_stream_kernel<NW, UNITS, PM, PROBE, GS, NT>;
    static int attr_dev[16];    // per instantiation, per device
    if (int e = kge_ensure_dyn_smem(reinterpret_cast<const void *>(k), smem, attr_dev)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * NW), smem, s, p);
    KGE_CHECK_LAUNCH();
    return 0;
}

template <int NW, int PM>
int hs_dispatch_units(const kge_hi_stream_params &p, int grid, int smem, int nt, hipStream_t s)
{
    if (nt == 4) {              // 128-query panels (kge_hi_stream_launch: PM = 0, one query per column)
        if (p.units == 13) return hs_launch<NW, 13, PM, 0, 0, 4>(p, grid, smem, s);
        if constexpr (PM == 0) {
            if (p.units == 26) return hs_launch<NW, 26, 0, 0, 0, 4>(p, grid, smem, s);
            return hs_launch<NW, 0, 0, 0, 0, 4>(p, grid, smem, s);
        }
        return KGE_EUNSUPPORTED;
    }
    if (NW == 4 && PM == 0 && p.units == 13 && !p.members) {
        switch (kge_env_int("KGE_HS_PROBE", 0)) {
        case 1: return hs_launch<4, 13, 0, 1>(p, grid, smem, s);
        case 2: return hs_launch<4, 13, 0, 2>(p, grid, smem, s);
        case 4: return hs_launch<4, 13, 0, 4>(p, grid, smem, s);
        case 8: return hs_launch<4, 13, 0, 8>(p, grid, smem, s);
        case 12: return hs_launch<4, 13, 0, 12>(p, grid, smem, s);
        case 13: return hs_launch<4, 13, 0, 13>(p, grid, smem, s);
        case 16: return hs_launch<4, 13, 0, 16>(p, grid, smem, s);
        case 17: return hs_launch<4, 13, 0, 17>(p, grid, smem, s);
        case 32: return hs_launch<4, 13, 0, 32>(p, grid, smem, s);     // (valid results: epilogue variants)
        case 64: return hs_launch<4, 13, 0, 64>(p, grid, smem, s);
        case 96: return hs_launch<4, 13, 0, 96>(p, grid, smem, s);
        default: break;
        }
    }
    if (p.members) {            // grouped columns (PM = 0, checked by the caller)
        if constexpr (PM == 0) {
            if (p.units == 13) return hs_launch<NW, 13, 0, 0, HS_GS>(p, grid, smem, s);
            if (p.units == 26) return hs_launch<NW, 26, 0, 0, HS_GS>(p, grid, smem, s);
            return hs_launch<NW, 0, 0, 0, HS_GS>(p, grid, smem, s);
        }
        return KGE_EUNSUPPORTED;
    }
    if (p.units == 13) return hs_launch<NW, 13, PM>(p, grid, smem, s);
    if (p.units == 26) return hs_launch<NW, 26, PM>(p, grid, smem, s);
    return hs_launch<NW, 0, PM>(p, grid, smem, s);
}

} // namespace

int kge_hi_stream_max_units(void) { return 32; }

// p.units, p.units_p, p.rows_p, p.B, pointers filled by the caller (kge_lp_split_count); this fills the work order
int kge_hi_stream_launch(kge_hi_stream_params p, int pm, int num_cus, hipStream_t s)
{
    if (p.units <= 0 || p.units > 32 || p.rows_p % 64 != 0 || p.rows_p < 64) return KGE_EINVAL;
    if (p.members && (pm != 0 || p.col_q || p.region_count)) return KGE_EINVAL;
    const int RS = p.units * 32 + 16;
    // (r06) 128-query panels for plain thresholds, one query per column: every candidate fragment feeds four MFMAs instead of three
    // (KGE_HS_NT=3: the 96-query panels of r05)
    const int nt = ((pm == 0 || (p.units == 13 && kge_env_int("KGE_HS_NT_PM", 0))) && !p.members &&
                    kge_env_int("KGE_HS_NT", HS_NT_DEFAULT) == 4 && kge_env_int("KGE_HS_PROBE", 0) == 0 &&
                    128 * RS + 8 * HS_WLIST * 8 + 128 * 20 <= 160 * 1024) ? 4 : 3;
    const int tq = 32 * nt;
    p.panel_bytes = (tq * RS + 15) / 16 * 16;
    // two 4-wave workgroups per CU while two panels (+ lists) fit the LDS; else one 8-wave workgroup
    const int extra = p.members ? HS_TQ * HS_GS * 20 : tq * 20;
    const int smem4 = p.panel_bytes + 4 * HS_WLIST * 8 + extra, smem8 = p.panel_bytes + 8 * HS_WLIST * 8 + extra;
    int nw = 2 * smem4 <= 160 * 1024 - 2048 ? 4 : 8;
    const int force = kge_env_int("KGE_HS_WAVES", 0);
    if (force == 4 && smem4 <= 160 * 1024) nw = 4;
    if (force == 8) nw = 8;
    if (nw == 8 && smem8 > 160 * 1024) return KGE_EUNSUPPORTED;
    const int tile_rows = nw * HS_WROWS;
    p.q_panels = (int)((p.q_rows + tq - 1) / tq);
    p.c_tiles = (int)((p.rows_p + tile_rows - 1) / tile_rows);
    p.n_items = (int64_t)p.q_panels * p.c_tiles;
    if (p.n_items == 0) return 0;
    const int per_cu = nw == 4 ? 2 : 1;
    const int64_t slots = (int64_t)num_cus * per_cu;
    int grid = (int)(p.n_items < slots ? p.n_items : slots);
    // panels interleaved under one candidate sweep: the largest power of two (<= 64) dividing the blocks per XCD, so that
    // a block keeps its panel for a whole sweep; small launches run panel-major
    p.qg = 1;
    if (grid >= 8 && p.n_items >= slots) {
        grid -= grid % 8;
        const int nbx = grid / 8;
        while (p.qg < 64 && nbx % (p.qg * 2) == 0) p.qg *= 2;
        const int cap_qg = kge_env_int("KGE_HS_QG", 64);
        while (p.qg > cap_qg && p.qg > 1) p.qg /= 2;
    }
    const int smem = nw == 4 ? smem4 : smem8;
    if (nw == 4) {
        if (pm == 1) return hs_dispatch_units<4, 1>(p, grid, smem, nt, s);
        if (pm == 2) return hs_dispatch_units<4, 2>(p, grid, smem, nt, s);
        return hs_dispatch_units<4, 0>(p, grid, smem, nt, s);
    }
    if (pm == 1) return hs_dispatch_units<8, 1>(p, grid, smem, nt, s);
    if (pm == 2) return hs_dispatch_units<8, 2>(p, grid, smem, nt, s);
    return hs_dispatch_units<8, 0>(p, grid, smem, nt, s);
}
