// Shared device helpers for libkge_hip.so (gfx950 only).
// Compiled with -ffp-contract=off: every fused multiply-add in this library is
// an explicit fmaf(), so the fp32 arithmetic is exactly what the source says
// (and what oracle/kge_oracle.c restates on the CPU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/kge_hip.h"

#define KGE_WAVE 64

#define KGE_CHECK_LAUNCH()                          \
    do {                                            \
        hipError_t e__ = hipGetLastError();         \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)

static inline hipStream_t kge_s(kge_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline bool kge_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Running maximum in a device scalar (non-negative floats compared as bit patterns).  Same-address atomics
// serialise at ~12 ns each on this part (they are resolved past the per-XCD L2s), so a wave first LOOKS:
// the scalar only grows, a stale (smaller) reading merely costs the atomic it would have saved.
__device__ __forceinline__ void kge_atomic_max_u32(unsigned *addr, unsigned v)
{
    if (v > __builtin_nontemporal_load(addr)) atomicMax(addr, v);
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- the scoring contract of kge_lp_desc (see include/kge_hip.h) ----------
__device__ __forceinline__ float lp_chain_dot(const float *__restrict__ a, const float *__restrict__ t,
                                              int K, float acc)
{
    // the MFMA kernel's accumulation order: 8-blocks ascending, and inside an
    // 8-block k = 0,4,1,5,2,6,3,7 (lane-half h of the wave supplies k = 4h + j
    // to MFMA j, and v_mfma_f32_32x32x2_f32 adds half 0's product first)
    int kb = 0;
    for (; kb + 8 <= K; kb += 8) { // full 8-blocks: 16 independent loads, then the dependent chain
        float av[8], tv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { av[j] = a[kb + j]; tv[j] = t[kb + j]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc = fmaf(av[j], tv[j], acc);
            acc = fmaf(av[4 + j], tv[4 + j], acc);
        }
    }
    if (kb < K) { // partial last block
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k0 = kb + j, k1 = kb + 4 + j;
            if (k0 < K) acc = fmaf(a[k0], t[k0], acc);
            if (k1 < K) acc = fmaf(a[k1], t[k1], acc);
        }
    }
    return acc;
}

__device__ __forceinline__ float lp_epilogue(int mode, float dot, float qn, float en)
{
    if (mode == KGE_LP_L2_EXPAND) {
        float d2 = fmaf(-2.0f, dot, qn + en);
        return -fmaxf(d2, 0.0f);
    }
    return dot;
}

// epilogue of the projection modes (include/kge_hip.h): `v0` = fmaf(-2, dot, qn + en)
__device__ __forceinline__ float lp_epilogue_proj(int mode, float v0, float x, float y, float p, float z)
{
    float v;
    if (mode == KGE_LP_L2_PROJH) v = fmaf(x, fmaf(x, z, p), v0);
    else v = fmaf(y, fmaf(y, z, fmaf(2.0f, x, p)), v0);
    return -fmaxf(v, 0.0f);
}
__device__ __forceinline__ float lp_epilogue_any(const kge_lp_desc &d, float dot, int64_t i, int64_t c)
{
    if (d.mode == KGE_LP_DOT) return dot;
    const float v0 = fmaf(-2.0f, dot, d.qn[i] + d.en[c]);
    if (d.mode == KGE_LP_L2_EXPAND) return -fmaxf(v0, 0.0f);
    const float x = d.scal[d.r_idx[i] * d.scal_ld + c];
    const float y = d.mode == KGE_LP_L2_PROJD ? d.yc[c] : 0.f;
    return lp_epilogue_proj(d.mode, v0, x, y, d.Wq[i * d.ldw], d.Wq[i * d.ldw + 1]);
}

// score of query i against LOCAL candidate c, any mode (scalar reference path
// used by the pair / filter kernels; bit-identical to the tile kernels).
__device__ __forceinline__ float lp_pair_score(const kge_lp_desc &d, int64_t i, int64_t c)
{
    if (KGE_LP_IS_MFMA(d.mode)) {
        float acc = lp_chain_dot(d.A0 + i * d.lda0, d.T0 + c * d.ldt0, d.K0, 0.0f);
        if (d.K1 > 0) acc = lp_chain_dot(d.A1 + i * d.lda1, d.T1 + c * d.ldt1, d.K1, acc);
        return lp_epilogue_any(d, acc, i, c);
    }
    const float *q = d.A0 + i * d.lda0;
    const float *t = d.T0 + c * d.ldt0;
    float acc = 0.0f;
    const bool l1 = d.mode == KGE_LP_L1_DIRECT;
    const float a = d.Wq ? d.scal[c * d.scal_ld + (d.scal_ld > 1 ? d.r_idx[i] : 0)] : 0.0f;
    const float *w = d.Wq ? d.Wq + i * d.ldw : nullptr;
    if (l1) {   // one add per aligned 4-group of k, the group as (|d0|+|d1|)+(|d2|+|d3|), absent k = 0
        for (int k = 0; k < d.K0; k += 4) {
            float m[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float diff = 0.0f;
                if (k + e < d.K0) {
                    diff = q[k + e] - t[k + e];
                    if (w) diff = fmaf(a, w[k + e], diff);
                }
                m[e] = fabsf(diff);
            }
            acc = acc + ((m[0] + m[1]) + (m[2] + m[3]));
        }
    } else {
        for (int k = 0; k < d.K0; ++k) {
            float diff = q[k] - t[k];
            if (w) diff = fmaf(a, w[k], diff);
            acc = fmaf(diff, diff, acc);
        }
    }
    return -acc;
}

// ---- wave-cooperative exact pair scores (MFMA modes) ------------------------------
// One lane per (query, candidate) pair runs the scalar chain above -- one
// accumulator in a fixed order, it cannot be split across lanes -- but the two
// rows of each of the wavefront's 64 pairs are fetched COOPERATIVELY, 32 k at a
// time, as 128-byte row segments (8 lanes x float4 per row, all of a chunk's
// loads in flight together) and handed to their lane through LDS (row stride 36
// floats: conflict-free b128 stores and loads).  A lane-per-row gather touches 64
// different cache lines per load instruction and is ~5x slower.
// Every lane of the wavefront must call; `qs`/`es` are this wavefront's own
// 64 x KGE_PS_LD floats of LDS.  Bit-identical to lp_pair_score.
#ifndef KGE_PS_KC_V
#define KGE_PS_KC_V 32      /* 8 float4 pieces per row: the piece -> (row, column) split is shifts, and hipcc keeps */
#define KGE_PS_LD_V 36      /* the pipelined loop at 88 VGPRs (40 / 44 hoisted 80 address registers and spilled)   */
#endif
constexpr int KGE_PS_KC = KGE_PS_KC_V, KGE_PS_LD = KGE_PS_LD_V;

static inline bool kge_lp_vec4(const kge_lp_desc &d)
{
    bool v = (d.K0 % 4 == 0) && (d.lda0 % 4 == 0) && (d.ldt0 % 4 == 0) && kge_aligned16(d.A0) && kge_aligned16(d.T0);
    if (d.K1 > 0)
        v = v && (d.K1 % 4 == 0) && (d.lda1 % 4 == 0) && (d.ldt1 % 4 == 0) && kge_aligned16(d.A1) && kge_aligned16(d.T1);
    return v;
}

// the direct modes' chains (lp_pair_score without the rank-1 term), one accumulator: L2 one fmaf per k in
// ascending order; L1 one add per aligned 4-group of k, the group as (|d0|+|d1|)+(|d2|+|d3|).  `a` / `t` must be
// readable (zero-filled) up to the next multiple of 4 -- the staged chunks below are.
template <bool L1>
__device__ __forceinline__ float lp_chain_direct(const float *__restrict__ a, const float *__restrict__ t, int K, float acc)
{
    if (L1) {
        for (int k = 0; k < K; k += 4) {
            const float4 av = *reinterpret_cast<const float4 *>(a + k), tv = *reinterpret_cast<const float4 *>(t + k);
            acc = acc + ((fabsf(av.x - tv.x) + fabsf(av.y - tv.y)) + (fabsf(av.z - tv.z) + fabsf(av.w - tv.w)));
        }
    } else {
        for (int k = 0; k < K; ++k) {
            const float diff = a[k] - t[k];
            acc = fmaf(diff, diff, acc);
        }
    }
    return acc;
}

// CH: 0 = the MFMA modes' dot chain (lp_chain_dot), 1 = L1 direct, 2 = L2 direct
template <bool VEC4, int CH = 0>
__device__ __forceinline__ float lp_staged_segment(const float *__restrict__ A, int64_t lda,
                                                   const float *__restrict__ T, int64_t ldt, int K, int qi, int ci,
                                                   float *qs, float *es, float acc)
{
    const int lane = threadIdx.x & 63;
    // Full chunks (16-byte aligned rows): software-pipelined -- the NEXT chunk's 16 row loads are issued before
    // the current chunk's sequential chain runs, so the chain (32 dependent FMAs fed from LDS) hides under the
    // loads' latency instead of following it: a pair costs one load latency plus the chains, not one per chunk.
    // The 16 in-flight float4 are NAMED scalars (macro-expanded): as arrays carried around the chunk loop hipcc
    // left them in scratch memory (272 B of private segment, 3.5x slower than no pipelining at all).
    static_assert(KGE_PS_KC == 32, "the fetch / store macros below are written out for 8 float4 pieces per row");
    int k0 = 0;
    if (VEC4 && K >= KGE_PS_KC) {
        float4 q0, q1, q2, q3, q4, q5, q6, q7, e0, e1, e2, e3, e4, e5, e6, e7;
#define KGE_PS_FETCH(IT, KK)                                                                                  \
    {                                                                                                         \
        const int idx_ = lane + 64 * IT, rr_ = idx_ >> 3, pc_ = idx_ & 7;                                     \
        const int rq_ = __shfl(qi, rr_, 64), rc_ = __shfl(ci, rr_, 64);                                      \
        q##IT = *reinterpret_cast<const float4 *>(A + (int64_t)rq_ * lda + (KK) + pc_ * 4);                   \
        e##IT = *reinterpret_cast<const float4 *>(T + (int64_t)rc_ * ldt + (KK) + pc_ * 4);                   \
    }
#define KGE_PS_STORE(IT)                                                                                      \
    {                                                                                                         \
        const int idx_ = lane + 64 * IT, rr_ = idx_ >> 3, pc_ = idx_ & 7;                                     \
        *reinterpret_cast<float4 *>(qs + rr_ * KGE_PS_LD + pc_ * 4) = q##IT;                                  \
        *reinterpret_cast<float4 *>(es + rr_ * KGE_PS_LD + pc_ * 4) = e##IT;                                  \
    }
#define KGE_PS_ALL(M, ...) M(0, ##__VA_ARGS__) M(1, ##__VA_ARGS__) M(2, ##__VA_ARGS__) M(3, ##__VA_ARGS__) \
                           M(4, ##__VA_ARGS__) M(5, ##__VA_ARGS__) M(6, ##__VA_ARGS__) M(7, ##__VA_ARGS__)
        KGE_PS_ALL(KGE_PS_FETCH, 0)
        for (; k0 + KGE_PS_KC <= K; k0 += KGE_PS_KC) {
            KGE_PS_ALL(KGE_PS_STORE)
            if (k0 + 2 * KGE_PS_KC <= K) { KGE_PS_ALL(KGE_PS_FETCH, k0 + KGE_PS_KC) }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); // same wave: LDS executes in order
            if (CH == 0) acc = lp_chain_dot(qs + lane * KGE_PS_LD, es + lane * KGE_PS_LD, KGE_PS_KC, acc);
            else acc = lp_chain_direct<CH == 1>(qs + lane * KGE_PS_LD, es + lane * KGE_PS_LD, KGE_PS_KC, acc);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
#undef KGE_PS_ALL
#undef KGE_PS_STORE
#undef KGE_PS_FETCH
    }
    for (; k0 < K; k0 += KGE_PS_KC) {      // the last, partial chunk (and everything when rows are not 16-byte aligned)
        const int kc = min(KGE_PS_KC, K - k0);
        {
            const int pieces = (kc + 3) >> 2;
            for (int idx = lane; idx < 64 * pieces; idx += 64) { // uniform trip count
                const int rr = idx / pieces, pc = idx - rr * pieces;
                const int rq = __shfl(qi, rr, 64), rc = __shfl(ci, rr, 64);
                const float *qp = A + (int64_t)rq * lda + k0 + pc * 4;
                const float *ep = T + (int64_t)rc * ldt + k0 + pc * 4;
                const int left = kc - pc * 4;
                float4 qv, ev;
                qv.x = qp[0]; ev.x = ep[0];
                qv.y = left > 1 ? qp[1] : 0.f; ev.y = left > 1 ? ep[1] : 0.f;
                qv.z = left > 2 ? qp[2] : 0.f; ev.z = left > 2 ? ep[2] : 0.f;
                qv.w = left > 3 ? qp[3] : 0.f; ev.w = left > 3 ? ep[3] : 0.f;
                *reinterpret_cast<float4 *>(qs + rr * KGE_PS_LD + pc * 4) = qv;
                *reinterpret_cast<float4 *>(es + rr * KGE_PS_LD + pc * 4) = ev;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (CH == 0) acc = lp_chain_dot(qs + lane * KGE_PS_LD, es + lane * KGE_PS_LD, kc, acc);
        else acc = lp_chain_direct<CH == 1>(qs + lane * KGE_PS_LD, es + lane * KGE_PS_LD, kc, acc);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    return acc;
}

// plain direct modes (no rank-1 term): -sum_k |q - t| resp. -sum_k (q - t)^2, bit-identical to lp_pair_score
template <bool VEC4, bool L1>
__device__ __forceinline__ float lp_pair_score_staged_direct(const kge_lp_desc &d, int qi, int ci, float *qs, float *es)
{
    return -lp_staged_segment<VEC4, L1 ? 1 : 2>(d.A0, d.lda0, d.T0, d.ldt0, d.K0, qi, ci, qs, es, 0.0f);
}

// MFMA modes only (KGE_LP_IS_MFMA); (qi, ci) must be valid rows on every lane
template <bool VEC4>
__device__ __forceinline__ float lp_pair_score_staged(const kge_lp_desc &d, int qi, int ci, float *qs, float *es)
{
    float acc = lp_staged_segment<VEC4>(d.A0, d.lda0, d.T0, d.ldt0, d.K0, qi, ci, qs, es, 0.0f);
    if (d.K1 > 0) acc = lp_staged_segment<VEC4>(d.A1, d.lda1, d.T1, d.ldt1, d.K1, qi, ci, qs, es, acc);
    return lp_epilogue_any(d, acc, qi, ci);
}

// tuning knob for experiments (env KGE_LP_TARGET_BLOCKS), default `dflt`
static inline int kge_env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

static inline int kge_lp_desc_check(const kge_lp_desc *d)
{
    if (!d) return KGE_EINVAL;
    if (d->mode < KGE_LP_DOT || d->mode > KGE_LP_L2_PROJD) return KGE_EINVAL;
    if (d->B < 0 || d->N < 0 || d->K0 <= 0 || d->K1 < 0) return KGE_EINVAL;
    if (d->B == 0 || d->N == 0) return 0; // empty problem: nothing is dereferenced
    if (!d->A0 || !d->T0) return KGE_EINVAL;
    if (d->K1 > 0 && (!d->A1 || !d->T1)) return KGE_EINVAL;
    if (d->K1 > 0 && d->mode != KGE_LP_DOT) return KGE_EINVAL;
    if (d->mode == KGE_LP_L2_EXPAND && (!d->qn || !d->en)) return KGE_EINVAL;
    if (d->mode >= KGE_LP_L2_PROJH) {
        if (!d->qn || !d->en || !d->Wq || d->ldw < 2 || !d->scal || d->scal_ld < d->N || !d->r_idx) return KGE_EINVAL;
        if (d->mode == KGE_LP_L2_PROJD && !d->yc) return KGE_EINVAL;
        return 0;
    }
    if (d->Wq && (!d->scal || d->scal_ld < 1 || (d->scal_ld > 1 && !d->r_idx))) return KGE_EINVAL;
    if (d->Wq && d->mode < KGE_LP_L1_DIRECT) return KGE_EINVAL;
    return 0;
}

// ---- free-running one-product count kernel (lp_hi_stream.hip), launched by kge_lp_split_count ----------------------
struct kge_hi_stream_params {
    const char *Ef;             // candidates: FRAGMENT-MAJOR hi table [rows_p / 32][units_p][64 lanes][16 B] (kge_lp_hi_rows, frag = 1)
    const char *Qh;             // queries: planar hi operand [q_rows][q_row_bytes]
    int64_t q_row_bytes;
    int units, units_p;         // k16 units holding data / units per 32-row group of Ef
    int64_t rows_p;             // candidate rows of Ef (a multiple of 64)
    int64_t q_rows;             // rows of Qh (a multiple of 96)
    int64_t B;                  // queries (thr / raw_count are indexed by query)
    const float2 *thr;
    const float4 *thr4;         // projection modes: (a_lo, a_hi, p_i, z_i)
    const float *X;             // projection modes: X (n_rel, ldx)
    int64_t ldx;
    const int64_t *r_idx;
    const float *yc;
    int32_t *raw_count;
    int32_t *list;
    int32_t cap;
    int32_t *list_count;
    float *overflow;
    const int32_t *col_q;       // optional: column -> query id (< 0: padding)
    const int32_t *members;     // grouped launch (r06): [q_rows][sets] query ids (< 0: unused) -- Qh row = one column of up to `sets` queries
    int32_t *region_count;      // optional [q_panels * 3], zeroed: the list is cut into REGIONS of region_cap entries, one per
    int32_t region_cap;         // (panel, 32-query sub-tile); entries land in their query's region, list_count is not touched
    const int64_t *true_idx;    // optional: GLOBAL id of the entity whose exact score is the query's threshold s_true ...
    int64_t c_base;             // ... local candidate c is global entity c_base + c
    // filled by kge_hi_stream_launch
    int q_panels, c_tiles, qg;  // qg: panels interleaved under one candidate sweep (a power of two dividing the blocks per XCD)
    int64_t n_items;
    int panel_bytes;
};
int kge_hi_stream_max_units(void);
int kge_hi_stream_launch(kge_hi_stream_params p, int pm, int num_cus, hipStream_t s);
// ... and its chunked-panel form for rows too long for a resident panel (lp_hi_chunk.hip; PM = 0, one global list)
int kge_hi_chunk_supported(int units);
int kge_hi_chunk_launch(kge_hi_stream_params p, int num_cus, hipStream_t s);

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: remember what was set per device
// (`cache`: 16 zero-initialised ints owned by the launch site), not once per process.
static inline int kge_ensure_dyn_smem(const void *func, int smem, int *cache)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = -1;
    if (dev >= 0 && __atomic_load_n(&cache[dev], __ATOMIC_RELAXED) >= smem) return 0;
    hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    if (dev >= 0) __atomic_store_n(&cache[dev], smem, __ATOMIC_RELAXED);   // (racing first calls both set the attribute: idempotent)
    return 0;
}
