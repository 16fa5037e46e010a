// Certified f16-split prefilter for the fused rank count (gfx950).
//
// The rank of a test fact only needs  #{c : s[i,c] >= s_true[i]}  (get_rank,
// utils/operations.py:37-61), not the scores.  For TransE-L2 through the norm
// expansion  s = -max(||q||^2 + ||e||^2 - 2 q.e, 0)  the comparison
// s[i,c] >= s_true[i] is  q.e - ||e_c||^2/2 >= (||q_i||^2 + s_true_i)/2 ;
// for DistMult / ComplEx (KGE_LP_DOT, K = K0 + K1 columns) it is  q.e >= s_true_i.
// This file evaluates the left side APPROXIMATELY on the f16 matrix cores
// (16x the fp32 MFMA rate) with a rigorous error bound eps_i:
//
//   x  = hi + lo + rho,  hi = f16(x), lo = f16(x - hi), |rho| <= 2^-22 |x|
//   q.e ~ sum_k  qh*eh + qh*el + ql*eh          (3 v_mfma_f32_32x32x16_f16, fp32 accumulate)
//
// and classifies every (query, candidate) pair against two per-query thresholds:
//   acc >= a_hi : certainly counted          acc < a_lo : certainly not counted
//   a_lo <= acc < a_hi : UNCERTAIN -> appended to a list and re-scored by the
//   exact scalar chain (kge_common.h: lp_pair_score_staged, bit-identical to the
//   fp32 MFMA tile kernel and to oracle/kge_oracle.c).
// raw_count[i] first receives #{acc >= a_lo}; kge_lp_split_recheck then takes 1
// off for every listed pair whose exact score is below s_true.  The resulting
// counts are therefore EXACTLY those of kge_lp_count_ge -- integer work stays
// bit-exact -- while ~99.9% of the pairs never touch the fp32 pipe.
//
// Error bound used for the thresholds (split_thr_kernel), per unit of
// P >= sum_k |q_k e_k| (+ |augmentation term|), P = ||q|| * max||e|| (+ max||e||^2 / 2):
//   split residual   3 * 2^-22            (ql*el dropped, rho_q*e, q*rho_e)
//   accumulation     c_acc * n_terms * 2^-24  (n_terms = 3 * 16 * units products; c_acc = 2 covers any fp32
//                                          adder, rounding or truncating, in any order; 1.25 when the
//                                          device passed kge_mfma_f16_selftest, measured 9/8)
//   exact chain      K * 2^-24            (the scalar fmaf chain it is compared with)
// plus absolute
// terms for f16 subnormal lo parts (flushed or not) and for the roundings of
// the threshold arithmetic itself.  tools/probe/mfma_probe.hip measures what the
// MFMA really does (two passes of acc + 8 products, addends truncated 24 bits
// below the largest, one RNE rounding; subnormals kept): <= 9 * 2^-24 per pass,
// inside the assumption.  tests/test_gpu_parity.py checks the counts against the
// exact kernel, also with the band shrunk 16x.
//
// Data layout: a split operand is [rows_p][units_p] cells of 64 bytes,
//   cell = [hi k0..7][hi k8..15][lo k0..7][lo k8..15]   (f16, k within the 16-unit)
// rows_p = rows rounded up to the tile (256 candidates / 192 queries; zero rows),
// units_p = k16 units rounded up to 2; one extra column carries the augmentation
// (L2: 1 for queries, -||e||^2/2 for candidates; DOT: a per-query guard value
// against 0); padding candidates hold -65504 there and can never count.  L2
// operands are scaled by 2^12 (|x| <= 4 is implied by the evaluator's norm guard,
// L2_EXPAND_LIMIT), DOT operands by a power of two derived on the device from
// the squared-norm maxima (split_scale).
//
// Kernel: block tile 256 candidates (MFMA rows) x 192 queries (MFMA columns =
// lanes), 8 waves of 64 x 96 (2x3 tiles of 32x32: 96 accumulator VGPRs, two waves
// per SIMD -- accumulators stay in architectural VGPRs, which the VALU epilogue
// can compare directly; AGPR accumulators would cost a v_accvgpr_read each).
// K is staged 32 at a time into double-buffered LDS by LDS-DMA
// (global_load_lds_dwordx4, pieces issued between the MFMA groups) with an XOR
// swizzle (16-byte chunk c of row r sits at chunk c ^ ((r>>1)&7)): the fragment
// ds_read_b128 are bank-conflict free without padding.  Queries on the lane axis
// make thresholds and counters per-lane constants, so the epilogue is two
// compares and an add-with-carry per element.
#include "kge_common.h"
#ifndef KGE_BUILD_NO_SLP
#error "build with -fno-slp-vectorize -DKGE_BUILD_NO_SLP=1 (torchkge_amd/csrc/build.py): SLP-packed v_pk_fma_f32 with a lane-crossing op_sel misreads beside co-executing MFMAs (profiles/r06/slp_bisect.txt)"
#endif

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int TQ = 192, TC = 256;               // 256 accumulator registers leave hipcc no slack: 4 x 3 tiles
constexpr int NT = 3;                           // 32x32 query tiles per wave (2 wave columns)
constexpr int E_STAGE_BYTES = TC * 128;         // candidate operand, one stage (2 k16 units)
constexpr int Q_STAGE_BYTES = TQ * 128;
constexpr int STAGE_BYTES = E_STAGE_BYTES + Q_STAGE_BYTES;
constexpr int UNC_CAP = 2048;                   // uncertain pairs buffered per tile
constexpr int GSETS = 4;                        // grouped columns: queries that share one query row (threshold sets per column)
// + per-panel (thr4, X row) of the projection modes, or the (a_lo, a_hi) sets + counters of a grouped panel
constexpr int SMEM_BYTES = 2 * STAGE_BYTES + 16 + UNC_CAP * 4 + TQ * 20 + TQ * GSETS * 12 + 16;
constexpr int SPLIT_SCALE_LOG2 = 12;

struct SplitParams {
    const char *Es, *Qs;  // split operands
    int row_bytes;        // units_p * 64
    int units;            // k16 units holding data (MFMA work)
    int stages;           // units_p / 2
    int64_t B, N;
    const float2 *thr;    // (a_lo, a_hi) per padded query, scaled like the accumulators
    const float4 *thr4;   // projection modes: (a_lo, a_hi, p_i, z_i) per padded query
    const float *X;       // projection modes: X (n_rel, ldx), ldx % 4 == 0, readable up to the padded tile edge
    int64_t ldx;
    const int64_t *r_idx; // row of X per query
    const float *yc;      // PROJD: y_c per (padded) candidate
    int32_t *raw_count;
    int32_t *list;        // cap x (query, candidate)
    int32_t cap;
    int32_t *list_count;  // device scalar
    float *overflow;      // set to 1 when a buffer or the list overflowed
    // Columns instead of queries (queries that share a key share the query ROW: its accumulators are computed once).
    // col_q: column -> query id (< 0: padding column) for a launch whose columns carry ONE query each (NULL: column
    // == query); members: [column][GSETS] query ids (< 0: unused set) for the grouped launch (template GS > 0).
    const int32_t *col_q, *members;
    int q_panels, c_tiles;
    int qg;               // query panels interleaved under one sweep of the candidate tiles (work order)
    int64_t n_items;
    int dbg;              // env KGE_SPLIT_DBG (timing probes, wrong results): 1 no global loads, 4 no epilogue,
                          // 16 no LDS fragment reads, 32 no barriers, 128 every block streams tile (0,0),
                          // 1024 hi*hi product only (one-product first level), 2048 half of the LDS-DMA pieces,
                          // 4096 (LV = 1) cycle stamps of the stage phases into the head of the pair list
                          // (a planar hi table would move half the bytes)
};

// ---- operand preparation ---------------------------------------------------
// power-of-two scale that puts rows of squared norm <= norm2max just inside f16 range
__device__ __forceinline__ float split_scale(float norm2max)
{
    const float m = sqrtf(norm2max);
    if (!(m > 0.f) || !(m < INFINITY)) return 1.0f;
    float e = floorf(log2f(16384.0f / m)) - 1.0f;      // one binade of slack for the roundings above
    e = fminf(fmaxf(e, -100.0f), 100.0f);
    return ldexpf(1.0f, (int)e);
}

struct SplitRowsParams {
    const float *X0, *X1;     // segment 1 optional (ComplEx: [Re | Im])
    int64_t ld0, ld1;
    int K0, K1;
    int64_t rows, rows_p;
    int aug_mode;             // 0 none; 1: aug[row]*aug_mul (candidates, L2); 2: aug_mul (queries, L2);
                              // 3: query guard column of the DOT mode; 4: 0 for real rows (candidates, DOT)
    const float *aug;
    float aug_mul;
    const float *nmax0, *nmax1;   // device scalars: squared-norm maxima -> scale (NULL: 2^12)
    int units_p;
    uint4 *out;
    float *cell_ss;           // optional [units_p][rows_p]: sum of squares of the cell's 16 data values (unscaled)
    const int64_t *row_index; // optional: output row r is built from source row row_index[r] of X0 / X1 / aug (gather)
};

// A block converts tiles of 16 rows x 16 k16 cells: 16 consecutive threads read one row's 1-KiB run (float4 loads where the
// cell lies inside one K-segment and is 16-byte aligned) and write its 16 cells; the cells' sums of squares go through
// LDS so that the unit-major cell_ss array is written in 64-byte runs as well (a thread-per-cell store there touches one
// cache line per cell: it cost as much as the split table itself).
__global__ __launch_bounds__(256) void split_rows_kernel(const SplitRowsParams p)
{
    __shared__ float ss_s[16][17];
    float scale = (float)(1 << SPLIT_SCALE_LOG2), nmax = 0.f;
    if (p.nmax0) {
        nmax = *p.nmax0 + (p.nmax1 ? *p.nmax1 : 0.f);
        scale = split_scale(nmax);
    }
    const int K = p.K0 + p.K1;
    const int tiles_u = (p.units_p + 15) / 16;
    const int64_t n_tiles = (p.rows_p / 16) * tiles_u;          // rows_p is a multiple of the count kernel's tile
    const int tr_ = threadIdx.x >> 4, tu_ = threadIdx.x & 15;
    const bool vec0 = (p.ld0 % 4 == 0) && ((size_t)p.X0 & 15) == 0;
    const bool vec1 = p.X1 && (p.ld1 % 4 == 0) && ((size_t)p.X1 & 15) == 0 && (p.K0 % 4 == 0);
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t row = (tile / tiles_u) * 16 + tr_;
        const int u = (int)(tile % tiles_u) * 16 + tu_;
        const int64_t srow = (p.row_index && row < p.rows) ? p.row_index[row] : row;    // (gathered source row)
        float ss = 0.f;
        if (u < p.units_p) {
            const int k0 = u * 16;
            float xs[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) xs[e] = 0.f;
            if (row < p.rows) {
                if (k0 + 16 <= p.K0 && vec0) {
                    const float4 *src = reinterpret_cast<const float4 *>(p.X0 + srow * p.ld0 + k0);
#pragma unroll
                    for (int v = 0; v < 4; ++v) { const float4 t = src[v]; xs[4 * v] = t.x; xs[4 * v + 1] = t.y; xs[4 * v + 2] = t.z; xs[4 * v + 3] = t.w; }
                } else if (k0 >= p.K0 && k0 + 16 <= K && vec1) {
                    const float4 *src = reinterpret_cast<const float4 *>(p.X1 + srow * p.ld1 + (k0 - p.K0));
#pragma unroll
                    for (int v = 0; v < 4; ++v) { const float4 t = src[v]; xs[4 * v] = t.x; xs[4 * v + 1] = t.y; xs[4 * v + 2] = t.z; xs[4 * v + 3] = t.w; }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int k = k0 + e;
                        if (k < p.K0) xs[e] = p.X0[srow * p.ld0 + k];
                        else if (k < K) xs[e] = p.X1[srow * p.ld1 + (k - p.K0)];
                    }
                }
            }
            union { _Float16 h[16]; uint4 v[2]; } hi, lo;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = k0 + e;
                float x = xs[e];
                if (row < p.rows) {
                    if (k < K) ss = fmaf(x, x, ss);
                    else if (k == K && p.aug_mode == 1) x = p.aug[srow] * p.aug_mul;
                    else if (k == K && p.aug_mode == 2) x = p.aug_mul;
                    else if (k == K && p.aug_mode == 3) x = 0.25f * (sqrtf(p.aug[srow]) + sqrtf(nmax) * 0.00390625f);
                }
                x *= scale;
                if (row < p.rows && k == K && p.aug_mode == 3) x = fmaxf(x, 1.0f);
                _Float16 h = (_Float16)x;                   // round to nearest even
                _Float16 l = (_Float16)(x - (float)h);      // x - hi is exact in fp32
                if (row >= p.rows && k == K && (p.aug_mode == 1 || p.aug_mode == 4)) {
                    // padding candidate: hi = lo = -65504 in the column that meets the queries' guard
                    // column drives its accumulator below every threshold (L2: <= -2*65504*2^12 against
                    // >= -16*2^24 for norm-guarded queries; DOT: <= -32752*S_q*||q|| against >= -16384*S_q*||q||)
                    h = (_Float16)(-65504.f);
                    l = (_Float16)(-65504.f);
                }
                hi.h[e] = h;
                lo.h[e] = l;
            }
            uint4 *o = p.out + (row * p.units_p + u) * 4;
            o[0] = hi.v[0]; o[1] = hi.v[1]; o[2] = lo.v[0]; o[3] = lo.v[1];
        }
        if (p.cell_ss) {        // (block-uniform)
            ss_s[tu_][tr_] = ss;
            __syncthreads();
            const int u2 = (int)(tile % tiles_u) * 16 + tr_;     // this thread now stores unit u2, row tu_ of the tile
            if (u2 < p.units_p) p.cell_ss[(int64_t)u2 * p.rows_p + (tile / tiles_u) * 16 + tu_] = ss_s[tr_][tu_];
            __syncthreads();
        }
    }
}

// ---- one-product level: PLANAR hi operands -----------------------------------------------------------------------
// [rows_p][units_p][32 bytes]: the f16 hi parts of the 16 values of a k16 unit (units_p a multiple of 4: one 128-byte
// row segment = one stage of the LV = 1 count kernel).  Two extra columns K, K+1: L2 candidates carry hi and lo of
// -||e||^2/2 there (queries 1, 1), so the norm term keeps its 2^-22 precision; DOT queries their guard column at K.
// Also emitted: dn2[row] = ||x - hi(x)||^2 (unscaled; the exact fp32 differences, summed) and its maximum.
struct HiRowsParams {
    const float *X0, *X1;
    int64_t ld0, ld1;
    int K0, K1;
    int64_t rows, rows_p;
    int aug_mode;             // as SplitRowsParams
    const float *aug;
    float aug_mul;
    const float *nmax0, *nmax1;
    int units_p;
    uint4 *out;
    float *dn2;               // optional (rows)
    float *dn2max;            // optional device scalar, max folded in
    const int64_t *row_index;
    const float *nm_bmax;     // optional [2][nm_blocks]: per-block squared-norm maxima of the two segments (dot_table_norm_max_kernel)
    int nm_blocks;            //   -- every block folds them into *nmax0 / *nmax1 on its way in, block 0 stores the two scalars
    float *dn_bmax;           // optional [gridDim.x]: the blocks' residual maxima as plain stores INSTEAD of the dn2max atomic
    const float *prev_nmax;   // hi_rows_frag_kernel<true> (r06, ONE pass over the table): the two squared-norm maxima a PREVIOUS
    float *nm_out;            //   evaluation measured fix the scale; this pass's maxima per block go to nm_out[2][gridDim.x] and
                              //   the consumer (dot_query_pipeline_kernel) raises the overflow flag if they ask for another scale
    int frag;                 // 1: FRAGMENT-MAJOR output [rows_p / 32][units_p][64][16 B] -- chunk (row % 32) + 32 * k-half of
                              // the 1-KiB block of (32-row group, unit): the A operand of v_mfma_f32_32x32x16_f16 in lane
                              // order, one coalesced global_load_dwordx4 per block (lp_hi_stream.hip)
};

__global__ __launch_bounds__(256) void hi_rows_kernel(const HiRowsParams p)
{
    __shared__ unsigned bmax[4];
    float scale = (float)(1 << SPLIT_SCALE_LOG2), nmax = 0.f;
    if (p.nm_bmax) {        // (as query_pipeline_kernel: block maxima -> the two scalars, folded into what they hold)
        __shared__ unsigned red[8];
        unsigned m0 = 0u, m1 = 0u;
        for (int j = threadIdx.x; j < p.nm_blocks; j += 256) {
            m0 = max(m0, __float_as_uint(p.nm_bmax[j]));
            m1 = max(m1, __float_as_uint(p.nm_bmax[p.nm_blocks + j]));
        }
        for (int off = 32; off > 0; off >>= 1) {
            m0 = max(m0, (unsigned)__shfl_xor((int)m0, off, 64));
            m1 = max(m1, (unsigned)__shfl_xor((int)m1, off, 64));
        }
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = m0; red[4 + (threadIdx.x >> 6)] = m1; }
        __syncthreads();
        m0 = max(max(red[0], red[1]), max(red[2], red[3]));
        m1 = max(max(red[4], red[5]), max(red[6], red[7]));
        const float n0 = __uint_as_float(max(m0, __float_as_uint(*p.nmax0)));
        const float n1 = p.nmax1 ? __uint_as_float(max(m1, __float_as_uint(*p.nmax1))) : 0.f;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            *const_cast<float *>(p.nmax0) = n0;
            if (p.nmax1) *const_cast<float *>(p.nmax1) = n1;
        }
        nmax = n0 + n1;
        scale = split_scale(nmax);
    } else if (p.nmax0) {
        nmax = *p.nmax0 + (p.nmax1 ? *p.nmax1 : 0.f);
        scale = split_scale(nmax);
    }
    const float inv2 = 1.0f / (scale * scale);
    const int K = p.K0 + p.K1;
    const int tr_ = threadIdx.x >> 4, tu_ = threadIdx.x & 15;
    const bool vec0 = (p.ld0 % 4 == 0) && ((size_t)p.X0 & 15) == 0;
    const bool vec1 = p.X1 && (p.ld1 % 4 == 0) && ((size_t)p.X1 & 15) == 0 && (p.K0 % 4 == 0);
    float dmax = 0.f;
    for (int64_t r0 = (int64_t)blockIdx.x * 16; r0 < p.rows_p; r0 += (int64_t)gridDim.x * 16) {
        const int64_t row = r0 + tr_;
        const bool real = row < p.rows;
        const int64_t srow = (p.row_index && real) ? p.row_index[row] : row;
        float dn = 0.f;
        for (int u = tu_; u < p.units_p; u += 16) {
            const int k0 = u * 16;
            float xs[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) xs[e] = 0.f;
            if (real) {
                if (k0 + 16 <= p.K0 && vec0) {
                    const float4 *src = reinterpret_cast<const float4 *>(p.X0 + srow * p.ld0 + k0);
#pragma unroll
                    for (int v = 0; v < 4; ++v) { const float4 t = src[v]; xs[4 * v] = t.x; xs[4 * v + 1] = t.y; xs[4 * v + 2] = t.z; xs[4 * v + 3] = t.w; }
                } else if (k0 >= p.K0 && k0 + 16 <= K && vec1) {
                    const float4 *src = reinterpret_cast<const float4 *>(p.X1 + srow * p.ld1 + (k0 - p.K0));
#pragma unroll
                    for (int v = 0; v < 4; ++v) { const float4 t = src[v]; xs[4 * v] = t.x; xs[4 * v + 1] = t.y; xs[4 * v + 2] = t.z; xs[4 * v + 3] = t.w; }
                } else if (k0 < K) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int k = k0 + e;
                        if (k < p.K0) xs[e] = p.X0[srow * p.ld0 + k];
                        else if (k < K) xs[e] = p.X1[srow * p.ld1 + (k - p.K0)];
                    }
                }
            }
            union { _Float16 h[16]; uint4 v[2]; } hi;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = k0 + e;
                float x = xs[e] * scale;
                _Float16 h = (_Float16)x;                   // round to nearest even
                if (real && k < K) {
                    const float d = x - (float)h;           // exact in fp32
                    dn = fmaf(d, d, dn);
                }
                if (k == K || k == K + 1) {                 // the augmentation columns
                    float a = 0.f;
                    if (real) {
                        if (p.aug_mode == 1) {              // L2 candidates: hi (column K) and lo (column K + 1) of -||e||^2/2
                            const float full = p.aug[srow] * p.aug_mul * scale;
                            const _Float16 fh = (_Float16)full;
                            a = k == K ? (float)fh : (float)(_Float16)(full - (float)fh);
                        } else if (p.aug_mode == 2) {       // L2 queries: 1 against both
                            a = p.aug_mul * scale;
                        } else if (p.aug_mode == 3 && k == K) {   // DOT queries: the guard column (see split_rows_kernel)
                            a = fmaxf(0.25f * (sqrtf(p.aug[srow]) + sqrtf(nmax) * 0.00390625f) * scale, 1.0f);
                        }
                    } else if ((p.aug_mode == 1 || (p.aug_mode == 4 && k == K))) {
                        a = -65504.f;                       // padding candidate: can never count
                    }
                    h = (_Float16)a;
                }
                hi.h[e] = h;
            }
            if (p.frag) {
                uint4 *o = p.out + (((row >> 5) * p.units_p + u) << 6) + (row & 31);
                o[0] = hi.v[0]; o[32] = hi.v[1];
            } else {
                uint4 *o = p.out + (row * p.units_p + u) * 2;
                o[0] = hi.v[0]; o[1] = hi.v[1];
            }
        }
        dn += __shfl_xor(dn, 8, 64);    // the 16 lanes of a row sit in one aligned group of the wavefront
        dn += __shfl_xor(dn, 4, 64);
        dn += __shfl_xor(dn, 2, 64);
        dn += __shfl_xor(dn, 1, 64);
        dn *= inv2 * 1.0001f;           // (fp32 summation error: K * 2^-24 relative)
        if (real && tu_ == 0 && p.dn2) p.dn2[row] = dn;
        if (real) dmax = fmaxf(dmax, dn);
    }
    if (p.dn2max || p.dn_bmax) {
        unsigned m = __float_as_uint(dmax);
        for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
        if ((threadIdx.x & 63) == 0) bmax[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned mm = max(max(bmax[0], bmax[1]), max(bmax[2], bmax[3]));
            if (p.dn_bmax) p.dn_bmax[blockIdx.x] = __uint_as_float(mm);
            else kge_atomic_max_u32(reinterpret_cast<unsigned *>(p.dn2max), mm);
        }
    }
}

// ---- the same table for DOT candidates, FRAGMENT-MAJOR, with coalesced traffic on both sides (r06) -----------------------
// hi_rows_kernel reads 64 bytes per lane (a wave's float4 load touches 16 B of 64 different 64-byte pieces) and writes a
// fragment block 64 bytes at a time: at ComplEx d = 512 on 4.59 M entities it moves 28.8 GB in 8.9 ms (3.2 TB/s), the
// address pipes being the limit, not HBM.  Here a block takes a 32-row group and walks the row in spans of 256 columns:
// a wave reads 1 KiB of ONE row per instruction (lane l: columns 4 l .. 4 l + 3), converts, and leaves the f16 values in
// an LDS tile laid out like the output -- [unit][k-half][row][16 B], a 16-byte pad per k-half block: the 16 lanes of a
// write pass hit 16 different 8-byte bank pairs -- from which every wave then copies whole 1-KiB fragment blocks to the
// table, one coalesced store per block.  Two LDS tiles, ONE barrier per span, the next span's loads in flight across it.
// Same hi values as hi_rows_kernel (aug_mode 4: guard column K = 0 for real rows, -65504 for padding rows); the residual
// sums are the same exact differences in another order (a bound input: any order, see the 1.0001 below).
// Needs float4-readable rows (K0, K1, ld % 4 == 0, 16-byte aligned bases).
constexpr int HF_HST = 512 + 16, HF_UST = 2 * HF_HST;     // LDS strides of a k-half block / a unit

// FUSED (r06): no norm pass in front -- the scale comes from the maxima of the PREVIOUS evaluation (p.prev_nmax; any power
// of two under which nothing overflows is a valid scale: the band is built from residuals measured HERE), the rows'
// squared norms are summed on the way (a bound input: any order) and their maxima left per block in p.nm_out for the
// query pipeline, which checks that they still ask for the scale that was used.
template <bool FUSED>
__global__ __launch_bounds__(256) void hi_rows_frag_kernel(const HiRowsParams p)
{
    __shared__ __attribute__((aligned(16))) unsigned char tile[2][16 * HF_UST];
    __shared__ unsigned bmax[4], nbmax[8];
    float scale, nmax;
    if (FUSED) {
        nmax = p.prev_nmax[0] + p.prev_nmax[1];
        scale = split_scale(nmax);
    } else {       // the block maxima of dot_table_norm_max_kernel -> the two scalars (as hi_rows_kernel)
        __shared__ unsigned red[8];
        unsigned m0 = 0u, m1 = 0u;
        for (int j = threadIdx.x; j < p.nm_blocks; j += 256) {
            m0 = max(m0, __float_as_uint(p.nm_bmax[j]));
            m1 = max(m1, __float_as_uint(p.nm_bmax[p.nm_blocks + j]));
        }
        for (int off = 32; off > 0; off >>= 1) {
            m0 = max(m0, (unsigned)__shfl_xor((int)m0, off, 64));
            m1 = max(m1, (unsigned)__shfl_xor((int)m1, off, 64));
        }
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = m0; red[4 + (threadIdx.x >> 6)] = m1; }
        __syncthreads();
        m0 = max(max(red[0], red[1]), max(red[2], red[3]));
        m1 = max(max(red[4], red[5]), max(red[6], red[7]));
        const float n0 = __uint_as_float(max(m0, __float_as_uint(*p.nmax0)));
        const float n1 = p.nmax1 ? __uint_as_float(max(m1, __float_as_uint(*p.nmax1))) : 0.f;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            *const_cast<float *>(p.nmax0) = n0;
            if (p.nmax1) *const_cast<float *>(p.nmax1) = n1;
        }
        nmax = n0 + n1;
        scale = split_scale(nmax);
    }
    const float inv2 = 1.0f / (scale * scale);
    const int K = p.K0 + p.K1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nspan = (p.units_p + 15) >> 4;
    const int64_t ngroups = p.rows_p >> 5;
    const int wo = (lane >> 2) * HF_UST + ((lane >> 1) & 1) * HF_HST + (lane & 1) * 8;     // this lane's 8 bytes of a tile row
    const int ro = (lane >> 5) * HF_HST + (lane & 31) * 16;                                 // this lane's chunk of a fragment block
    float4 v[8];
    float dn[8], nn0[8], nn1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dn[j] = nn0[j] = nn1[j] = 0.f;
    float dmax = 0.f, n0max = 0.f, n1max = 0.f;
    auto load = [&](int64_t g, int s) {
        const int col = s * 256 + 4 * lane;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t row = g * 32 + wv * 8 + j;
            v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < p.rows && col < K)
                v[j] = col < p.K0 ? *reinterpret_cast<const float4 *>(p.X0 + row * p.ld0 + col)
                                  : *reinterpret_cast<const float4 *>(p.X1 + row * p.ld1 + (col - p.K0));
        }
    };
    int64_t g = blockIdx.x;
    int s = 0, it = 0;
    bool have = g < ngroups;
    if (have) load(g, s);
    while (have) {                      // (block-uniform)
        unsigned char *tl = tile[it & 1];
        const int col = s * 256 + 4 * lane;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool real = g * 32 + wv * 8 + j < p.rows;
            const float xs[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
            if (FUSED) {        // (padding lanes hold zeros)
                const float ss = fmaf(xs[0], xs[0], fmaf(xs[1], xs[1], fmaf(xs[2], xs[2], xs[3] * xs[3])));
                if (col < p.K0) nn0[j] += ss; else nn1[j] += ss;
            }
            _Float16 h[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = xs[e] * scale;
                h[e] = (_Float16)x;                         // round to nearest even
                if (real && col < K) {                      // (K % 4 == 0: a float4 is data or padding as a whole)
                    const float d = x - (float)h[e];        // exact in fp32
                    dn[j] = fmaf(d, d, dn[j]);
                }
            }
            if (col == K && !real) h[0] = (_Float16)(-65504.f);      // the guard column of a padding candidate: can never count
            union { _Float16 hh[4]; uint2 u; } pk;
            pk.hh[0] = h[0]; pk.hh[1] = h[1]; pk.hh[2] = h[2]; pk.hh[3] = h[3];
            *reinterpret_cast<uint2 *>(tl + wo + (wv * 8 + j) * 16) = pk.u;
        }
        int64_t g2 = g;
        int s2 = s + 1;
        if (s2 == nspan) { s2 = 0; g2 += gridDim.x; }
        const bool have2 = g2 < ngroups;
        if (have2) load(g2, s2);
        __syncthreads();                // the tile is complete; the other tile's readers of two spans ago are long past
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) {
            const int ul = wv * 4 + uu, u = s * 16 + ul;
            if (u < p.units_p)
                p.out[((g * p.units_p + u) << 6) + lane] = *reinterpret_cast<const uint4 *>(tl + ul * HF_UST + ro);
        }
        if (s == nspan - 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float a = dn[j];
                for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
                a *= inv2 * 1.0001f;    // (fp32 summation error: K * 2^-24 relative)
                if (g * 32 + wv * 8 + j < p.rows) dmax = fmaxf(dmax, a);
                dn[j] = 0.f;
                if (FUSED) {
                    float b0 = nn0[j], b1 = nn1[j];
                    for (int off = 32; off > 0; off >>= 1) { b0 += __shfl_xor(b0, off, 64); b1 += __shfl_xor(b1, off, 64); }
                    n0max = fmaxf(n0max, b0 * 1.0001f);     // (summation order: K * 2^-24 relative)
                    n1max = fmaxf(n1max, b1 * 1.0001f);
                    nn0[j] = nn1[j] = 0.f;
                }
            }
        }
        g = g2; s = s2; have = have2; ++it;
    }
    unsigned m = __float_as_uint(dmax);
    if (lane == 0) { bmax[wv] = m; nbmax[wv] = __float_as_uint(n0max); nbmax[4 + wv] = __float_as_uint(n1max); }   // (wave-uniform)
    __syncthreads();
    if (threadIdx.x == 0) {
        p.dn_bmax[blockIdx.x] = __uint_as_float(max(max(bmax[0], bmax[1]), max(bmax[2], bmax[3])));
        if (FUSED) {
            p.nm_out[blockIdx.x] = __uint_as_float(max(max(nbmax[0], nbmax[1]), max(nbmax[2], nbmax[3])));
            p.nm_out[gridDim.x + blockIdx.x] = __uint_as_float(max(max(nbmax[4], nbmax[5]), max(nbmax[6], nbmax[7])));
        }
    }
}

// Squared-norm maxima of the rows of one or two tables ([Re | Im] segments of a DOT candidate table) in ONE sweep, any
// summation order (they fix the operand scale and bound the error band; no score contains them): row_sqnorm_any_kernel
// for both segments without the per-row outputs, the blocks' maxima as plain stores (bmax[2][gridDim.x]) -- the consumer
// (hi_rows_kernel, nm_bmax) reduces them: no same-address atomics, no zero-fill of the scalars.
__global__ __launch_bounds__(256) void dot_table_norm_max_kernel(const float *__restrict__ X0, int64_t ld0, int K0,
                                                                 const float *__restrict__ X1, int64_t ld1, int K1,
                                                                 int64_t rows, float *bmax_out)
{
    __shared__ unsigned wmax[8];
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
    float big0 = 0.f, big1 = 0.f;
    for (int sg = 0; sg < (X1 ? 2 : 1); ++sg) {
        const float *X = sg ? X1 : X0;
        const int64_t ld = sg ? ld1 : ld0;
        const int K = sg ? K1 : K0;
        const bool vec = (K % 4 == 0) && (ld % 4 == 0) && ((size_t)X & 15) == 0;
        float big = 0.f;
        for (int64_t r0 = (int64_t)blockIdx.x * 64; r0 < rows; r0 += (int64_t)gridDim.x * 64) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            if (vec) {
                for (int k = sub * 4; k < K; k += 64) {
                    float4 t[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[j] = *reinterpret_cast<const float4 *>(X + min(r0 + grp * 4 + j, rows - 1) * ld + k);
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
                big = __uint_as_float(max(__float_as_uint(big), __float_as_uint(a)));   // (rows past the end repeat the last row)
            }
        }
        if (sg) big1 = big; else big0 = big;
    }
    unsigned m0 = __float_as_uint(big0), m1 = __float_as_uint(big1);
    for (int off = 32; off > 0; off >>= 1) {
        m0 = max(m0, (unsigned)__shfl_xor((int)m0, off, 64));
        m1 = max(m1, (unsigned)__shfl_xor((int)m1, off, 64));
    }
    if ((threadIdx.x & 63) == 0) { wmax[threadIdx.x >> 6] = m0; wmax[4 + (threadIdx.x >> 6)] = m1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        bmax_out[blockIdx.x] = __uint_as_float(max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])));
        bmax_out[gridDim.x + blockIdx.x] = __uint_as_float(max(max(wmax[4], wmax[5]), max(wmax[6], wmax[7])));
    }
}

// ---- candidate-table preparation of the L2 one-product sweep in ONE pass (r05) -----------------------------------
// What kge_row_sqnorm (en, its maximum) + a zero-fill + kge_lp_hi_rows_frag (hi table, residual maximum) do in three
// launches with the table read twice: a 256-thread block stages 16 rows in LDS; lanes 0..15 run the rows' sequential
// ||e||^2 chains (row_sqnorm_kernel's: acc = fmaf(x_k, x_k, acc), k ascending -- the score contains en, same bits)
// while all threads convert the data units to f16 hi parts (thread = row + 16 * unit: a unit's 16 rows are one 256-byte
// run of the fragment-major table) and sum their residuals; the two augmentation columns follow once en is known.
struct TablePrepParams {
    const float *X;
    int64_t ld, rows, rows_p;
    int K, units_p;
    float *en;                // (rows)
    float *en_max;            // device scalar, max folded in
    uint4 *out;               // fragment-major hi table
    float *dn2max;            // device scalar, max folded in
    float *block_max;         // optional [2][gridDim.x]: the blocks' two maxima as plain stores INSTEAD of the atomics -- ~900
                              // same-address device-scope atomics per scalar serialise for ~50 us at the end of the kernel
                              // (measured in situ: 65 against 14 us); the consumer (kge_lp_query_pipeline) reduces them
    int dbg;                  // env KGE_TP_DBG (timing probes, wrong results): 1 no norm chain, 2 no conversions, 4 no table reads
};

__global__ __launch_bounds__(256) void table_prep_l2_kernel(const TablePrepParams p)
{
    extern __shared__ __attribute__((aligned(16))) float tp_smem[];
    const int K = p.K, LDS_LD = ((K + 3) & ~3) + 4;         // (row stride: 16-byte aligned, 4 floats of padding)
    float *xs = tp_smem;                                    // [16][LDS_LD]
    float *dnp = xs + 16 * LDS_LD;                          // [16][17] partial residual sums
    float *ens = dnp + 16 * 17;                             // [16]
    __shared__ unsigned bmax[2];
    const int tid = threadIdx.x, r = tid & 15, uu = tid >> 4;
    const float scale = (float)(1 << SPLIT_SCALE_LOG2);
    const float inv2 = 1.0f / (scale * scale);
    const int units_d = (K + 15) >> 4;                      // units that hold data columns
    float emax_b = 0.f, dmax_b = 0.f;
    for (int64_t r0 = (int64_t)blockIdx.x * 16; r0 < p.rows_p; r0 += (int64_t)gridDim.x * 16) {
        // stage the 16 rows (rows past the table: zeros)
        const int nv = K >> 2;                              // float4 per row (K % 4 == 0, checked by the host)
        for (int idx = tid; idx < 16 * nv; idx += 256) {
            const int rr = idx / nv, c = idx - rr * nv;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r0 + rr < p.rows && !(p.dbg & 4)) v = *reinterpret_cast<const float4 *>(p.X + (r0 + rr) * p.ld + c * 4);
            *reinterpret_cast<float4 *>(xs + rr * LDS_LD + c * 4) = v;
        }
        __syncthreads();
        const int64_t row = r0 + r;
        const bool real = row < p.rows;
        if (tid < 16 && !(p.dbg & 1)) {                     // the sequential chain of ||e||^2
            const float *x = xs + r * LDS_LD;
            float acc = 0.f;
            for (int k = 0; k < K; k += 4) {
                const float4 t = *reinterpret_cast<const float4 *>(x + k);
                acc = fmaf(t.x, t.x, acc);
                acc = fmaf(t.y, t.y, acc);
                acc = fmaf(t.z, t.z, acc);
                acc = fmaf(t.w, t.w, acc);
            }
            ens[r] = acc;
            if (real) {
                p.en[row] = acc;
                emax_b = __uint_as_float(max(__float_as_uint(emax_b), __float_as_uint(acc)));
            }
        }
        // data units: hi parts + residuals (the units holding an augmentation column are finished below)
        float dn = 0.f;
        for (int u = uu; u < p.units_p; u += 16) {
            const int k0 = u * 16;
            if (k0 + 16 <= K && !(p.dbg & 2)) {
                union { _Float16 h[16]; uint4 v[2]; } hi;
                const float *x = xs + r * LDS_LD + k0;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float xsj = x[e] * scale;
                    const _Float16 h = (_Float16)xsj;
                    const float d = xsj - (float)h;
                    dn = fmaf(d, d, dn);
                    hi.h[e] = h;
                }
                uint4 *o = p.out + (((row >> 5) * p.units_p + u) << 6) + (row & 31);
                o[0] = hi.v[0]; o[32] = hi.v[1];
            }
        }
        dnp[r * 17 + uu] = dn;
        __syncthreads();
        // units that straddle / follow K: data tail, the two augmentation columns (hi and lo of -||e||^2/2), zeros
        for (int u = (K >> 4) + uu; u < p.units_p; u += 16) {
            const int k0 = u * 16;
            union { _Float16 h[16]; uint4 v[2]; } hi;
            float dt = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = k0 + e;
                float a = 0.f;
                if (k < K) {
                    a = xs[r * LDS_LD + k] * scale;
                    const float d = a - (float)(_Float16)a;
                    dt = fmaf(d, d, dt);
                } else if (k == K || k == K + 1) {
                    if (real) {
                        const float full = ens[r] * -0.5f * scale;
                        const _Float16 fh = (_Float16)full;
                        a = k == K ? (float)fh : (float)(_Float16)(full - (float)fh);
                    } else {
                        a = -65504.f;                       // padding candidate: can never count
                    }
                }
                hi.h[e] = (_Float16)a;
            }
            if (u == (K >> 4)) dnp[r * 17 + 16] = dt;           // (the one unit that may hold a data tail; always present)
            uint4 *o = p.out + (((row >> 5) * p.units_p + u) << 6) + (row & 31);
            o[0] = hi.v[0]; o[32] = hi.v[1];
        }
        __syncthreads();
        if (tid < 16 && real) {
            float t = 0.f;
            for (int j = 0; j < 17; ++j) t += dnp[r * 17 + j];
            t *= inv2 * 1.0002f;                            // (fp32 summation error, any order: K * 2^-24 relative)
            dmax_b = fmaxf(dmax_b, t);
        }
        __syncthreads();
    }
    // one atomic per block for each maximum
    if (tid < 64) {
        unsigned m0 = __float_as_uint(emax_b), m1 = __float_as_uint(dmax_b);
        for (int off = 8; off > 0; off >>= 1) {
            m0 = max(m0, (unsigned)__shfl_xor((int)m0, off, 64));
            m1 = max(m1, (unsigned)__shfl_xor((int)m1, off, 64));
        }
        if (tid == 0) { bmax[0] = m0; bmax[1] = m1; }
    }
    __syncthreads();
    if (tid == 0) {
        if (p.block_max) {
            p.block_max[blockIdx.x] = __uint_as_float(bmax[0]);
            p.block_max[gridDim.x + blockIdx.x] = __uint_as_float(bmax[1]);
        } else {
            if (p.en_max) kge_atomic_max_u32(reinterpret_cast<unsigned *>(p.en_max), bmax[0]);
            if (p.dn2max) kge_atomic_max_u32(reinterpret_cast<unsigned *>(p.dn2max), bmax[1]);
        }
    }
}

// e2pref[u] = max over rows of the squared norm of the row's first (u+1)*16 data columns: with the
// same prefix norms of a query, || q[:k] || * sqrt(e2pref) bounds every partial sum the MFMA
// accumulator holds while it works through unit u (Cauchy-Schwarz on the prefix) -- the error band
// then charges each unit with ITS magnitude instead of the full ||q|| ||e|| (about half of it).
__global__ __launch_bounds__(1024) void prefix_max_kernel(const float *__restrict__ cell_ss, int64_t rows_p, int64_t rows,
                                                          int units_p, float *e2pref)
{
    // block maxima in LDS, ONE global atomic per (block, unit): same-line atomics serialise at ~12 ns each
    __shared__ unsigned smax[128];
    const int lane = threadIdx.x & 63;
    for (int u = threadIdx.x; u < units_p; u += blockDim.x) smax[u] = 0u;
    __syncthreads();
    for (int64_t r0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) - lane; r0 < rows;
         r0 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = r0 + lane;
        float prefix = 0.f;
        for (int u = 0; u < units_p; ++u) {
            if (r < rows) prefix += cell_ss[(int64_t)u * rows_p + r];
            unsigned m = __float_as_uint(prefix);       // sums of squares: >= 0, ordered like their bit patterns
            for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
            if (lane == 0) atomicMax(&smax[u], m);
        }
    }
    __syncthreads();
    for (int u = threadIdx.x; u < units_p; u += blockDim.x) kge_atomic_max_u32(reinterpret_cast<unsigned *>(e2pref) + u, smax[u]);
}

// One step of the per-query magnitude sum: prefix += cell sum;  amag += sqrt(prefix * e2pref[u])
__device__ __forceinline__ void split_amag_step(float &prefix, float &amag, float ss, float e2u)
{
    prefix = prefix + ss;
    amag = amag + sqrtf(prefix * e2u);
}

__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ x, int64_t n, float *max_io)
{
    // The maximum is taken on the BIT PATTERNS of |x| as unsigned integers: for non-negative floats that is the
    // float order, +inf sorts above every finite value and every NaN above +inf -- a NaN element therefore
    // reaches max_io as a NaN (fmaxf would skip it), and the threshold kernels that consume the scalar turn a
    // non-finite maximum into their overflow flag -> exact path (the SAD prefilter quantised a NaN
    // element to a finite value).  One atomic per block (thousands of same-address atomics serialise in the L2:
    // 96 us for 14 MB when every wave issued its own).
    __shared__ unsigned wmax[4];
    unsigned u = 0u;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const int64_t n4 = n >> 2;
        const uint4 *x4 = reinterpret_cast<const uint4 *>(x);
        for (int64_t i = tid; i < n4; i += nth) {
            const uint4 v = x4[i];
            u = max(max(u, v.x & 0x7fffffffu), max(v.y & 0x7fffffffu, max(v.z & 0x7fffffffu, v.w & 0x7fffffffu)));
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += nth) u = max(u, __float_as_uint(x[i]) & 0x7fffffffu);
    } else {
        for (int64_t i = tid; i < n; i += nth) u = max(u, __float_as_uint(x[i]) & 0x7fffffffu);
    }
    for (int off = 32; off > 0; off >>= 1) u = max(u, (unsigned)__shfl_xor((int)u, off, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = u;
    __syncthreads();
    if (threadIdx.x == 0) {
        u = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        kge_atomic_max_u32(reinterpret_cast<unsigned *>(max_io), u);
    }
}

// Self-test of the accumulation model behind c_acc = 1.25 (tools/probe/mfma_probe.hip is the long
// version): v_mfma_f32_32x32x16_f16 computes each output as two passes  acc <- acc + sum of 8 products,
// the 9 addends of a pass aligned to the largest exponent and truncated (toward zero) 24 bits below
// it, the sum exact, one round-to-nearest-even at the end; f16 subnormal inputs are kept.  Hence at
// most 9 * 2^-24 * |running magnitude| of error per 8 products.  The vectors below tell this model
// apart from sequential fp32 adds, from wider / narrower alignment and from other roundings.
__global__ void mfma_selftest_kernel(const float *__restrict__ ab, const float *__restrict__ c, float *out, int n)
{
    const int lane = threadIdx.x, half = lane >> 5;
    for (int t = 0; t < n; ++t) {
        f16x8 fa, fb;
        for (int j = 0; j < 8; ++j) {
            fa[j] = (_Float16)ab[t * 32 + half * 8 + j];
            fb[j] = (_Float16)ab[t * 32 + 16 + half * 8 + j];
        }
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = c[t];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
        if (lane == 0) out[t] = acc[0];
    }
}

struct SplitThrParams {
    int mode;                       // KGE_LP_L2_EXPAND or KGE_LP_DOT
    const float *qn0, *qn1;         // per-query squared norms (segment 1 optional)
    const float *s_true;
    const float *qmax0, *qmax1;     // device scalars (DOT: scale of the query operand)
    const float *emax0, *emax1;     // device scalars: max ||e||^2 per segment
    int64_t B, Bp;
    int K, units;
    float eps_scale;
    float2 *thr;
    int32_t *list_count;
    float *overflow;
    const float *pz;                // projection modes: (p_i, z_i) pairs, stride ldw
    int64_t ldw;
    const float *xabsmax, *yabsmax; // device scalars >= max |X|, max |y_c|
    float4 *thr4;
    float c_acc;                    // accumulation-error coefficient per product (2: any adder; 1.25: measured model)
    int K0;                         // columns of the first K-segment (K - K0 of the second)
    const float *q_cell_ss;         // optional [units_p][ss_ld] cell sums of the queries (kge_lp_split_rows) ...
    const int64_t *ss_index;        // ... read at column ss_index[i] instead of i (query columns); ss_ld = their row stride
    int64_t ss_ld;
    const float *e2pref;            // ... and prefix squared-norm maxima of the candidates (kge_lp_split_prefix_max)
    int units_p;
    int level;                      // 1: thresholds of the one-product sweep (q_dn2, de2max; L2_EXPAND and DOT modes)
    const float *q_dn2;             // ||q_i - hi(q_i)||^2, read at q_dn2_index[i] when given (query columns)
    const int64_t *q_dn2_index;
    const float *de2max;            // device scalar >= max_c ||e_c - hi(e_c)||^2
    const float *tp_bmax;           // optional [2][tp_blocks]: block maxima left by kge_lp_table_prep_l2 (emax0, de2max): folded
    int tp_blocks;                  // into the two scalars by every block on its way in, stored by block 0
    float *emax_out, *de2max_out;
    int32_t *zero_i32;              // optional: zero_n int32 zeroed by this launch (the region counters of the sweep's list)
    int zero_n;
    int q_scale_per_query;          // DOT, level 1: the query operand of row i is scaled by split_scale(||q_i||^2), its own norm
                                    // (kge_lp_dot_query_pipeline), not by the batch maximum's; qn0 then holds the total
};

// (a_lo, a_hi) of the plain L2 expansion, unscaled half-width logic shared by split_thr_kernel and the fused
// query pipeline (which must produce the same thresholds bit for bit)
// The count kernel reads "v >= a_lo" off the sign of v - a_lo, which is wrong only for v = -0, a_lo = +0:
// a zero threshold is moved down to the next normal number (widening the band is always safe).
__device__ __forceinline__ float split_nonzero_lo(float lo) { return lo == 0.f ? -1.17549435e-38f : lo; }

// amag: the sum over the k16 units of the bound on the accumulator's magnitude in that unit (split_amag_step,
// times 1.003 for the cross terms and f16 roundings), or < 0 when the prefix norms are not at hand: every unit is
// then charged with the full ||q|| ||e||.
__device__ __forceinline__ float split_acc_err(float amag, float aug_mag, float mag, int units, float c_acc,
                                               float adds_per_unit = 48.0f)
{
    const float two24 = 5.9604645e-8f;
    const float sum_mag = amag >= 0.f ? amag * 1.003f + aug_mag : (float)units * mag;
    return c_acc * adds_per_unit * two24 * sum_mag;  // 48 (one-product level: 16) additions per unit, each within c_acc * 2^-24 of the magnitude
}
// Rounding error of the exact fp32 chain the counts are defined by: one fmaf rounding per element, each within
// 2^-24 of the partial sum it produces (running error bound) -- 16 per unit against the same prefix magnitudes,
// or gamma_K * ||q|| ||e|| without them.
__device__ __forceinline__ float split_chain_err(float amag, float mag, int K)
{
    const float two24 = 5.9604645e-8f;
    return amag >= 0.f ? 16.16f * two24 * amag * 1.003f : 1.01f * (float)K * two24 * mag;
}

__device__ __forceinline__ float2 split_thr_l2(float q, float st, float em, int K, int units, float c_acc, float eps_scale,
                                               float amag)
{
    const float two22 = 2.3841858e-7f;
    const float eps_rel = 3.01f * two22;             // split residual
    const float enrm = sqrtf(em) * 1.000001f, qnrm = sqrtf(q) * 1.000001f;
    const float out_scale = (float)(1 << SPLIT_SCALE_LOG2) * (float)(1 << SPLIT_SCALE_LOG2);
    const float u = -st;                             // count c iff v_c <= u, v = ||q||^2 + ||e||^2 - 2 q.e
    const float mag = qnrm * enrm + 0.5f * em;       // >= sum of |products|
    const float eps_dot = split_acc_err(amag, 0.5f * em, mag, units, c_acc) + split_chain_err(amag, mag, K) + eps_rel * mag +
                          2.5e-7f * (qnrm + enrm) + 4e-9f;
    const float eps_v = (2.0f * eps_dot + 4.0f * two22 * (q + em + fabsf(u))) * eps_scale;
    const float mid = 0.5f * (q - u);
    const float hw = 0.5f * eps_v + two22 * (fabsf(q) + fabsf(u));
    return make_float2(split_nonzero_lo((mid - hw) * out_scale), (mid + hw) * out_scale);
}

// ONE-PRODUCT level (LV = 1 of the count kernel): acc = sum_k qh*eh (+ the two-term augmentation column), i.e. the
// split residual is no longer 3 * 2^-22 of the magnitude but the operands' own f16 rounding residuals,
//     q.e - qh.eh = dq.e + qh.de,    |.| <= ||dq|| ||e|| + ||qh|| ||de||,   ||qh|| <= ||q|| + ||dq||,
// with dq = q - hi(q) MEASURED per query (dq2 = ||dq||^2, exact differences summed in fp32) and de2m >= max_c ||de_c||^2:
// ~4.7e-4 of ||q|| max||e|| at K = 200 (a third of the candidates' values round up, a third down ...), 8 x the band of the
// three-product sweep -- which buys one MFMA per k16 unit instead of three and half the operand bytes.  The augmentation
// column -||e||^2/2 rides as TWO columns (hi, lo: residual 2^-22); the accumulation term has 16 additions per unit.
__device__ __forceinline__ float split_hi_resid(float qnrm, float enrm, float em_aug, float dq2, float de2m)
{
    const float two22 = 2.3841858e-7f;
    const float dqn = sqrtf(dq2) * 1.0001f, den = sqrtf(de2m) * 1.0001f;
    return (dqn * enrm + (qnrm + dqn) * den) * 1.0005f + 1.01f * two22 * em_aug;
}

__device__ __forceinline__ float2 split_thr_l2_hi(float q, float st, float em, int K, int units, float c_acc, float eps_scale,
                                                  float dq2, float de2m)
{
    const float two22 = 2.3841858e-7f;
    const float enrm = sqrtf(em) * 1.000001f, qnrm = sqrtf(q) * 1.000001f;
    const float out_scale = (float)(1 << SPLIT_SCALE_LOG2) * (float)(1 << SPLIT_SCALE_LOG2);
    const float u = -st;                             // count c iff v_c <= u, v = ||q||^2 + ||e||^2 - 2 q.e
    const float mag = qnrm * enrm + 0.5f * em;       // >= sum of |products|
    const float eps_dot = split_acc_err(-1.0f, 0.5f * em, mag, units, c_acc, 16.0f) + split_chain_err(-1.0f, mag, K) +
                          split_hi_resid(qnrm, enrm, 0.5f * em, dq2, de2m) + 2.5e-7f * (qnrm + enrm) + 4e-9f;
    const float eps_v = (2.0f * eps_dot + 4.0f * two22 * (q + em + fabsf(u))) * eps_scale;
    const float mid = 0.5f * (q - u);
    const float hw = 0.5f * eps_v + two22 * (fabsf(q) + fabsf(u));
    return make_float2(split_nonzero_lo((mid - hw) * out_scale), (mid + hw) * out_scale);
}

__global__ void split_thr_kernel(const SplitThrParams p)
{
    const float two24_c = 5.9604645e-8f;
    const float two22 = 2.3841858e-7f;
    float em, de2m = 0.f;
    if (p.tp_bmax) {        // (as query_pipeline_kernel: the table preparation's per-block maxima -> the two scalars)
        __shared__ unsigned red[8];
        unsigned m0 = 0u, m1 = 0u;
        for (int j = threadIdx.x; j < p.tp_blocks; j += blockDim.x) {
            m0 = max(m0, __float_as_uint(p.tp_bmax[j]));
            m1 = max(m1, __float_as_uint(p.tp_bmax[p.tp_blocks + j]));
        }
        for (int off = 32; off > 0; off >>= 1) {
            m0 = max(m0, (unsigned)__shfl_xor((int)m0, off, 64));
            m1 = max(m1, (unsigned)__shfl_xor((int)m1, off, 64));
        }
        const int wv_ = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { red[wv_] = m0; red[4 + wv_] = m1; }
        __syncthreads();
        m0 = max(max(red[0], red[1]), max(red[2], red[3]));
        m1 = max(max(red[4], red[5]), max(red[6], red[7]));
        em = __uint_as_float(max(m0, __float_as_uint(*p.emax0)));
        de2m = __uint_as_float(max(m1, p.de2max ? __float_as_uint(*p.de2max) : 0u));
        __syncthreads();
        if (blockIdx.x == 0 && threadIdx.x == 0) { *p.emax_out = em; if (p.de2max_out) *p.de2max_out = de2m; }
    } else {
        em = *p.emax0 + (p.emax1 ? *p.emax1 : 0.f);
        if (p.level == 1) de2m = *p.de2max;
    }
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < p.zero_n; j += gridDim.x * blockDim.x) p.zero_i32[j] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *p.list_count = 0;
        // non-finite norms (diverged embeddings): f16 operands would hold inf / NaN and the
        // comparisons would silently fail -- hand the count back to the exact fp32 path
        const float qm = p.qmax0 ? *p.qmax0 + (p.qmax1 ? *p.qmax1 : 0.f) : 0.f;
        if (!(em < INFINITY) || !(qm < INFINITY)) *p.overflow = 1.0f;
    }
    // accumulation: 48*units fp32 additions, c_acc = 2 (adders that truncate instead of rounding, any order)
    // or 1.25 when kge_mfma_f16_selftest confirmed the measured behaviour (<= 9/8 per product, see below);
    // exact chain: K fmaf roundings (gamma_K <= 1.01 K u); split residual 3 * 2^-22 * (1 + 2^-10)
    const float eps_rel = 3.01f * two22;             // split residual
    const float enrm = sqrtf(em) * 1.000001f;
    const bool have_pref = p.q_cell_ss && p.e2pref;
    // two K-segments whose first is not a multiple of 8 long: the exact chain's 8-blocks of the second segment
    // straddle the k16 cells, a partial sum may reach one cell ahead of its own -- one more full magnitude
    // covers the 16 roundings per unit that are then charged to the earlier (smaller) prefix
    const float straddle = (have_pref && p.K0 % 8 != 0 && p.K0 < p.K) ? 16.16f * two24_c : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.Bp; i += (int64_t)gridDim.x * blockDim.x) {
        if (i >= p.B) {
            if (p.mode >= KGE_LP_L2_PROJH) p.thr4[i] = make_float4(INFINITY, INFINITY, 0.f, 0.f);
            else p.thr[i] = make_float2(INFINITY, INFINITY);
            continue;
        }
        const float q = p.qn0[i] + (p.qn1 ? p.qn1[i] : 0.f);
        const float qnrm = sqrtf(q) * 1.000001f;
        float amag = -1.0f;
        if (have_pref) {
            float prefix = 0.f;
            amag = 0.f;
            for (int u = 0; u < p.units; ++u) split_amag_step(prefix, amag, p.q_cell_ss[(int64_t)u * p.ss_ld + (p.ss_index ? p.ss_index[i] : i)], p.e2pref[u]);
        }
        const float dq2 = p.level == 1 ? p.q_dn2[p.q_dn2_index ? p.q_dn2_index[i] : i] : 0.f;
        if (p.mode == KGE_LP_L2_EXPAND && p.level == 1) {
            p.thr[i] = split_thr_l2_hi(q, p.s_true[i], em, p.K, p.units, p.c_acc, p.eps_scale, dq2, de2m);
        } else if (p.mode == KGE_LP_L2_EXPAND) {
            p.thr[i] = split_thr_l2(q, p.s_true[i], em, p.K, p.units, p.c_acc, p.eps_scale, amag);
        } else if (p.mode >= KGE_LP_L2_PROJH) {
            const float out_scale = (float)(1 << SPLIT_SCALE_LOG2) * (float)(1 << SPLIT_SCALE_LOG2);
            const float u = -p.s_true[i];                    // count c iff v_c <= u, v = ||q||^2 + ||e||^2 - 2 q.e
            const float mag = qnrm * enrm + 0.5f * em;       // >= sum of |products|
            const float eps_dot = p.level == 1
                ? split_acc_err(-1.0f, 0.5f * em, mag, p.units, p.c_acc, 16.0f) + split_chain_err(-1.0f, mag, p.K) +
                  split_hi_resid(qnrm, enrm, 0.5f * em, dq2, de2m) + 2.5e-7f * (qnrm + enrm) + 4e-9f
                : split_acc_err(amag, 0.5f * em, mag, p.units, p.c_acc) + split_chain_err(amag, mag, p.K) +
                  eps_rel * mag + 2.5e-7f * (qnrm + enrm) + 4e-9f;
            const float eps_v = (2.0f * eps_dot + 4.0f * two22 * (q + em + fabsf(u))) * p.eps_scale;
            const float mid = 0.5f * (q - u);
            float hw = 0.5f * eps_v + two22 * (fabsf(q) + fabsf(u));
            if (p.mode >= KGE_LP_L2_PROJH) {
                // + the projection term corr = x (x z + p)  resp.  y (y z + 2 g + p): it is computed exactly in fp32
                // by both paths but enters in a different association -> a few ulps of its largest possible size
                const float pi = p.pz[i * p.ldw], zi = p.pz[i * p.ldw + 1];
                // |X[r_i, c]| = |w_i . e_c| <= ||w_i|| max||e||  (Cauchy-Schwarz; ||w_i||^2 = z_i + 2 for TransH, z_i for TransD) when
                // no measured maximum is given: the term below is 2^-22 of cmax, a looser bound costs nothing
                const float wn2 = p.mode == KGE_LP_L2_PROJH ? zi + 2.0f : zi;
                const float xm = p.xabsmax ? *p.xabsmax : sqrtf(fmaxf(wn2, 0.f) * em) * 1.000001f;
                const float ym = p.mode == KGE_LP_L2_PROJD ? *p.yabsmax : 0.f;
                const float cmax = p.mode == KGE_LP_L2_PROJH ? xm * (xm * fabsf(zi) + fabsf(pi))
                                                             : ym * (ym * fabsf(zi) + 2.0f * xm + fabsf(pi));
                hw += 8.0f * two22 * cmax * p.eps_scale + two22 * cmax;
                p.thr4[i] = make_float4(split_nonzero_lo((mid - hw) * out_scale), (mid + hw) * out_scale, pi, zi);
                continue;
            }
        } else {
            // count c iff dot_c >= s_true; both operands carry their own power-of-two scale
            const float qm = p.q_scale_per_query ? q : *p.qmax0 + (p.qmax1 ? *p.qmax1 : 0.f);
            const float out_scale = split_scale(qm) * split_scale(em);
            const float st = p.s_true[i];
            const float sqk = sqrtf((float)p.K);
            const float eps_abs = 1.4901161e-8f * sqk * (sqrtf(qm) * enrm + sqrtf(em) * qnrm) + 1e-30f;
            const float eps_dot = p.level == 1
                ? (split_acc_err(-1.0f, 0.f, qnrm * enrm, p.units, p.c_acc, 16.0f) + split_chain_err(-1.0f, qnrm * enrm, p.K) +
                   split_hi_resid(qnrm, enrm, 0.f, dq2, de2m) + eps_abs) * p.eps_scale
                : (split_acc_err(amag, 0.f, qnrm * enrm, p.units, p.c_acc) +
                   split_chain_err(amag, qnrm * enrm, p.K) + straddle * qnrm * enrm +
                   eps_rel * qnrm * enrm + eps_abs) * p.eps_scale;
            const float hw = eps_dot + two22 * fabsf(st);
            p.thr[i] = make_float2(split_nonzero_lo((st - hw) * out_scale), (st + hw) * out_scale);
        }
    }
}

// ---- fused query side of one TransE-L2 batch -----------------------------------
// One wavefront per 64 queries does what lp_prep + row_sqnorm + pair_scores + split_rows(Q) + split_thr
// do in five launches: q = e_src +- r (written for the later exact kernels), ||q||^2 and the exact true
// score by the SAME sequential chains (one lane per query, rows staged cooperatively through LDS), the
// two thresholds and the f16 split row.  Bit-identical outputs to the separate kernels.
struct QueryPipeParams {
    int tail;                       // 1: q = E[h] + R[r], true = t;  0: q = E[t] - R[r], true = h;
                                    // 2: both sides in one batch -- queries [0, Bh) tail side, [Bh, 2 Bh) head side
    int64_t Bh;                     // facts per side (tail == 2: B = 2 Bh)
    const float *E, *R;
    int d;
    const int64_t *h, *t, *r;
    int64_t B, Bp;
    const float *en;                // ||E[c]||^2
    const float *emax;              // device scalar max ||e||^2
    float *qmax_io;                 // device scalar, max ||q||^2 folded in (may be NULL)
    float c_acc, eps_scale;
    int units, units_p;
    float *Q, *qn, *s_true;
    float2 *thr;
    _Float16 *Qs;
    int32_t *list_count;
    const float *e2pref;            // optional: prefix squared-norm maxima of the entity table (tighter error band)
    const int32_t *qs_row;          // optional: row of Qs that receives query i's split cells (< 0: none -- a query whose
                                    // row another query of the same key already provides); NULL: row i
    int level;                      // 1: one-product level -- Qs is a PLANAR hi operand (units_p = kge_lp_hi_units), two
                                    // augmentation columns, thresholds from the measured residual ||q - hi(q)||
    const float *de2max;            // level 1: device scalar >= max_c ||e_c - hi(e_c)||^2
    float *q_dn2;                   // level 1, optional: ||q_i - hi(q_i)||^2 per query (for a later kge_lp_split_count
                                    // that recomputes the thresholds: thr_ready = 0)
    const float *tp_bmax;           // optional [2][tp_blocks]: the block maxima kge_lp_table_prep_l2 left instead of its atomics:
    int tp_blocks;                  // every block reduces them (emax, de2max), block 0 stores the two scalars
    float *emax_out, *de2max_out;
    int32_t *zero_i32;              // optional: zero_n int32 zeroed by this launch (the batch's rank counters)
    int64_t zero_n;
    int dbg;                        // env KGE_QP_DBG (timing probes, wrong results): 1 no chains, 2 no split cells, 4 no Q store,
                                    // 8 no row loads after the first chunk, 16 no block-maxima reduction, 32 no final atomic
};

template <int QPW>   // queries per wavefront: their chains run on lanes 0..QPW-1, loads / stores use all 64 lanes
__global__ __launch_bounds__(256) void query_pipeline_kernel(const QueryPipeParams p)
{
    // rows staged cooperatively 48 k at a time (row stride 52 floats: conflict-free b128), the two
    // sequential chains run one lane per query; few queries per wavefront = many wavefronts in flight
    // (the chains are latency bound)
    constexpr int KC = 48, LD = 52;
    // (4 independent wavefronts per block, each on its own LDS slice: they only share the final atomic)
    __shared__ __attribute__((aligned(16))) float qs_all[4 * QPW * LD];
    __shared__ __attribute__((aligned(16))) float ts_all[4 * QPW * LD];
    __shared__ unsigned wmax[4];
    __shared__ float dnp_all[4 * QPW * 8];          // level 1: residual sums per (row, 8-column group) of the current chunk
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float *qs = qs_all + wv * QPW * LD, *ts = ts_all + wv * QPW * LD, *dnp = dnp_all + wv * QPW * 8;
    const int d = p.d, kpad = p.units_p * 16;
    if (blockIdx.x == 0 && threadIdx.x == 0) *p.list_count = 0;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < p.zero_n; j += (int64_t)gridDim.x * 256) p.zero_i32[j] = 0;
    float em, de2m = 0.f;
    if (p.tp_bmax && !(p.dbg & 16)) {        // the table preparation's block maxima -> the two scalars (values >= 0: ordered like their bits)
        __shared__ unsigned red[8];
        unsigned m0 = 0u, m1 = 0u;
        for (int j = threadIdx.x; j < p.tp_blocks; j += 256) {
            m0 = max(m0, __float_as_uint(p.tp_bmax[j]));
            m1 = max(m1, __float_as_uint(p.tp_bmax[p.tp_blocks + j]));
        }
        for (int off = 32; off > 0; off >>= 1) {
            m0 = max(m0, (unsigned)__shfl_xor((int)m0, off, 64));
            m1 = max(m1, (unsigned)__shfl_xor((int)m1, off, 64));
        }
        if (lane == 0) { red[wv] = m0; red[4 + wv] = m1; }
        __syncthreads();
        m0 = max(max(red[0], red[1]), max(red[2], red[3]));
        m1 = max(max(red[4], red[5]), max(red[6], red[7]));
        // (folded into what the scalars already hold -- the guard vector is zeroed per evaluation, other shards may add)
        em = __uint_as_float(max(m0, __float_as_uint(*p.emax)));
        de2m = __uint_as_float(max(m1, p.de2max ? __float_as_uint(*p.de2max) : 0u));
        __syncthreads();
        if (blockIdx.x == 0 && threadIdx.x == 0) { *p.emax_out = em; if (p.de2max_out) *p.de2max_out = de2m; }
    } else {
        em = *p.emax;
        if (p.level == 1) de2m = *p.de2max;
    }
    float qbig = 0.f;
    const int64_t ngroups = (p.Bp + QPW - 1) / QPW;
    for (int64_t grp = (int64_t)blockIdx.x * 4 + wv; grp < ngroups; grp += (int64_t)gridDim.x * 4) {
        const int64_t i = grp * QPW + lane;
        const bool valid = lane < QPW && i < p.B;
        const int64_t ic = valid ? i : 0;
        const bool tl = p.tail == 2 ? ic < p.Bh : p.tail == 1;          // this query's side
        const int64_t fi = (p.tail == 2 && ic >= p.Bh) ? ic - p.Bh : ic; // its fact
        const int64_t src = tl ? p.h[fi] : p.t[fi], tru = tl ? p.t[fi] : p.h[fi], ri = p.r[fi];
        const int tli = tl ? 1 : 0;
        float qn = 0.f, acc = 0.f;
        float amag = 0.f;                                        // split_thr's magnitude sum
        float dn = 0.f;                                          // level 1: || (q - hi(q)) * 2^12 ||^2
        // Software-pipelined staging: the three row loads of the NEXT chunk are issued before this chunk's two
        // sequential chains run (they are the latency of this kernel: 48 dependent FMA pairs per chunk), so a
        // group of queries costs one load latency plus its chains instead of one load latency per chunk.
        constexpr int NP = KC / 4, ITS = (QPW * NP + 63) / 64;     // 16-byte pieces per full row chunk; passes per chunk
        float4 pe[ITS], pr[ITS], pt[ITS];
#define KGE_QP_FETCH(K0)                                                                                     \
    {                                                                                                        \
        const int pcs_ = max(0, min(KC, d - (K0))) >> 2;                                                     \
        _Pragma("unroll") for (int it = 0; it < ITS; ++it) {                                                 \
            const int idx = it * 64 + lane;                                                                  \
            const bool act = idx < QPW * pcs_;                                                               \
            const int rr = act ? idx / pcs_ : 0, pc = act ? idx - rr * pcs_ : 0;                             \
            const int64_t s_ = __shfl(src, rr, 64), r_ = __shfl(ri, rr, 64), t_ = __shfl(tru, rr, 64);       \
            if (act) {                                                                                       \
                pe[it] = *reinterpret_cast<const float4 *>(p.E + s_ * d + (K0) + pc * 4);                    \
                pr[it] = *reinterpret_cast<const float4 *>(p.R + r_ * d + (K0) + pc * 4);                    \
                pt[it] = *reinterpret_cast<const float4 *>(p.E + t_ * d + (K0) + pc * 4);                    \
            }                                                                                                \
        }                                                                                                    \
    }
        KGE_QP_FETCH(0)
        for (int k0 = 0; k0 < kpad; k0 += KC) {
            const int kc = max(0, min(KC, d - k0));              // data columns of this chunk
            const int pieces = kc >> 2;
#pragma unroll
            for (int it = 0; it < ITS; ++it) {                   // uniform trip count (shuffles inside)
                const int idx = it * 64 + lane;
                const bool act = idx < QPW * pieces;
                const int rr = act ? idx / pieces : 0, pc = act ? idx - rr * pieces : 0;
                const bool tl_ = __shfl(tli, rr, 64) != 0;
                if (!act) continue;
                const float4 e4 = pe[it], r4 = pr[it], t4 = pt[it];
                float4 q4;                                       // lp_prep_kernel, translation.py:105-125
                q4.x = tl_ ? e4.x + r4.x : e4.x - r4.x;
                q4.y = tl_ ? e4.y + r4.y : e4.y - r4.y;
                q4.z = tl_ ? e4.z + r4.z : e4.z - r4.z;
                q4.w = tl_ ? e4.w + r4.w : e4.w - r4.w;
                const int64_t row = grp * QPW + rr;
                if (row < p.B && !(p.dbg & 4)) *reinterpret_cast<float4 *>(p.Q + row * d + k0 + pc * 4) = q4;
                *reinterpret_cast<float4 *>(qs + rr * LD + pc * 4) = q4;
                *reinterpret_cast<float4 *>(ts + rr * LD + pc * 4) = t4;
            }
            if (k0 + KC < kpad && !(p.dbg & 8)) KGE_QP_FETCH(k0 + KC)
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (kc > 0 && lane < QPW && !(p.dbg & 1)) {
                const float *x = qs + lane * LD;
                // row_sqnorm_kernel's chain; its value at the end of every k16 cell is the prefix squared norm
                // of the magnitude sum (split_thr_kernel adds up cell sums instead: equal up to rounding, and
                // the band carries a 1.003 factor)
                // Whole k16 cells: the cell's 16 query and 16 true-entity values come in with 8 b128 LDS reads, then the
                // two dependent chains run side by side (||q||^2 in ascending k; the true score in the tile kernel's
                // order, 8-blocks ascending and k = 0,4,1,5,2,6,3,7 inside) -- one LDS round trip per cell instead of
                // one per element / per 8-block and chain.  Same operations in the same order: same bits.
                const float *tt = ts + lane * LD;
                int k = 0;
                for (; k + 16 <= kc; k += 16) {
                    float xv[16], tv[16];
#pragma unroll
                    for (int j4 = 0; j4 < 4; ++j4) {
                        const float4 v = *reinterpret_cast<const float4 *>(x + k + 4 * j4);
                        const float4 w = *reinterpret_cast<const float4 *>(tt + k + 4 * j4);
                        xv[4 * j4] = v.x; xv[4 * j4 + 1] = v.y; xv[4 * j4 + 2] = v.z; xv[4 * j4 + 3] = v.w;
                        tv[4 * j4] = w.x; tv[4 * j4 + 1] = w.y; tv[4 * j4 + 2] = w.z; tv[4 * j4 + 3] = w.w;
                    }
#pragma unroll
                    for (int j = 0; j < 16; ++j) qn = fmaf(xv[j], xv[j], qn);
#pragma unroll
                    for (int b8 = 0; b8 < 16; b8 += 8) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc = fmaf(xv[b8 + j], tv[b8 + j], acc);
                            acc = fmaf(xv[b8 + 4 + j], tv[b8 + 4 + j], acc);
                        }
                    }
                    if (p.e2pref) amag = amag + sqrtf(qn * p.e2pref[(k0 + k) >> 4]);
                }
                const int ktail = k;
                for (; k < kc; ++k) {
                    qn = fmaf(x[k], x[k], qn);
                    if (p.e2pref && (((k0 + k) & 15) == 15 || k0 + k == d - 1))
                        amag = amag + sqrtf(qn * p.e2pref[(k0 + k) >> 4]);
                }
                if (ktail < kc) acc = lp_chain_dot(x + ktail, tt + ktail, kc - ktail, acc);   // the pair kernel's chain
            }
            // split cells of this chunk: 8 consecutive k of one row per lane and pass
            const int ngr = (p.dbg & 2) ? 0 : min(KC, kpad - k0) >> 3;
            for (int idx = lane; idx < QPW * ngr; idx += 64) {
                const int rr = idx / ngr, gq = idx - rr * ngr;
                const int64_t row = grp * QPW + rr;
                union { _Float16 h[8]; uint4 v; } hi, lo;
                float dsum = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = k0 + gq * 8 + e;
                    float xv = 0.f;
                    if (row < p.B) xv = k < d ? qs[rr * LD + gq * 8 + e] : ((k == d || (p.level == 1 && k == d + 1)) ? 1.0f : 0.f);
                    xv *= (float)(1 << SPLIT_SCALE_LOG2);
                    const _Float16 hh = (_Float16)xv;
                    hi.h[e] = hh;
                    const float dd = xv - (float)hh;             // exact in fp32 (0 in the augmentation / padding columns)
                    lo.h[e] = (_Float16)dd;
                    dsum = fmaf(dd, dd, dsum);
                }
                // level 1: the residual ||q - hi(q)||^2 is a BOUND of the error band (any summation order, 1.0001 for it):
                // summed here on all 64 lanes -- on the chain lanes its 5 operations per element were 70 % of their work
                if (p.level == 1) dnp[rr * 8 + gq] = dsum;
                const int kk = k0 + gq * 8, u = kk >> 4, hf = (kk >> 3) & 1;
                const int64_t dst = p.qs_row ? (row < p.B ? (int64_t)p.qs_row[row] : -1) : row;
                if (dst >= 0 && p.level == 1) {      // planar hi operand: 32 bytes per unit
                    uint4 *cell = reinterpret_cast<uint4 *>(p.Qs) + (dst * p.units_p + u) * 2;
                    cell[hf] = hi.v;
                } else if (dst >= 0) {
                    uint4 *cell = reinterpret_cast<uint4 *>(p.Qs) + (dst * p.units_p + u) * 4;
                    cell[hf] = hi.v;
                    cell[2 + hf] = lo.v;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (p.level == 1 && lane < QPW)
                for (int gq = 0; gq < ngr; ++gq) dn += dnp[lane * 8 + gq];
        }
#undef KGE_QP_FETCH
        if (lane < QPW && i < p.Bp) {
            if (valid) {
                const float st = lp_epilogue(KGE_LP_L2_EXPAND, acc, qn, p.en[tru]);
                p.qn[i] = qn;
                p.s_true[i] = st;
                if (p.e2pref) {     // units past the data (the augmentation column alone in its unit)
                    for (int u = (d + 15) >> 4; u < p.units; ++u) amag = amag + sqrtf(qn * p.e2pref[u]);
                } else {
                    amag = -1.0f;
                }
                if (p.level == 1) {
                    const float inv2 = 1.0f / ((float)(1 << SPLIT_SCALE_LOG2) * (float)(1 << SPLIT_SCALE_LOG2));
                    const float dq2 = dn * inv2 * 1.0001f;
                    if (p.q_dn2) p.q_dn2[i] = dq2;
                    p.thr[i] = split_thr_l2_hi(qn, st, em, d, p.units, p.c_acc, p.eps_scale, dq2, de2m);
                } else {
                    p.thr[i] = split_thr_l2(qn, st, em, d, p.units, p.c_acc, p.eps_scale, amag);
                }
                qbig = __uint_as_float(max(__float_as_uint(qbig), __float_as_uint(qn)));
            } else {
                p.thr[i] = make_float2(INFINITY, INFINITY);
            }
        }
    }
    if (p.qmax_io && !(p.dbg & 32)) {    // one atomic per block (same-address atomics serialise at ~12 ns each)
        unsigned m = __float_as_uint(qbig);
        for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
        if (lane == 0) wmax[wv] = m;
        __syncthreads();
        if (threadIdx.x == 0)
            kge_atomic_max_u32(reinterpret_cast<unsigned *>(p.qmax_io), max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])));
    }
}

// ---- fused query side of one DistMult / ComplEx batch on the one-product level (r05) ------------------------------
// What lp_prep + pair_scores + two any-order norm passes + their sum + kge_lp_hi_rows(is_query) + the threshold kernel
// (+ two fills) do in nine launches.  The DOT modes' operands carry a power-of-two scale taken from a squared-norm
// MAXIMUM (arbitrary magnitudes, unlike the unit-ball rows of the L2 modes) -- batch-wide in the separate kernels, which
// is what kept this side at two sweeps over Q with a device-wide reduction between them.  Here every query row carries
// ITS OWN scale S_i = split_scale(||q_i||^2): the count kernels never see a scale (the thresholds of query i are
// multiplied by S_i * S_e like its accumulators), the band only gets tighter (its absolute term then holds ||q_i|| where
// it held max ||q||), and the guard column of the padding candidates is built from ||q_i|| alone.
// Per group of QPW queries of a wavefront: (1) ||q||^2 in any order (a bound: 16 lanes per row, the source rows read
// once -- they are read again, from the L1 / L2, by) (2) the TransE pipeline's chunk loop: q = e (x) r written for the
// later exact kernels and staged in LDS with the true entity's rows, the exact true score by the pair kernel's chain
// (one lane per query; segment [Re | Im] after segment), the planar f16 hi operand and its measured residual on all lanes.
struct DotPipeParams {
    int tail;                       // as QueryPipeParams
    int64_t Bh;
    const float *E0, *E1, *R0, *R1; // entity / relation tables (ComplEx: Re, Im; DistMult: E1 = R1 = NULL)
    int d;                          // columns per segment (K = d resp. 2 d), d % 8 == 0
    const int64_t *h, *t, *r;
    int64_t B, Bp;
    const float *emax0, *emax1;     // device scalars: max ||row||^2 of the candidate table's segments
    const float *de2max;            // device scalar >= max_c ||e_c - hi(e_c)||^2
    float *qmax_io;                 // device scalar, max ||q||^2 folded in (may be NULL)
    float c_acc, eps_scale;
    int units, units_p;
    float *Q0, *Q1, *qn, *s_true, *q_dn2;
    float2 *thr;
    _Float16 *Qh;
    int32_t *list_count;
    float *overflow;
    int32_t *zero_i32;
    int64_t zero_n;
    const float *dn_bmax;           // optional [dn_blocks]: block maxima of the candidate table's residuals (kge_lp_dot_table_prep):
    int dn_blocks;                  // folded into *de2max by every block on its way in, stored by block 0
    const float *nm_bmax;           // optional [2][nm_blocks]: squared-norm maxima per block of kge_lp_dot_table_prep_fused -- folded
    int nm_blocks;                  // into *emax0 / *emax1 the same way
    float *prev_nmax;               // optional [2]: the maxima the NEXT one-pass table preparation takes its scale from (stored by
                                    // block 0); with nm_bmax: the ones THIS table was scaled by -- another scale: *overflow = 1
};

// thresholds of one DOT query on the one-product level, operand scales s_q (its own) and s_e
__device__ __forceinline__ float2 split_thr_dot_hi(float q, float st, float em, int K, int units, float c_acc, float eps_scale,
                                                   float dq2, float de2m, float qm, float s_q, float s_e)
{
    const float two22 = 2.3841858e-7f;
    const float enrm = sqrtf(em) * 1.000001f, qnrm = sqrtf(q) * 1.000001f;
    const float out_scale = s_q * s_e;
    const float sqk = sqrtf((float)K);
    const float eps_abs = 1.4901161e-8f * sqk * (sqrtf(qm) * enrm + sqrtf(em) * qnrm) + 1e-30f;
    const float eps_dot = (split_acc_err(-1.0f, 0.f, qnrm * enrm, units, c_acc, 16.0f) + split_chain_err(-1.0f, qnrm * enrm, K) +
                           split_hi_resid(qnrm, enrm, 0.f, dq2, de2m) + eps_abs) * eps_scale;
    const float hw = eps_dot + two22 * fabsf(st);
    return make_float2(split_nonzero_lo((st - hw) * out_scale), (st + hw) * out_scale);
}

template <int QPW, bool CPLX>
__global__ __launch_bounds__(256) void dot_query_pipeline_kernel(const DotPipeParams p)
{
    constexpr int KC = 48, LD = 52;
    __shared__ __attribute__((aligned(16))) float qs_all[4 * QPW * LD];
    __shared__ __attribute__((aligned(16))) float ts_all[4 * QPW * LD];
    __shared__ float dnp_all[4 * QPW * 8];
    __shared__ float qn_all[4 * QPW], sc_all[4 * QPW];
    __shared__ unsigned wmax[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float *qs = qs_all + wv * QPW * LD, *ts = ts_all + wv * QPW * LD, *dnp = dnp_all + wv * QPW * 8;
    float *qn_s = qn_all + wv * QPW, *sc_s = sc_all + wv * QPW;
    const int d = p.d, nseg = CPLX ? 2 : 1, K = nseg * d, kpad = p.units_p * 16;
    if (blockIdx.x == 0 && threadIdx.x == 0) *p.list_count = 0;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < p.zero_n; j += (int64_t)gridDim.x * 256) p.zero_i32[j] = 0;
    float em;
    if (p.nm_bmax) {        // (as query_pipeline_kernel: block maxima -> the scalars, folded into what they hold)
        __shared__ unsigned nred[8];
        unsigned m0 = 0u, m1 = 0u;
        for (int j = threadIdx.x; j < p.nm_blocks; j += 256) {
            m0 = max(m0, __float_as_uint(p.nm_bmax[j]));
            m1 = max(m1, __float_as_uint(p.nm_bmax[p.nm_blocks + j]));
        }
        for (int off = 32; off > 0; off >>= 1) {
            m0 = max(m0, (unsigned)__shfl_xor((int)m0, off, 64));
            m1 = max(m1, (unsigned)__shfl_xor((int)m1, off, 64));
        }
        if (lane == 0) { nred[wv] = m0; nred[4 + wv] = m1; }
        __syncthreads();
        m0 = max(max(nred[0], nred[1]), max(nred[2], nred[3]));
        m1 = max(max(nred[4], nred[5]), max(nred[6], nred[7]));
        const float n0 = __uint_as_float(max(m0, __float_as_uint(*p.emax0)));
        const float n1 = p.emax1 ? __uint_as_float(max(m1, __float_as_uint(*p.emax1))) : 0.f;
        em = n0 + n1;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            *const_cast<float *>(p.emax0) = n0;
            if (p.emax1) *const_cast<float *>(p.emax1) = n1;
            if (p.prev_nmax) {
                // the table was converted under split_scale(prev): thresholds and table agree only under the same scale
                // (2: not the list -- the caller runs the same path again, now under the maxima stored below)
                if (split_scale(p.prev_nmax[0] + p.prev_nmax[1]) != split_scale(em)) *p.overflow = 2.0f;
                p.prev_nmax[0] = n0; p.prev_nmax[1] = n1;
            }
        }
    } else {
        em = *p.emax0 + (p.emax1 ? *p.emax1 : 0.f);
        if (p.prev_nmax && blockIdx.x == 0 && threadIdx.x == 0) {
            p.prev_nmax[0] = *p.emax0;
            p.prev_nmax[1] = p.emax1 ? *p.emax1 : 0.f;
        }
    }
    float de2m;
    if (p.dn_bmax) {
        __shared__ unsigned red[4];
        unsigned m = 0u;
        for (int j = threadIdx.x; j < p.dn_blocks; j += 256) m = max(m, __float_as_uint(p.dn_bmax[j]));
        for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
        if (lane == 0) red[wv] = m;
        __syncthreads();
        m = max(max(red[0], red[1]), max(red[2], red[3]));
        de2m = __uint_as_float(max(m, __float_as_uint(*p.de2max)));
        __syncthreads();
        if (blockIdx.x == 0 && threadIdx.x == 0) *const_cast<float *>(p.de2max) = de2m;
    } else {
        de2m = *p.de2max;
    }
    const float s_e = split_scale(em);
    if (blockIdx.x == 0 && threadIdx.x == 0 && !(em < INFINITY)) *p.overflow = 1.0f;
    float qbig = 0.f;
    const int nch_seg = (d + KC - 1) / KC, nch = nseg * nch_seg;
    const int64_t ngroups = (p.Bp + QPW - 1) / QPW;
    for (int64_t grp = (int64_t)blockIdx.x * 4 + wv; grp < ngroups; grp += (int64_t)gridDim.x * 4) {
        const int64_t i = grp * QPW + lane;
        const bool valid = lane < QPW && i < p.B;
        const int64_t ic = valid ? i : 0;
        const bool tl = p.tail == 2 ? ic < p.Bh : p.tail == 1;
        const int64_t fi = (p.tail == 2 && ic >= p.Bh) ? ic - p.Bh : ic;
        const int64_t src = tl ? p.h[fi] : p.t[fi], tru = tl ? p.t[fi] : p.h[fi], ri = p.r[fi];
        const int tli = tl ? 1 : 0;
        // ---- (1) ||q||^2, any order: 16 lanes per row, four rows of the group at a time
        {
            const int sub = lane & 15;
#pragma unroll 1
            for (int rb = 0; rb < QPW; rb += 4) {
                const int rr = rb + (lane >> 4);
                const int64_t s_ = __shfl(src, rr, 64), r_ = __shfl(ri, rr, 64);
                const bool tl_ = __shfl(tli, rr, 64) != 0;
                float ss = 0.f;
                for (int k = sub * 4; k < d; k += 64) {
                    if (CPLX) {
                        const float4 re = *reinterpret_cast<const float4 *>(p.E0 + s_ * d + k);
                        const float4 im = *reinterpret_cast<const float4 *>(p.E1 + s_ * d + k);
                        const float4 rr4 = *reinterpret_cast<const float4 *>(p.R0 + r_ * d + k);
                        const float4 ir4 = *reinterpret_cast<const float4 *>(p.R1 + r_ * d + k);
#define KGE_DP_SS(C)                                                                                         \
    {                                                                                                        \
        const float q0_ = tl_ ? re.C * rr4.C - im.C * ir4.C : rr4.C * re.C + ir4.C * im.C;                   \
        const float q1_ = tl_ ? re.C * ir4.C + im.C * rr4.C : rr4.C * im.C - ir4.C * re.C;                   \
        ss = fmaf(q0_, q0_, ss);                                                                             \
        ss = fmaf(q1_, q1_, ss);                                                                             \
    }
                        KGE_DP_SS(x) KGE_DP_SS(y) KGE_DP_SS(z) KGE_DP_SS(w)
#undef KGE_DP_SS
                    } else {
                        const float4 e4 = *reinterpret_cast<const float4 *>(p.E0 + s_ * d + k);
                        const float4 r4 = *reinterpret_cast<const float4 *>(p.R0 + r_ * d + k);
                        const float q0 = e4.x * r4.x, q1 = e4.y * r4.y, q2 = e4.z * r4.z, q3 = e4.w * r4.w;
                        ss = fmaf(q0, q0, ss); ss = fmaf(q1, q1, ss); ss = fmaf(q2, q2, ss); ss = fmaf(q3, q3, ss);
                    }
                }
                ss += __shfl_xor(ss, 8, 64); ss += __shfl_xor(ss, 4, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 1, 64);
                if (sub == 0) qn_s[rr] = ss;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        const float qn = lane < QPW ? qn_s[lane] : 0.f;
        const float s_q = split_scale(qn);
        if (lane < QPW) sc_s[lane] = s_q;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        // ---- (2) the chunk loop
        float acc = 0.f, dn = 0.f;
        constexpr int NP = KC / 4, ITS = (QPW * NP + 63) / 64;
        float4 pa[ITS], pb[ITS], pc_[ITS], pd[ITS], pt[ITS];
#define KGE_DP_FETCH(C)                                                                                      \
    {                                                                                                        \
        const int sg_ = (C) / nch_seg, kk0_ = ((C) - sg_ * nch_seg) * KC;                                    \
        const int pcs_ = max(0, min(KC, d - kk0_)) >> 2;                                                     \
        const float *tt_ = (CPLX && sg_ != 0) ? p.E1 : p.E0;                                                 \
        _Pragma("unroll") for (int it = 0; it < ITS; ++it) {                                                 \
            const int idx = it * 64 + lane;                                                                  \
            const bool act = idx < QPW * pcs_;                                                               \
            const int rr = act ? idx / pcs_ : 0, pc = act ? idx - rr * pcs_ : 0;                             \
            const int64_t s_ = __shfl(src, rr, 64), r_ = __shfl(ri, rr, 64), t_ = __shfl(tru, rr, 64);       \
            if (act) {                                                                                       \
                const int64_t ko_ = kk0_ + pc * 4;                                                           \
                pa[it] = *reinterpret_cast<const float4 *>(p.E0 + s_ * d + ko_);                             \
                pc_[it] = *reinterpret_cast<const float4 *>(p.R0 + r_ * d + ko_);                            \
                if (CPLX) {                                                                                  \
                    pb[it] = *reinterpret_cast<const float4 *>(p.E1 + s_ * d + ko_);                         \
                    pd[it] = *reinterpret_cast<const float4 *>(p.R1 + r_ * d + ko_);                         \
                }                                                                                            \
                pt[it] = *reinterpret_cast<const float4 *>(tt_ + t_ * d + ko_);                              \
            }                                                                                                \
        }                                                                                                    \
    }
        KGE_DP_FETCH(0)
#pragma unroll 1
        for (int c = 0; c < nch; ++c) {
            const int sg = c / nch_seg, k0 = (c - sg * nch_seg) * KC;
            const int kc = min(KC, d - k0), pieces = kc >> 2;
            float *Qg = sg == 0 ? p.Q0 : p.Q1;
#pragma unroll
            for (int it = 0; it < ITS; ++it) {
                const int idx = it * 64 + lane;
                const bool act = idx < QPW * pieces;
                const int rr = act ? idx / pieces : 0, pc = act ? idx - rr * pieces : 0;
                const bool tl_ = __shfl(tli, rr, 64) != 0;
                if (!act) continue;
                const float4 t4 = pt[it];
                float4 q4;
                if (CPLX) {      // lp_prep_kernel, bilinear.py:514-515 (tail) / :521-522 (head): the same operations
                    const float4 re = pa[it], im = pb[it], rr4 = pc_[it], ir4 = pd[it];
                    if (sg == 0) {
                        q4.x = tl_ ? re.x * rr4.x - im.x * ir4.x : rr4.x * re.x + ir4.x * im.x;
                        q4.y = tl_ ? re.y * rr4.y - im.y * ir4.y : rr4.y * re.y + ir4.y * im.y;
                        q4.z = tl_ ? re.z * rr4.z - im.z * ir4.z : rr4.z * re.z + ir4.z * im.z;
                        q4.w = tl_ ? re.w * rr4.w - im.w * ir4.w : rr4.w * re.w + ir4.w * im.w;
                    } else {
                        q4.x = tl_ ? re.x * ir4.x + im.x * rr4.x : rr4.x * im.x - ir4.x * re.x;
                        q4.y = tl_ ? re.y * ir4.y + im.y * rr4.y : rr4.y * im.y - ir4.y * re.y;
                        q4.z = tl_ ? re.z * ir4.z + im.z * rr4.z : rr4.z * im.z - ir4.z * re.z;
                        q4.w = tl_ ? re.w * ir4.w + im.w * rr4.w : rr4.w * im.w - ir4.w * re.w;
                    }
                } else {         // bilinear.py:247-267
                    const float4 e4 = pa[it], r4 = pc_[it];
                    q4.x = e4.x * r4.x; q4.y = e4.y * r4.y; q4.z = e4.z * r4.z; q4.w = e4.w * r4.w;
                }
                const int64_t row = grp * QPW + rr;
                if (row < p.B) *reinterpret_cast<float4 *>(Qg + row * d + k0 + pc * 4) = q4;
                else q4 = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4 *>(qs + rr * LD + pc * 4) = q4;
                *reinterpret_cast<float4 *>(ts + rr * LD + pc * 4) = t4;
            }
            if (c + 1 < nch) KGE_DP_FETCH(c + 1)
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (lane < QPW) {       // the exact true score: the pair kernel's chain (lp_chain_dot), continued over the segments
                const float *x = qs + lane * LD, *tt = ts + lane * LD;
                int k = 0;
                for (; k + 16 <= kc; k += 16) {
                    float xv[16], tv[16];
#pragma unroll
                    for (int j4 = 0; j4 < 4; ++j4) {
                        const float4 v = *reinterpret_cast<const float4 *>(x + k + 4 * j4);
                        const float4 w = *reinterpret_cast<const float4 *>(tt + k + 4 * j4);
                        xv[4 * j4] = v.x; xv[4 * j4 + 1] = v.y; xv[4 * j4 + 2] = v.z; xv[4 * j4 + 3] = v.w;
                        tv[4 * j4] = w.x; tv[4 * j4 + 1] = w.y; tv[4 * j4 + 2] = w.z; tv[4 * j4 + 3] = w.w;
                    }
#pragma unroll
                    for (int b8 = 0; b8 < 16; b8 += 8) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc = fmaf(xv[b8 + j], tv[b8 + j], acc);
                            acc = fmaf(xv[b8 + 4 + j], tv[b8 + 4 + j], acc);
                        }
                    }
                }
                if (k < kc) acc = lp_chain_dot(x + k, tt + k, kc - k, acc);     // (kc % 8 == 0: one more 8-block)
            }
            // the chunk's hi cells: 8 consecutive k of one row per lane and pass, the row's own scale
            const int ngr = kc >> 3;
            for (int idx = lane; idx < QPW * ngr; idx += 64) {
                const int rr = idx / ngr, gq = idx - rr * ngr;
                const int64_t row = grp * QPW + rr;
                const float sc = sc_s[rr];
                const float4 v0 = *reinterpret_cast<const float4 *>(qs + rr * LD + gq * 8);
                const float4 v1 = *reinterpret_cast<const float4 *>(qs + rr * LD + gq * 8 + 4);
                union { _Float16 h[8]; uint4 v; } hi;
                float dsum = 0.f;
#define KGE_DP_CV(E, X)                                                                                      \
    {                                                                                                        \
        const float xsj = (X) * sc;                                                                          \
        const _Float16 hh = (_Float16)xsj;                                                                   \
        const float dd = xsj - (float)hh;                                                                    \
        dsum = fmaf(dd, dd, dsum);                                                                           \
        hi.h[E] = hh;                                                                                        \
    }
                KGE_DP_CV(0, v0.x) KGE_DP_CV(1, v0.y) KGE_DP_CV(2, v0.z) KGE_DP_CV(3, v0.w)
                KGE_DP_CV(4, v1.x) KGE_DP_CV(5, v1.y) KGE_DP_CV(6, v1.z) KGE_DP_CV(7, v1.w)
#undef KGE_DP_CV
                dnp[rr * 8 + gq] = dsum;
                const int kk = sg * d + k0 + gq * 8, u = kk >> 4, hf = (kk >> 3) & 1;
                if (row < p.Bp) reinterpret_cast<uint4 *>(p.Qh)[(row * p.units_p + u) * 2 + hf] = hi.v;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (lane < QPW)
                for (int gq = 0; gq < ngr; ++gq) dn += dnp[lane * 8 + gq];
        }
#undef KGE_DP_FETCH
        // the cells behind the data: the guard column at K (see hi_rows_kernel, aug_mode 3 -- with this row's own norm), zeros
        const int ntail = (kpad - K) >> 3;
        for (int idx = lane; idx < QPW * ntail; idx += 64) {
            const int rr = idx / ntail, g = idx - rr * ntail;
            const int64_t row = grp * QPW + rr;
            union { _Float16 h[8]; uint4 v; } hi;
#pragma unroll
            for (int e = 0; e < 8; ++e) hi.h[e] = (_Float16)0.f;
            if (g == 0 && row < p.B) {
                const float qr = sqrtf(qn_s[rr]);
                hi.h[0] = (_Float16)fmaxf(0.25f * (qr + qr * 0.00390625f) * sc_s[rr], 1.0f);
            }
            const int kk = K + g * 8, u = kk >> 4, hf = (kk >> 3) & 1;
            if (row < p.Bp) reinterpret_cast<uint4 *>(p.Qh)[(row * p.units_p + u) * 2 + hf] = hi.v;
        }
        if (lane < QPW && i < p.Bp) {
            if (valid) {
                p.qn[i] = qn;
                p.s_true[i] = acc;
                const float dq2 = dn * (1.0f / (s_q * s_q)) * 1.0001f;
                if (p.q_dn2) p.q_dn2[i] = dq2;
                p.thr[i] = split_thr_dot_hi(qn, acc, em, K, p.units, p.c_acc, p.eps_scale, dq2, de2m, qn, s_q, s_e);
                if (!(qn < INFINITY)) *p.overflow = 1.0f;
                qbig = __uint_as_float(max(__float_as_uint(qbig), __float_as_uint(qn)));
            } else {
                p.thr[i] = make_float2(INFINITY, INFINITY);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");     // (qn_s / sc_s are rewritten by the next group)
    }
    if (p.qmax_io) {
        unsigned m = __float_as_uint(qbig);
        for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
        if (lane == 0) wmax[wv] = m;
        __syncthreads();
        if (threadIdx.x == 0)
            kge_atomic_max_u32(reinterpret_cast<unsigned *>(p.qmax_io), max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])));
    }
}

// ---- the count kernel --------------------------------------------------------
// GS > 0: GROUPED columns -- every column (one split query row) carries up to GS queries that share the row but
// have their own true entity, hence their own thresholds: the MFMA sweep runs once per column, the epilogue once
// per (column, set); thresholds and counters of the panel's sets live in LDS.
// LV = 1: the ONE-PRODUCT level.  Operands are PLANAR hi tables (kge_lp_hi_rows: 32 bytes per k16 unit), a 128-byte stage
// row holds FOUR units, and a stage is four MFMA groups qh*eh instead of 2 x 3 -- a third of the matrix work and half
// the operand bytes per (pair, unit); same tile, same LDS-DMA staging and swizzle, same epilogue against thresholds
// that carry the operands' measured f16 residuals (split_thr_l2_hi).  p.units / p.stages then count hi units / 4-unit stages.
template <int NWAVES, bool DBG, int PM, int GS = 0, int LV = 0>   // PM: 0 plain thresholds, 1 TransH projection term, 2 TransD
__global__ __launch_bounds__(64 * NWAVES, 1) void lp_split_count_kernel(const SplitParams p)
{
    static_assert(GS == 0 || GS == GSETS, "grouped columns carry GSETS threshold sets");
    const int dbg = DBG ? p.dbg : 0;                                // probes compile away in the product kernel
    constexpr int NTHREADS = 64 * NWAVES;
    constexpr int MT = TC / 32 / (NWAVES / 2);                      // 32x32 candidate tiles per wave
    constexpr int EJ = TC * 8 / NTHREADS, QJ = TQ * 8 / NTHREADS;   // staged 16-byte chunks per thread
    constexpr int SROWS = NTHREADS / 8;                             // rows covered by one staging pass
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *unc_cnt = reinterpret_cast<int *>(smem + 2 * STAGE_BYTES);
    unsigned *unc_list = reinterpret_cast<unsigned *>(smem + 2 * STAGE_BYTES + 16);
    float4 *pthr = reinterpret_cast<float4 *>(smem + 2 * STAGE_BYTES + 16 + UNC_CAP * 4);   // PM: per query of the panel
    int *prow = reinterpret_cast<int *>(pthr + TQ);
    float2 *gthr = reinterpret_cast<float2 *>(prow + TQ);                  // GS: [set][TQ] thresholds ...
    int *gcnt = reinterpret_cast<int *>(gthr + (GS ? GS : 1) * TQ);         // ... [set][TQ] counters ...
    int *gsets = gcnt + (GS ? GS : 1) * TQ;                                 // ... and the number of sets in use
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: the LDS-DMA targets become SALU arithmetic)
    const int wr = wid >> 1, wc = wid & 1, l31 = lane & 31, half = lane >> 5;   // waves: (NWAVES/2) x 2

    // Work order.  The (query panel, candidate tile) items are listed with QG (= p.qg: 4 or 16) query panels
    // interleaved under a sweep of the candidate tiles:  (g*4+0, ct) (g*4+1, ct) .. (g*4+3, ct)
    // (g*4+0, ct+1) ...;  XCD x (the blocks with bid % 8 == x, one per CU) owns an eighth of the
    // list and its blocks take the positions  start + loc, start + loc + nbx, ...  So at any moment
    // the ~32 CUs that share an L2 work on 4 query panels x 8 candidate tiles: every split row
    // that enters the L2 is used by 8 (queries) or 4 (candidates) CUs before it is evicted, and a
    // block stays on one query panel for a whole sweep (its rank counters live in registers).
    const int QG = p.qg;
    const int nb = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3;
    const int nbx = (nb - xcd + 7) >> 3;                                   // blocks on this XCD
    const int nx = nb < 8 ? nb : 8;                                        // XCD slots in use
    const int64_t x_begin = p.n_items * xcd / nx, x_end = p.n_items * (xcd + 1) / nx;
    const int64_t item_begin = x_begin + loc;
    const int nitems = item_begin < x_end ? (int)((x_end - item_begin + nbx - 1) / nbx) : 0;
    if (nitems <= 0) return;
    const int per_group = QG * p.c_tiles, n_groups = (p.q_panels + QG - 1) / QG;
    auto item_qp_ct = [&](int i, int &qp, int &ct) __attribute__((always_inline)) {   // i-th item of this block
        const int idx = (int)item_begin + i * nbx;
        const int grp = min(idx / per_group, n_groups - 1);
        const int r = idx - grp * per_group, gsz = min(QG, p.q_panels - grp * QG);
        ct = r / gsz;
        qp = grp * QG + (r - ct * gsz);
    };
    const int S = p.stages, G = nitems * S;
    if (tid == 0) { *unc_cnt = 0; if (GS) *gsets = 0; }
    if (GS) __syncthreads();

    // staging: 8 lanes cover one 128-byte row segment of a stage
    const int srow = tid >> 3, scs = tid & 7;
    const int sch = scs ^ ((srow >> 1) & 7);        // global chunk that lands in LDS chunk slot scs
    const int64_t rstep = (int64_t)SROWS * p.row_bytes;
    static_assert(EJ == 4 && QJ == 3, "the stage body below places 4 + 3 LDS-DMA pieces per wave");
    int pf_it = 0, pf_s = 0;
    const char *pfE = nullptr, *pfQ = nullptr;
    auto pf_new_item = [&]() __attribute__((always_inline)) {
        int qp, ct;
        item_qp_ct(pf_it, qp, ct);
        if (dbg & 128) qp = ct = 0;     // every block streams tile (0,0): cache-ceiling probe
        const int64_t q0 = (int64_t)qp * TQ, c0 = (int64_t)ct * TC;
        pfE = p.Es + (c0 + srow) * p.row_bytes + sch * 16;
        pfQ = p.Qs + (q0 + srow) * p.row_bytes + sch * 16;
    };
    // global -> LDS directly (LDS-DMA, 1 KiB per wave-instruction: the 64 lanes' 16-byte pieces land
    // at consecutive LDS addresses, which is exactly this wave's 8 rows x 128 bytes of the stage)
    auto dma = [&](const char *g, char *l) __attribute__((always_inline)) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                         (__attribute__((address_space(3))) void *)l, 16, 0, 0);
    };

    // fragment addressing: row r of a tile, chunk (u*4 + piece*2 + half) ^ ((r>>1)&7)
    const int sw = (l31 >> 1) & 7;
    const int a_row = (wr * (MT * 32) + l31) * 128;
    const int b_row = E_STAGE_BYTES + (wc * 96 + l31) * 128;
    // byte addresses inside a stage of this lane's fragments: [k16 unit][hi / lo]
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    unsigned a_off[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) a_off[u][pc] = lds0 + a_row + ((u * 4 + pc * 2 + half) ^ sw) * 16;
    unsigned h_off[4];      // LV = 1: the hi fragment of unit j of a stage is chunk 2 j + half
#pragma unroll
    for (int j = 0; j < 4; ++j) h_off[j] = lds0 + a_row + ((j * 2 + half) ^ sw) * 16;
    // the query fragments sit a wave-uniform distance behind the candidate fragments (scalar register)
    const unsigned b_delta = __builtin_amdgcn_readfirstlane(b_row - a_row);

    f32x16 acc[MT][NT];
    int cnt[NT] = {0, 0, 0};
    float alo[NT] = {0.f, 0.f, 0.f}, ahi[NT] = {0.f, 0.f, 0.f};
    auto load_panel = [&](int64_t q0) __attribute__((always_inline)) {
        if (GS) {   // thresholds of every (column, set) of the panel + zeroed counters; sets in use (block-wide max)
            int used = 0;
            for (int idx = tid; idx < TQ * GS; idx += NTHREADS) {
                const int c = idx / GS, gs = idx - c * GS;
                const int q = p.members[(q0 + c) * GS + gs];
                float2 t = make_float2(INFINITY, INFINITY);
                if (q >= 0) {
                    if (PM) { const float4 t4 = p.thr4[q]; t = make_float2(t4.x, t4.y); }
                    else t = p.thr[q];
                }
                gthr[gs * TQ + c] = t;
                gcnt[gs * TQ + c] = 0;
                if (q >= 0) used = max(used, gs + 1);
                if (PM && gs == 0) {    // the queries of a column share the key, hence relation and projection scalars
                    pthr[c] = q >= 0 ? p.thr4[q] : make_float4(INFINITY, INFINITY, 0.f, 0.f);
                    prow[c] = (int)p.r_idx[max(q, 0)];
                }
            }
            if (used > 0) atomicMax(gsets, used);
            return;
        }
        if (PM) {   // thresholds + projection scalars + X row of the panel's queries live in LDS
            if (tid < TQ) {
                const int64_t q = p.col_q ? (int64_t)p.col_q[q0 + tid] : q0 + tid;
                pthr[tid] = q >= 0 ? p.thr4[q] : make_float4(INFINITY, INFINITY, 0.f, 0.f);
                prow[tid] = (int)p.r_idx[min(max(q, (int64_t)0), p.B - 1)];
            }
            return;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int64_t col = q0 + wc * 96 + nt * 32 + l31;
            const int64_t q = p.col_q ? (int64_t)p.col_q[col] : col;
            const float2 t = q >= 0 ? p.thr[q] : make_float2(INFINITY, INFINITY);
            alo[nt] = t.x;
            ahi[nt] = t.y;
        }
    };
    auto flush_counts = [&](int64_t q0) __attribute__((always_inline)) {
        if (GS) {   // (same idx -> thread mapping as load_panel: a thread flushes and re-zeroes its own entries)
            for (int idx = tid; idx < TQ * GS; idx += NTHREADS) {
                const int c = idx / GS, gs = idx - c * GS;
                const int q = p.members[(q0 + c) * GS + gs];
                const int v = gcnt[gs * TQ + c];
                if (q >= 0 && v != 0) atomicAdd(&p.raw_count[q], v);
            }
            __syncthreads();
            if (tid == 0) *gsets = 0;
            __syncthreads();
            return;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int v = cnt[nt] + __shfl_xor(cnt[nt], 32, 64);
            const int64_t col = q0 + wc * 96 + nt * 32 + l31;
            const int64_t q = p.col_q ? (int64_t)p.col_q[col] : col;
            if (half == 0 && v != 0 && q >= 0 && q < p.B) atomicAdd(&p.raw_count[q], v);
            cnt[nt] = 0;
        }
    };

    int64_t cur_q0;
    {
        int qp, ct;
        item_qp_ct(0, qp, ct);
        cur_q0 = (int64_t)qp * TQ;
    }
    load_panel(cur_q0);
    pf_new_item();
    {   // stage 0 of the first tile
        char *nE = smem + wid * 1024, *nQ = nE + E_STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < EJ; ++j) dma(pfE + j * rstep, nE + j * SROWS * 128);
#pragma unroll
        for (int j = 0; j < QJ; ++j) dma(pfQ + j * rstep, nQ + j * SROWS * 128);
        if (++pf_s == S) {
            pf_s = 0;
            if (++pf_it < nitems) pf_new_item();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    // Fragment loads are inline asm with hand-placed waits: hipcc tracks the LDS-DMA instructions as
    // FLAT accesses, after which every lgkmcnt wait it inserts is a full drain -- it then waits for the
    // fragment loads it has just issued (before MFMAs that do not use them) instead of letting them
    // fly under the next MFMA group.  KGE_SWAIT ties the registers to the wait, so no use can move above it.
    static_assert(MT == 2 && NT == 3, "KGE_SLOAD / KGE_SWAIT are written out for 2 x 3 tiles per wave");
#define KGE_DSR(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(DST) : "v"(ADDR) : "memory")
#define KGE_SLOAD(AH, AL, BH, BL, BASE, U)                                                          \
    {                                                                                               \
        const unsigned ah_ = (BASE) + a_off[U][0], al_ = (BASE) + a_off[U][1];                      \
        const unsigned bh_ = ah_ + b_delta, bl_ = al_ + b_delta;                                    \
        KGE_DSR(AH[0], ah_, 0); KGE_DSR(BH[0], bh_, 0); KGE_DSR(AH[1], ah_, 4096);                  \
        KGE_DSR(BH[1], bh_, 4096); KGE_DSR(BH[2], bh_, 8192);                                       \
        KGE_DSR(BL[0], bl_, 0); KGE_DSR(BL[1], bl_, 4096); KGE_DSR(BL[2], bl_, 8192);               \
        KGE_DSR(AL[0], al_, 0); KGE_DSR(AL[1], al_, 4096);                                          \
        /* the address registers stay live past the last load: no destination may be allocated on them */ \
        asm volatile("" :: "v"(ah_), "v"(al_), "v"(bh_), "v"(bl_));                                 \
    }
#define KGE_SWAIT(AH, AL, BH, BL)                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(AH[0]), "+v"(AH[1]), "+v"(AL[0]), "+v"(AL[1]),      \
                 "+v"(BH[0]), "+v"(BH[1]), "+v"(BH[2]), "+v"(BL[0]), "+v"(BL[1]), "+v"(BL[2]) :: "memory");
#define KGE_HLOAD(AH, BH, BASE, J) /* LV = 1: the hi fragments of unit J of the stage at BASE */     \
    {                                                                                               \
        const unsigned ah_ = (BASE) + h_off[J];                                                     \
        const unsigned bh_ = ah_ + b_delta;                                                         \
        KGE_DSR(AH[0], ah_, 0); KGE_DSR(BH[0], bh_, 0); KGE_DSR(AH[1], ah_, 4096);                  \
        KGE_DSR(BH[1], bh_, 4096); KGE_DSR(BH[2], bh_, 8192);                                       \
        asm volatile("" :: "v"(ah_), "v"(bh_));                                                     \
    }
#define KGE_HWAIT(AH, BH)                                                                           \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(AH[0]), "+v"(AH[1]), "+v"(BH[0]), "+v"(BH[1]), "+v"(BH[2]) :: "memory");
#define KGE_SMMA_P(A, B, C) /* one of the three split products over the wave's MT x NT tiles, given C */ \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                               \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                           \
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[mt], B[nt], C, 0, 0, 0);
#define KGE_SMMA_PA(A, B)                                                                           \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                               \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                           \
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[mt], B[nt], acc[mt][nt], 0, 0, 0);

    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    f16x8 zero8;
#pragma unroll
    for (int r = 0; r < 8; ++r) zero8[r] = (_Float16)0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = zero16;

    f16x8 ah0[MT], al0[MT], bh0[NT], bl0[NT], ah1[MT], al1[MT], bh1[NT], bl1[NT];
    if (LV == 1) { KGE_HLOAD(ah0, bh0, 0u, 0) } else { KGE_SLOAD(ah0, al0, bh0, bl0, 0u, 0) }
    int it = 0, s = 0;
    // probe 4096 (DBG kernel, LV = 1): cycle stamps of the phases of the first KGE_TL_STAGES stages of block 0, waves 0 and 4
    // (partners on one SIMD), written to the head of the pair list (combine with 64: no list flush) -- tools/split_timeline.py
    constexpr int KGE_TL_STAGES = 48;
    unsigned long long tsv[16];
#define KGE_TS(I) if (DBG && tl) tsv[I] = __builtin_readcyclecounter();
    for (int g = 0; g < G; ++g) {
        const bool tl = DBG && LV == 1 && (dbg & 4096) && bid == 0 && (wid == 0 || wid == 4) && g < KGE_TL_STAGES;
        const int buf = g & 1;
        const bool more = g + 1 < G;
        const unsigned sb = buf * STAGE_BYTES, sb_next = (buf ^ 1) * STAGE_BYTES;
        const int nunits = min(2, p.units - 2 * s);
        const bool two = nunits == 2;              // the last stage of a tile may hold a single k16 unit
        // In the product kernel the LDS-DMA pieces are issued unconditionally (after the block's last
        // stage they re-fetch valid rows into the buffer nobody reads): a branch around them makes
        // hipcc drain lgkmcnt to 0 at the join, i.e. wait for the fragment loads it just issued.
        const bool pf = DBG ? (more && !(dbg & 1)) : true;
        // LDS-DMA of the next stage into the other buffer, one piece at a time between the MFMA
        // groups: an LDS-DMA instruction holds the issuing wave for 60+ cycles, which hides behind
        // matrix work only if the pieces are spread over the stage (and the other wave of the SIMD
        // is in its MFMAs)
        char *nE = smem + (buf ^ 1) * STAGE_BYTES + wid * 1024, *nQ = nE + E_STAGE_BYTES;
        const char *gE = pfE + pf_s * 128, *gQ = pfQ + pf_s * 128;

        // (grouped columns: on a tile's last stage the next stage's first fragments are fetched AFTER the multi-pass
        // epilogue -- their registers are what its temporaries need; the LDS buffer stays valid through the next stage)
        const bool defer_frag = GS != 0 && s == S - 1;
        if constexpr (LV == 1) {
            // ONE product per k16 unit, four units per stage: [u0] dma E0 E1 | frag u1 [u1] dma E2 E3 | frag u2 [u2] dma Q |
            // frag u3 -- barrier -- frag u0 of the next stage [u3].  Every fragment overwrite has an LDS wait (or an MFMA
            // group) between it and the MFMAs that last read those registers.
            const int nu = min(4, p.units - 4 * s);
            const bool frag = !DBG || !(dbg & 16);      // (probe: no fragment loads after the first)
            const char *gE1 = gE + rstep;
            KGE_TS(0)
            KGE_HWAIT(ah0, bh0)
            KGE_TS(1)
            if (s == 0) { KGE_SMMA_P(ah0, bh0, zero16) } else { KGE_SMMA_PA(ah0, bh0) }
            __builtin_amdgcn_sched_barrier(0);
            KGE_TS(2)
            if (pf) { dma(gE, nE); dma(gE1, nE + SROWS * 128); }
            __builtin_amdgcn_sched_barrier(0);
            KGE_TS(3)
            if (frag) { KGE_HLOAD(ah1, bh1, sb, 1) }
            asm volatile("" :: "v"(gE), "v"(gE1));
            __builtin_amdgcn_sched_barrier(0);
            KGE_TS(4)
            KGE_HWAIT(ah1, bh1)
            KGE_TS(5)
            if (nu > 1) { KGE_SMMA_PA(ah1, bh1) }
            __builtin_amdgcn_sched_barrier(0);
            KGE_TS(6)
            if (pf) { dma(gE + 2 * rstep, nE + 2 * SROWS * 128); dma(gE + 3 * rstep, nE + 3 * SROWS * 128); }
            __builtin_amdgcn_sched_barrier(0);
            KGE_TS(7)
            if (frag) { KGE_HLOAD(ah0, bh0, sb, 2) }
            __builtin_amdgcn_sched_barrier(0);
            KGE_HWAIT(ah0, bh0)
            KGE_TS(8)
            if (nu > 2) { KGE_SMMA_PA(ah0, bh0) }
            __builtin_amdgcn_sched_barrier(0);
            KGE_TS(9)
            if (pf) { dma(gQ, nQ); dma(gQ + rstep, nQ + SROWS * 128); dma(gQ + 2 * rstep, nQ + 2 * SROWS * 128); }
            if (more && pf) {
                if (++pf_s == S) {
                    pf_s = 0;
                    if (++pf_it < nitems) pf_new_item();
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            KGE_TS(10)
            if (frag) { KGE_HLOAD(ah1, bh1, sb, 3) }
            __builtin_amdgcn_sched_barrier(0);
            KGE_HWAIT(ah1, bh1)
            KGE_TS(11)
            if (!DBG || !(dbg & 512)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of the next stage landed in LDS
            KGE_TS(12)
            if (!DBG || !(dbg & 32)) __syncthreads();
            KGE_TS(13)
            if (more && !defer_frag && frag) { KGE_HLOAD(ah0, bh0, sb_next, 0) }
            __builtin_amdgcn_sched_barrier(0);
            if (nu > 3) { KGE_SMMA_PA(ah1, bh1) }
            __builtin_amdgcn_sched_barrier(0);
            KGE_TS(14)
        } else {
            // (this stage's first k16 fragments were fetched behind the previous stage's barrier, below)
            KGE_SWAIT(ah0, al0, bh0, bl0)
            if (s == 0) { KGE_SMMA_P(ah0, bh0, zero16) } else { KGE_SMMA_PA(ah0, bh0) }
            __builtin_amdgcn_sched_barrier(0);
            const char *gE1 = gE + rstep;
            if (pf) { dma(gE, nE); dma(gE1, nE + SROWS * 128); }
            __builtin_amdgcn_sched_barrier(0);
            if (!(dbg & 16) || g == 0) { KGE_SLOAD(ah1, al1, bh1, bl1, sb, 1) }
            // (hipcc would drain lgkmcnt before a fragment load whose destination reuses the address
            // registers of an LDS-DMA still in flight: keep those registers occupied until here)
            asm volatile("" :: "v"(gE), "v"(gE1));
            __builtin_amdgcn_sched_barrier(0);
            const bool hi1 = DBG && (dbg & 1024), halfdma = DBG && (dbg & 2048);
            if (!hi1) { KGE_SMMA_PA(ah0, bl0) }
            __builtin_amdgcn_sched_barrier(0);
            if (pf && !halfdma) { dma(gE + 2 * rstep, nE + 2 * SROWS * 128); dma(gE + 3 * rstep, nE + 3 * SROWS * 128); }
            __builtin_amdgcn_sched_barrier(0);
            if (!hi1) { KGE_SMMA_PA(al0, bh0) }
            __builtin_amdgcn_sched_barrier(0);
            if (pf) { dma(gQ, nQ); if (!halfdma) { dma(gQ + rstep, nQ + SROWS * 128); dma(gQ + 2 * rstep, nQ + 2 * SROWS * 128); } }
            if (more && pf) {
                if (++pf_s == S) {
                    pf_s = 0;
                    if (++pf_it < nitems) pf_new_item();
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            KGE_SWAIT(ah1, al1, bh1, bl1)
            if (two) { KGE_SMMA_PA(ah1, bh1) }
            __builtin_amdgcn_sched_barrier(0);
            // The stage's last two MFMA groups run BEHIND the barrier, next to the fetch of the next
            // stage's first fragments: right after a barrier all 8 waves read LDS at once (80 KB), and
            // without matrix work in flight the MFMA pipe would idle for that long.
            if (!(dbg & 512)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the next stage landed in LDS
            if (dbg & 512) __builtin_amdgcn_s_barrier();   // probe: barrier without waiting for the DMA pieces
            else if (!(dbg & 32)) __syncthreads();
            if (more && !(dbg & 16) && !defer_frag) { KGE_SLOAD(ah0, al0, bh0, bl0, sb_next, 0) }
            __builtin_amdgcn_sched_barrier(0);
            if (two && !hi1) { KGE_SMMA_PA(ah1, bl1) }
            __builtin_amdgcn_sched_barrier(0);
            if (two && !hi1) { KGE_SMMA_PA(al1, bh1) }
            __builtin_amdgcn_sched_barrier(0);

        }

        const bool tile_done = s == S - 1;
        if (tile_done && !(dbg & 4)) {
            int qp_cur, ct;
            item_qp_ct(it, qp_cur, ct);
            const int64_t c0 = (int64_t)ct * TC;
            // opaque to the optimiser: otherwise the ~100 list-entry constants below are
            // hoisted out of the tile loop and held in registers across the MFMA stream
            int cl_base = wr * (MT * 32) + 4 * half, ql_base = wc * 96 + l31;
            asm volatile("" : "+v"(cl_base), "+v"(ql_base));
            if (GS) {
                // Grouped columns: the accumulators of a column are compared with the thresholds of each of its
                // queries in turn (block-uniform number of sets; the accumulators stay intact, w lives in temporaries).
                const int nsets = *gsets;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int ql = ql_base + nt * 32;
                    if (PM) {   // acc <- acc - 2^23 * corr once per element: the corrected value serves every set
                        const float4 t4 = pthr[ql];
                        const float p_n = t4.z, z_n = t4.w;
                        const float *xrow = p.X + (int64_t)prow[ql] * p.ldx + c0 + wr * (MT * 32) + 4 * half;
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            float4 x4[4], y4[4];
#pragma unroll
                            for (int g4 = 0; g4 < 4; ++g4) {
                                x4[g4] = *reinterpret_cast<const float4 *>(xrow + mt * 32 + 8 * g4);
                                if (PM == 2)
                                    y4[g4] = *reinterpret_cast<const float4 *>(p.yc + c0 + wr * (MT * 32) + 4 * half + mt * 32 + 8 * g4);
                            }
#pragma unroll
                            for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const float xe = e == 0 ? x4[g4].x : (e == 1 ? x4[g4].y : (e == 2 ? x4[g4].z : x4[g4].w));
                                    float corr;
                                    if (PM == 1) {
                                        corr = xe * fmaf(xe, z_n, p_n);
                                    } else {
                                        const float ye = e == 0 ? y4[g4].x : (e == 1 ? y4[g4].y : (e == 2 ? y4[g4].z : y4[g4].w));
                                        corr = ye * fmaf(ye, z_n, fmaf(2.0f, xe, p_n));
                                    }
                                    acc[mt][nt][g4 * 4 + e] = fmaf(corr, -8388608.0f, acc[mt][nt][g4 * 4 + e]);
                                }
                        }
                    }
                    for (int gs = 0; gs < nsets; ++gs) {
                        const float2 th = gthr[gs * TQ + ql];
                        const f32x2 nlo2 = {-th.x, -th.x};
                        const float hwf = th.y - th.x;
                        const unsigned hwb = hwf >= 0.f ? __float_as_uint(hwf) : 0u;
                        unsigned smask = 0u;
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                            for (int g4 = 0; g4 < 4; ++g4) {
                                const f32x2 w01 = (f32x2){acc[mt][nt][g4 * 4 + 0], acc[mt][nt][g4 * 4 + 1]} + nlo2;
                                const f32x2 w23 = (f32x2){acc[mt][nt][g4 * 4 + 2], acc[mt][nt][g4 * 4 + 3]} + nlo2;
                                const unsigned b0 = __float_as_uint(w01.x), b1 = __float_as_uint(w01.y);
                                const unsigned b2 = __float_as_uint(w23.x), b3 = __float_as_uint(w23.y);
                                smask = __builtin_amdgcn_alignbit(smask, b0, 31);
                                smask = __builtin_amdgcn_alignbit(smask, b1, 31);
                                smask = __builtin_amdgcn_alignbit(smask, b2, 31);
                                smask = __builtin_amdgcn_alignbit(smask, b3, 31);
                                const unsigned mq = min(min(min(b0, b1), b2), b3);
                                if (__ballot(mq <= hwb)) {
#define KGE_GLIST(BITS, E)                                                                              \
    if ((BITS) <= hwb) {                                                                                \
        const int idx = atomicAdd(unc_cnt, 1);                                                          \
        if (idx < UNC_CAP)                                                                              \
            unc_list[idx] = ((unsigned)(cl_base + mt * 32 + (E) + 8 * g4) << 10) | ((unsigned)gs << 8) | (unsigned)ql; \
    }
                                    KGE_GLIST(b0, 0) KGE_GLIST(b1, 1) KGE_GLIST(b2, 2) KGE_GLIST(b3, 3)
#undef KGE_GLIST
                                }
                                __builtin_amdgcn_sched_barrier(0);   // one quad at a time: its temporaries die here
                            }
                        }
                        // the two lane halves hold the two row halves of the same column
                        int v = 32 - __popc(smask);
                        v += __shfl_xor(v, 32, 64);
                        if (half == 0 && v != 0) atomicAdd(&gcnt[gs * TQ + ql], v);
                    }
                }
            }
#pragma unroll
            for (int nt = 0; nt < (GS ? 0 : NT); ++nt) {
                float lo_n = alo[nt], hi_n = ahi[nt], p_n = 0.f, z_n = 0.f;
                const float *xrow = nullptr;
                if (PM) {
                    const int ql = wc * 96 + nt * 32 + l31;
                    const float4 t4 = pthr[ql];
                    lo_n = t4.x; hi_n = t4.y; p_n = t4.z; z_n = t4.w;
                    xrow = p.X + (int64_t)prow[ql] * p.ldx + c0 + wr * (MT * 32) + 4 * half;
                }
                asm volatile("" : "+v"(lo_n), "+v"(hi_n));   // (keeps the derived constants out of the MFMA loop's registers)
                const f32x2 nlo2 = {-lo_n, -lo_n};
                // band width as bits; padding queries carry a_lo = a_hi = +inf (inf - inf = NaN, of either sign):
                // width 0 there, so that none of their pairs can be listed
                const float hwf = hi_n - lo_n;
                const unsigned hwb = hwf >= 0.f ? __float_as_uint(hwf) : 0u;
                unsigned smask = 0u;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    float4 x4[4], y4[4];
                    if (PM) {   // the 4 quads' gathers X[r_i, c..c+3] (and y_c) issued together
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            x4[g4] = *reinterpret_cast<const float4 *>(xrow + mt * 32 + 8 * g4);
                            if (PM == 2)
                                y4[g4] = *reinterpret_cast<const float4 *>(p.yc + c0 + wr * (MT * 32) + 4 * half + mt * 32 + 8 * g4);
                        }
                    }
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {   // 4 accumulator registers = rows 8*g4 + 4*half + {0..3}
                        float vq[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            vq[e] = acc[mt][nt][g4 * 4 + e];
                            if (PM) {   // acc - 2^23 * corr: the comparison of v = qn + en - 2 dot + corr against u
                                const float xe = e == 0 ? x4[g4].x : (e == 1 ? x4[g4].y : (e == 2 ? x4[g4].z : x4[g4].w));
                                float corr;
                                if (PM == 1) {
                                    corr = xe * fmaf(xe, z_n, p_n);
                                } else {
                                    const float ye = e == 0 ? y4[g4].x : (e == 1 ? y4[g4].y : (e == 2 ? y4[g4].z : y4[g4].w));
                                    corr = ye * fmaf(ye, z_n, fmaf(2.0f, xe, p_n));
                                }
                                vq[e] = fmaf(corr, -8388608.0f, vq[e]);
                            }
                        }
                        // 2.25 VALU per element instead of 3 VALU + 2 SALU (compare / add-with-carry /
                        // compare / mask logic): w = v - a_lo as packed f32 adds; the SIGN bits of the 32
                        // w's of this lane's query are shifted into one register (v >= a_lo  <=>  sign clear:
                        // a float subtraction never gets the sign wrong, and a_lo is never +-0) and counted
                        // with one popcount per 32 elements; a pair is uncertain (a_lo <= v < a_hi) only if
                        // 0 <= w <= fl(a_hi - a_lo) -- rounding is monotonic -- i.e. iff the bits of w,
                        // read as unsigned, are <= those of the band width: one unsigned min3 + min per quad.
                        const f32x2 w01 = (f32x2){vq[0], vq[1]} + nlo2, w23 = (f32x2){vq[2], vq[3]} + nlo2;
                        const unsigned b0 = __float_as_uint(w01.x), b1 = __float_as_uint(w01.y);
                        const unsigned b2 = __float_as_uint(w23.x), b3 = __float_as_uint(w23.y);
                        smask = __builtin_amdgcn_alignbit(smask, b0, 31);
                        smask = __builtin_amdgcn_alignbit(smask, b1, 31);
                        smask = __builtin_amdgcn_alignbit(smask, b2, 31);
                        smask = __builtin_amdgcn_alignbit(smask, b3, 31);
                        const unsigned mq = min(min(min(b0, b1), b2), b3);
                        const unsigned long long any = __ballot(mq <= hwb);
                        // (w replaces v in the accumulator registers -- they are dead after the epilogue, the next
                        // tile starts from C = 0 -- so the epilogue needs no registers of its own)
                        acc[mt][nt][g4 * 4 + 0] = w01.x; acc[mt][nt][g4 * 4 + 1] = w01.y;
                        acc[mt][nt][g4 * 4 + 2] = w23.x; acc[mt][nt][g4 * 4 + 3] = w23.y;
                        if (any) { // some lane holds an uncertain pair among these 4 rows: list them
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                // 0 <= w <= fl(a_hi - a_lo): every pair of the band, and only counted pairs
                                if (__float_as_uint(acc[mt][nt][g4 * 4 + e]) <= hwb) {
                                    const int cl = cl_base + mt * 32 + e + 8 * g4;
                                    const int idx = atomicAdd(unc_cnt, 1);
                                    if (idx < UNC_CAP)
                                        unc_list[idx] = ((unsigned)cl << 8) | (unsigned)(ql_base + nt * 32);
                                }
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);   // one quad at a time: its temporaries die here
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                cnt[nt] += 32 - __popc(smask);      // the 32 elements (2 tiles x 16) of this lane's query
            }
            __syncthreads();
            if (wid == 0 && !(dbg & 64)) { // hand this tile's uncertain pairs to the global list
                const int n = *unc_cnt;
                const int nc = min(n, UNC_CAP);
                if (n > 0) {
                    int base = 0;
                    // (a degenerate all-tied model can push the counter past 2^31: positions are compared as
                    // unsigned, so nothing is ever written outside the list, and the sticky overflow flag makes
                    // the caller redo the count on the exact kernel)
                    if (lane == 0) base = atomicAdd(p.list_count, nc);
                    base = __shfl(base, 0, 64);
                    if (n > UNC_CAP && lane == 0) *p.overflow = 1.0f;
                    for (int i = lane; i < nc; i += 64) {
                        const unsigned e = unc_list[i];
                        const int pos = base + i;
                        if ((unsigned)pos < (unsigned)p.cap) {
                            // (grouped: [cl:8][set:2][column:8]; columns with one query: [cl][column:8], via col_q if given)
                            int32_t qid;
                            if (GS) qid = p.members[(cur_q0 + (e & 255u)) * GS + ((e >> 8) & 3u)];
                            else qid = p.col_q ? p.col_q[cur_q0 + (e & 255u)] : (int32_t)(cur_q0 + (e & 255u));
                            // (one 8-byte store per pair: consecutive lanes fill consecutive entries)
                            reinterpret_cast<int2 *>(p.list)[pos] = make_int2(qid, (int32_t)(c0 + (e >> (GS ? 10 : 8))));
                        } else {
                            *p.overflow = 1.0f;
                        }
                    }
                    if (lane == 0) *unc_cnt = 0;
                }
            }
            if (GS && more && !(dbg & 16)) {
                if (LV == 1) { KGE_HLOAD(ah0, bh0, sb_next, 0) } else { KGE_SLOAD(ah0, al0, bh0, bl0, sb_next, 0) }
            }
            if (more) { // query panel change (block-uniform): flush counters, load the next thresholds
                int qp_next, ct_next;
                item_qp_ct(it + 1, qp_next, ct_next);
                const int64_t next_q0 = (int64_t)qp_next * TQ;
                if (next_q0 != cur_q0) {
                    flush_counts(cur_q0);
                    cur_q0 = next_q0;
                    load_panel(cur_q0);
                }
            }
        }
        if (DBG && tl) {
            tsv[15] = __builtin_readcyclecounter();     // (behind the epilogue of a tile's last stage)
            if (lane == 0) {
                unsigned long long *o = reinterpret_cast<unsigned long long *>(p.list) + ((wid >> 2) * KGE_TL_STAGES + g) * 16;
#pragma unroll
                for (int i = 0; i < 16; ++i) o[i] = tsv[i];
            }
        }
        if (++s == S) { s = 0; ++it; }
    }
#undef KGE_TS
#undef KGE_SMMA_PA
#undef KGE_SMMA_P
#undef KGE_HLOAD
#undef KGE_HWAIT
#undef KGE_SLOAD
#undef KGE_SWAIT
#undef KGE_DSR
    flush_counts(cur_q0);
}

// Exact re-scoring of the listed pairs: one lane per pair, rows staged cooperatively
// (kge_common.h: lp_pair_score_staged).
template <bool VEC4>
__global__ __launch_bounds__(64, 2) void split_recheck_kernel(const kge_lp_desc d, const float *__restrict__ s_true,
                                                           const int32_t *__restrict__ list, int32_t cap,
                                                           const int32_t *__restrict__ list_count, int32_t *raw_count,
                                                           float *list_stat)
{
    __shared__ __attribute__((aligned(16))) float qs[64 * KGE_PS_LD];
    __shared__ __attribute__((aligned(16))) float es[64 * KGE_PS_LD];
    const int lane = threadIdx.x;
    const int n = (int)min((unsigned)*list_count, (unsigned)cap);   // (a count past the capacity means overflow: the caller redoes the count)
    if (list_stat && blockIdx.x == 0 && lane == 0) atomicAdd(list_stat, (float)n);   // pairs re-scored per evaluation (level policy)
    const int ngroups = (n + 63) >> 6;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int pi = grp * 64 + lane;
        const bool valid = pi < n;
        const int pj = valid ? pi : grp * 64;       // idle lanes shadow the group's first pair
        const int qi = list[2 * pj], ci = list[2 * pj + 1];
        const float sc = lp_pair_score_staged<VEC4>(d, qi, ci, qs, es);
        if (valid && !(sc >= s_true[qi])) atomicSub(&raw_count[qi], 1);
    }
}

// ---- exact re-scoring REGION BY REGION (r05) -------------------------------------------------------------------------
// The free-running sweep can leave its uncertain pairs in regions of the list, one per (query panel, 32-query sub-tile)
// (kge_split_args.region_count).  A block takes a region: the sub-tile's 32 query rows go to LDS ONCE (fp32, row stride an
// odd number of 16-byte pieces: conflict-free b128 reads at per-lane rows), then every pair costs the candidate row
// alone -- staged cooperatively like lp_staged_segment's -- instead of both rows: the recheck is bound by the L2's row
// bandwidth (8.8 TB/s of 1.6 KB per pair at cfg2), so half the bytes is most of half the time.  Same chains (lp_chain_dot
// on 32-column chunks, segment after segment), same epilogue: same bits as lp_pair_score.  Rows must be float4-readable
// (kge_lp_vec4).
__device__ __forceinline__ float recheck_e_segment(const float *__restrict__ T, int64_t ldt, int K, int ci,
                                                   const float *__restrict__ qrow, float *es, float acc)
{
    const int lane = threadIdx.x & 63;
    // (ONE 32-column chunk in flight per wavefront, as lp_staged_segment: with two -- 175 VGPRs -- the kernel took 83 us
    // instead of 68, profiles/r05/region_recheck_ab.txt)
    int k0 = 0;
    if (K >= KGE_PS_KC) {
        float4 e0, e1, e2, e3, e4, e5, e6, e7;
#define KGE_RR_FETCH(IT, KK)                                                                                  \
    {                                                                                                         \
        const int idx_ = lane + 64 * IT, rr_ = idx_ >> 3, pc_ = idx_ & 7;                                     \
        const int rc_ = __shfl(ci, rr_, 64);                                                                  \
        e##IT = *reinterpret_cast<const float4 *>(T + (int64_t)rc_ * ldt + (KK) + pc_ * 4);                   \
    }
#define KGE_RR_STORE(IT)                                                                                      \
    {                                                                                                         \
        const int idx_ = lane + 64 * IT, rr_ = idx_ >> 3, pc_ = idx_ & 7;                                     \
        *reinterpret_cast<float4 *>(es + rr_ * KGE_PS_LD + pc_ * 4) = e##IT;                                  \
    }
#define KGE_RR_ALL(M, ...) M(0, ##__VA_ARGS__) M(1, ##__VA_ARGS__) M(2, ##__VA_ARGS__) M(3, ##__VA_ARGS__) \
                           M(4, ##__VA_ARGS__) M(5, ##__VA_ARGS__) M(6, ##__VA_ARGS__) M(7, ##__VA_ARGS__)
        KGE_RR_ALL(KGE_RR_FETCH, 0)
        for (; k0 + KGE_PS_KC <= K; k0 += KGE_PS_KC) {
            KGE_RR_ALL(KGE_RR_STORE)
            if (k0 + 2 * KGE_PS_KC <= K) { KGE_RR_ALL(KGE_RR_FETCH, k0 + KGE_PS_KC) }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            acc = lp_chain_dot(qrow + k0, es + lane * KGE_PS_LD, KGE_PS_KC, acc);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
#undef KGE_RR_ALL
#undef KGE_RR_STORE
#undef KGE_RR_FETCH
    }
    for (; k0 < K; k0 += KGE_PS_KC) {      // the last, partial chunk (K % 4 == 0)
        const int kc = min(KGE_PS_KC, K - k0);
        const int pieces = kc >> 2;
        for (int idx = lane; idx < 64 * pieces; idx += 64) {
            const int rr = idx / pieces, pc = idx - rr * pieces;
            const int rc = __shfl(ci, rr, 64);
            *reinterpret_cast<float4 *>(es + rr * KGE_PS_LD + pc * 4) =
                *reinterpret_cast<const float4 *>(T + (int64_t)rc * ldt + k0 + pc * 4);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        acc = lp_chain_dot(qrow + k0, es + lane * KGE_PS_LD, kc, acc);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    return acc;
}

constexpr int RR_ROWS = 32, RR_PANEL = 96;      // queries per region / per panel of the free-running sweep (lp_hi_stream.hip)

// Rows longer than one LDS segment (r06; K > 256 or so -- DistMult / ComplEx d = 400: the recheck was 20 % of cfg4's step, all
// of it row fetches): the region's query rows pass through LDS in SEGMENTS of seg_cols logical columns of [A0 | A1] while the
// chains of up to RR_G pair groups per wave (a batch of NWV * 64 * RR_G pairs: a whole region, typically) rest in registers
// between segments -- the sequential chain is cut, not reordered: same bits.  One segment (K <= seg_cols) is r05's form.
constexpr int RR_G = 4;

template <int NWV>
__global__ __launch_bounds__(64 * NWV) void split_recheck_regions_kernel(const kge_lp_desc d, const float *__restrict__ s_true,
                                                                        const int32_t *__restrict__ list, int32_t region_cap,
                                                                        const int32_t *__restrict__ region_count, int n_regions,
                                                                        int ldq, int seg_cols, int32_t *raw_count, float *list_stat,
                                                                        int32_t *list_count)
{
    extern __shared__ __attribute__((aligned(16))) float rr_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float *qrows = rr_smem;
    float *es = rr_smem + RR_ROWS * ldq + wv * 64 * KGE_PS_LD;
    const int K = d.K0 + d.K1;
    int n_block = 0;
    // logical columns [c0, c1) of the region's 32 query rows -> LDS (row stride ldq)
    auto stage = [&](int64_t q0, int c0, int c1) {
        const int np = (c1 - c0) >> 2;
        for (int idx = tid; idx < RR_ROWS * np; idx += 64 * NWV) {
            const int rr = idx / np, pc = idx - rr * np;
            const int64_t q = min(q0 + rr, d.B - 1);
            const int col = c0 + pc * 4;        // (K0 % 4 == 0: a piece lies in one segment of the operand)
            const float4 v = col < d.K0 ? *reinterpret_cast<const float4 *>(d.A0 + q * d.lda0 + col)
                                        : *reinterpret_cast<const float4 *>(d.A1 + q * d.lda1 + (col - d.K0));
            *reinterpret_cast<float4 *>(qrows + rr * ldq + pc * 4) = v;
        }
    };
    // the chain of one pair over the staged columns [c0, c1): operand segment 0, then 1
    auto chain = [&](float acc, int ci, const float *qrow, int c0, int c1) -> float {
        if (c0 < d.K0) acc = recheck_e_segment(d.T0 + c0, d.ldt0, min(c1, d.K0) - c0, ci, qrow, es, acc);
        if (c1 > d.K0 && d.K1 > 0) {
            const int s0 = max(c0, d.K0);
            acc = recheck_e_segment(d.T1 + (s0 - d.K0), d.ldt1, c1 - s0, ci, qrow + (s0 - c0), es, acc);
        }
        return acc;
    };
    for (int reg = blockIdx.x; reg < n_regions; reg += gridDim.x) {
        const int n = (int)min((unsigned)region_count[reg], (unsigned)region_cap);   // (past the capacity: overflow flagged by the sweep)
        if (n == 0) continue;                   // (block-uniform)
        n_block += n;
        const int64_t q0 = (int64_t)(reg / 3) * RR_PANEL + (reg % 3) * RR_ROWS;
        const int2 *ent = reinterpret_cast<const int2 *>(list) + (int64_t)reg * region_cap;
        if (K <= seg_cols) {
            __syncthreads();                    // the previous region's readers are done
            stage(q0, 0, K);
            __syncthreads();
            for (int c0 = wv * 64; c0 < n; c0 += NWV * 64) {
                const int pi = c0 + lane;
                const bool valid = pi < n;
                const int2 e = ent[valid ? pi : c0];       // idle lanes shadow the group's first pair
                const int qi = e.x, ci = e.y;
                const float acc = chain(0.0f, ci, qrows + (int)(qi - q0) * ldq, 0, K);
                const float sc = lp_epilogue_any(d, acc, qi, ci);
                if (valid && !(sc >= s_true[qi])) atomicSub(&raw_count[qi], 1);
            }
            continue;
        }
        for (int b0 = 0; b0 < n; b0 += NWV * 64 * RR_G) {       // (block-uniform trip counts: barriers inside)
            float acc[RR_G];
            int qi[RR_G], ci[RR_G];
#pragma unroll
            for (int g = 0; g < RR_G; ++g) {
                const int c0 = b0 + (g * NWV + wv) * 64;
                const int pi = c0 + lane;
                const int2 e = ent[min(pi < n ? pi : c0, n - 1)];   // idle lanes shadow the group's first pair (idle groups: the last pair)
                qi[g] = e.x; ci[g] = e.y; acc[g] = 0.0f;
            }
            for (int c0 = 0; c0 < K; c0 += seg_cols) {
                const int c1 = min(K, c0 + seg_cols);
                __syncthreads();                // the previous segment's / region's readers are done
                stage(q0, c0, c1);
                __syncthreads();
#pragma unroll
                for (int g = 0; g < RR_G; ++g)
                    if (b0 + (g * NWV + wv) * 64 < n)       // (wave-uniform)
                        acc[g] = chain(acc[g], ci[g], qrows + (int)(qi[g] - q0) * ldq, c0, c1);
            }
#pragma unroll
            for (int g = 0; g < RR_G; ++g) {
                const int pi = b0 + (g * NWV + wv) * 64 + lane;
                if (pi < n) {
                    const float sc = lp_epilogue_any(d, acc[g], qi[g], ci[g]);
                    if (!(sc >= s_true[qi[g]])) atomicSub(&raw_count[qi[g]], 1);
                }
            }
        }
    }
    if (tid == 0 && n_block > 0) {              // pairs re-scored per evaluation (level policy) / the list's length
        if (list_stat) atomicAdd(list_stat, (float)n_block);
        if (list_count) atomicAdd(list_count, n_block);
    }
}

template <int NWAVES, bool DBG, int PM, int GS = 0, int LV = 0>
int launch_split(const SplitParams &p, int grid, hipStream_t s)
{
    auto k = lp_split_count_kernel<NWAVES, DBG, PM, GS, LV>;
    static int attr_dev[16];    // per instantiation, per device
    if (int e = kge_ensure_dyn_smem(reinterpret_cast<const void *>(k), SMEM_BYTES, attr_dev)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * NWAVES), SMEM_BYTES, s, p);
    KGE_CHECK_LAUNCH();
    return 0;
}

int split_num_cus()
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

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

} // namespace

extern "C" int kge_lp_split_units(int K, int with_aug)
{
    return (int)round_up((K + (with_aug ? 1 : 0) + 15) / 16, 2);
}

extern "C" int64_t kge_lp_split_rows_padded(int64_t rows, int is_query) { return round_up(rows, is_query ? TQ : TC); }

extern "C" int kge_lp_split_rows(const float *X0, int64_t ld0, int K0, const float *X1, int64_t ld1, int K1,
                                 int64_t rows, int is_query, int aug_mode, const float *aug, float aug_mul,
                                 const float *norm2max0, const float *norm2max1, void *out, float *cell_ss,
                                 const int64_t *row_index, kge_stream_t stream)
{
    if (rows < 0 || K0 <= 0 || K1 < 0 || ld0 < K0 || (K1 > 0 && ld1 < K1) || aug_mode < 0 || aug_mode > 4)
        return KGE_EINVAL;
    if (rows == 0 && is_query) return 0;
    if ((rows > 0 && !X0) || (rows > 0 && K1 > 0 && !X1) || !out || ((aug_mode == 1 || aug_mode == 3) && rows > 0 && !aug))
        return KGE_EINVAL;
    SplitRowsParams p;
    p.X0 = X0; p.X1 = X1; p.ld0 = ld0; p.ld1 = ld1; p.K0 = K0; p.K1 = K1;
    p.rows = rows;
    p.rows_p = kge_lp_split_rows_padded(rows, is_query);
    p.aug_mode = aug_mode; p.aug = aug; p.aug_mul = aug_mul;
    p.nmax0 = norm2max0; p.nmax1 = norm2max1;
    p.units_p = kge_lp_split_units(K0 + K1, aug_mode != 0);
    p.out = reinterpret_cast<uint4 *>(out);
    p.cell_ss = cell_ss;
    p.row_index = row_index;
    const int64_t total = (p.rows_p / 16) * ((p.units_p + 15) / 16);     // tiles of 16 rows x 16 cells
    if (total == 0) return 0;
    const int grid = (int)(total < 65536 ? total : 65536);
    hipLaunchKernelGGL(split_rows_kernel, dim3(grid), dim3(256), 0, kge_s(stream), p);
    KGE_CHECK_LAUNCH();
    return 0;
}

/* units of a PLANAR hi operand (one-product level): k16 units of K + 2 columns, rounded up to 4 (one 128-byte stage) */
extern "C" int kge_lp_hi_units(int K) { return (int)round_up((K + 2 + 15) / 16, 4); }

static int hi_rows_impl(const float *X0, int64_t ld0, int K0, const float *X1, int64_t ld1, int K1, int64_t rows,
                        int is_query, int aug_mode, const float *aug, float aug_mul, const float *norm2max0,
                        const float *norm2max1, void *out, float *dn2, float *dn2max, const int64_t *row_index,
                        int frag, kge_stream_t stream)
{
    if (rows < 0 || K0 <= 0 || K1 < 0 || ld0 < K0 || (K1 > 0 && ld1 < K1) || aug_mode < 1 || aug_mode > 4) return KGE_EINVAL;
    if (rows == 0 && is_query) return 0;
    if ((rows > 0 && !X0) || (rows > 0 && K1 > 0 && !X1) || !out || ((aug_mode == 1 || aug_mode == 3) && rows > 0 && !aug))
        return KGE_EINVAL;
    HiRowsParams p;
    p.X0 = X0; p.X1 = X1; p.ld0 = ld0; p.ld1 = ld1; p.K0 = K0; p.K1 = K1;
    p.rows = rows;
    p.rows_p = kge_lp_split_rows_padded(rows, is_query);
    p.aug_mode = aug_mode; p.aug = aug; p.aug_mul = aug_mul;
    p.nmax0 = norm2max0; p.nmax1 = norm2max1;
    p.units_p = kge_lp_hi_units(K0 + K1);
    p.out = reinterpret_cast<uint4 *>(out);
    p.dn2 = dn2; p.dn2max = dn2max; p.row_index = row_index;
    p.frag = frag;
    p.nm_bmax = nullptr; p.nm_blocks = 0; p.dn_bmax = nullptr; p.prev_nmax = nullptr; p.nm_out = nullptr;
    const int64_t blocks = p.rows_p / 16;
    if (blocks == 0) return 0;
    // every block ends with ONE same-address atomic (dn2max), and those serialise at ~20-30 ns each (measured r05 on the
    // fused table preparation: 1,824 of them cost 50 us): with a maximum to fold, two blocks per CU walk the rows
    const int64_t cap = dn2max ? 2 * (int64_t)split_num_cus() : 65536;
    hipLaunchKernelGGL(hi_rows_kernel, dim3((int)(blocks < cap ? blocks : cap)), dim3(256), 0, kge_s(stream), p);
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_lp_hi_rows(const float *X0, int64_t ld0, int K0, const float *X1, int64_t ld1, int K1, int64_t rows,
                              int is_query, int aug_mode, const float *aug, float aug_mul, const float *norm2max0,
                              const float *norm2max1, void *out, float *dn2, float *dn2max, const int64_t *row_index,
                              kge_stream_t stream)
{
    return hi_rows_impl(X0, ld0, K0, X1, ld1, K1, rows, is_query, aug_mode, aug, aug_mul, norm2max0, norm2max1, out, dn2,
                        dn2max, row_index, 0, stream);
}

/* the CANDIDATE operand of the free-running one-product kernel: same values, fragment-major layout (kge_split_args.es_frag) */
extern "C" int kge_lp_hi_rows_frag(const float *X0, int64_t ld0, int K0, const float *X1, int64_t ld1, int K1, int64_t rows,
                                   int aug_mode, const float *aug, float aug_mul, const float *norm2max0,
                                   const float *norm2max1, void *out, float *dn2, float *dn2max, kge_stream_t stream)
{
    return hi_rows_impl(X0, ld0, K0, X1, ld1, K1, rows, 0, aug_mode, aug, aug_mul, norm2max0, norm2max1, out, dn2, dn2max,
                        nullptr, 1, stream);
}

/* Candidate side of a DOT problem on the one-product level in TWO launches (r05): the squared-norm maxima of the table's one
 * or two segments in one sweep (block maxima, no atomics, no zero-fill), then the hi table (planar or fragment-major), whose
 * blocks fold those maxima into *norm2max0_io / *norm2max1_io (block 0 stores the scalars) and leave their residual maxima
 * per block in dn_block_max[kge_lp_dot_table_prep_blocks(rows, 1)] for kge_lp_dot_query_pipeline to fold into *de2max.
 * ws: 2 * kge_lp_dot_table_prep_blocks(rows, 0) floats of scratch. */
extern "C" int kge_lp_dot_table_prep_blocks(int64_t rows, int which)
{
    if (which == 0) { const int64_t b = (rows + 63) / 64; return (int)(b < 2048 ? (b > 0 ? b : 1) : 2048); }
    const int64_t b = kge_lp_split_rows_padded(rows, 0) / 16;
    return (int)(b < 4096 ? (b > 0 ? b : 1) : 4096);
}

extern "C" int kge_lp_dot_table_prep(const float *X0, int64_t ld0, int K0, const float *X1, int64_t ld1, int K1, int64_t rows,
                                     int frag, float *norm2max0_io, float *norm2max1_io, void *out, float *dn_block_max,
                                     float *ws, kge_stream_t stream)
{
    if (rows <= 0 || K0 <= 0 || K1 < 0 || ld0 < K0 || (K1 > 0 && ld1 < K1)) return KGE_EINVAL;
    if (!X0 || (K1 > 0 && (!X1 || !norm2max1_io)) || !norm2max0_io || !out || !dn_block_max || !ws) return KGE_EINVAL;
    const int nb = kge_lp_dot_table_prep_blocks(rows, 0);
    hipLaunchKernelGGL(dot_table_norm_max_kernel, dim3(nb), dim3(256), 0, kge_s(stream), X0, ld0, K0,
                       K1 > 0 ? X1 : nullptr, ld1, K1, rows, ws);
    KGE_CHECK_LAUNCH();
    HiRowsParams p;
    p.X0 = X0; p.X1 = K1 > 0 ? X1 : nullptr; p.ld0 = ld0; p.ld1 = ld1; p.K0 = K0; p.K1 = K1;
    p.rows = rows;
    p.rows_p = kge_lp_split_rows_padded(rows, 0);
    p.aug_mode = 4; p.aug = nullptr; p.aug_mul = 0.f;
    p.nmax0 = norm2max0_io; p.nmax1 = K1 > 0 ? norm2max1_io : nullptr;
    p.units_p = kge_lp_hi_units(K0 + K1);
    p.out = reinterpret_cast<uint4 *>(out);
    p.dn2 = nullptr; p.dn2max = nullptr; p.row_index = nullptr;
    p.frag = frag ? 1 : 0;
    p.nm_bmax = ws; p.nm_blocks = nb; p.dn_bmax = dn_block_max; p.prev_nmax = nullptr; p.nm_out = nullptr;
    // fragment-major tables of float4-readable rows: the coalesced kernel (KGE_HIROWS_OLD=1: the general one, for A/B runs)
    static const int old_only = getenv("KGE_HIROWS_OLD") ? atoi(getenv("KGE_HIROWS_OLD")) : 0;
    const bool vec = K0 % 4 == 0 && K1 % 4 == 0 && ld0 % 4 == 0 && ((size_t)X0 & 15) == 0 &&
                     (K1 == 0 || (ld1 % 4 == 0 && ((size_t)X1 & 15) == 0));
    if (frag && vec && !old_only)
        hipLaunchKernelGGL(hi_rows_frag_kernel<false>, dim3(kge_lp_dot_table_prep_blocks(rows, 1)), dim3(256), 0, kge_s(stream), p);
    else
        hipLaunchKernelGGL(hi_rows_kernel, dim3(kge_lp_dot_table_prep_blocks(rows, 1)), dim3(256), 0, kge_s(stream), p);
    KGE_CHECK_LAUNCH();
    return 0;
}

/* 1 if kge_lp_split_count takes a fragment-major candidate table (es_frag = 1) for K columns on the one-product level */
extern "C" int kge_lp_hi_stream_supported(int K)
{
    const int units = (K + 2 + 15) / 16;
    return (units <= kge_hi_stream_max_units() || kge_hi_chunk_supported(units)) ? 1 : 0;
}

/* Candidate-table preparation of the L2 one-product sweep in one pass: en[row] = ||X[row]||^2 by kge_row_sqnorm's
 * sequential chain (same bits), *en_max_io = max(., max en), the fragment-major hi table of kge_lp_hi_rows_frag(aug_mode 1,
 * aug = en, aug_mul = -0.5) and *dn2max_io = max(., max_row ||x - hi(x)||^2) -- one launch, the table read once.
 * K % 4 == 0, ld % 4 == 0, X 16-byte aligned (else KGE_EUNSUPPORTED: use the separate entry points). */
extern "C" int kge_lp_table_prep_blocks(int64_t rows)
{
    const int64_t blocks = kge_lp_split_rows_padded(rows, 0) / 16;
    return (int)(blocks < 65536 ? blocks : 65536);
}

extern "C" int kge_lp_table_prep_l2(const float *X, int64_t ld, int64_t rows, int K, float *en, float *en_max_io,
                                    void *out, float *dn2max_io, float *block_max, kge_stream_t stream)
{
    if (rows < 0 || K <= 0 || ld < K) return KGE_EINVAL;
    if (rows == 0) return 0;
    if (!X || !en || !out) return KGE_EINVAL;
    if (K % 4 != 0 || ld % 4 != 0 || !kge_aligned16(X) || K > 8192) return KGE_EUNSUPPORTED;
    TablePrepParams p;
    p.X = X; p.ld = ld; p.rows = rows; p.rows_p = kge_lp_split_rows_padded(rows, 0);
    p.K = K; p.units_p = kge_lp_hi_units(K);
    p.en = en; p.en_max = en_max_io; p.out = reinterpret_cast<uint4 *>(out); p.dn2max = dn2max_io;
    p.block_max = block_max;
    p.dbg = kge_env_int("KGE_TP_DBG", 0);
    const int lds_ld = ((K + 3) & ~3) + 4;
    const int smem = (16 * lds_ld + 16 * 17 + 16) * 4;
    const int64_t blocks = p.rows_p / 16;
    auto k = table_prep_l2_kernel;
    static int attr_dev[16];
    if (smem > 48 * 1024)
        if (int e = kge_ensure_dyn_smem(reinterpret_cast<const void *>(k), smem, attr_dev)) return e;
    hipLaunchKernelGGL(k, dim3((int)(blocks < 65536 ? blocks : 65536)), dim3(256), smem, kge_s(stream), p);
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_lp_split_prefix_max(const float *cell_ss, int64_t rows, int is_query, int units_p, float *e2pref,
                                       kge_stream_t stream)
{
    if (rows < 0 || units_p <= 0) return KGE_EINVAL;
    if (rows == 0) return 0;
    if (!cell_ss || !e2pref) return KGE_EINVAL;
    const int64_t rows_p = kge_lp_split_rows_padded(rows, is_query);
    if (units_p > 128) return KGE_EINVAL;       // (K <= 2031)
    const int64_t want = (rows + 1023) / 1024;
    hipLaunchKernelGGL(prefix_max_kernel, dim3((int)(want < 512 ? want : 512)), dim3(1024), 0, kge_s(stream), cell_ss,
                       rows_p, rows, units_p, e2pref);
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_lp_split_count(const kge_lp_desc *d, const kge_split_args *a, const float *s_true,
                                  int32_t *raw_count, kge_stream_t stream)
{
    int rc = kge_lp_desc_check(d);
    if (rc) return rc;
    if (!KGE_LP_IS_MFMA(d->mode)) return KGE_EINVAL;
    const bool proj = d->mode >= KGE_LP_L2_PROJH;
    if (d->B == 0 || d->N == 0) return 0;
    if (proj && (!a || (d->mode == KGE_LP_L2_PROJD && !a->yabsmax) || d->scal_ld % 4 != 0 ||
                 !kge_aligned16(d->scal)))
        return KGE_EINVAL;
    if (!a || !a->Qs || !a->Es || !s_true || !a->emax0 || !a->thr || !raw_count || !a->list || a->cap <= 0 ||
        !a->list_count || !a->overflow)
        return KGE_EINVAL;
    const bool pq = a->q_scale_per_query != 0;     // (DOT, one-product level: per-query operand scales, qn0 = the total norm)
    if (pq && (d->mode != KGE_LP_DOT || a->level != 1 || !a->qn0)) return KGE_EINVAL;
    if (d->mode == KGE_LP_DOT && !pq && (!a->qn0 || !a->qmax0 || (d->K1 > 0 && (!a->qn1 || !a->qmax1 || !a->emax1))))
        return KGE_EINVAL;
    if (d->mode == KGE_LP_DOT && d->K1 > 0 && !a->emax1) return KGE_EINVAL;
    if (d->B > INT32_MAX || d->N > INT32_MAX) return KGE_EINVAL;
    if (a->level != 0 && a->level != 1) return KGE_EINVAL;
    const bool lv1 = a->level == 1;     // one-product level: planar hi operands, plain thresholds
    if (lv1 && !a->thr_ready && (!a->q_dn2 || !a->de2max)) return KGE_EINVAL;
    hipStream_t s = kge_s(stream);
    const int K = d->K0 + d->K1;
    const int units_p = lv1 ? kge_lp_hi_units(K) : kge_lp_split_units(K, 1);
    const int units = lv1 ? (K + 2 + 15) / 16 : (K + 1 + 15) / 16;
    const int64_t Bp = kge_lp_split_rows_padded(d->B, 1);
    SplitThrParams t;
    t.mode = d->mode;
    t.qn0 = d->mode != KGE_LP_DOT ? d->qn : a->qn0;
    t.qn1 = (d->mode == KGE_LP_DOT && d->K1 > 0 && !pq) ? a->qn1 : nullptr;
    t.s_true = s_true;
    t.qmax0 = a->qmax0; t.qmax1 = (d->K1 > 0 && !pq) ? a->qmax1 : nullptr;
    t.q_scale_per_query = pq ? 1 : 0;
    // (thresholds recomputed = another sweep on these operands: the list's region counters start from zero like *list_count)
    t.zero_i32 = a->region_count; t.zero_n = a->region_count ? kge_lp_split_regions(d->B) : 0;
    t.emax0 = a->emax0; t.emax1 = (d->mode == KGE_LP_DOT && d->K1 > 0) ? a->emax1 : nullptr;
    t.B = d->B; t.Bp = Bp; t.K = K; t.units = units;
    t.eps_scale = a->eps_scale;
    t.thr = reinterpret_cast<float2 *>(a->thr);
    t.thr4 = reinterpret_cast<float4 *>(a->thr);
    t.pz = d->Wq; t.ldw = d->ldw;
    t.xabsmax = a->xabsmax; t.yabsmax = a->yabsmax;
    t.c_acc = a->accum_model == 1 ? 1.25f : 2.0f;
    t.q_cell_ss = a->q_cell_ss; t.e2pref = a->e2pref; t.units_p = units_p; t.K0 = d->K0;
    t.ss_index = a->q_cell_ss_index; t.ss_ld = a->q_cell_ss_index ? a->q_cell_ss_ld : Bp;
    if (a->q_cell_ss_index && a->q_cell_ss_ld <= 0) return KGE_EINVAL;
    t.level = a->level; t.q_dn2 = a->q_dn2; t.q_dn2_index = a->q_dn2_index; t.de2max = a->de2max;
    t.tp_bmax = a->tp_block_max; t.tp_blocks = a->tp_blocks;
    t.emax_out = const_cast<float *>(a->emax0); t.de2max_out = const_cast<float *>(a->de2max);
    if (a->tp_block_max && (a->tp_blocks <= 0 || a->emax1 || a->thr_ready)) return KGE_EINVAL;
    if (lv1) { t.q_cell_ss = nullptr; t.e2pref = nullptr; }
    t.list_count = a->list_count;
    t.overflow = a->overflow;
    if (!a->thr_ready) {    // (the fused query pipeline has already written thr and zeroed list_count)
        hipLaunchKernelGGL(split_thr_kernel, dim3((int)((Bp + 255) / 256)), dim3(256), 0, s, t);
        KGE_CHECK_LAUNCH();
    }

    const void *Es = a->Es, *Qs = a->Qs;
    float *thr = a->thr, *overflow = a->overflow;
    int32_t *list = a->list, *list_count = a->list_count;
    const int32_t cap = a->cap;
    SplitParams p;
    p.Es = reinterpret_cast<const char *>(Es);
    p.Qs = reinterpret_cast<const char *>(Qs);
    p.row_bytes = lv1 ? units_p * 32 : units_p * 64;
    p.units = units;
    p.stages = lv1 ? units_p / 4 : units_p / 2;
    p.B = d->B;
    p.N = d->N;
    p.thr = reinterpret_cast<const float2 *>(thr);
    p.thr4 = reinterpret_cast<const float4 *>(thr);
    p.X = d->scal; p.ldx = d->scal_ld; p.r_idx = d->r_idx; p.yc = d->yc;
    p.raw_count = raw_count;
    p.list = list;
    p.cap = cap;
    p.list_count = list_count;
    p.overflow = overflow;
    p.col_q = nullptr; p.members = nullptr;
    p.c_tiles = (int)((d->N + TC - 1) / TC);
    p.dbg = kge_env_int("KGE_SPLIT_DBG", 0);
    // Query panels interleaved under one sweep of the candidate tiles (work order, see the kernel).  Measured r03
    // (profiles/r03/split_work_order_sweep.txt): at K = 200 (172 KiB per panel) 16 panels per XCD cut the L2-miss
    // traffic from 646 to 295 MB per evaluate and are ~0.5 % FASTER in the sustained, power-capped state (4: 0.819,
    // 8: 0.812, 16: 0.807 ms per evaluate; 32 thrashes the 4 MiB L2 slice); at K = 400 (320 KiB per panel) 4 / 8 / 16
    // are within noise with 4 ahead -> 16 while 16 panels stay below ~3 MiB, else 4.
    p.qg = kge_env_int("KGE_SPLIT_QG", (int64_t)TQ * p.row_bytes * 16 <= (3 << 20) ? 16 : 4);
    const int slots = split_num_cus();
    if (a->es_frag) {
        // the free-running one-product kernel (lp_hi_stream.hip): fragment-major candidate table, resident query panel
        // long rows: the panel streamed in chunks (lp_hi_chunk.hip); KGE_HC_FORCE=1: also where the resident panel would fit (A/B)
        const bool chunked = units > kge_hi_stream_max_units() ||
                             (kge_env_int("KGE_HC_FORCE", 0) && kge_hi_chunk_supported(units) && !a->members && !a->region_count &&
                              d->mode < KGE_LP_L2_PROJH);
        const bool grouped = a->members && a->n_multi_p > 0;         // r06: + a second launch over the grouped columns
        if (!lv1 || (chunked && !kge_hi_chunk_supported(units))) return KGE_EINVAL;
        if ((a->members != nullptr) != (a->n_multi_p > 0) || (grouped && (chunked || proj || a->region_count || !a->col_q)))
            return KGE_EINVAL;
        if (grouped && (a->n_multi_p % TQ || GSETS != 4)) return KGE_EINVAL;
        kge_hi_stream_params h;
        h.Ef = reinterpret_cast<const char *>(Es);
        h.Qh = reinterpret_cast<const char *>(Qs);
        h.q_row_bytes = p.row_bytes;
        h.units = units; h.units_p = units_p;
        h.rows_p = kge_lp_split_rows_padded(d->N, 0);
        h.q_rows = a->col_q ? a->n_single_p : Bp;
        if (a->col_q && (a->n_single_p <= 0 || a->n_single_p % TQ)) return KGE_EINVAL;
        h.B = d->B;
        h.thr = p.thr; h.thr4 = p.thr4;
        h.X = p.X; h.ldx = p.ldx; h.r_idx = p.r_idx; h.yc = p.yc;
        h.raw_count = raw_count; h.list = list; h.cap = cap; h.list_count = list_count; h.overflow = overflow;
        h.col_q = a->col_q;
        h.members = nullptr;
        h.region_count = nullptr; h.region_cap = 0;
        if (a->region_count) {      // the list cut into regions (kge_lp_split_recheck_regions takes them)
            if (a->col_q || !kge_lp_split_regions_supported(d)) return KGE_EINVAL;
            const int n_regions = kge_lp_split_regions(d->B);
            h.region_count = a->region_count;
            h.region_cap = cap / n_regions;
            if (h.region_cap <= 0) return KGE_EINVAL;
        }
        h.true_idx = a->true_idx; h.c_base = d->c_base;
        const int pm = d->mode == KGE_LP_L2_PROJH ? 1 : (d->mode == KGE_LP_L2_PROJD ? 2 : 0);
        if (chunked) {
            if (pm != 0 || h.region_count) return KGE_EUNSUPPORTED;
            return kge_hi_chunk_launch(h, slots, s);
        }
        if (!grouped) return kge_hi_stream_launch(h, pm, slots, s);
        // columns: the single-query ones (col_q), then the grouped ones (members) -- two launches over the same candidate table
        if (a->n_single_p > 0) {
            rc = kge_hi_stream_launch(h, pm, slots, s);
            if (rc) return rc;
        }
        h.Qh = reinterpret_cast<const char *>(Qs) + a->n_single_p * (int64_t)p.row_bytes;
        h.q_rows = a->n_multi_p;
        h.col_q = nullptr;
        h.members = a->members;
        return kge_hi_stream_launch(h, 0, slots, s);
    }
    if (a->col_q || a->members) {
        // Columns instead of queries: Qs holds n_single_p rows that carry one query each (col_q), then n_multi_p rows
        // that carry up to GSETS queries of one key each (members); both counts multiples of the query panel.
        if (a->n_single_p < 0 || a->n_multi_p < 0 || a->n_single_p % TQ || a->n_multi_p % TQ ||
            (a->n_single_p > 0 && !a->col_q) || (a->n_multi_p > 0 && !a->members))
            return KGE_EINVAL;
        const int pm = d->mode == KGE_LP_L2_PROJH ? 1 : (d->mode == KGE_LP_L2_PROJD ? 2 : 0);
        if (a->n_single_p > 0) {
            p.col_q = a->col_q;
            p.q_panels = (int)(a->n_single_p / TQ);
            p.n_items = (int64_t)p.q_panels * p.c_tiles;
            const int grid = (int)(p.n_items < slots ? p.n_items : slots);
            rc = lv1 ? (pm == 1 ? launch_split<8, false, 1, 0, 1>(p, grid, s)
                                : (pm == 2 ? launch_split<8, false, 2, 0, 1>(p, grid, s) : launch_split<8, false, 0, 0, 1>(p, grid, s)))
                     : (pm == 1 ? launch_split<8, false, 1>(p, grid, s)
                                : (pm == 2 ? launch_split<8, false, 2>(p, grid, s) : launch_split<8, false, 0>(p, grid, s)));
            if (rc) return rc;
        }
        if (a->n_multi_p > 0) {
            p.col_q = nullptr;
            p.members = a->members;
            p.Qs = reinterpret_cast<const char *>(Qs) + a->n_single_p * (int64_t)p.row_bytes;
            p.q_panels = (int)(a->n_multi_p / TQ);
            p.n_items = (int64_t)p.q_panels * p.c_tiles;
            const int grid = (int)(p.n_items < slots ? p.n_items : slots);
            rc = lv1 ? (pm == 1 ? launch_split<8, false, 1, GSETS, 1>(p, grid, s)
                                : (pm == 2 ? launch_split<8, false, 2, GSETS, 1>(p, grid, s) : launch_split<8, false, 0, GSETS, 1>(p, grid, s)))
                     : (pm == 1 ? launch_split<8, false, 1, GSETS>(p, grid, s)
                                : (pm == 2 ? launch_split<8, false, 2, GSETS>(p, grid, s) : launch_split<8, false, 0, GSETS>(p, grid, s)));
        }
        return rc;
    }
    p.q_panels = (int)((d->B + TQ - 1) / TQ);
    p.n_items = (int64_t)p.q_panels * p.c_tiles;
    const int grid = (int)(p.n_items < slots ? p.n_items : slots);
    if (lv1 && d->mode == KGE_LP_L2_PROJH) return launch_split<8, false, 1, 0, 1>(p, grid, s);
    if (lv1 && d->mode == KGE_LP_L2_PROJD) return launch_split<8, false, 2, 0, 1>(p, grid, s);
    if (lv1) return p.dbg ? launch_split<8, true, 0, 0, 1>(p, grid, s) : launch_split<8, false, 0, 0, 1>(p, grid, s);
    if (d->mode == KGE_LP_L2_PROJH) return launch_split<8, false, 1>(p, grid, s);
    if (d->mode == KGE_LP_L2_PROJD) return launch_split<8, false, 2>(p, grid, s);
    return p.dbg ? launch_split<8, true, 0>(p, grid, s) : launch_split<8, false, 0>(p, grid, s);
}

/* threshold sets per grouped column (kge_split_args.members) */
extern "C" int kge_lp_split_group_sets(void) { return GSETS; }

extern "C" int kge_lp_split_recheck(const kge_lp_desc *d, const float *s_true, const int32_t *list, int32_t cap,
                                    const int32_t *list_count, int32_t *raw_count, float *list_stat, kge_stream_t stream)
{
    int rc = kge_lp_desc_check(d);
    if (rc) return rc;
    if (d->B == 0 || d->N == 0) return 0;
    if (!s_true || !list || cap <= 0 || !list_count || !raw_count) return KGE_EINVAL;
    if (!KGE_LP_IS_MFMA(d->mode)) return KGE_EINVAL;
    const bool vec4 = kge_lp_vec4(*d);
    const int grid = split_num_cus() * kge_env_int("KGE_SPLIT_RECHECK_WAVES", 160 * 1024 / (2 * 64 * KGE_PS_LD * 4));
    if (vec4)
        hipLaunchKernelGGL(split_recheck_kernel<true>, dim3(grid), dim3(64), 0, kge_s(stream), *d, s_true, list, cap,
                           list_count, raw_count, list_stat);
    else
        hipLaunchKernelGGL(split_recheck_kernel<false>, dim3(grid), dim3(64), 0, kge_s(stream), *d, s_true, list,
                           cap, list_count, raw_count, list_stat);
    KGE_CHECK_LAUNCH();
    return 0;
}

/* regions of the list of n queries' sweep: 3 per panel of 96 queries (kge_split_args.region_count) */
extern "C" int kge_lp_split_regions(int64_t B)
{
    return (int)(kge_lp_split_rows_padded(B, 1) / RR_PANEL) * 3;
}

// LDS segment of the region recheck: logical columns of the query rows resident at a time (a multiple of 32, the chunk of the
// candidate-row staging), from the byte budget of the 32 rows (KGE_REGION_MAX_BYTES, default 36 KiB: at K = 200 the region's
// 26 KB of query rows leave six wavefronts per CU; measured r05 -- profiles/r05/region_recheck_ab.txt -- a WHOLE 52 KB row
// block at K = 400 left four and was slower than no regions; r06 passes longer rows through in segments instead)
static int region_seg_cols(int K)
{
    const int max_ld = kge_env_int("KGE_REGION_MAX_BYTES", 36 * 1024) / (RR_ROWS * 4);
    const int ld_full = K + (((K >> 2) & 1) ? 0 : 4);
    if (ld_full <= max_ld) return K;
    // longer rows: SMALLER segments than the budget of a whole row block -- more wavefronts per CU is what the kernel lives on
    // (cfg4, DistMult d = 400, same box: no regions 2.107 ms per evaluate, 36 KiB segments 2.083, 20 KiB 2.031, 12 KiB 2.036;
    // profiles/r06/region_segments_ab.txt)
    const int seg_ld = min(max_ld, kge_env_int("KGE_REGION_SEG_BYTES", 20 * 1024) / (RR_ROWS * 4));
    int seg = ((seg_ld - 4) / 32) * 32;
    return seg < 32 ? 32 : seg;
}

/* 1 if kge_lp_split_count / kge_lp_split_recheck_regions take a list cut into regions for this problem */
extern "C" int kge_lp_split_regions_supported(const kge_lp_desc *d)
{
    if (kge_lp_desc_check(d) || !KGE_LP_IS_MFMA(d->mode) || !kge_lp_vec4(*d)) return 0;
    const int K = d->K0 + d->K1;
    // (rows of the free-running kernel's range: the chunked-panel kernel of longer rows keeps one global list; segments of the
    // operand must not cut a 16-byte piece: K0 % 4 == 0 is part of kge_lp_vec4)
    if ((K + 2 + 15) / 16 > 32) return 0;
    if (region_seg_cols(K) < K && kge_env_int("KGE_REGION_SEGMENTS", 1) == 0) return 0;
    return 1;
}

extern "C" int kge_lp_split_recheck_regions(const kge_lp_desc *d, const float *s_true, const int32_t *list, int32_t cap,
                                            const int32_t *region_count, int32_t *raw_count, float *list_stat,
                                            int32_t *list_count, kge_stream_t stream)
{
    int rc = kge_lp_desc_check(d);
    if (rc) return rc;
    if (d->B == 0 || d->N == 0) return 0;
    if (!s_true || !list || cap <= 0 || !region_count || !raw_count) return KGE_EINVAL;
    if (!kge_lp_split_regions_supported(d)) return KGE_EINVAL;
    const int n_regions = kge_lp_split_regions(d->B);
    const int32_t region_cap = cap / n_regions;
    if (region_cap <= 0) return KGE_EINVAL;
    const int K = d->K0 + d->K1;
    const int seg = region_seg_cols(K);
    const int ldq = seg + (((seg >> 2) & 1) ? 0 : 4);   // floats: a multiple of 4, an odd number of 16-byte pieces
    const int nwv = kge_env_int("KGE_RECHECK_REGION_WAVES", 2);
    const int smem = (RR_ROWS * ldq + (nwv == 4 ? 4 : 2) * 64 * KGE_PS_LD) * 4;
    const int want = split_num_cus() * 6;
    const int grid = n_regions < want ? n_regions : want;
    static int attr2[16], attr4[16];     // per device
    if (nwv == 4) {
        auto k = split_recheck_regions_kernel<4>;
        if (int e = kge_ensure_dyn_smem(reinterpret_cast<const void *>(k), smem, attr4)) return e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), smem, kge_s(stream), *d, s_true, list, region_cap, region_count, n_regions,
                           ldq, seg, raw_count, list_stat, list_count);
    } else {
        auto k = split_recheck_regions_kernel<2>;
        if (int e = kge_ensure_dyn_smem(reinterpret_cast<const void *>(k), smem, attr2)) return e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(128), smem, kge_s(stream), *d, s_true, list, region_cap, region_count, n_regions,
                           ldq, seg, raw_count, list_stat, list_count);
    }
    KGE_CHECK_LAUNCH();
    return 0;
}

/* *max_io = max(*max_io, max_i |x[i]|)  (device scalar, non-negative; the bound on the projection gather term) */
extern "C" int kge_absmax(const float *x, int64_t n, float *max_io, kge_stream_t stream)
{
    if (n < 0 || !max_io) return KGE_EINVAL;
    if (n == 0) return 0;
    if (!x) return KGE_EINVAL;
    const int64_t want = (n + 1023) / 1024;
    const int grid = (int)(want < 1024 ? want : 1024);
    hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, kge_s(stream), x, n, max_io);
    KGE_CHECK_LAUNCH();
    return 0;
}


/* 1 if this device's v_mfma_f32_32x32x16_f16 accumulates as modelled (see mfma_selftest_kernel), 0 if
 * not, negative / positive error codes as usual.  Synchronises; call it once, outside any capture. */
extern "C" int kge_mfma_f16_selftest(void)
{
    constexpr int NT_ = 8;
    float ab[NT_][32], c[NT_], expect[NT_];
    for (int t = 0; t < NT_; ++t) { for (int k = 0; k < 32; ++k) ab[t][k] = 0.f; c[t] = 0.f; }
    auto A = [&](int t, int k) -> float & { return ab[t][k]; };
    auto Bv = [&](int t, int k) -> float & { return ab[t][16 + k]; };
    // 0: [2^24, 1 x15]: exact 2^24+15 -> RNE 2^24+16 (sequential fp32 adds would give 2^24)
    for (int k = 0; k < 16; ++k) { A(0, k) = 1.f; Bv(0, k) = 1.f; } A(0, 0) = 4096.f; Bv(0, 0) = 4096.f; expect[0] = 16777232.f;
    // 1: C = 2^24 + [1,1,1]: 2^24+3 -> RNE 2^24+4 (truncation would give +2)
    A(1, 0) = A(1, 1) = A(1, 2) = 1.f; Bv(1, 0) = Bv(1, 1) = Bv(1, 2) = 1.f; c[1] = 16777216.f; expect[1] = 16777220.f;
    // 2: C = 2^24 + [1]: tie -> even
    A(2, 0) = 1.f; Bv(2, 0) = 1.f; c[2] = 16777216.f; expect[2] = 16777216.f;
    // 3: [2^24, -2^24, 1 x14]: addends 24 bits below the largest survive -> 14
    for (int k = 0; k < 16; ++k) { A(3, k) = 1.f; Bv(3, k) = 1.f; } A(3, 0) = 4096.f; Bv(3, 0) = 4096.f; A(3, 1) = 4096.f; Bv(3, 1) = -4096.f; expect[3] = 14.f;
    // 4: [2^24, -2^24, 0.5 x6 | 0.5 x8]: 25 bits below is cut in the first pass, the second pass is exact -> 4
    for (int k = 0; k < 16; ++k) { A(4, k) = 0.5f; Bv(4, k) = 1.f; } A(4, 0) = 4096.f; Bv(4, 0) = 4096.f; A(4, 1) = 4096.f; Bv(4, 1) = -4096.f; expect[4] = 4.f;
    // 5: [2^24, 1, 0.5]: no sticky bit -> tie to even 2^24
    A(5, 0) = 4096.f; Bv(5, 0) = 4096.f; A(5, 1) = 1.f; Bv(5, 1) = 1.f; A(5, 2) = 0.5f; Bv(5, 2) = 1.f; expect[5] = 16777216.f;
    // 6: C = 2^24, [1 | 0.5]: the second pass aligns to 2^24 as well
    A(6, 0) = 1.f; Bv(6, 0) = 1.f; A(6, 8) = 0.5f; Bv(6, 8) = 1.f; c[6] = 16777216.f; expect[6] = 16777216.f;
    // 7: f16 subnormal input 2^-20 * 2^10: kept
    A(7, 0) = 9.5367431640625e-07f; Bv(7, 0) = 1024.f; expect[7] = 0.0009765625f;
    float *d_ab = nullptr, *d_c = nullptr, *d_out = nullptr, got[NT_];
    hipError_t e = hipMalloc(&d_ab, sizeof(ab));
    if (e == hipSuccess) e = hipMalloc(&d_c, sizeof(c));
    if (e == hipSuccess) e = hipMalloc(&d_out, sizeof(got));
    if (e == hipSuccess) e = hipMemcpy(d_ab, ab, sizeof(ab), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_c, c, sizeof(c), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, 0, d_ab, d_c, d_out, NT_);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(got, d_out, sizeof(got), hipMemcpyDeviceToHost);
    if (d_ab) (void)hipFree(d_ab);
    if (d_c) (void)hipFree(d_c);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return (int)e;
    for (int t = 0; t < NT_; ++t)
        if (got[t] != expect[t]) return 0;
    return 1;
}

/* The same candidate side in ONE launch and ONE pass over the table (r06): the operand scale is the one the squared-norm
 * maxima of a PREVIOUS evaluation ask for (prev_nmax[2], device; kge_lp_dot_query_pipeline keeps them), the fragment-major
 * hi table, its residual maxima per block (dn_block_max) and THIS pass's squared-norm maxima per block
 * (nm_block_max[2][kge_lp_dot_table_prep_blocks(rows, 1)]) come out; hand both arrays and prev_nmax to
 * kge_lp_dot_query_pipeline, which folds them, raises *overflow when the table has outgrown (or fallen below) the scale
 * that was used -- the caller redoes that evaluation on another path -- and stores the new maxima for the next call.
 * KGE_EINVAL unless the rows are float4-readable (K0, K1, ld % 4 == 0, 16-byte aligned): then kge_lp_dot_table_prep. */
extern "C" int kge_lp_dot_table_prep_fused(const float *X0, int64_t ld0, int K0, const float *X1, int64_t ld1, int K1,
                                           int64_t rows, const float *prev_nmax, void *out, float *dn_block_max,
                                           float *nm_block_max, kge_stream_t stream)
{
    if (rows <= 0 || K0 <= 0 || K1 < 0 || ld0 < K0 || (K1 > 0 && ld1 < K1)) return KGE_EINVAL;
    if (!X0 || (K1 > 0 && !X1) || !prev_nmax || !out || !dn_block_max || !nm_block_max) return KGE_EINVAL;
    const bool vec = K0 % 4 == 0 && K1 % 4 == 0 && ld0 % 4 == 0 && ((size_t)X0 & 15) == 0 &&
                     (K1 == 0 || (ld1 % 4 == 0 && ((size_t)X1 & 15) == 0));
    if (!vec) return KGE_EINVAL;
    HiRowsParams p;
    p.X0 = X0; p.X1 = K1 > 0 ? X1 : nullptr; p.ld0 = ld0; p.ld1 = ld1; p.K0 = K0; p.K1 = K1;
    p.rows = rows;
    p.rows_p = kge_lp_split_rows_padded(rows, 0);
    p.aug_mode = 4; p.aug = nullptr; p.aug_mul = 0.f;
    p.nmax0 = nullptr; p.nmax1 = nullptr;
    p.units_p = kge_lp_hi_units(K0 + K1);
    p.out = reinterpret_cast<uint4 *>(out);
    p.dn2 = nullptr; p.dn2max = nullptr; p.row_index = nullptr;
    p.frag = 1;
    p.nm_bmax = nullptr; p.nm_blocks = 0; p.dn_bmax = dn_block_max; p.prev_nmax = prev_nmax; p.nm_out = nm_block_max;
    hipLaunchKernelGGL(hi_rows_frag_kernel<true>, dim3(kge_lp_dot_table_prep_blocks(rows, 1)), dim3(256), 0, kge_s(stream), p);
    KGE_CHECK_LAUNCH();
    return 0;
}

/* TransE-L2 query side of one batch in ONE launch (what kge_lp_prep + kge_row_sqnorm + kge_lp_pair_scores
 * (true scores) + kge_lp_split_rows(queries) + the threshold kernel of kge_lp_split_count do separately),
 * bit-identical outputs.  Q (B,d), qn (B), s_true (B), Qs (split operand), thr (2*Bp floats), *list_count = 0.
 * The candidate table must be the whole entity table (no shard).  Then call kge_lp_split_count with
 * thr_ready = 1. */
// The query side of one DistMult (E1 = R1 = NULL) / ComplEx batch on the one-product level in one launch, per-query operand
// scales (dot_query_pipeline_kernel).  emax0 / emax1 / de2max must hold their final values when the launch runs.
extern "C" int kge_lp_dot_query_pipeline(int side, const float *E0, const float *E1, const float *R0, const float *R1, int d,
                                         const int64_t *h, const int64_t *t, const int64_t *r, int64_t B,
                                         const float *emax0, const float *emax1, const float *de2max, float *qmax_io,
                                         int accum_model, float eps_scale, float *Q0, float *Q1, float *qn, float *s_true,
                                         void *Qh, float *thr, float *q_dn2, int32_t *list_count, float *overflow,
                                         int32_t *zero_i32, int64_t zero_n, const float *dn_block_max, int dn_blocks,
                                         const float *nm_block_max, int nm_blocks, float *prev_nmax, kge_stream_t stream)
{
    if (dn_block_max && dn_blocks <= 0) return KGE_EINVAL;
    if (nm_block_max && nm_blocks <= 0) return KGE_EINVAL;
    const bool both = side == KGE_SIDE_BOTH, cplx = E1 != nullptr;
    if ((side != KGE_SIDE_TAIL && side != KGE_SIDE_HEAD && !both) || d <= 0 || d > 4096 || B < 0) return KGE_EINVAL;
    if (B == 0) return 0;
    if (!E0 || !R0 || !h || !t || !r || !emax0 || !de2max || !Q0 || !qn || !s_true || !Qh || !thr || !list_count || !overflow)
        return KGE_EINVAL;
    if (cplx && (!R1 || !Q1 || !emax1)) return KGE_EINVAL;
    if (!cplx && (R1 || Q1)) return KGE_EINVAL;
    if (d % 8 != 0 || !kge_aligned16(E0) || !kge_aligned16(R0) || (cplx && (!kge_aligned16(E1) || !kge_aligned16(R1))))
        return KGE_EINVAL;      // float4 staging, hi cells of 8 columns inside one segment
    if (zero_n < 0 || (zero_n > 0 && !zero_i32)) return KGE_EINVAL;
    DotPipeParams p;
    p.tail = both ? 2 : (side == KGE_SIDE_TAIL ? 1 : 0);
    p.Bh = B;
    p.E0 = E0; p.E1 = E1; p.R0 = R0; p.R1 = R1; p.d = d; p.h = h; p.t = t; p.r = r;
    p.B = both ? 2 * B : B; p.Bp = kge_lp_split_rows_padded(p.B, 1);
    p.emax0 = emax0; p.emax1 = cplx ? emax1 : nullptr; p.de2max = de2max; p.qmax_io = qmax_io;
    p.c_acc = accum_model == 1 ? 1.25f : 2.0f; p.eps_scale = eps_scale;
    const int K = cplx ? 2 * d : d;
    p.units = (K + 2 + 15) / 16; p.units_p = kge_lp_hi_units(K);
    p.Q0 = Q0; p.Q1 = Q1; p.qn = qn; p.s_true = s_true; p.q_dn2 = q_dn2;
    p.thr = reinterpret_cast<float2 *>(thr);
    p.Qh = reinterpret_cast<_Float16 *>(Qh);
    p.list_count = list_count; p.overflow = overflow;
    p.zero_i32 = zero_i32; p.zero_n = zero_n;
    p.dn_bmax = dn_block_max; p.dn_blocks = dn_blocks;
    p.nm_bmax = nm_block_max; p.nm_blocks = nm_blocks; p.prev_nmax = prev_nmax;
    // queries per wavefront: 16 -- or 4 for a small batch (the kernel is a latency chain per group: fewer than two groups of
    // 16 per SIMD leave most of the chip idle while ~400 wavefronts walk 10 chunks each)
    const int qpw = kge_env_int("KGE_DQPIPE_QPW", p.Bp / 16 < 2048 ? 4 : 16);
    const int64_t groups = (p.Bp + qpw - 1) / qpw, blocks = (groups + 3) / 4;
    const int grid = (int)(blocks < 256 * 16 ? blocks : 256 * 16);
    if (qpw == 4) {
        if (cplx) hipLaunchKernelGGL((dot_query_pipeline_kernel<4, true>), dim3(grid), dim3(256), 0, kge_s(stream), p);
        else hipLaunchKernelGGL((dot_query_pipeline_kernel<4, false>), dim3(grid), dim3(256), 0, kge_s(stream), p);
    } else {
        if (cplx) hipLaunchKernelGGL((dot_query_pipeline_kernel<16, true>), dim3(grid), dim3(256), 0, kge_s(stream), p);
        else hipLaunchKernelGGL((dot_query_pipeline_kernel<16, false>), dim3(grid), dim3(256), 0, kge_s(stream), p);
    }
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_lp_query_pipeline(int side, const float *E, const float *R, int d, const int64_t *h,
                                     const int64_t *t, const int64_t *r, int64_t B, const float *en,
                                     const float *emax, float *qmax_io, int accum_model, float eps_scale, float *Q,
                                     float *qn, float *s_true, void *Qs, float *thr, int32_t *list_count,
                                     const float *e2pref, const int32_t *qs_row, int level, const float *de2max,
                                     float *q_dn2, const float *tp_block_max, int tp_blocks, int32_t *zero_i32,
                                     int64_t zero_n, kge_stream_t stream)
{
    if (level != 0 && level != 1) return KGE_EINVAL;
    if (level == 1 && !de2max) return KGE_EINVAL;
    const bool both = side == KGE_SIDE_BOTH;
    if ((side != KGE_SIDE_TAIL && side != KGE_SIDE_HEAD && !both) || d <= 0 || d > 4096 || B < 0) return KGE_EINVAL;
    if (B == 0) return 0;
    if (!E || !R || !h || !t || !r || !en || !emax || !Q || !qn || !s_true || !Qs || !thr || !list_count) return KGE_EINVAL;
    QueryPipeParams p;
    p.tail = both ? 2 : (side == KGE_SIDE_TAIL ? 1 : 0);
    p.Bh = B;
    p.E = E; p.R = R; p.d = d; p.h = h; p.t = t; p.r = r;
    p.B = both ? 2 * B : B; p.Bp = kge_lp_split_rows_padded(p.B, 1);
    p.en = en; p.emax = emax; p.qmax_io = qmax_io;
    p.c_acc = accum_model == 1 ? 1.25f : 2.0f; p.eps_scale = eps_scale;
    p.units = (d + 1 + 15) / 16; p.units_p = kge_lp_split_units(d, 1);
    p.level = level; p.de2max = de2max; p.q_dn2 = q_dn2;
    p.tp_bmax = tp_block_max; p.tp_blocks = tp_blocks;
    p.emax_out = const_cast<float *>(emax); p.de2max_out = const_cast<float *>(de2max);
    if (tp_block_max && tp_blocks <= 0) return KGE_EINVAL;
    if (zero_n < 0 || (zero_n > 0 && !zero_i32)) return KGE_EINVAL;
    p.zero_i32 = zero_i32; p.zero_n = zero_n;
    p.dbg = kge_env_int("KGE_QP_DBG", 0);
    if (level == 1) { p.units = (d + 2 + 15) / 16; p.units_p = kge_lp_hi_units(d); }
    p.Q = Q; p.qn = qn; p.s_true = s_true;
    p.thr = reinterpret_cast<float2 *>(thr);
    p.Qs = reinterpret_cast<_Float16 *>(Qs);
    p.list_count = list_count;
    p.e2pref = e2pref;
    p.qs_row = qs_row;
    if (d % 4 != 0 || !kge_aligned16(E) || !kge_aligned16(R)) return KGE_EINVAL;   // float4 staging
    const int qpw = kge_env_int("KGE_QPIPE_QPW", 16);
    const int64_t groups = (p.Bp + qpw - 1) / qpw, blocks = (groups + 3) / 4;
    const int grid = (int)(blocks < 256 * 16 ? blocks : 256 * 16);
    if (qpw == 8) hipLaunchKernelGGL(query_pipeline_kernel<8>, dim3(grid), dim3(256), 0, kge_s(stream), p);
    else if (qpw == 32) hipLaunchKernelGGL(query_pipeline_kernel<32>, dim3(grid), dim3(256), 0, kge_s(stream), p);
    else hipLaunchKernelGGL(query_pipeline_kernel<16>, dim3(grid), dim3(256), 0, kge_s(stream), p);
    KGE_CHECK_LAUNCH();
    return 0;
}
