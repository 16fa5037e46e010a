// K4: rank / filter kernels (integer results, bit-exact to the reference) and
// the entry points that dispatch the all-candidates scorers (gfx950).
//   get_rank                      utils/operations.py:37-61
//   get_true_targets/filter_scores utils/modeling.py:53-102
//   the rank/filter half of LinkPredictionEvaluator.evaluate  evaluation.py:290-300
// HBM-bound streaming over the (B,N) matrix when it is materialised; the fused
// path (pair scores + count_ge + filter_sub + finalize) never materialises it.
#include "kge_common.h"

int kge_lp_gemm_run(const kge_lp_desc *d, float *out, int64_t ldo, const float *s_true,
                    int32_t *raw_count, hipStream_t s);
int kge_lp_direct_count_cols(const kge_lp_desc *d, const float *s_true, int32_t *raw_count, const int64_t *rep,
                             const int32_t *col_q, int64_t n_single_p, const int32_t *members, int64_t n_multi_p,
                             hipStream_t s);
int kge_lp_direct_run(const kge_lp_desc *d, float *out, int64_t ldo, const float *s_true,
                      int32_t *raw_count, hipStream_t s);

namespace {

constexpr int RB = 256; // threads per row-block

__device__ __forceinline__ int block_sum_i(int v, int *sh)
{
    v = wave_sum_i(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    int t = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    return t;
}

__device__ __forceinline__ int count_row(const float *__restrict__ row, int64_t N, float tv, bool low)
{
    int c = 0;
    const int64_t head = ((16 - (reinterpret_cast<uintptr_t>(row) & 15)) & 15) >> 2; // floats to 16B
    const int64_t nh = head < N ? head : N;
    for (int64_t j = threadIdx.x; j < nh; j += blockDim.x) c += low ? (row[j] <= tv) : (row[j] >= tv);
    const int64_t n4 = (N - nh) >> 2;
    const float4 *r4 = reinterpret_cast<const float4 *>(row + nh);
    for (int64_t j = threadIdx.x; j < n4; j += blockDim.x) {
        const float4 v = r4[j];
        if (low) c += (v.x <= tv) + (v.y <= tv) + (v.z <= tv) + (v.w <= tv);
        else c += (v.x >= tv) + (v.y >= tv) + (v.z >= tv) + (v.w >= tv);
    }
    for (int64_t j = nh + (n4 << 2) + threadIdx.x; j < N; j += blockDim.x)
        c += low ? (row[j] <= tv) : (row[j] >= tv);
    return c;
}

__global__ __launch_bounds__(RB) void get_rank_kernel(const float *__restrict__ scores, int64_t ld,
                                                      const int64_t *__restrict__ true_idx, int64_t B,
                                                      int64_t N, int low, int64_t *rank)
{
    __shared__ int sh[RB / 64];
    for (int64_t i = blockIdx.x; i < B; i += gridDim.x) {
        const float *row = scores + i * ld;
        const float tv = row[true_idx[i]];
        const int c = block_sum_i(count_row(row, N, tv, low != 0), sh);
        if (threadIdx.x == 0) rank[i] = c;
    }
}

__global__ __launch_bounds__(RB) void filtered_rank_kernel(const float *__restrict__ scores, int64_t ld,
                                                           const int64_t *__restrict__ true_idx,
                                                           const int64_t *__restrict__ seg_lo,
                                                           const int64_t *__restrict__ seg_hi,
                                                           const int32_t *__restrict__ targets, int64_t B,
                                                           int64_t N, int64_t *rank, int64_t *filt)
{
    __shared__ int sh[RB / 64];
    for (int64_t i = blockIdx.x; i < B; i += gridDim.x) {
        const float *row = scores + i * ld;
        const int64_t ti = true_idx[i];
        const float tv = row[ti];
        const int raw = block_sum_i(count_row(row, N, tv, false), sh);
        int sub = 0, found = 0;
        const int neg_inf_counts = (-INFINITY >= tv) ? 1 : 0;
        for (int64_t j = seg_lo[i] + threadIdx.x; j < seg_hi[i]; j += blockDim.x) {
            const int64_t c = targets[j];
            if (c == ti) { found = 1; continue; }
            if (c >= 0 && c < N) sub += ((row[c] >= tv) ? 1 : 0) - neg_inf_counts;
        }
        sub = block_sum_i(sub, sh);
        found = block_sum_i(found, sh);
        if (threadIdx.x == 0) {
            rank[i] = raw;
            filt[i] = found ? raw - sub : raw;
        }
    }
}

// Ranks of `rows` queries whose score rows arrive as P RANK-MAJOR tiles (the receive buffer of the score all-to-all of
// the entity-sharded path, kge_hip_coll.h): tile p = (m, per) holds the scores of global candidates [p*per, (p+1)*per),
// row i of every tile belongs to query q_first + i.  Same arithmetic as filtered_rank_kernel on the (B, N) matrix the
// tiles would concatenate to -- no re-layout, every tile row read as one contiguous segment.  Results go straight to
// the (4, ld) result matrix of kge_rank_finalize_both (query q < B: tail side of fact q, else head side of fact q - B).
__global__ __launch_bounds__(RB) void filtered_rank_tiles_kernel(const float *__restrict__ tiles, int64_t m, int64_t per,
                                                                 int P, int64_t N, const int64_t *__restrict__ true_idx,
                                                                 const int64_t *__restrict__ seg_lo,
                                                                 const int64_t *__restrict__ seg_hi,
                                                                 const int32_t *__restrict__ targets, int64_t rows,
                                                                 int64_t q_first, int64_t B, int64_t *out, int64_t ld,
                                                                 int64_t off, const int64_t *__restrict__ pos,
                                                                 const float *__restrict__ own, int own_rank)
{
    __shared__ int sh[RB / 64];
    // tile p: the receive buffer's block p -- or, for the caller's own rank, the block it scored itself, read where the
    // scorer wrote it (`own`): the own block never needs to be copied
    auto tile = [&](int64_t p) -> const float * { return (own && p == own_rank) ? own : tiles + p * m * per; };
    for (int64_t i = blockIdx.x; i < rows; i += gridDim.x) {
        const int64_t ti = true_idx[i];
        const int64_t tp = ti / per;
        const float tv = tile(tp)[i * per + (ti - tp * per)];
        int c = 0;
        for (int p = 0; p < P; ++p) {
            const int64_t w = N - (int64_t)p * per;      // candidates this tile really holds (the last one may be short)
            if (w <= 0) break;
            c += count_row(tile(p) + i * per, w < per ? w : per, tv, false);
        }
        const int raw = block_sum_i(c, sh);
        int sub = 0, found = 0;
        const int neg_inf_counts = (-INFINITY >= tv) ? 1 : 0;
        for (int64_t j = seg_lo[i] + threadIdx.x; j < seg_hi[i]; j += blockDim.x) {
            const int64_t cc = targets[j];
            if (cc == ti) { found = 1; continue; }
            if (cc >= 0 && cc < N) {
                const int64_t cp = cc / per;
                sub += ((tile(cp)[i * per + (cc - cp * per)] >= tv) ? 1 : 0) - neg_inf_counts;
            }
        }
        sub = block_sum_i(sub, sh);
        found = block_sum_i(found, sh);
        if (threadIdx.x == 0) {
            const int64_t q = q_first + i;
            const bool tail = q < B;
            const int64_t j = off + (tail ? q : q - B);
            const int64_t f = pos ? pos[j] : j;
            out[(tail ? 1 : 0) * ld + f] = raw;
            out[(tail ? 3 : 2) * ld + f] = found ? raw - sub : raw;
        }
    }
}

__global__ __launch_bounds__(RB) void filter_scores_kernel(float *scores, int64_t ld,
                                                           const int64_t *__restrict__ true_idx,
                                                           const int64_t *__restrict__ seg_lo,
                                                           const int64_t *__restrict__ seg_hi,
                                                           const int32_t *__restrict__ targets, int64_t B,
                                                           int64_t N)
{
    __shared__ int sh[RB / 64];
    for (int64_t i = blockIdx.x; i < B; i += gridDim.x) {
        const int64_t lo = seg_lo[i], hi = seg_hi[i];
        // true_idx == NULL (get_true_targets(..., true_idx=None), modeling.py:83-84): mask every known target
        const int64_t ti = true_idx ? true_idx[i] : -1;
        if (true_idx) {
            int found = 0;
            for (int64_t j = lo + threadIdx.x; j < hi; j += blockDim.x) found |= (targets[j] == ti);
            found = block_sum_i(found, sh);
            if (!found) continue; // KeyError / set.remove KeyError: row untouched (modeling.py:87-88)
        }
        float *row = scores + i * ld;
        for (int64_t j = lo + threadIdx.x; j < hi; j += blockDim.x) {
            const int64_t c = targets[j];
            if (c != ti && c >= 0 && c < N) row[c] = -INFINITY;
        }
    }
}

__global__ void filter_lookup_kernel(const int64_t *__restrict__ keys, int64_t n_keys,
                                     const int64_t *__restrict__ offsets, const int64_t *__restrict__ key1,
                                     const int64_t *__restrict__ key2, int64_t n_key2, int64_t B,
                                     int64_t *seg_lo, int64_t *seg_hi)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t key = key1[i] * n_key2 + key2[i];
        int64_t lo = 0, hi = n_keys;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (keys[mid] < key) lo = mid + 1; else hi = mid;
        }
        if (lo < n_keys && keys[lo] == key) { seg_lo[i] = offsets[lo]; seg_hi[i] = offsets[lo + 1]; }
        else { seg_lo[i] = 0; seg_hi[i] = 0; }
    }
}

__global__ void filter_lookup_both_kernel(const int64_t *__restrict__ keys_t, int64_t n_t,
                                          const int64_t *__restrict__ offs_t, const int64_t *__restrict__ keys_h,
                                          int64_t n_h, const int64_t *__restrict__ offs_h, int64_t base_h,
                                          const int64_t *__restrict__ h, const int64_t *__restrict__ t,
                                          const int64_t *__restrict__ r, int64_t n_key2, int64_t B,
                                          int64_t *seg_lo, int64_t *seg_hi, int64_t *true_idx)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * B; i += (int64_t)gridDim.x * blockDim.x) {
        const bool tail = i < B;
        const int64_t f = tail ? i : i - B;
        const int64_t *keys = tail ? keys_t : keys_h, *offs = tail ? offs_t : offs_h;
        const int64_t n_keys = tail ? n_t : n_h, base = tail ? 0 : base_h;
        const int64_t key = (tail ? h[f] : t[f]) * n_key2 + r[f];
        int64_t lo = 0, hi = n_keys;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (keys[mid] < key) lo = mid + 1; else hi = mid;
        }
        if (lo < n_keys && keys[lo] == key) { seg_lo[i] = offs[lo] + base; seg_hi[i] = offs[lo + 1] + base; }
        else { seg_lo[i] = 0; seg_hi[i] = 0; }
        true_idx[i] = tail ? t[f] : h[f];
    }
}

__global__ void pair_scores_kernel(const kge_lp_desc d, const int64_t *__restrict__ qi,
                                   const int64_t *__restrict__ ci, int64_t P, float *out)
{
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = qi ? qi[p] : p;
        const int64_t c = ci[p] - d.c_base;
        out[p] = (c >= 0 && c < d.N) ? lp_pair_score(d, i, c) : 0.f;
    }
}

// MFMA modes: one wavefront per block, rows staged cooperatively (lp_pair_score_staged)
template <bool VEC4, int DIRECT = 0>
__global__ __launch_bounds__(64, 2) void pair_scores_staged_kernel(const kge_lp_desc d, const int64_t *__restrict__ qi,
                                                                const int64_t *__restrict__ ci, int64_t P, float *out)
{
    __shared__ __attribute__((aligned(16))) float qs[64 * KGE_PS_LD];
    __shared__ __attribute__((aligned(16))) float es[64 * KGE_PS_LD];
    const int lane = threadIdx.x;
    const int64_t ngroups = (P + 63) >> 6;
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int64_t p = grp * 64 + lane;
        int64_t i = 0, c = -1;
        if (p < P) { i = qi ? qi[p] : p; c = ci[p] - d.c_base; }
        const bool ok = p < P && c >= 0 && c < d.N;
        float sc;
        if (DIRECT) sc = lp_pair_score_staged_direct<VEC4, DIRECT == 1>(d, ok ? (int)i : 0, ok ? (int)c : 0, qs, es);
        else sc = lp_pair_score_staged<VEC4>(d, ok ? (int)i : 0, ok ? (int)c : 0, qs, es);
        if (p < P) out[p] = ok ? sc : 0.f;
    }
}

// 8 lanes per query (8 queries per wavefront); the lanes of a group stride over
// the query's filter segment and score each listed candidate with the same
// arithmetic as the tile kernels.  Most segments hold a handful of entities.
__global__ __launch_bounds__(256) void filter_sub_kernel(const kge_lp_desc d, const float *__restrict__ s_true,
                                                         const int64_t *__restrict__ true_idx,
                                                         const int64_t *__restrict__ seg_lo,
                                                         const int64_t *__restrict__ seg_hi,
                                                         const int32_t *__restrict__ targets,
                                                         int32_t *sub_out, int32_t *found_out)
{
    constexpr int LPQ = 8;
    const int sub_lane = threadIdx.x & (LPQ - 1);
    const int64_t group = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LPQ;
    const int64_t ngroups = (int64_t)gridDim.x * 256 / LPQ;
    const int64_t rounds = (d.B + ngroups - 1) / ngroups;   // uniform trip count: shuffles need all lanes
    for (int64_t rd = 0; rd < rounds; ++rd) {
        const int64_t i = group + rd * ngroups;
        int sub = 0, found = 0;
        if (i < d.B) {
            const float tv = s_true[i];
            const int64_t ti = true_idx[i];
            const int neg_inf_counts = (-INFINITY >= tv) ? 1 : 0;
            for (int64_t j = seg_lo[i] + sub_lane; j < seg_hi[i]; j += LPQ) {
                const int64_t cg = targets[j];
                const int64_t c = cg - d.c_base;
                if (c < 0 || c >= d.N) continue;
                if (cg == ti) { found = 1; continue; }
                sub += ((lp_pair_score(d, i, c) >= tv) ? 1 : 0) - neg_inf_counts;
            }
        }
#pragma unroll
        for (int o = LPQ / 2; o > 0; o >>= 1) {
            sub += __shfl_xor(sub, o, 64);
            found += __shfl_xor(found, o, 64);
        }
        if (i < d.B && sub_lane == 0) { sub_out[i] = sub; found_out[i] = found ? 1 : 0; }
    }
}

// MFMA modes: the same 8-lanes-per-query walk, but every round's 64 (query, candidate)
// pairs are scored through the cooperative row staging (one wavefront per block)
template <bool VEC4>
__global__ __launch_bounds__(64, 2) void filter_sub_staged_kernel(const kge_lp_desc d, const float *__restrict__ s_true,
                                                               const int64_t *__restrict__ true_idx,
                                                               const int64_t *__restrict__ seg_lo,
                                                               const int64_t *__restrict__ seg_hi,
                                                               const int32_t *__restrict__ targets,
                                                               int32_t *sub_out, int32_t *found_out)
{
    constexpr int LPQ = 8;
    __shared__ __attribute__((aligned(16))) float qs[64 * KGE_PS_LD];
    __shared__ __attribute__((aligned(16))) float es[64 * KGE_PS_LD];
    const int lane = threadIdx.x, sub_lane = lane & (LPQ - 1);
    const int64_t nq = (d.B + 7) >> 3;                 // groups of 8 queries
    for (int64_t qg = blockIdx.x; qg < nq; qg += gridDim.x) {
        const int64_t i = qg * 8 + (lane >> 3);
        const bool live = i < d.B;
        int sub = 0, found = 0;
        const float tv = live ? s_true[i] : 0.f;
        const int64_t ti = live ? true_idx[i] : -1;
        const int64_t lo = live ? seg_lo[i] : 0, hi = live ? seg_hi[i] : 0;
        const int neg_inf_counts = (-INFINITY >= tv) ? 1 : 0;
        int len = (int)(hi - lo);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) len = max(len, __shfl_xor(len, o, 64));
        for (int j0 = 0; j0 < len; j0 += LPQ) {        // wave-uniform trip count
            const int64_t j = lo + j0 + sub_lane;
            bool score = false;
            int64_t c = 0;
            if (j < hi) {
                const int64_t cg = targets[j];
                c = cg - d.c_base;
                if (c >= 0 && c < d.N) {
                    if (cg == ti) found = 1;
                    else score = true;
                }
            }
            if (__ballot(score) == 0ull) continue;      // this round lists only true entities / other shards' candidates
            const float sc = lp_pair_score_staged<VEC4>(d, score ? (int)i : 0, score ? (int)c : 0, qs, es);
            if (score) sub += ((sc >= tv) ? 1 : 0) - neg_inf_counts;
        }
#pragma unroll
        for (int o = LPQ / 2; o > 0; o >>= 1) {
            sub += __shfl_xor(sub, o, 64);
            found += __shfl_xor(found, o, 64);
        }
        if (live && sub_lane == 0) { sub_out[i] = sub; found_out[i] = found ? 1 : 0; }
    }
}

// ---- filter correction, grouped and flattened (kge_lp_filter_sub_grouped) ----------------------
// Real link-prediction test splits are heavy-tailed: many queries share a key -- (h, r) on the tail
// side, (t, r) on the head side -- and a hub key's filter list holds thousands of entities (FB15k-237:
// gender / nationality / profession).  A key fixes BOTH the filter list and the query row, so the
// exact scores of a list are the same for every query of that key.  Instead of walking each query's
// list (8 lanes per query, the wavefront looping to its longest list: the r01 kernel), the lists the
// batch touches are scored ONCE per key into fs[] (indexed like targets[]), all (key, target) pairs
// flattened over the whole grid, and every query then only COMPARES its true score with its list's
// scores.  Work is bounded by the size of the target array, whatever the skew.
//   claim[T]  : smallest query index whose segment starts at that target position (0xffffffff: none)
//   woff[B+1] : exclusive prefix sum of the claimed segments' lengths (the flattened work list)
// (only the entries the batch touches are reset: O(B), not O(n_targets), and no memset node in a captured graph)
__global__ void fsub_reset_kernel(const int64_t *__restrict__ seg_lo, const int64_t *__restrict__ seg_hi, int64_t B,
                                  unsigned *claim)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x)
        if (seg_hi[i] > seg_lo[i]) claim[seg_lo[i]] = 0xffffffffu;
}
__global__ void fsub_claim_kernel(const int64_t *__restrict__ seg_lo, const int64_t *__restrict__ seg_hi, int64_t B,
                                  unsigned *claim)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x)
        if (seg_hi[i] > seg_lo[i]) atomicMin(&claim[seg_lo[i]], (unsigned)i);
}

// exclusive prefix sum of the claimed segments' lengths, reduce-then-scan over blocks of FS_SCAN_T queries:
//   fsub_len_kernel   len[i] (0 for queries that are not their segment's leader) + one sum per block
//   fsub_bscan_kernel exclusive scan of the block sums (one block; any number of block sums)
//   fsub_off_kernel   woff[i] = block base + exclusive scan inside the block;  woff[B] = total
constexpr int FS_SCAN_T = 1024;
__device__ __forceinline__ int64_t fsub_shfl_up64(int64_t v, int o)
{
    const unsigned vlo = __shfl_up((unsigned)(v & 0xffffffffll), o, 64);
    const unsigned vhi = __shfl_up((unsigned)((uint64_t)v >> 32), o, 64);
    return (int64_t)(((uint64_t)vhi << 32) | vlo);
}
// inclusive scan of one value per thread over a block of FS_SCAN_T threads; returns (inclusive, block total)
__device__ __forceinline__ int64_t fsub_block_scan(int64_t v, int64_t *wsum, int64_t &total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int64_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int64_t u = fsub_shfl_up64(inc, o);
        if (lane >= o) inc += u;
    }
    __syncthreads();
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int64_t before = 0, tot = 0;
    for (int w = 0; w < FS_SCAN_T / 64; ++w) {
        const int64_t x = wsum[w];
        if (w < wv) before += x;
        tot += x;
    }
    total = tot;
    return before + inc;
}
__global__ __launch_bounds__(FS_SCAN_T) void fsub_len_kernel(const int64_t *__restrict__ seg_lo,
                                                             const int64_t *__restrict__ seg_hi, int64_t B,
                                                             const unsigned *__restrict__ claim, int64_t *woff,
                                                             int64_t *bsum)
{
    __shared__ int64_t wsum[FS_SCAN_T / 64];
    const int64_t i = (int64_t)blockIdx.x * FS_SCAN_T + threadIdx.x;
    int64_t l = 0;
    if (i < B) {
        const int64_t lo = seg_lo[i], hi = seg_hi[i];
        if (hi > lo && claim[lo] == (unsigned)i) l = hi - lo;
        woff[i] = l;     // (turned into the offset by fsub_off_kernel)
    }
    int64_t total;
    fsub_block_scan(l, wsum, total);
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}
__global__ __launch_bounds__(FS_SCAN_T) void fsub_bscan_kernel(int64_t *bsum, int64_t nb, int64_t *woff_total)
{
    __shared__ int64_t wsum[FS_SCAN_T / 64];
    int64_t carry = 0;
    for (int64_t base = 0; base < nb; base += FS_SCAN_T) {
        const int64_t b = base + threadIdx.x;
        const int64_t v = b < nb ? bsum[b] : 0;
        int64_t total;
        const int64_t inc = fsub_block_scan(v, wsum, total);
        if (b < nb) bsum[b] = carry + inc - v;
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *woff_total = carry;
}
__global__ __launch_bounds__(FS_SCAN_T) void fsub_off_kernel(int64_t B, int64_t *woff, const int64_t *__restrict__ bsum)
{
    __shared__ int64_t wsum[FS_SCAN_T / 64];
    const int64_t i = (int64_t)blockIdx.x * FS_SCAN_T + threadIdx.x;
    const int64_t l = i < B ? woff[i] : 0;
    int64_t total;
    const int64_t inc = fsub_block_scan(l, wsum, total);
    if (i < B) woff[i] = bsum[blockIdx.x] + inc - l;
}

// scores of the flattened (claimed key, target) pairs: one lane per pair, 64 pairs per wavefront round
template <bool STAGED, bool VEC4, int DIRECT = 0>   // DIRECT: 0 MFMA modes / scalar; 1 plain L1 direct, 2 plain L2 direct (staged)
__global__ __launch_bounds__(64, 2) void fsub_score_kernel(const kge_lp_desc d, const int64_t *__restrict__ seg_lo,
                                                        const int32_t *__restrict__ targets,
                                                        const int64_t *__restrict__ woff, float *fs)
{
    __shared__ __attribute__((aligned(16))) float qs[STAGED ? 64 * KGE_PS_LD : 4];
    __shared__ __attribute__((aligned(16))) float es[STAGED ? 64 * KGE_PS_LD : 4];
    const int lane = threadIdx.x;
    const int64_t W = woff[d.B];
    for (int64_t w0 = (int64_t)blockIdx.x * 64; w0 < W; w0 += (int64_t)gridDim.x * 64) {
        const int64_t w = w0 + lane;
        const bool valid = w < W;
        int64_t lo = 0, hi = d.B; // first index with woff[idx] > w, minus one (zero-length entries are skipped)
        if (valid) {
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (woff[mid + 1] <= w) lo = mid + 1; else hi = mid;
            }
        }
        const int64_t i = lo;
        int64_t j = 0, c = -1;
        if (valid) {
            j = seg_lo[i] + (w - woff[i]);
            c = (int64_t)targets[j] - d.c_base;
        }
        const bool ok = valid && c >= 0 && c < d.N;
        float sc;
        if (STAGED && DIRECT) sc = lp_pair_score_staged_direct<VEC4, DIRECT == 1>(d, ok ? (int)i : 0, ok ? (int)c : 0, qs, es);
        else if (STAGED) sc = lp_pair_score_staged<VEC4>(d, ok ? (int)i : 0, ok ? (int)c : 0, qs, es);
        else sc = ok ? lp_pair_score(d, i, c) : 0.f;
        if (ok) fs[j] = sc;
    }
}

// Comparison of every query's true score with the scores of its list.  Lists of up to FS_SHORT entries: one
// wavefront per query, 8 independent loads per lane in flight (a hub list walked 64 entries per dependent
// round was ~1 us per round).  Longer lists (hub keys: thousands of entities, shared by hundreds of queries):
// `long_q` names those queries and one 256-thread BLOCK takes each; without it the wavefront loops.
constexpr int FS_SHORT = 512;
__device__ __forceinline__ void fsub_cmp(const kge_lp_desc &d, const int32_t *__restrict__ targets,
                                         const float *__restrict__ fs, int64_t j, int64_t hi, int64_t ti, float tv,
                                         int neg_inf_counts, int &sub, int &found)
{
    if (j >= hi) return;
    const int64_t cg = targets[j];
    const int64_t c = cg - d.c_base;
    if (c < 0 || c >= d.N) return;
    if (cg == ti) { found = 1; return; }
    sub += ((fs[j] >= tv) ? 1 : 0) - neg_inf_counts;
}
// the two compare kernels of the filter correction: short lists (a wavefront per query) and hub lists (a block per
// query); bodies as device functions so that ONE launch can run both side by side (fsub_count_both_kernel)
__device__ __forceinline__ void fsub_count_short(const kge_lp_desc &d, const float *__restrict__ s_true,
                                                 const int64_t *__restrict__ true_idx, const int64_t *__restrict__ seg_lo,
                                                 const int64_t *__restrict__ seg_hi, const int32_t *__restrict__ targets,
                                                 const float *__restrict__ fs, int skip_long, int32_t *sub_out,
                                                 int32_t *found_out, int bid, int nblk)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)bid * 4 + (threadIdx.x >> 6), nwaves = (int64_t)nblk * 4;
    for (int64_t i = wave; i < d.B; i += nwaves) {
        const int64_t lo = seg_lo[i], hi = seg_hi[i];
        if (skip_long && hi - lo > FS_SHORT) continue;      // the hub-list body writes this query
        int sub = 0, found = 0;
        if (hi > lo) {
            const float tv = s_true[i];
            const int64_t ti = true_idx[i];
            const int neg_inf_counts = (-INFINITY >= tv) ? 1 : 0;
            for (int64_t j0 = lo; j0 < hi; j0 += FS_SHORT) {
#pragma unroll
                for (int u = 0; u < FS_SHORT / 64; ++u)
                    fsub_cmp(d, targets, fs, j0 + u * 64 + lane, hi, ti, tv, neg_inf_counts, sub, found);
            }
            sub = wave_sum_i(sub);
            found = wave_sum_i(found);
        }
        if (lane == 0) { sub_out[i] = sub; found_out[i] = found ? 1 : 0; }
    }
}
__device__ __forceinline__ void fsub_count_long(const kge_lp_desc &d, const float *__restrict__ s_true,
                                                const int64_t *__restrict__ true_idx, const int64_t *__restrict__ seg_lo,
                                                const int64_t *__restrict__ seg_hi, const int32_t *__restrict__ targets,
                                                const float *__restrict__ fs, const int64_t *__restrict__ long_q,
                                                int64_t n_long, int32_t *sub_out, int32_t *found_out, int bid, int nblk,
                                                int *sh)
{
    for (int64_t q = bid; q < n_long; q += nblk) {
        const int64_t i = long_q[q];
        const int64_t lo = seg_lo[i], hi = seg_hi[i];
        const float tv = s_true[i];
        const int64_t ti = true_idx[i];
        const int neg_inf_counts = (-INFINITY >= tv) ? 1 : 0;
        int sub = 0, found = 0;
        for (int64_t j0 = lo; j0 < hi; j0 += 1024) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                fsub_cmp(d, targets, fs, j0 + u * 256 + threadIdx.x, hi, ti, tv, neg_inf_counts, sub, found);
        }
        sub = block_sum_i(sub, sh);
        found = block_sum_i(found, sh);
        if (threadIdx.x == 0) { sub_out[i] = sub; found_out[i] = found ? 1 : 0; }
    }
}
__global__ __launch_bounds__(256) void fsub_count_kernel(const kge_lp_desc d, const float *__restrict__ s_true,
                                                         const int64_t *__restrict__ true_idx,
                                                         const int64_t *__restrict__ seg_lo,
                                                         const int64_t *__restrict__ seg_hi,
                                                         const int32_t *__restrict__ targets,
                                                         const float *__restrict__ fs, int skip_long,
                                                         int32_t *sub_out, int32_t *found_out)
{
    fsub_count_short(d, s_true, true_idx, seg_lo, seg_hi, targets, fs, skip_long, sub_out, found_out, blockIdx.x, gridDim.x);
}
// ONE launch for both: the first gridDim.x - short_blocks blocks run the hub-list body, the rest the short-list body (they write disjoint
// queries) -- the two kernels were 17 + 17 us back to back, each far from filling the GPU
__global__ __launch_bounds__(256) void fsub_count_both_kernel(const kge_lp_desc d, const float *__restrict__ s_true,
                                                              const int64_t *__restrict__ true_idx,
                                                              const int64_t *__restrict__ seg_lo,
                                                              const int64_t *__restrict__ seg_hi,
                                                              const int32_t *__restrict__ targets,
                                                              const float *__restrict__ fs,
                                                              const int64_t *__restrict__ long_q, int64_t n_long,
                                                              int short_blocks, int32_t *sub_out, int32_t *found_out)
{
    __shared__ int sh[4];
    const int long_blocks = (int)gridDim.x - short_blocks;     // the hub-list blocks come FIRST: they are the long ones
    if ((int)blockIdx.x < long_blocks)
        fsub_count_long(d, s_true, true_idx, seg_lo, seg_hi, targets, fs, long_q, n_long, sub_out, found_out,
                        (int)blockIdx.x, long_blocks, sh);
    else
        fsub_count_short(d, s_true, true_idx, seg_lo, seg_hi, targets, fs, 1, sub_out, found_out,
                         (int)blockIdx.x - long_blocks, short_blocks);
}

__global__ void rank_finalize_kernel(const int32_t *__restrict__ raw, const int32_t *__restrict__ sub,
                                     const int32_t *__restrict__ found, int64_t B, int64_t *rank, int64_t *filt)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = raw[i];
        rank[i] = r;
        filt[i] = found[i] ? r - sub[i] : r;
    }
}

// 2B-query batches (tail-side queries first): ranks straight into the evaluator's (4, n) result rows
// [head raw, tail raw, head filtered, tail filtered] at column off + fact
__global__ void rank_finalize_both_kernel(const int32_t *__restrict__ raw, const int32_t *__restrict__ sub,
                                          const int32_t *__restrict__ found, int64_t B, int64_t *out, int64_t ld,
                                          int64_t off, const int64_t *__restrict__ pos,
                                          float *__restrict__ guard, float *flags, int zero_guard,
                                          int64_t *const *out_indirect)
{
    // (r06) out_indirect: the result matrix of THIS launch is *out_indirect -- a device-visible pointer the host stores there
    // before the launch (a hipGraph replays the launch with the pointer of the day): pinned host memory, so that the ranks
    // need no copy of their own; the flags then sit behind the four rows, as in the evaluator's packed buffer
    if (out_indirect) {
        out = *out_indirect;
        if (flags) flags = reinterpret_cast<float *>(out + 4 * ld);
    }
    // the evaluation's two guard decisions ride the last finalize (instead of an add + a copy node of their own):
    // flags[0] = max ||q||^2 + max ||e||^2 (norm-expansion guard), flags[1] = overflow of the uncertain-pair list
    if (flags && blockIdx.x == 0 && threadIdx.x == 0) {
        flags[0] = guard[0] + guard[1];
        flags[1] = guard[2];
        flags[2] = guard[6];        // pairs the split prefilter re-scored in this evaluation (kge_lp_split_recheck list_stat)
        // ... and the guard vector is left ZEROED for the next evaluation (nobody reads it after this point): the next
        // evaluate() starts without a fill node of its own
        if (zero_guard) {
#pragma unroll
            for (int j = 0; j < 8; ++j) guard[j] = 0.f;
        }
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * B; i += (int64_t)gridDim.x * blockDim.x) {
        const bool tail = i < B;
        const int64_t j = off + (tail ? i : i - B);
        const int64_t f = pos ? pos[j] : j;        // facts processed in another order (e.g. sorted by relation)
        const int64_t r = raw[i];
        out[(tail ? 1 : 0) * ld + f] = r;
        out[(tail ? 3 : 2) * ld + f] = found[i] ? r - sub[i] : r;
    }
}

// generic per-query candidate matrices: one wavefront per (query, candidate)
__global__ __launch_bounds__(256) void lp_batched_kernel(int mode, const float *__restrict__ q, int64_t ldq,
                                                         const float *__restrict__ cand, int64_t stride_b,
                                                         int64_t stride_n, int64_t B, int64_t N, int K,
                                                         float *out, int64_t ldo)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t total = B * N;
    for (int64_t pidx = wave; pidx < total; pidx += (int64_t)gridDim.x * 4) {
        const int64_t i = pidx / N, c = pidx - i * N;
        const float *qq = q + i * ldq;
        const float *cc = cand + i * stride_b + c * stride_n;
        float acc = 0.f;
        for (int k = lane; k < K; k += 64) {
            if (mode == KGE_LP_DOT) acc = fmaf(qq[k], cc[k], acc);
            else {
                const float diff = qq[k] - cc[k];
                acc = (mode == KGE_LP_L1_DIRECT) ? acc + fabsf(diff) : fmaf(diff, diff, acc);
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) out[i * ldo + c] = (mode == KGE_LP_DOT) ? acc : -acc;
    }
}

// top-k of each row in the strict order (score descending, index ascending):
// pass j finds the largest element that is strictly after the (j-1)-th pick, so
// nothing is marked or copied; k passes over a row that stays in L2.  NaNs are
// never selected (as with `>`-based comparison); exhausted rows yield (-inf, -1).
__global__ __launch_bounds__(RB) void topk_kernel(const float *__restrict__ scores, int64_t ld, int64_t B,
                                                  int64_t N, int k, int64_t *out_idx, float *out_val)
{
    __shared__ float sv[RB / 64];
    __shared__ int64_t si[RB / 64];
    for (int64_t i = blockIdx.x; i < B; i += gridDim.x) {
        const float *row = scores + i * ld;
        float last_v = INFINITY;
        int64_t last_i = -1;
        for (int j = 0; j < k; ++j) {
            float bv = -INFINITY;
            int64_t bi = -1;
            for (int64_t c = threadIdx.x; c < N; c += blockDim.x) {
                const float v = row[c];
                const bool after = (v < last_v) || (v == last_v && c > last_i);   // not picked yet
                const bool better = (v > bv) || (v == bv && (bi < 0 || c < bi));
                if (after && better && v == v) { bv = v; bi = c; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64);
                const int64_t oi = __shfl_xor(bi, o, 64);
                if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
            }
            __syncthreads();
            if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
            __syncthreads();
            bv = sv[0]; bi = si[0];
            for (int w = 1; w < RB / 64; ++w)
                if (si[w] >= 0 && (bi < 0 || sv[w] > bv || (sv[w] == bv && si[w] < bi))) { bv = sv[w]; bi = si[w]; }
            if (threadIdx.x == 0) { out_idx[i * k + j] = bi; out_val[i * k + j] = bi >= 0 ? bv : -INFINITY; }
            if (bi < 0) { last_v = -INFINITY; last_i = N; } else { last_v = bv; last_i = bi; }
        }
    }
}

// Top-k of a row CHUNK: the columns are candidates [c_base, c_base + C) of a larger candidate set (one tile of the
// entity table, or one entity shard), processed tile by tile so that only (B, C) scores ever exist.  Optionally the
// known targets of the row's filter segment that fall into the chunk are masked first (filter_scores with
// true_idx = None, utils/modeling.py:83-84 as used by inference.py:146, :241) -- in place, the tile is scratch.
// Output slot `col_off` of a (B, ldo) buffer: the tile's k best as (score, GLOBAL id), order (score descending, id
// ascending).  The same kernel MERGES partial lists: `ids_in` then names the candidates of the columns (entries
// with id < 0 are padding and never selected); partial lists laid out chunk after chunk keep the id-ascending tie
// order because every chunk's list is itself in that order and chunks are ascending id ranges.
__global__ __launch_bounds__(RB) void topk_chunk_kernel(float *__restrict__ scores, int64_t ld, int64_t B, int64_t C,
                                                        int64_t c_base, int k, const int64_t *__restrict__ seg_lo,
                                                        const int64_t *__restrict__ seg_hi,
                                                        const int32_t *__restrict__ targets,
                                                        const int64_t *__restrict__ ids_in, int64_t ld_ids,
                                                        int64_t *out_idx, float *out_val, int64_t ldo, int64_t col_off)
{
    __shared__ float sv[RB / 64];
    __shared__ int64_t si[RB / 64];
    for (int64_t i = blockIdx.x; i < B; i += gridDim.x) {
        float *row = scores + i * ld;
        if (targets) {
            for (int64_t j = seg_lo[i] + threadIdx.x; j < seg_hi[i]; j += blockDim.x) {
                const int64_t t = (int64_t)targets[j] - c_base;
                if (t >= 0 && t < C) row[t] = -INFINITY;
            }
            __syncthreads();
        }
        const int64_t *ids = ids_in ? ids_in + i * ld_ids : nullptr;
        float last_v = INFINITY;
        int64_t last_i = -1;
        for (int j = 0; j < k; ++j) {
            float bv = -INFINITY;
            int64_t bi = -1;
            for (int64_t c = threadIdx.x; c < C; c += blockDim.x) {
                const float v = row[c];
                const bool after = (v < last_v) || (v == last_v && c > last_i);   // not picked yet
                const bool better = (v > bv) || (v == bv && (bi < 0 || c < bi));
                const bool real = ids ? ids[c] >= 0 : true;
                if (after && better && v == v && real) { bv = v; bi = c; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64);
                const int64_t oi = __shfl_xor(bi, o, 64);
                if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
            }
            __syncthreads();
            if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
            __syncthreads();
            bv = sv[0]; bi = si[0];
            for (int w = 1; w < RB / 64; ++w)
                if (si[w] >= 0 && (bi < 0 || sv[w] > bv || (sv[w] == bv && si[w] < bi))) { bv = sv[w]; bi = si[w]; }
            if (threadIdx.x == 0) {
                out_idx[i * ldo + col_off + j] = bi < 0 ? -1 : (ids ? ids[bi] : bi + c_base);
                out_val[i * ldo + col_off + j] = bi >= 0 ? bv : -INFINITY;
            }
            if (bi < 0) { last_v = -INFINITY; last_i = C; } else { last_v = bv; last_i = bi; }
        }
        __syncthreads();
    }
}

// The same selection in ONE pass over the tile for k <= KMAX (r04): a wavefront per row, every lane keeps the KMAX best
// of its columns (lane, lane + 64, ...) as a sorted register list -- a compare-exchange chain per visited element whose
// list it enters --, then the 64 lists are merged by k rounds of a wave arg-max over the list heads (the winner's list
// shifts up).  Order (score descending, id ascending), NaN never selected, -inf entries fill up in id order, padding
// ids (< 0, merge mode) skipped: output identical to topk_chunk_kernel's, which re-read the whole tile k times.
template <int KMAX>
__global__ __launch_bounds__(RB) void topk_chunk_reg_kernel(float *__restrict__ scores, int64_t ld, int64_t B, int64_t C,
                                                            int64_t c_base, int k, const int64_t *__restrict__ seg_lo,
                                                            const int64_t *__restrict__ seg_hi,
                                                            const int32_t *__restrict__ targets,
                                                            const int64_t *__restrict__ ids_in, int64_t ld_ids,
                                                            int64_t *out_idx, float *out_val, int64_t ldo, int64_t col_off)
{
    constexpr int EMPTY = 0x7fffffff;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int64_t i0 = (int64_t)blockIdx.x * (RB / 64); i0 < B; i0 += (int64_t)gridDim.x * (RB / 64)) {
        const int64_t i = i0 + wv;
        const bool active = i < B;
        float *row = scores + (active ? i : 0) * ld;
        if (targets) {
            if (active)
                for (int64_t j = seg_lo[i] + lane; j < seg_hi[i]; j += 64) {
                    const int64_t t = (int64_t)targets[j] - c_base;
                    if (t >= 0 && t < C) row[t] = -INFINITY;
                }
            __syncthreads();
        }
        if (!active) continue;      // (no block-wide barrier below this point)
        const int64_t *ids = ids_in ? ids_in + i * ld_ids : nullptr;
        float lv[KMAX];
        int li[KMAX];
#pragma unroll
        for (int j = 0; j < KMAX; ++j) { lv[j] = -INFINITY; li[j] = EMPTY; }
        // Wave-wide pruning threshold: after 16 / 64 / 256 full iterations the k-th best entry of the 64 lists is selected
        // (on copies).  Every later element has a LARGER id than all entries seen so far (iteration t covers ids
        // [64 t, 64 t + 63]), so one whose score does not exceed that k-th score already has k entries ahead of it in
        // the (score descending, id ascending) order and can never be selected: after the first thousand columns only
        // ~k ln(C / 1024) elements per ROW still run the insertion chain, and the scan is bandwidth bound.
        float tau = 0.f;
        bool tau_on = false;
        auto refresh_tau = [&]() __attribute__((always_inline)) {    // (every lane active)
            float cv[KMAX];
            int ci[KMAX];
#pragma unroll
            for (int j = 0; j < KMAX; ++j) { cv[j] = lv[j]; ci[j] = li[j]; }
            float bv = -INFINITY;
            int bi = EMPTY;
            for (int j = 0; j < k; ++j) {
                bv = cv[0];
                bi = ci[0];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const float ov = __shfl_xor(bv, o, 64);
                    const int oi = __shfl_xor(bi, o, 64);
                    if (oi != EMPTY && (bi == EMPTY || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
                }
                if (bi != EMPTY && ci[0] == bi) {
#pragma unroll
                    for (int q = 0; q + 1 < KMAX; ++q) { cv[q] = cv[q + 1]; ci[q] = ci[q + 1]; }
                    cv[KMAX - 1] = -INFINITY;
                    ci[KMAX - 1] = EMPTY;
                }
            }
            tau_on = bi != EMPTY;       // k real entries exist: their k-th score prunes
            tau = bv;
        };
        auto visit = [&](float v, int64_t c, bool real) __attribute__((always_inline)) {
            // enters the list iff it beats the list's last entry (strictly, or at equal score by the smaller column)
            if (real && (!tau_on || v > tau) && (v > lv[KMAX - 1] || (v == lv[KMAX - 1] && (int)c < li[KMAX - 1]))) {
                float cv = v;
                int ci = (int)c;
#pragma unroll
                for (int j = 0; j < KMAX; ++j) {
                    const bool gt = cv > lv[j] || (cv == lv[j] && ci < li[j]);
                    const float tv = gt ? lv[j] : cv;
                    const int ti = gt ? li[j] : ci;
                    lv[j] = gt ? cv : lv[j];
                    li[j] = gt ? ci : li[j];
                    cv = tv;
                    ci = ti;
                }
            }
        };
        // full steps of UN iterations (every lane active): the UN loads are issued together, then visited in id order
        constexpr int UN = 8;
        int64_t t = 0;                                  // iteration = 64 consecutive columns
        const int64_t t_full = C / (64 * UN) * UN;      // iterations covered by full steps
        for (; t < t_full; t += UN) {
            if (t == 16 || t == 64 || t == 256) refresh_tau();
            float v[UN];
            bool real[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int64_t c = (t + u) * 64 + lane;
                v[u] = row[c];
                real[u] = ids ? ids[c] >= 0 : true;
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) visit(v[u], (t + u) * 64 + lane, real[u]);
        }
        for (int64_t c = t * 64 + lane; c < C; c += 64) visit(row[c], c, ids ? ids[c] >= 0 : true);
        for (int j = 0; j < k; ++j) {
            float bv = lv[0];
            int bi = li[0];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (oi != EMPTY && (bi == EMPTY || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
            }
            if (lane == 0) {
                out_idx[i * ldo + col_off + j] = bi == EMPTY ? -1 : (ids ? ids[bi] : (int64_t)bi + c_base);
                out_val[i * ldo + col_off + j] = bi != EMPTY ? bv : -INFINITY;
            }
            if (bi != EMPTY && li[0] == bi) {     // this lane's head was taken: its list moves up
#pragma unroll
                for (int q = 0; q + 1 < KMAX; ++q) { lv[q] = lv[q + 1]; li[q] = li[q + 1]; }
                lv[KMAX - 1] = -INFINITY;
                li[KMAX - 1] = EMPTY;
            }
        }
    }
}

inline int grid1d(int64_t n, int per_block)
{
    int64_t b = (n + per_block - 1) / per_block;
    const int64_t cap = 256 * 16;
    return (int)(b < cap ? (b > 0 ? b : 1) : cap);
}

} // namespace

extern "C" int kge_get_rank(const float *scores, int64_t ld, const int64_t *true_idx, int64_t B, int64_t N,
                            int low_values, int64_t *rank, kge_stream_t stream)
{
    if (B < 0 || N <= 0 || ld < N) return KGE_EINVAL;
    if (B == 0) return 0;
    if (!scores || !true_idx || !rank) return KGE_EINVAL;
    hipLaunchKernelGGL(get_rank_kernel, dim3(grid1d(B, 1)), dim3(RB), 0, kge_s(stream), scores, ld, true_idx, B, N,
                       low_values, rank);
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_filter_lookup(const int64_t *keys, int64_t n_keys, const int64_t *offsets,
                                 const int64_t *key1, const int64_t *key2, int64_t n_key2, int64_t B,
                                 int64_t *seg_lo, int64_t *seg_hi, kge_stream_t stream)
{
    if (B < 0 || n_keys < 0 || n_key2 <= 0) return KGE_EINVAL;
    if (B == 0) return 0;
    if (!key1 || !key2 || !seg_lo || !seg_hi || (n_keys > 0 && (!keys || !offsets))) return KGE_EINVAL;
    hipLaunchKernelGGL(filter_lookup_kernel, dim3(grid1d(B, 256)), dim3(256), 0, kge_s(stream), keys, n_keys,
                       offsets, key1, key2, n_key2, B, seg_lo, seg_hi);
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_filter_lookup_both(const int64_t *keys_t, int64_t n_keys_t, const int64_t *offsets_t,
                                      const int64_t *keys_h, int64_t n_keys_h, const int64_t *offsets_h,
                                      int64_t targets_base_h, const int64_t *h, const int64_t *t, const int64_t *r,
                                      int64_t n_key2, int64_t B, int64_t *seg_lo, int64_t *seg_hi,
                                      int64_t *true_idx, kge_stream_t stream)
{
    if (B < 0 || n_keys_t < 0 || n_keys_h < 0 || n_key2 <= 0 || targets_base_h < 0) return KGE_EINVAL;
    if (B == 0) return 0;
    if (!h || !t || !r || !seg_lo || !seg_hi || !true_idx || (n_keys_t > 0 && (!keys_t || !offsets_t)) ||
        (n_keys_h > 0 && (!keys_h || !offsets_h)))
        return KGE_EINVAL;
    hipLaunchKernelGGL(filter_lookup_both_kernel, dim3(grid1d(2 * B, 256)), dim3(256), 0, kge_s(stream), keys_t,
                       n_keys_t, offsets_t, keys_h, n_keys_h, offsets_h, targets_base_h, h, t, r, n_key2, B, seg_lo,
                       seg_hi, true_idx);
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_filter_scores(float *scores, int64_t ld, const int64_t *true_idx, const int64_t *seg_lo,
                                 const int64_t *seg_hi, const int32_t *targets, int64_t B, int64_t N,
                                 kge_stream_t stream)
{
    if (B < 0 || N <= 0 || ld < N) return KGE_EINVAL;
    if (B == 0) return 0;
    if (!scores || !seg_lo || !seg_hi) return KGE_EINVAL; // true_idx may be NULL: filter all known targets
    hipLaunchKernelGGL(filter_scores_kernel, dim3(grid1d(B, 1)), dim3(RB), 0, kge_s(stream), scores, ld, true_idx,
                       seg_lo, seg_hi, targets, B, N);
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_filtered_rank_from_scores(const float *scores, int64_t ld, const int64_t *true_idx,
                                             const int64_t *seg_lo, const int64_t *seg_hi,
                                             const int32_t *targets, int64_t B, int64_t N, int64_t *rank,
                                             int64_t *filt_rank, kge_stream_t stream)
{
    if (B < 0 || N <= 0 || ld < N) return KGE_EINVAL;
    if (B == 0) return 0;
    if (!scores || !true_idx || !seg_lo || !seg_hi || !rank || !filt_rank) return KGE_EINVAL;
    hipLaunchKernelGGL(filtered_rank_kernel, dim3(grid1d(B, 1)), dim3(RB), 0, kge_s(stream), scores, ld, true_idx,
                       seg_lo, seg_hi, targets, B, N, rank, filt_rank);
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_filtered_rank_from_tiles(const float *tiles, int64_t m, int64_t per, int world, int64_t N,
                                            const int64_t *true_idx, const int64_t *seg_lo, const int64_t *seg_hi,
                                            const int32_t *targets, int64_t rows, int64_t q_first, int64_t B,
                                            int64_t *out, int64_t ld, int64_t off, const int64_t *pos,
                                            const float *own, int own_rank, kge_stream_t stream)
{
    if (rows < 0 || rows > m || per <= 0 || world < 1 || N <= 0 || N > (int64_t)world * per) return KGE_EINVAL;
    if (own && (own_rank < 0 || own_rank >= world)) return KGE_EINVAL;
    if (B < 0 || off < 0 || ld < off + B || q_first < 0 || q_first + rows > 2 * B) return KGE_EINVAL;
    if (rows == 0) return 0;
    if ((!tiles && !(own && world == 1)) || !true_idx || !seg_lo || !seg_hi || !out) return KGE_EINVAL;
    hipLaunchKernelGGL(filtered_rank_tiles_kernel, dim3(grid1d(rows, 1)), dim3(RB), 0, kge_s(stream), tiles, m, per,
                       world, N, true_idx, seg_lo, seg_hi, targets, rows, q_first, B, out, ld, off, pos, own, own_rank);
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_lp_scores(const kge_lp_desc *d, float *out, int64_t ldo, kge_stream_t stream)
{
    int rc = kge_lp_desc_check(d);
    if (rc) return rc;
    if (d->B == 0 || d->N == 0) return 0;
    if (!out || ldo < d->N) return KGE_EINVAL;
    if (KGE_LP_IS_MFMA(d->mode)) return kge_lp_gemm_run(d, out, ldo, nullptr, nullptr, kge_s(stream));
    return kge_lp_direct_run(d, out, ldo, nullptr, nullptr, kge_s(stream));
}

extern "C" int kge_lp_count_ge(const kge_lp_desc *d, const float *s_true, int32_t *raw_count, kge_stream_t stream)
{
    int rc = kge_lp_desc_check(d);
    if (rc) return rc;
    if (d->B == 0 || d->N == 0) return 0;
    if (!s_true || !raw_count) return KGE_EINVAL;
    if (KGE_LP_IS_MFMA(d->mode)) return kge_lp_gemm_run(d, nullptr, 0, s_true, raw_count, kge_s(stream));
    return kge_lp_direct_run(d, nullptr, 0, s_true, raw_count, kge_s(stream));
}

extern "C" int kge_lp_count_ge_cols(const kge_lp_desc *d, const float *s_true, int32_t *raw_count, const int64_t *rep,
                                    const int32_t *col_q, int64_t n_single_p, const int32_t *members, int64_t n_multi_p,
                                    kge_stream_t stream)
{
    int rc = kge_lp_desc_check(d);
    if (rc) return rc;
    if (d->B == 0 || d->N == 0) return 0;
    return kge_lp_direct_count_cols(d, s_true, raw_count, rep, col_q, n_single_p, members, n_multi_p, kge_s(stream));
}

extern "C" int kge_lp_pair_scores(const kge_lp_desc *d, const int64_t *qi, const int64_t *ci, int64_t P,
                                  float *out, kge_stream_t stream)
{
    int rc = kge_lp_desc_check(d);
    if (rc) return rc;
    if (P < 0) return KGE_EINVAL;
    if (P == 0) return 0;
    if (!ci || !out) return KGE_EINVAL;
    if (KGE_LP_IS_MFMA(d->mode) && d->B > 0 && d->N > 0 && d->B <= INT32_MAX && d->N <= INT32_MAX) {
        const int64_t groups = (P + 63) / 64;
        const int grid = (int)(groups < 256 * 14 ? groups : 256 * 14);
        if (kge_lp_vec4(*d))
            hipLaunchKernelGGL(pair_scores_staged_kernel<true>, dim3(grid), dim3(64), 0, kge_s(stream), *d, qi, ci, P, out);
        else
            hipLaunchKernelGGL(pair_scores_staged_kernel<false>, dim3(grid), dim3(64), 0, kge_s(stream), *d, qi, ci, P, out);
    } else if (!KGE_LP_IS_MFMA(d->mode) && !d->Wq && kge_lp_vec4(*d) && d->B > 0 && d->N > 0 && d->B <= INT32_MAX &&
               d->N <= INT32_MAX) {        // plain L1 / L2 direct: staged rows, the ascending-k chain of lp_pair_score
        const int64_t groups = (P + 63) / 64;
        const int grid = (int)(groups < 256 * 14 ? groups : 256 * 14);
        if (d->mode == KGE_LP_L1_DIRECT)
            hipLaunchKernelGGL((pair_scores_staged_kernel<true, 1>), dim3(grid), dim3(64), 0, kge_s(stream), *d, qi, ci, P, out);
        else
            hipLaunchKernelGGL((pair_scores_staged_kernel<true, 2>), dim3(grid), dim3(64), 0, kge_s(stream), *d, qi, ci, P, out);
    } else {
        hipLaunchKernelGGL(pair_scores_kernel, dim3(grid1d(P, 64)), dim3(64), 0, kge_s(stream), *d, qi, ci, P, out);
    }
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_lp_filter_sub(const kge_lp_desc *d, const float *s_true, const int64_t *true_idx,
                                 const int64_t *seg_lo, const int64_t *seg_hi, const int32_t *targets,
                                 int32_t *sub, int32_t *found, kge_stream_t stream)
{
    int rc = kge_lp_desc_check(d);
    if (rc) return rc;
    if (d->B == 0) return 0;
    if (!s_true || !true_idx || !seg_lo || !seg_hi || !sub || !found) return KGE_EINVAL;
    if (KGE_LP_IS_MFMA(d->mode) && d->N > 0 && d->B <= INT32_MAX && d->N <= INT32_MAX) {
        const int64_t groups = (d->B + 7) / 8;
        const int grid = (int)(groups < 256 * 14 ? groups : 256 * 14);
        if (kge_lp_vec4(*d))
            hipLaunchKernelGGL(filter_sub_staged_kernel<true>, dim3(grid), dim3(64), 0, kge_s(stream), *d, s_true,
                               true_idx, seg_lo, seg_hi, targets, sub, found);
        else
            hipLaunchKernelGGL(filter_sub_staged_kernel<false>, dim3(grid), dim3(64), 0, kge_s(stream), *d, s_true,
                               true_idx, seg_lo, seg_hi, targets, sub, found);
    } else {
        hipLaunchKernelGGL(filter_sub_kernel, dim3(grid1d(d->B, 32)), dim3(256), 0, kge_s(stream), *d, s_true, true_idx,
                           seg_lo, seg_hi, targets, sub, found);
    }
    KGE_CHECK_LAUNCH();
    return 0;
}

static inline int64_t fsub_align(int64_t x) { return (x + 255) & ~(int64_t)255; }
static inline int64_t fsub_nblocks(int64_t B) { return (B + FS_SCAN_T - 1) / FS_SCAN_T; }

extern "C" int64_t kge_lp_filter_sub_ws_bytes(int64_t B, int64_t n_targets)
{
    if (B < 0 || n_targets < 0) return 0;
    return fsub_align(n_targets * 4) * 2 + fsub_align((B + 1) * 8) + fsub_align((fsub_nblocks(B) + 1) * 8);
}

// scoring of the flattened work list + the per-query comparison (shared by the two entry points below)
static int fsub_score_and_count(const kge_lp_desc *d, const float *s_true, const int64_t *true_idx,
                                const int64_t *seg_lo, const int64_t *seg_hi, const int32_t *targets, int64_t n_pairs_max,
                                const int64_t *woff, const int64_t *long_q, int64_t n_long, float *fs, int32_t *sub,
                                int32_t *found, hipStream_t st)
{
    if (n_pairs_max > 0 && d->N > 0) {
        const int64_t groups = (n_pairs_max + 63) / 64;      // upper bound of the flattened work (the exact total is read on the device)
        const int grid = (int)(groups < 256 * 14 ? groups : 256 * 14);
        if (KGE_LP_IS_MFMA(d->mode)) {
            if (kge_lp_vec4(*d))
                hipLaunchKernelGGL((fsub_score_kernel<true, true>), dim3(grid), dim3(64), 0, st, *d, seg_lo, targets, woff, fs);
            else
                hipLaunchKernelGGL((fsub_score_kernel<true, false>), dim3(grid), dim3(64), 0, st, *d, seg_lo, targets, woff, fs);
        } else if (!d->Wq && kge_lp_vec4(*d)) {   // plain L1 / L2 direct: the same cooperative row staging, ascending-k chain
            if (d->mode == KGE_LP_L1_DIRECT)
                hipLaunchKernelGGL((fsub_score_kernel<true, true, 1>), dim3(grid), dim3(64), 0, st, *d, seg_lo, targets, woff, fs);
            else
                hipLaunchKernelGGL((fsub_score_kernel<true, true, 2>), dim3(grid), dim3(64), 0, st, *d, seg_lo, targets, woff, fs);
        } else {
            hipLaunchKernelGGL((fsub_score_kernel<false, false>), dim3(grid), dim3(64), 0, st, *d, seg_lo, targets, woff, fs);
        }
    }
    if (long_q && n_long > 0) {
        const int sb = grid1d(d->B, 4), lb = (int)(n_long < 256 * 16 ? n_long : 256 * 16);
        hipLaunchKernelGGL(fsub_count_both_kernel, dim3(sb + lb), dim3(256), 0, st, *d, s_true, true_idx, seg_lo, seg_hi,
                           targets, fs, long_q, n_long, sb, sub, found);
    } else {
        hipLaunchKernelGGL(fsub_count_kernel, dim3(grid1d(d->B, 4)), dim3(256), 0, st, *d, s_true, true_idx, seg_lo, seg_hi,
                           targets, fs, long_q ? 1 : 0, sub, found);
    }
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_lp_filter_sub_grouped(const kge_lp_desc *d, const float *s_true, const int64_t *true_idx,
                                         const int64_t *seg_lo, const int64_t *seg_hi, const int32_t *targets,
                                         int64_t n_targets, int32_t *sub, int32_t *found, void *ws, int64_t ws_bytes,
                                         kge_stream_t stream)
{
    int rc = kge_lp_desc_check(d);
    if (rc) return rc;
    if (d->B == 0) return 0;
    if (!s_true || !true_idx || !seg_lo || !seg_hi || !sub || !found || n_targets < 0) return KGE_EINVAL;
    if (n_targets > 0 && (!targets || !ws || ws_bytes < kge_lp_filter_sub_ws_bytes(d->B, n_targets))) return KGE_EINVAL;
    if (d->B > INT32_MAX || d->N > INT32_MAX) return KGE_EINVAL;
    hipStream_t st = kge_s(stream);
    char *w8 = static_cast<char *>(ws);
    unsigned *claim = reinterpret_cast<unsigned *>(w8);
    float *fs = reinterpret_cast<float *>(w8 + fsub_align(n_targets * 4));
    int64_t *woff = reinterpret_cast<int64_t *>(w8 + 2 * fsub_align(n_targets * 4));
    int64_t *bsum = reinterpret_cast<int64_t *>(w8 + 2 * fsub_align(n_targets * 4) + fsub_align((d->B + 1) * 8));
    if (n_targets > 0 && d->N > 0) {
        const int nb = (int)fsub_nblocks(d->B);
        hipLaunchKernelGGL(fsub_reset_kernel, dim3(grid1d(d->B, 256)), dim3(256), 0, st, seg_lo, seg_hi, d->B, claim);
        hipLaunchKernelGGL(fsub_claim_kernel, dim3(grid1d(d->B, 256)), dim3(256), 0, st, seg_lo, seg_hi, d->B, claim);
        hipLaunchKernelGGL(fsub_len_kernel, dim3(nb), dim3(FS_SCAN_T), 0, st, seg_lo, seg_hi, d->B, claim, woff, bsum);
        hipLaunchKernelGGL(fsub_bscan_kernel, dim3(1), dim3(FS_SCAN_T), 0, st, bsum, (int64_t)nb, woff + d->B);
        hipLaunchKernelGGL(fsub_off_kernel, dim3(nb), dim3(FS_SCAN_T), 0, st, d->B, woff, bsum);
    }
    return fsub_score_and_count(d, s_true, true_idx, seg_lo, seg_hi, targets, n_targets, woff, nullptr, 0, fs, sub, found, st);
}

extern "C" int kge_lp_filter_sub_planned(const kge_lp_desc *d, const float *s_true, const int64_t *true_idx,
                                         const int64_t *seg_lo, const int64_t *seg_hi, const int32_t *targets,
                                         int64_t n_targets, const int64_t *woff, int64_t n_pairs,
                                         const int64_t *long_q, int64_t n_long, float *fs, int32_t *sub, int32_t *found,
                                         kge_stream_t stream)
{
    int rc = kge_lp_desc_check(d);
    if (rc) return rc;
    if (d->B == 0) return 0;
    if (!s_true || !true_idx || !seg_lo || !seg_hi || !sub || !found || !woff || n_targets < 0 || n_pairs < 0 || n_long < 0)
        return KGE_EINVAL;
    if (n_targets > 0 && (!targets || !fs)) return KGE_EINVAL;
    if (n_long > 0 && !long_q) return KGE_EINVAL;
    if (d->B > INT32_MAX || d->N > INT32_MAX) return KGE_EINVAL;
    return fsub_score_and_count(d, s_true, true_idx, seg_lo, seg_hi, targets, n_pairs, woff, long_q, n_long, fs, sub, found,
                                kge_s(stream));
}

extern "C" int kge_rank_finalize(const int32_t *raw, const int32_t *sub, const int32_t *found, int64_t B,
                                 int64_t *rank, int64_t *filt_rank, kge_stream_t stream)
{
    if (B < 0) return KGE_EINVAL;
    if (B == 0) return 0;
    if (!raw || !sub || !found || !rank || !filt_rank) return KGE_EINVAL;
    hipLaunchKernelGGL(rank_finalize_kernel, dim3(grid1d(B, 256)), dim3(256), 0, kge_s(stream), raw, sub, found, B,
                       rank, filt_rank);
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_rank_finalize_both(const int32_t *raw, const int32_t *sub, const int32_t *found, int64_t B,
                                      int64_t *out, int64_t ld, int64_t off, const int64_t *pos, float *guard,
                                      float *flags, int zero_guard, int64_t *const *out_indirect, kge_stream_t stream)
{
    if (B < 0 || off < 0 || ld < off + B) return KGE_EINVAL;
    if (B == 0) return 0;
    if (!raw || !sub || !found || (!out && !out_indirect) || (flags && !guard)) return KGE_EINVAL;
    hipLaunchKernelGGL(rank_finalize_both_kernel, dim3(grid1d(2 * B, 256)), dim3(256), 0, kge_s(stream), raw, sub,
                       found, B, out, ld, off, pos, guard, flags, (flags && zero_guard) ? 1 : 0, out_indirect);
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_lp_scores_batched(int mode, const float *q, int64_t ldq, const float *cand,
                                     int64_t stride_b, int64_t stride_n, int64_t B, int64_t N, int K,
                                     float *out, int64_t ldo, kge_stream_t stream)
{
    if (mode != KGE_LP_DOT && mode != KGE_LP_L1_DIRECT && mode != KGE_LP_L2_DIRECT) return KGE_EINVAL;
    if (B < 0 || N < 0 || K <= 0 || ldo < N) return KGE_EINVAL;
    if (B == 0 || N == 0) return 0;
    if (!q || !cand || !out) return KGE_EINVAL;
    hipLaunchKernelGGL(lp_batched_kernel, dim3(grid1d(B * N, 4)), dim3(256), 0, kge_s(stream), mode, q, ldq, cand,
                       stride_b, stride_n, B, N, K, out, ldo);
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_topk_chunk(float *scores, int64_t ld, int64_t B, int64_t C, int64_t c_base, int k,
                              const int64_t *seg_lo, const int64_t *seg_hi, const int32_t *targets,
                              const int64_t *ids_in, int64_t ld_ids, int64_t *out_idx, float *out_val, int64_t ldo,
                              int64_t col_off, kge_stream_t stream)
{
    if (B < 0 || C <= 0 || ld < C || k <= 0 || col_off < 0 || ldo < col_off + k) return KGE_EINVAL;
    if (B == 0) return 0;
    if (!scores || !out_idx || !out_val) return KGE_EINVAL;
    if (targets && (!seg_lo || !seg_hi)) return KGE_EINVAL;
    if (ids_in && ld_ids < C) return KGE_EINVAL;
    // k <= 32 (and columns that fit an int): the single-pass register selection; larger k: k passes over the tile
    static const int reg_topk = kge_env_int("KGE_TOPK_REG", 1);
    if (reg_topk && k <= 32 && C < 0x7fffffff) {
        const dim3 grid(grid1d(B, RB / 64)), block(RB);
        if (k <= 8)
            hipLaunchKernelGGL(topk_chunk_reg_kernel<8>, grid, block, 0, kge_s(stream), scores, ld, B, C, c_base, k, seg_lo, seg_hi,
                               targets, ids_in, ld_ids, out_idx, out_val, ldo, col_off);
        else if (k <= 16)
            hipLaunchKernelGGL(topk_chunk_reg_kernel<16>, grid, block, 0, kge_s(stream), scores, ld, B, C, c_base, k, seg_lo, seg_hi,
                               targets, ids_in, ld_ids, out_idx, out_val, ldo, col_off);
        else
            hipLaunchKernelGGL(topk_chunk_reg_kernel<32>, grid, block, 0, kge_s(stream), scores, ld, B, C, c_base, k, seg_lo, seg_hi,
                               targets, ids_in, ld_ids, out_idx, out_val, ldo, col_off);
        KGE_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(topk_chunk_kernel, dim3(grid1d(B, 1)), dim3(RB), 0, kge_s(stream), scores, ld, B, C, c_base, k,
                       seg_lo, seg_hi, targets, ids_in, ld_ids, out_idx, out_val, ldo, col_off);
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_topk(const float *scores, int64_t ld, int64_t B, int64_t N, int k, int64_t *out_idx,
                        float *out_val, kge_stream_t stream)
{
    if (B < 0 || N <= 0 || ld < N || k <= 0) return KGE_EINVAL;
    if (B == 0) return 0;
    if (!scores || !out_idx || !out_val) return KGE_EINVAL;
    hipLaunchKernelGGL(topk_kernel, dim3(grid1d(B, 1)), dim3(RB), 0, kge_s(stream), scores, ld, B, N, k, out_idx, out_val);
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int kge_abi_version(void) { return 32; }

__global__ __launch_bounds__(256) void copy_i64_indirect_kernel(const int64_t *__restrict__ src, int64_t n, int64_t *const *dst_ind)
{
    int64_t *dst = *dst_ind;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

/* dst[0 .. n) = src[0 .. n) with dst = *dst_indirect read on the device when the launch runs (see kge_rank_finalize_both's
 * out_indirect): the packed result of an evaluation whose finalize launches had to scatter (facts processed in another
 * order) leaves for pinned host memory as ONE coalesced pass at the end of the captured graph instead of a copy of its own. */
extern "C" int kge_copy_i64_indirect(const int64_t *src, int64_t n, int64_t *const *dst_indirect, kge_stream_t stream)
{
    if (n < 0 || (n > 0 && (!src || !dst_indirect))) return KGE_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(copy_i64_indirect_kernel, dim3(grid1d(n, 256)), dim3(256), 0, kge_s(stream), src, n, dst_indirect);
    KGE_CHECK_LAUNCH();
    return 0;
}

/* *dev = the device-visible address of pinned (hipHostMalloc / hipHostRegister) host memory -- what a kernel may be handed as
 * kge_rank_finalize_both's *out_indirect.  KGE_EINVAL when the memory is not mapped into the device's address space. */
extern "C" int kge_host_device_pointer(void *host, void **dev)
{
    if (!host || !dev) return KGE_EINVAL;
    void *d = nullptr;
    if (hipHostGetDevicePointer(&d, host, 0) != hipSuccess || !d) { (void)hipGetLastError(); return KGE_EINVAL; }
    *dev = d;
    return 0;
}
extern "C" const char *kge_build_arch(void) { return "gfx950"; }
