// Device-side construction of the evaluator's static integer structures (gfx950) -- SURVEY 8f N4.
//
//   kge_filter_index_build   the sorted-key CSR of the filter sets (keys, offsets, targets) from the full graph's
//                            (key1, key2, value) triples: what KnowledgeGraph builds as dict-of-sets in a per-fact
//                            Python loop (data_structures.py:386-397) and filter_scores walks row by row
//                            (utils/modeling.py:91-102)
//   kge_filter_plan_build    the grouping of a batch's filter correction (filter_index.FilterPlan): first query of every
//                            distinct filter list, flattened work offsets, the long-list queries
//   kge_column_plan_build    the batch's distinct query rows (filter_index.ColumnPlan): columns with one query, grouped
//                            columns of up to `sets` queries ordered by decreasing size, the query -> column maps
//
// Until r03 these were ATen sort / unique / cumsum / scatter compositions: correct, but the FIRST evaluate() of a
// process paid ~0.5-1 s for loading ATen's sort / unique kernel families.  Here: rocPRIM radix sorts and scans on the
// caller's stream plus a few flag / scatter kernels -- all in this library's own code object.  Bit-identical outputs to
// the ATen compositions (tests/test_gpu_parity.py::test_device_built_index_and_plans_equal_the_torch_builds).
#include "kge_common.h"
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

namespace {

typedef unsigned long long u64;

inline size_t a256(size_t x) { return (x + 255) / 256 * 256; }
inline int grid_for(int64_t n, int cap = 4096) { const int64_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g < cap ? g : cap)); }
inline int bits_of(u64 x) { int b = 0; while (x) { ++b; x >>= 1; } return b < 1 ? 1 : b; }

#define KGE_GRID_STRIDE(i, n) for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// ---- filter index ------------------------------------------------------------------------------------------
// composite sort key: ((key1 * n_key2 + key2) << vbits) | value -- ONE radix sort of 64-bit keys gives the
// lexicographic (key, value) order of FilterIndex.from_triples
__global__ void fi_pack_kernel(const int64_t *__restrict__ k1, const int64_t *__restrict__ k2, const int64_t *__restrict__ v,
                               int64_t n, u64 n_key2, int vbits, u64 *__restrict__ c)
{
    KGE_GRID_STRIDE(j, n) c[j] = (((u64)k1[j] * n_key2 + (u64)k2[j]) << vbits) | (u64)v[j];
}

// flags[j] = (new key) << 32 | (new (key, value) pair)
__global__ void fi_flag_kernel(const u64 *__restrict__ c, int64_t n, int vbits, u64 *__restrict__ flags)
{
    KGE_GRID_STRIDE(j, n) {
        const u64 cur = c[j];
        const u64 prev = j > 0 ? c[j - 1] : ~cur;
        const u64 keep = (j == 0 || cur != prev) ? 1ull : 0ull;
        const u64 nk = (j == 0 || (cur >> vbits) != (prev >> vbits)) ? 1ull : 0ull;
        flags[j] = (nk << 32) | keep;
    }
}

// pos = exclusive scan of flags: low 32 bits = kept pairs before j, high 32 = keys before j
__global__ void fi_scatter_kernel(const u64 *__restrict__ c, const u64 *__restrict__ flags, const u64 *__restrict__ pos,
                                  int64_t n, int vbits, u64 n_key2, int64_t key2_span, int64_t *__restrict__ keys,
                                  int64_t *__restrict__ offsets, int32_t *__restrict__ targets, int64_t *__restrict__ counts)
{
    KGE_GRID_STRIDE(j, n) {
        const u64 f = flags[j], p = pos[j], cur = c[j];
        const int64_t pk = (int64_t)(p & 0xffffffffull), pn = (int64_t)(p >> 32);
        if (f & 1ull) targets[pk] = (int32_t)(cur & ((1ull << vbits) - 1ull));
        if (f >> 32) {
            const u64 ck = cur >> vbits;
            keys[pn] = (int64_t)(ck / n_key2) * key2_span + (int64_t)(ck % n_key2);
            offsets[pn] = pk;
        }
        if (j == n - 1) {
            const int64_t n_kept = pk + (int64_t)(f & 1ull), n_keys = pn + (int64_t)(f >> 32);
            offsets[n_keys] = n_kept;
            counts[0] = n_keys;
            counts[1] = n_kept;
        }
    }
}

__global__ void i64_max3_kernel(const int64_t *__restrict__ a, const int64_t *__restrict__ b, const int64_t *__restrict__ c,
                                int64_t n, u64 *out)
{
    u64 m0 = 0, m1 = 0, m2 = 0;
    KGE_GRID_STRIDE(j, n) {
        m0 = max(m0, (u64)a[j]);      // (a negative id reads as a huge value: the caller rejects it)
        m1 = max(m1, (u64)b[j]);
        m2 = max(m2, (u64)c[j]);
    }
    for (int off = 32; off > 0; off >>= 1) {
        m0 = max(m0, (u64)__shfl_xor((long long)m0, off, 64));
        m1 = max(m1, (u64)__shfl_xor((long long)m1, off, 64));
        m2 = max(m2, (u64)__shfl_xor((long long)m2, off, 64));
    }
    if ((threadIdx.x & 63) == 0) { atomicMax(out, m0); atomicMax(out + 1, m1); atomicMax(out + 2, m2); }
}

size_t sort_keys_temp(int64_t n, int bits)
{
    size_t b = 0;
    u64 *nul = nullptr;
    (void)rocprim::radix_sort_keys(nullptr, b, nul, nul, (size_t)n, 0u, (unsigned)bits);
    return b;
}
size_t scan_u64_temp(int64_t n)
{
    size_t b = 0;
    u64 *nul = nullptr;
    (void)rocprim::exclusive_scan(nullptr, b, nul, nul, (u64)0, (size_t)n, rocprim::plus<u64>());
    return b;
}
size_t sort_pairs_temp(int64_t n, int bits)
{
    size_t b = 0;
    u64 *nul = nullptr;
    unsigned *nv = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, b, nul, nul, rocprim::counting_iterator<unsigned>(0), nv, (size_t)n, 0u, (unsigned)bits);
    return b;
}
size_t sort_small_temp(int64_t n)
{
    size_t b = 0;
    unsigned *nul = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, b, nul, nul, rocprim::counting_iterator<unsigned>(0), nul, (size_t)n, 0u, 8u);
    return b;
}
size_t scan_max_temp(int64_t n)
{
    size_t b = 0;
    int *nul = nullptr;
    (void)rocprim::inclusive_scan(nullptr, b, nul, nul, (size_t)n, rocprim::maximum<int>());
    return b;
}

// ---- filter plan -------------------------------------------------------------------------------------------
__global__ void fp_fill_kernel(int *__restrict__ first, int64_t n) { KGE_GRID_STRIDE(j, n) first[j] = 0x7fffffff; }

__global__ void fp_claim_kernel(const int64_t *__restrict__ seg_lo, const int64_t *__restrict__ seg_hi, int64_t n,
                                int *__restrict__ first)
{
    KGE_GRID_STRIDE(i, n) if (seg_hi[i] > seg_lo[i]) atomicMin(&first[seg_lo[i]], (int)i);
}

// packed[i] = (long list) << 40 | (list length if query i is the first of the batch with that list, else 0)
__global__ void fp_owned_kernel(const int64_t *__restrict__ seg_lo, const int64_t *__restrict__ seg_hi, int64_t n,
                                const int *__restrict__ first, int64_t long_len, u64 *__restrict__ packed)
{
    KGE_GRID_STRIDE(i, n) {
        const int64_t ln = seg_hi[i] - seg_lo[i];
        const bool owner = ln > 0 && first[seg_lo[i]] == (int)i;
        packed[i] = ((ln > long_len ? 1ull : 0ull) << 40) | (owner ? (u64)ln : 0ull);
    }
}

__global__ void fp_emit_kernel(const u64 *__restrict__ packed, const u64 *__restrict__ pos, int64_t n,
                               int64_t *__restrict__ woff, int64_t *__restrict__ long_q, int64_t *__restrict__ counts)
{
    KGE_GRID_STRIDE(i, n) {
        const u64 f = packed[i], p = pos[i];
        woff[i] = (int64_t)(p & ((1ull << 40) - 1ull));
        if (f >> 40) long_q[p >> 40] = i;
        if (i == n - 1) {
            const int64_t n_pairs = (int64_t)((p + f) & ((1ull << 40) - 1ull));
            woff[n] = n_pairs;
            counts[0] = n_pairs;
            counts[1] = (int64_t)((p + f) >> 40);
        }
    }
}

// ---- column plan -------------------------------------------------------------------------------------------
__global__ void cp_key_kernel(const int64_t *__restrict__ h, const int64_t *__restrict__ t, const int64_t *__restrict__ r,
                              int64_t B, u64 n_ent, u64 n_rel, int relation_major, u64 *__restrict__ key)
{
    KGE_GRID_STRIDE(q, 2 * B) {
        const bool tail = q < B;
        const int64_t f = tail ? q : q - B;
        const u64 e = (u64)(tail ? h[f] : t[f]), rr = (u64)r[f];
        const u64 k = relation_major ? rr * n_ent + e : e * n_rel + rr;
        key[q] = tail ? k : k + n_ent * n_rel;          // side bit: the two sides never share rows
    }
}

// start[j] = j at the first position of a key run, else 0 (an inclusive max-scan turns it into "start of my run")
__global__ void cp_head_kernel(const u64 *__restrict__ ks, int64_t n, int *__restrict__ start)
{
    KGE_GRID_STRIDE(j, n) start[j] = (j == 0 || ks[j] != ks[j - 1]) ? (int)j : 0;
}

// flags[j] = (run head) << 32 | (chunk head): a chunk = up to `sets` consecutive queries of one key
__global__ void cp_chunk_flag_kernel(const int *__restrict__ start, int64_t n, int sets, u64 *__restrict__ flags)
{
    KGE_GRID_STRIDE(j, n) {
        const int pos = (int)j - start[j];
        flags[j] = ((pos == 0 ? 1ull : 0ull) << 32) | ((pos % sets == 0) ? 1ull : 0ull);
    }
}

// inc = inclusive scan of flags; chunk[j] = (low 32 bits) - 1; cstart[chunk] = j at chunk heads; totals at the end
__global__ void cp_chunk_id_kernel(const u64 *__restrict__ flags, const u64 *__restrict__ inc, int64_t n,
                                   int *__restrict__ chunk, int *__restrict__ cstart, int64_t *__restrict__ counts)
{
    KGE_GRID_STRIDE(j, n) {
        const int c = (int)(inc[j] & 0xffffffffull) - 1;
        chunk[j] = c;
        if (flags[j] & 1ull) cstart[c] = (int)j;
        if (j == n - 1) {
            cstart[c + 1] = (int)n;
            counts[0] = c + 1;                          // chunks (columns)
            counts[2] = (int64_t)(inc[j] >> 32);        // distinct keys
        }
    }
}

// per chunk c < n_chunks (device scalar counts[0]): single flag, and the sort key of the grouped columns
// (fullest first: sets - size; single / unused entries sort last)
__global__ void cp_size_kernel(const int *__restrict__ cstart, const int64_t *__restrict__ counts, int64_t n, int sets,
                               int *__restrict__ single, unsigned *__restrict__ okey)
{
    const int64_t nc = counts[0];
    KGE_GRID_STRIDE(c, n) {
        int s = 0;
        if (c < nc) s = cstart[c + 1] - cstart[c];
        single[c] = (s == 1) ? 1 : 0;
        okey[c] = (s >= 2) ? (unsigned)(sets - s) : (unsigned)(sets + 1);
    }
}

__global__ void cp_rank_kernel(const unsigned *__restrict__ order2, int64_t n, int *__restrict__ col2,
                               const int *__restrict__ single, const int *__restrict__ col1, int64_t *__restrict__ counts)
{
    KGE_GRID_STRIDE(k, n) {
        col2[order2[k]] = (int)k;
        if (k == n - 1) counts[1] = (int64_t)col1[k] + single[k];      // single-query columns
    }
}

__global__ void cp_prefill_kernel(int32_t *__restrict__ col_q, int64_t n1p, int32_t *__restrict__ members, int64_t n2ps,
                                  int64_t *__restrict__ rep, int64_t nrep)
{
    KGE_GRID_STRIDE(j, max(max(n1p, n2ps), nrep)) {
        if (j < n1p) col_q[j] = -1;
        if (j < n2ps) members[j] = -1;
        if (j < nrep) rep[j] = 0;
    }
}

__global__ void cp_scatter_kernel(const unsigned *__restrict__ qs, const int *__restrict__ chunk, const int *__restrict__ cstart,
                                  const int *__restrict__ single, const int *__restrict__ col1, const int *__restrict__ col2,
                                  int64_t n, int sets, int64_t n_single_p, int32_t *__restrict__ col_q,
                                  int32_t *__restrict__ members, int32_t *__restrict__ qs_row, int64_t *__restrict__ col_of_q,
                                  int64_t *__restrict__ rep)
{
    KGE_GRID_STRIDE(j, n) {
        const int q = (int)qs[j], c = chunk[j], slot = (int)j - cstart[c];
        int64_t row;
        if (single[c]) {
            row = col1[c];
            col_q[row] = q;
        } else {
            const int64_t cm = col2[c];
            members[cm * sets + slot] = q;
            row = n_single_p + cm;
        }
        col_of_q[q] = row;
        qs_row[q] = slot == 0 ? (int32_t)row : -1;
        if (slot == 0) rep[row] = q;
    }
}

struct ColWs {
    u64 *key_in, *key_out, *flags, *inc;
    unsigned *qs, *okey_in, *okey_out, *order2;
    int *start, *chunk, *cstart, *single, *col1, *col2;
    void *temp;
    size_t temp_bytes, total;
};

ColWs col_ws(void *ws, int64_t n, int bits)
{
    ColWs w;
    char *p = reinterpret_cast<char *>(ws);
    size_t off = 0;
    auto take = [&](size_t bytes) { char *q = p ? p + off : nullptr; off += a256(bytes); return q; };
    w.key_in = (u64 *)take(n * 8); w.key_out = (u64 *)take(n * 8); w.flags = (u64 *)take(n * 8); w.inc = (u64 *)take(n * 8);
    w.qs = (unsigned *)take(n * 4); w.okey_in = (unsigned *)take(n * 4); w.okey_out = (unsigned *)take(n * 4);
    w.order2 = (unsigned *)take(n * 4);
    w.start = (int *)take(n * 4); w.chunk = (int *)take(n * 4); w.cstart = (int *)take((n + 1) * 4);
    w.single = (int *)take(n * 4); w.col1 = (int *)take(n * 4); w.col2 = (int *)take(n * 4);
    size_t t = sort_pairs_temp(n, bits);
    t = t > sort_small_temp(n) ? t : sort_small_temp(n);
    t = t > scan_u64_temp(n) ? t : scan_u64_temp(n);
    t = t > scan_max_temp(n) ? t : scan_max_temp(n);
    w.temp_bytes = t;
    w.temp = take(t);
    w.total = off;
    return w;
}

} // namespace

/* maxima of three id arrays in one launch (out: 3 uint64 device scalars, caller-zeroed) */
extern "C" int kge_i64_max3(const int64_t *a, const int64_t *b, const int64_t *c, int64_t n, int64_t *out, kge_stream_t stream)
{
    if (n < 0 || !out) return KGE_EINVAL;
    if (n == 0) return 0;
    if (!a || !b || !c) return KGE_EINVAL;
    hipLaunchKernelGGL(i64_max3_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, kge_s(stream), a, b, c, n,
                       reinterpret_cast<u64 *>(out));
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t kge_filter_index_ws_bytes(int64_t n)
{
    if (n <= 0 || n > 0x7fffffffll) return 0;
    const size_t t = sort_keys_temp(n, 64) > scan_u64_temp(n) ? sort_keys_temp(n, 64) : scan_u64_temp(n);
    return (int64_t)(4 * a256((size_t)n * 8) + a256(t));
}

extern "C" int kge_filter_index_build(const int64_t *key1, const int64_t *key2, const int64_t *values, int64_t n,
                                      int64_t n_key1, int64_t n_key2, int64_t n_values, int64_t key2_span,
                                      int64_t *keys, int64_t *offsets, int32_t *targets, int64_t *counts, void *ws,
                                      int64_t ws_bytes, kge_stream_t stream)
{
    if (n < 0 || n > 0x7fffffffll || n_key1 <= 0 || n_key2 <= 0 || n_values <= 0 || key2_span < n_key2) return KGE_EINVAL;
    if (n_values > 0x7fffffffll) return KGE_EINVAL;
    if (n == 0) return 0;
    if (!key1 || !key2 || !values || !keys || !offsets || !targets || !counts || !ws) return KGE_EINVAL;
    const int vbits = bits_of((u64)n_values - 1);
    const int kbits = bits_of((u64)n_key1 * (u64)n_key2 - 1);
    if (vbits + kbits > 64 || (u64)n_key1 > (~0ull) / (u64)n_key2) return KGE_EUNSUPPORTED;   // (the caller falls back to two stable sorts)
    if (ws_bytes < kge_filter_index_ws_bytes(n)) return KGE_EINVAL;
    hipStream_t s = kge_s(stream);
    char *w = reinterpret_cast<char *>(ws);
    const size_t a = a256((size_t)n * 8);
    u64 *c_in = (u64 *)w, *c_out = (u64 *)(w + a), *flags = (u64 *)(w + 2 * a), *pos = (u64 *)(w + 3 * a);
    void *temp = w + 4 * a;
    size_t temp_bytes = (size_t)ws_bytes - 4 * a;
    const int g = grid_for(n);
    hipLaunchKernelGGL(fi_pack_kernel, dim3(g), dim3(256), 0, s, key1, key2, values, n, (u64)n_key2, vbits, c_in);
    KGE_CHECK_LAUNCH();
    hipError_t e = rocprim::radix_sort_keys(temp, temp_bytes, c_in, c_out, (size_t)n, 0u, (unsigned)(vbits + kbits), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(fi_flag_kernel, dim3(g), dim3(256), 0, s, c_out, n, vbits, flags);
    KGE_CHECK_LAUNCH();
    temp_bytes = (size_t)ws_bytes - 4 * a;
    e = rocprim::exclusive_scan(temp, temp_bytes, flags, pos, (u64)0, (size_t)n, rocprim::plus<u64>(), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(fi_scatter_kernel, dim3(g), dim3(256), 0, s, c_out, flags, pos, n, vbits, (u64)n_key2, key2_span, keys,
                       offsets, targets, counts);
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t kge_filter_plan_ws_bytes(int64_t n, int64_t n_targets)
{
    if (n <= 0 || n_targets < 0) return 0;
    return (int64_t)(a256((size_t)(n_targets + 1) * 4) + 2 * a256((size_t)n * 8) + a256(scan_u64_temp(n)));
}

extern "C" int kge_filter_plan_build(const int64_t *seg_lo, const int64_t *seg_hi, int64_t n, int64_t n_targets,
                                     int64_t long_len, int64_t *woff, int64_t *long_q, int64_t *counts, void *ws,
                                     int64_t ws_bytes, kge_stream_t stream)
{
    if (n < 0 || n > 0x7fffffffll || n_targets < 0 || n_targets >= (1ll << 39) || long_len < 0) return KGE_EINVAL;
    if (n == 0) return 0;
    if (!seg_lo || !seg_hi || !woff || !long_q || !counts || !ws || ws_bytes < kge_filter_plan_ws_bytes(n, n_targets))
        return KGE_EINVAL;
    hipStream_t s = kge_s(stream);
    char *w = reinterpret_cast<char *>(ws);
    const size_t af = a256((size_t)(n_targets + 1) * 4), ap = a256((size_t)n * 8);
    int *first = (int *)w;
    u64 *packed = (u64 *)(w + af), *pos = (u64 *)(w + af + ap);
    void *temp = w + af + 2 * ap;
    size_t temp_bytes = (size_t)ws_bytes - af - 2 * ap;
    hipLaunchKernelGGL(fp_fill_kernel, dim3(grid_for(n_targets + 1)), dim3(256), 0, s, first, n_targets + 1);
    KGE_CHECK_LAUNCH();
    const int g = grid_for(n);
    hipLaunchKernelGGL(fp_claim_kernel, dim3(g), dim3(256), 0, s, seg_lo, seg_hi, n, first);
    KGE_CHECK_LAUNCH();
    hipLaunchKernelGGL(fp_owned_kernel, dim3(g), dim3(256), 0, s, seg_lo, seg_hi, n, first, long_len, packed);
    KGE_CHECK_LAUNCH();
    hipError_t e = rocprim::exclusive_scan(temp, temp_bytes, packed, pos, (u64)0, (size_t)n, rocprim::plus<u64>(), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(fp_emit_kernel, dim3(g), dim3(256), 0, s, packed, pos, n, woff, long_q, counts);
    KGE_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t kge_column_plan_ws_bytes(int64_t n_queries, int64_t n_ent, int64_t n_rel)
{
    if (n_queries <= 0 || n_queries > 0x7fffffffll || n_ent <= 0 || n_rel <= 0) return 0;
    return (int64_t)col_ws(nullptr, n_queries, bits_of(2ull * (u64)n_ent * (u64)n_rel - 1)).total;
}

/* Phase A: everything up to the column numbering; counts (3 int64, device) = [columns, single-query columns, distinct keys].
 * The host reads counts (it sizes col_q / members by the padded column counts), then calls phase B on the same ws. */
extern "C" int kge_column_plan_build(const int64_t *h, const int64_t *t, const int64_t *r, int64_t B, int64_t n_ent,
                                     int64_t n_rel, int sets, int relation_major, int64_t *counts, void *ws,
                                     int64_t ws_bytes, kge_stream_t stream)
{
    if (B < 0 || 2 * B > 0x7fffffffll || n_ent <= 0 || n_rel <= 0 || sets < 1 || sets > 64) return KGE_EINVAL;
    if (B == 0) return 0;
    if ((u64)n_ent > (~0ull) / (2ull * (u64)n_rel)) return KGE_EUNSUPPORTED;
    const int64_t n = 2 * B;
    if (!h || !t || !r || !counts || !ws || ws_bytes < kge_column_plan_ws_bytes(n, n_ent, n_rel)) return KGE_EINVAL;
    const int bits = bits_of(2ull * (u64)n_ent * (u64)n_rel - 1);
    ColWs w = col_ws(ws, n, bits);
    hipStream_t s = kge_s(stream);
    const int g = grid_for(n);
    hipLaunchKernelGGL(cp_key_kernel, dim3(g), dim3(256), 0, s, h, t, r, B, (u64)n_ent, (u64)n_rel, relation_major, w.key_in);
    KGE_CHECK_LAUNCH();
    size_t tb = w.temp_bytes;
    hipError_t e = rocprim::radix_sort_pairs(w.temp, tb, w.key_in, w.key_out, rocprim::counting_iterator<unsigned>(0), w.qs,
                                             (size_t)n, 0u, (unsigned)bits, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(cp_head_kernel, dim3(g), dim3(256), 0, s, w.key_out, n, w.start);
    KGE_CHECK_LAUNCH();
    tb = w.temp_bytes;
    int *run_start = w.col2;        // (col2 itself is written last, by cp_rank_kernel)
    e = rocprim::inclusive_scan(w.temp, tb, w.start, run_start, (size_t)n, rocprim::maximum<int>(), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(cp_chunk_flag_kernel, dim3(g), dim3(256), 0, s, run_start, n, sets, w.flags);
    KGE_CHECK_LAUNCH();
    tb = w.temp_bytes;
    e = rocprim::inclusive_scan(w.temp, tb, w.flags, w.inc, (size_t)n, rocprim::plus<u64>(), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(cp_chunk_id_kernel, dim3(g), dim3(256), 0, s, w.flags, w.inc, n, w.chunk, w.cstart, counts);
    KGE_CHECK_LAUNCH();
    hipLaunchKernelGGL(cp_size_kernel, dim3(g), dim3(256), 0, s, w.cstart, counts, n, sets, w.single, w.okey_in);
    KGE_CHECK_LAUNCH();
    tb = w.temp_bytes;
    e = rocprim::exclusive_scan(w.temp, tb, w.single, w.col1, 0, (size_t)n, rocprim::plus<int>(), s);
    if (e != hipSuccess) return (int)e;
    tb = w.temp_bytes;
    e = rocprim::radix_sort_pairs(w.temp, tb, w.okey_in, w.okey_out, rocprim::counting_iterator<unsigned>(0), w.order2,
                                  (size_t)n, 0u, (unsigned)bits_of((u64)sets + 1), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(cp_rank_kernel, dim3(g), dim3(256), 0, s, w.order2, n, w.col2, w.single, w.col1, counts);
    KGE_CHECK_LAUNCH();
    return 0;
}

/* Phase B: the outputs of filter_index.ColumnPlan.  col_q (n_single_p, int32), members (n_multi_p * sets, int32),
 * qs_row (2B int32), col_of_q (2B int64), rep (max(n_single_p + n_multi_p, 1) int64). */
extern "C" int kge_column_plan_emit(int64_t B, int64_t n_ent, int64_t n_rel, int sets, int64_t n_single_p,
                                    int64_t n_multi_p, int32_t *col_q, int32_t *members, int32_t *qs_row,
                                    int64_t *col_of_q, int64_t *rep, void *ws, int64_t ws_bytes, kge_stream_t stream)
{
    if (B < 0 || 2 * B > 0x7fffffffll || n_ent <= 0 || n_rel <= 0 || sets < 1 || n_single_p < 0 || n_multi_p < 0)
        return KGE_EINVAL;
    if (B == 0) return 0;
    const int64_t n = 2 * B;
    if (!col_q || !members || !qs_row || !col_of_q || !rep || !ws || ws_bytes < kge_column_plan_ws_bytes(n, n_ent, n_rel))
        return KGE_EINVAL;
    ColWs w = col_ws(ws, n, bits_of(2ull * (u64)n_ent * (u64)n_rel - 1));
    hipStream_t s = kge_s(stream);
    const int64_t nrep = n_single_p + n_multi_p > 0 ? n_single_p + n_multi_p : 1;
    const int64_t n1p = n_single_p > 0 ? n_single_p : 1, n2ps = (n_multi_p > 0 ? n_multi_p : 1) * sets;
    hipLaunchKernelGGL(cp_prefill_kernel, dim3(grid_for(nrep > n2ps ? (nrep > n1p ? nrep : n1p) : (n2ps > n1p ? n2ps : n1p))),
                       dim3(256), 0, s, col_q, n1p, members, n2ps, rep, nrep);
    KGE_CHECK_LAUNCH();
    hipLaunchKernelGGL(cp_scatter_kernel, dim3(grid_for(n)), dim3(256), 0, s, w.qs, w.chunk, w.cstart, w.single, w.col1, w.col2,
                       n, sets, n_single_p, col_q, members, qs_row, col_of_q, rep);
    KGE_CHECK_LAUNCH();
    return 0;
}
