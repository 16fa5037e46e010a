// Ordering of small-integer ids for the backward's sorted gradient reduction (gfx950).
//
// score_triples_bwd writes one gradient row per (triple, operand) and kge_segment_sum_rows adds the rows of equal
// target id with one atomic row-add per run -- which needs the positions of [k0 | k1] ordered by id.  The ids are
// entity / relation indices (< 2^key_bits), so a device radix sort over key_bits bits of (id, position) pairs does it in
// two or three passes: rocPRIM's radix_sort_pairs on 32-bit keys and a counting iterator for the positions.
// (kge_key_hist / kge_key_scatter in score_triples.hip are the counting-sort alternative: their wave-aggregated atomics
// walk up to 64 distinct ids per wavefront one after the other, 113 us per id stream at B = 32768 against ~25 here.)
#include "kge_common.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

namespace {

__global__ void key_pack_kernel(const int64_t *__restrict__ k0, int64_t n0, const int64_t *__restrict__ k1, int64_t n1,
                                unsigned *__restrict__ keys)
{
    const int64_t n = n0 + n1;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x)
        keys[j] = (unsigned)(j < n0 ? k0[j] : k1[j - n0]);
}

__global__ void perm_widen_kernel(const unsigned *__restrict__ pos, int64_t n, int64_t *__restrict__ perm)
{
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x)
        perm[j] = (int64_t)pos[j];
}

inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

size_t sort_temp_bytes(int64_t n, int key_bits)
{
    size_t bytes = 0;
    unsigned *nul = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, nul, nul, rocprim::counting_iterator<unsigned>(0), nul, (size_t)n, 0u,
                                    (unsigned)key_bits);
    return bytes;
}

} // namespace

/* bytes of workspace kge_key_sort needs for n ids of key_bits bits */
extern "C" int64_t kge_key_sort_ws_bytes(int64_t n, int key_bits)
{
    if (n <= 0 || key_bits <= 0 || key_bits > 32 || n > 0x7fffffffll) return 0;
    return (int64_t)(3 * align256((size_t)n * 4) + align256(sort_temp_bytes(n, key_bits)));
}

/* perm[j] = position (into [k0 | k1]) of the j-th id in ascending id order (stable); ids must be < 2^key_bits */
extern "C" int kge_key_sort(const int64_t *k0, int64_t n0, const int64_t *k1, int64_t n1, int key_bits, int64_t *perm,
                            void *ws, int64_t ws_bytes, kge_stream_t stream)
{
    if (n0 < 0 || n1 < 0 || !perm || (n0 > 0 && !k0) || (n1 > 0 && !k1) || key_bits <= 0 || key_bits > 32) return KGE_EINVAL;
    const int64_t n = n0 + n1;
    if (n == 0) return 0;
    if (n > 0x7fffffffll || !ws || ws_bytes < kge_key_sort_ws_bytes(n, key_bits)) return KGE_EINVAL;
    hipStream_t s = kge_s(stream);
    char *w = reinterpret_cast<char *>(ws);
    const size_t a = align256((size_t)n * 4);
    unsigned *keys_in = reinterpret_cast<unsigned *>(w), *keys_out = reinterpret_cast<unsigned *>(w + a),
             *pos_out = reinterpret_cast<unsigned *>(w + 2 * a);
    void *temp = w + 3 * a;
    size_t temp_bytes = (size_t)ws_bytes - 3 * a;
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(key_pack_kernel, dim3(grid), dim3(256), 0, s, k0, n0, k1, n1, keys_in);
    KGE_CHECK_LAUNCH();
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, rocprim::counting_iterator<unsigned>(0),
                                             pos_out, (size_t)n, 0u, (unsigned)key_bits, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(perm_widen_kernel, dim3(grid), dim3(256), 0, s, pos_out, n, perm);
    KGE_CHECK_LAUNCH();
    return 0;
}
