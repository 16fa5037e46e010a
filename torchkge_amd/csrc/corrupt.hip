// K5: the integer half of BernoulliNegativeSampler.corrupt_batch /
// UniformNegativeSampler.corrupt_batch (sampling.py:313-325, :206-221), gfx950.
//
//   neg_heads = heads.repeat(n_neg); neg_tails = tails.repeat(n_neg)
//   neg_heads[mask == 1] = draws_h      (k   values, consumed in position order)
//   neg_tails[mask == 0] = draws_t      (n-k values, consumed in position order)
//
// as a pure function of (heads, tails, mask, draws_h, draws_t): an exclusive
// prefix sum of the mask gives every position its index into the draw arrays.
// Three small HBM-bound launches (block counts, scan of counts, scatter); no
// device->host sync (the reference needs one for mask.sum().item()).
#include "kge_common.h"

namespace {

constexpr int CT = 256;           // threads per block
constexpr int CE = 4;             // elements per thread
constexpr int CB = CT * CE;       // elements per block

__global__ __launch_bounds__(CT) void corrupt_count_kernel(const uint8_t *__restrict__ mask, int64_t n,
                                                           int32_t *block_counts)
{
    __shared__ int sh[CT / 64];
    const int64_t base = (int64_t)blockIdx.x * CB;
    int c = 0;
#pragma unroll
    for (int e = 0; e < CE; ++e) {
        const int64_t j = base + threadIdx.x * CE + e;
        if (j < n) c += mask[j] != 0;
    }
    c = wave_sum_i(c);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int i = 0; i < CT / 64; ++i) t += sh[i];
        block_counts[blockIdx.x] = t;
    }
}

// exclusive scan of block_counts in place (single block, serial carry over chunks)
__global__ __launch_bounds__(CT) void corrupt_scan_kernel(int32_t *block_counts, int64_t nb)
{
    __shared__ int sh[CT];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += CT) {
        const int64_t j = base + threadIdx.x;
        const int v = j < nb ? block_counts[j] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < CT; o <<= 1) { // Hillis-Steele inclusive scan
            int add = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += add;
            __syncthreads();
        }
        const int incl = sh[threadIdx.x];
        if (j < nb) block_counts[j] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == CT - 1) carry += incl;
        __syncthreads();
    }
}

__global__ __launch_bounds__(CT) void corrupt_scatter_kernel(const int64_t *__restrict__ heads,
                                                             const int64_t *__restrict__ tails,
                                                             const uint8_t *__restrict__ mask,
                                                             const int64_t *__restrict__ draws_h,
                                                             const int64_t *__restrict__ draws_t, int64_t B,
                                                             int64_t n, const int32_t *__restrict__ block_base,
                                                             int64_t *neg_heads, int64_t *neg_tails)
{
    __shared__ int sh[CT / 64];
    const int64_t base = (int64_t)blockIdx.x * CB;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int m[CE], c = 0;
#pragma unroll
    for (int e = 0; e < CE; ++e) {
        const int64_t j = base + threadIdx.x * CE + e;
        m[e] = (j < n) ? (mask[j] != 0) : 0;
        c += m[e];
    }
    // exclusive scan of per-thread counts: within wave, then across waves
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) sh[w] = incl;
    __syncthreads();
    int wbase = 0;
    for (int i = 0; i < w; ++i) wbase += sh[i];
    int ph = block_base[blockIdx.x] + wbase + incl - c; // #ones before this thread's first element
#pragma unroll
    for (int e = 0; e < CE; ++e) {
        const int64_t j = base + threadIdx.x * CE + e;
        if (j < n) {
            const int64_t b = j % B;
            if (m[e]) { neg_heads[j] = draws_h[ph]; neg_tails[j] = tails[b]; }
            else      { neg_heads[j] = heads[b];    neg_tails[j] = draws_t[j - ph]; }
            ph += m[e];
        }
    }
}

} // namespace

extern "C" int64_t kge_corrupt_ws_elems(int64_t n) { return (n + CB - 1) / CB + 1; }

extern "C" int kge_corrupt_scatter(const int64_t *heads, const int64_t *tails, const uint8_t *mask,
                                   const int64_t *draws_h, const int64_t *draws_t, int64_t B, int64_t n_neg,
                                   int64_t *neg_heads, int64_t *neg_tails, int32_t *ws, kge_stream_t stream)
{
    if (B < 0 || n_neg < 1) return KGE_EINVAL;
    const int64_t n = B * n_neg;
    if (n == 0) return 0;
    if (!heads || !tails || !mask || !neg_heads || !neg_tails || !ws) return KGE_EINVAL;
    // draws_h / draws_t may legitimately be empty (all-zero / all-one mask)
    const int64_t nb = (n + CB - 1) / CB;
    hipStream_t s = kge_s(stream);
    hipLaunchKernelGGL(corrupt_count_kernel, dim3((unsigned)nb), dim3(CT), 0, s, mask, n, ws);
    KGE_CHECK_LAUNCH();
    hipLaunchKernelGGL(corrupt_scan_kernel, dim3(1), dim3(CT), 0, s, ws, nb);
    KGE_CHECK_LAUNCH();
    hipLaunchKernelGGL(corrupt_scatter_kernel, dim3((unsigned)nb), dim3(CT), 0, s, heads, tails, mask, draws_h,
                       draws_t, B, n, ws, neg_heads, neg_tails);
    KGE_CHECK_LAUNCH();
    return 0;
}
