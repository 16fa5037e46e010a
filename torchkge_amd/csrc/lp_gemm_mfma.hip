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
// Tiling: 256 threads = 4 waves (2x2), block tile 128 queries x 128 candidates,
// wave tile 64x64 = 2x2 MFMA 32x32 tiles (64 accumulator VGPRs), BK = 32 staged
// through double-buffered LDS (row stride 36 floats: conflict-free
// ds_read_b128 / ds_write_b128).  Each block walks `tiles_per_block` candidate
// tiles so rank counts are reduced in registers before one atomic per row.
#include "kge_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BM = 128, BN = 128, BK = 32, LDS_LD = BK + 4;
constexpr int NTHREADS = 256;
constexpr int TILE_FLOATS = BM * LDS_LD;                 // one operand tile
constexpr int SMEM_BYTES = 4 * TILE_FLOATS * 4 + 3 * BM * 4; // 2 bufs x (A,B) + row counters, s_true, qn

struct GemmParams {
    kge_lp_desc d;
    float *out;
    int64_t ldo;
    const float *s_true;
    int *raw_count;
    int row_panels, col_tiles, tiles_per_block, col_chunks;
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

// De-interleave 8 consecutive k into [k0,k2,k4,k6 | k1,k3,k5,k7] so that lane
// half h of a wave reads (ds_read_b128) the four k = 2j+h it feeds to MFMA j:
// the accumulation order stays ascending in k.
__device__ __forceinline__ void lds_store8(float *dst, const float (&v)[8])
{
    *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[2], v[4], v[6]);
    *reinterpret_cast<float4 *>(dst + 4) = make_float4(v[1], v[3], v[5], v[7]);
}

template <bool VEC4, bool WRITE, bool COUNT, int MODE>
__global__ __launch_bounds__(NTHREADS, 2) void lp_gemm_kernel(const GemmParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const kge_lp_desc &d = p.d;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int l31 = lane & 31, half = lane >> 5;

    // XCD-aware logical block id: blocks that run on one XCD (bid % 8) get a
    // contiguous range of logical ids = the same candidate chunk, so the T
    // tiles they stream are shared through that XCD's L2.
    const int nblk_grid = gridDim.x;
    const int bid = blockIdx.x;
    const int xq = nblk_grid >> 3, xr = nblk_grid & 7, xcd = bid & 7, loc = bid >> 3;
    const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + loc;
    const int rp = lid % p.row_panels;
    const int cc = lid / p.row_panels;
    const int64_t row0 = (int64_t)rp * BM;
    const int tile_begin = cc * p.tiles_per_block;
    const int tile_end = min(tile_begin + p.tiles_per_block, p.col_tiles);
    const int ntiles = tile_end - tile_begin;
    if (ntiles <= 0) return;

    const int nb0 = (d.K0 + 7) >> 3, nb1 = (d.K1 + 7) >> 3;
    const int steps0 = (nb0 + 3) >> 2, steps1 = (nb1 + 3) >> 2;
    const int S = steps0 + steps1;
    const int G = ntiles * S;

    // staging assignment: chunk = 8 consecutive k of one row
    const int srow0 = tid >> 2, skc = tid & 3; // rows srow0 and srow0+64
    float stA[2][8], stB[2][8];

    auto prefetch = [&](int g) {
        const int ti = g / S, s = g - ti * S;
        const bool seg1 = s >= steps0;
        const int kb0 = (seg1 ? s - steps0 : s) << 2; // first 8-block of this step
        const int K = seg1 ? d.K1 : d.K0;
        const float *A = seg1 ? d.A1 : d.A0;
        const float *T = seg1 ? d.T1 : d.T0;
        const int64_t lda = seg1 ? d.lda1 : d.lda0, ldt = seg1 ? d.ldt1 : d.ldt0;
        const int k = (kb0 + skc) << 3;
        const int64_t col0 = (int64_t)(tile_begin + ti) * BN;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t ra = row0 + srow0 + 64 * j, rb = col0 + srow0 + 64 * j;
            g_load8<VEC4>(A, lda, ra, ra < d.B, k, K, stA[j]);
            g_load8<VEC4>(T, ldt, rb, rb < d.N, k, K, stB[j]);
        }
    };
    auto stage_store = [&](int buf) {
        float *As = smem + buf * 2 * TILE_FLOATS;
        float *Bs = As + TILE_FLOATS;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            lds_store8(As + (srow0 + 64 * j) * LDS_LD + skc * 8, stA[j]);
            lds_store8(Bs + (srow0 + 64 * j) * LDS_LD + skc * 8, stB[j]);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    int cnt[2][16];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) cnt[mt][r] = 0;

    // per-row vectors of this panel live in LDS (read as broadcasts in the epilogue)
    int *rc = reinterpret_cast<int *>(smem + 4 * TILE_FLOATS);
    float *st_s = smem + 4 * TILE_FLOATS + BM;
    float *qn_s = st_s + BM;
    if (tid < BM) {
        const int64_t row = row0 + tid;
        rc[tid] = 0;
        st_s[tid] = (COUNT && row < d.B) ? p.s_true[row] : 0.f;
        qn_s[tid] = (MODE == KGE_LP_L2_EXPAND && row < d.B) ? d.qn[row] : 0.f;
    }

    prefetch(0);
    stage_store(0);
    __syncthreads();

    for (int g = 0; g < G; ++g) {
        const int buf = g & 1;
        if (g + 1 < G) prefetch(g + 1);

        const int ti = g / S, s = g - ti * S;
        const bool seg1 = s >= steps0;
        const int kb0 = (seg1 ? s - steps0 : s) << 2;
        const int nblk = min(4, (seg1 ? nb1 : nb0) - kb0);

        const float *Ab = smem + buf * 2 * TILE_FLOATS + (wr * 64 + l31) * LDS_LD + half * 4;
        const float *Bb = smem + buf * 2 * TILE_FLOATS + TILE_FLOATS + (wc * 64 + l31) * LDS_LD + half * 4;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            if (blk < nblk) {
                const float4 a0 = *reinterpret_cast<const float4 *>(Ab + blk * 8);
                const float4 a1 = *reinterpret_cast<const float4 *>(Ab + 32 * LDS_LD + blk * 8);
                const float4 b0 = *reinterpret_cast<const float4 *>(Bb + blk * 8);
                const float4 b1 = *reinterpret_cast<const float4 *>(Bb + 32 * LDS_LD + blk * 8);
#define KGE_MFMA4(AX, BX)                                                                      \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.AX, b0.BX, acc[0][0], 0, 0, 0);         \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.AX, b1.BX, acc[0][1], 0, 0, 0);         \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.AX, b0.BX, acc[1][0], 0, 0, 0);         \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.AX, b1.BX, acc[1][1], 0, 0, 0);
                KGE_MFMA4(x, x)
                KGE_MFMA4(y, y)
                KGE_MFMA4(z, z)
                KGE_MFMA4(w, w)
#undef KGE_MFMA4
            }
        }

        if (s == S - 1) { // tile finished: epilogue
            const int64_t col0 = (int64_t)(tile_begin + ti) * BN + wc * 64;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int64_t col = col0 + nt * 32 + l31;
                const bool col_ok = col < d.N;
                const float en = (MODE == KGE_LP_L2_EXPAND && col_ok) ? d.en[col] : 0.f;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int lrow = wr * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        const float qn = (MODE == KGE_LP_L2_EXPAND) ? qn_s[lrow] : 0.f;
                        const float sc = lp_epilogue(MODE, acc[mt][nt][r], qn, en);
                        if (WRITE) {
                            const int64_t row = row0 + lrow;
                            if (col_ok && row < d.B) p.out[row * p.ldo + col] = sc;
                        }
                        if (COUNT) cnt[mt][r] += (col_ok && sc >= st_s[lrow]) ? 1 : 0;
                        acc[mt][nt][r] = 0.f;
                    }
                }
            }
        }

        if (g + 1 < G) stage_store(buf ^ 1);
        __syncthreads();
    }

    if (COUNT) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (cnt[mt][r]) atomicAdd(&rc[wr * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half], cnt[mt][r]);
        __syncthreads();
        if (tid < BM) {
            const int64_t row = row0 + tid;
            const int v = rc[tid];
            if (row < d.B && v) atomicAdd(&p.raw_count[row], v);
        }
    }
}

template <bool VEC4, bool WRITE, bool COUNT, int MODE>
int launch(const GemmParams &p, int grid, hipStream_t s)
{
    auto k = lp_gemm_kernel<VEC4, WRITE, COUNT, MODE>;
    static bool attr_set = false; // per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(NTHREADS), SMEM_BYTES, s, p);
    KGE_CHECK_LAUNCH();
    return 0;
}

template <bool WRITE, bool COUNT>
int dispatch(const GemmParams &p, bool vec4, int grid, hipStream_t s)
{
    if (p.d.mode == KGE_LP_DOT)
        return vec4 ? launch<true, WRITE, COUNT, KGE_LP_DOT>(p, grid, s)
                    : launch<false, WRITE, COUNT, KGE_LP_DOT>(p, grid, s);
    return vec4 ? launch<true, WRITE, COUNT, KGE_LP_L2_EXPAND>(p, grid, s)
                : launch<false, WRITE, COUNT, KGE_LP_L2_EXPAND>(p, grid, s);
}

} // namespace

// Internal entry shared by kge_lp_scores / kge_lp_count_ge for the MFMA modes.
int kge_lp_gemm_run(const kge_lp_desc *d, float *out, int64_t ldo, const float *s_true,
                    int32_t *raw_count, hipStream_t s)
{
    if (d->B == 0 || d->N == 0) return 0;
    GemmParams p;
    p.d = *d;
    p.out = out;
    p.ldo = ldo;
    p.s_true = s_true;
    p.raw_count = raw_count;
    p.row_panels = (int)((d->B + BM - 1) / BM);
    p.col_tiles = (int)((d->N + BN - 1) / BN);
    // enough blocks to fill 256 CUs x 2 resident blocks a few times over, but
    // as many candidate tiles per block as that allows (fewer count atomics,
    // longer software pipeline).
    const int target_blocks = 2048;
    int chunks = (target_blocks + p.row_panels - 1) / p.row_panels;
    if (chunks > p.col_tiles) chunks = p.col_tiles;
    if (chunks < 1) chunks = 1;
    p.tiles_per_block = (p.col_tiles + chunks - 1) / chunks;
    p.col_chunks = (p.col_tiles + p.tiles_per_block - 1) / p.tiles_per_block;
    const int grid = p.row_panels * p.col_chunks;

    bool vec4 = (d->K0 % 4 == 0) && (d->lda0 % 4 == 0) && (d->ldt0 % 4 == 0) &&
                kge_aligned16(d->A0) && kge_aligned16(d->T0);
    if (d->K1 > 0)
        vec4 = vec4 && (d->K1 % 4 == 0) && (d->lda1 % 4 == 0) && (d->ldt1 % 4 == 0) &&
               kge_aligned16(d->A1) && kge_aligned16(d->T1);

    const bool w = out != nullptr, c = raw_count != nullptr;
    if (w && c) return dispatch<true, true>(p, vec4, grid, s);
    if (w) return dispatch<true, false>(p, vec4, grid, s);
    if (c) return dispatch<false, true>(p, vec4, grid, s);
    return KGE_EINVAL;
}
