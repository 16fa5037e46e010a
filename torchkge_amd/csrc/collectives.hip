// libkge_hip_coll.so: the exchange step of the entity-sharded link-prediction path on RCCL (include/kge_hip_coll.h).
// One process per GPU; collectives are enqueued on the caller's stream and never synchronise.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <string.h>
#include "../../include/kge_hip.h"
#include "../../include/kge_hip_coll.h"

static_assert(sizeof(ncclUniqueId) == KGE_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");

static inline int kge_nccl(ncclResult_t r) { return r == ncclSuccess ? 0 : 1000 + (int)r; }

namespace {
// gathered (P, B, per) -> full (B, ld): float4 along the candidate axis where the shapes allow, one row per block row
__global__ __launch_bounds__(256) void unshard_kernel(const float *__restrict__ g, float *__restrict__ full, int64_t ld,
                                                      int64_t B, int64_t per, int64_t N, int world)
{
    const int64_t total = B * (int64_t)world * per;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / ((int64_t)world * per), col = i - row * (int64_t)world * per;
        const int64_t p = col / per, c = col - p * per;
        if (col < N) full[row * ld + col] = g[(p * B + row) * per + c];
    }
}
} // namespace

extern "C" int kge_comm_unique_id(void *id128)
{
    if (!id128) return KGE_EINVAL;
    ncclUniqueId id;
    const int rc = kge_nccl(ncclGetUniqueId(&id));
    if (rc == 0) memcpy(id128, &id, sizeof(id));
    return rc;
}

extern "C" int kge_comm_init(kge_comm_t *comm, int world, int rank, const void *id128)
{
    if (!comm || !id128 || world < 1 || rank < 0 || rank >= world) return KGE_EINVAL;
    // compiled against /opt/rocm's rccl.h, but the loader may have resolved librccl.so.1 to another copy (under Python:
    // torch's bundled one): refuse a runtime whose major version differs from the header's (ncclUniqueId / comm ABI)
    int v = 0;
    if (ncclGetVersion(&v) != ncclSuccess || v / 10000 != NCCL_VERSION_CODE / 10000) return 1000 + (int)ncclInvalidUsage;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c;
    const int rc = kge_nccl(ncclCommInitRank(&c, world, id, rank));
    if (rc == 0) *comm = (kge_comm_t)c;
    return rc;
}

extern "C" int kge_comm_destroy(kge_comm_t comm)
{
    if (!comm) return KGE_EINVAL;
    return kge_nccl(ncclCommDestroy((ncclComm_t)comm));
}

extern "C" int kge_allgather_scores(kge_comm_t comm, int world, const float *local, float *gathered, float *full,
                                    int64_t ld_full, int64_t B, int64_t n_per, int64_t N, void *stream)
{
    if (!comm || world < 1 || B < 0 || n_per < 0 || N < 0 || N > (int64_t)world * n_per) return KGE_EINVAL;
    if (B == 0 || n_per == 0) return 0;
    if (!local || !gathered || (full && ld_full < N)) return KGE_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int rc = kge_nccl(ncclAllGather(local, gathered, (size_t)(B * n_per), ncclFloat, (ncclComm_t)comm, s));
    if (rc != 0 || !full) return rc;
    const int64_t total = B * (int64_t)world * n_per;
    const int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(unshard_kernel, dim3(grid), dim3(256), 0, s, gathered, full, ld_full, B, n_per, N, world);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

extern "C" int kge_alltoall_scores(kge_comm_t comm, int world, int rank, const float *local, float *recv, int64_t m,
                                   int64_t n_per, int recv_own, void *stream)
{
    if (!comm || world < 1 || rank < 0 || rank >= world || m < 0 || n_per < 0) return KGE_EINVAL;
    if (m == 0 || n_per == 0) return 0;
    if (!local || (!recv && (world > 1 || recv_own))) return KGE_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t blk = (size_t)(m * n_per);
    // the own block never touches the fabric: one device-to-device copy on the same stream -- or none at all when the
    // caller ranks it where it lies (recv_own = 0; kge_filtered_rank_from_tiles' `own`)
    if (recv_own) {
        hipError_t e = hipMemcpyAsync(recv + (size_t)rank * blk, local + (size_t)rank * blk, blk * sizeof(float),
                                      hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return (int)e;
    }
    if (world == 1) return 0;
    int rc = kge_nccl(ncclGroupStart());
    if (rc) return rc;
    for (int j = 0; j < world && rc == 0; ++j) {
        if (j == rank) continue;
        rc = kge_nccl(ncclSend(local + (size_t)j * blk, blk, ncclFloat, j, (ncclComm_t)comm, s));
        if (rc == 0) rc = kge_nccl(ncclRecv(recv + (size_t)j * blk, blk, ncclFloat, j, (ncclComm_t)comm, s));
    }
    const int rc2 = kge_nccl(ncclGroupEnd());
    return rc ? rc : rc2;
}

extern "C" int kge_allreduce_ranks(kge_comm_t comm, int64_t *ranks, int64_t n, void *stream)
{
    if (!comm || n < 0 || (n > 0 && !ranks)) return KGE_EINVAL;
    if (n == 0) return 0;
    return kge_nccl(ncclAllReduce(ranks, ranks, (size_t)n, ncclInt64, ncclSum, (ncclComm_t)comm,
                                  reinterpret_cast<hipStream_t>(stream)));
}

extern "C" int kge_allreduce_counts(kge_comm_t comm, int32_t *counts, int64_t n, void *stream)
{
    if (!comm || n < 0 || (n > 0 && !counts)) return KGE_EINVAL;
    if (n == 0) return 0;
    return kge_nccl(ncclAllReduce(counts, counts, (size_t)n, ncclInt32, ncclSum, (ncclComm_t)comm,
                                  reinterpret_cast<hipStream_t>(stream)));
}

extern "C" int kge_allreduce_sum_f32(kge_comm_t comm, float *x, int64_t n, void *stream)
{
    if (!comm || n < 0 || (n > 0 && !x)) return KGE_EINVAL;
    if (n == 0) return 0;
    return kge_nccl(ncclAllReduce(x, x, (size_t)n, ncclFloat, ncclSum, (ncclComm_t)comm,
                                  reinterpret_cast<hipStream_t>(stream)));
}
