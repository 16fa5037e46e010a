/*
 * kge_hip_coll.h -- C ABI of libkge_hip_coll.so: the exchange step of the ENTITY-SHARDED link-prediction
 * path (SURVEY.md section 8e / 8b "C1"), for hosts that drive the sharded path without torch.distributed.
 * One process per GPU; rank p holds rows [p*ceil(N/P), ...) of every entity table, scores ITS candidates
 * with the kernels of kge_hip.h (kge_lp_scores / kge_lp_count_ge / kge_lp_filter_sub with c_base = first
 * owned row), and the ranks meet in exactly one of the collectives below (RCCL over xGMI):
 *
 *   kge_allgather_scores   the partial score tiles (B, n_per) of every rank -> (B, N) on every rank: the
 *                          collective BASELINE.json's north_star names; followed by
 *                          kge_filtered_rank_from_scores on the full rows.  Replaces nothing in the
 *                          reference (it has no distributed code): the loop being sharded is
 *                          LinkPredictionEvaluator.evaluate, evaluation.py:263-308.
 *   kge_alltoall_scores    the same scores, but each rank receives only the rows of the queries IT ranks (1/world of
 *                          the all-gather's bytes, ranking work split world ways): the exchange the evaluator's
 *                          exchange='scores' uses since r04; followed by kge_filtered_rank_from_tiles.
 *   kge_allreduce_counts   ranks are sums over candidates: every rank counts on its shard and ONE int32
 *                          SUM all-reduce of the (3, B) partial counts (raw >=-count, filter correction,
 *                          found flag) gives bit-identical ranks -- 12 B per query instead of 4 N.
 *
 * A separate shared object so that libkge_hip.so itself does not depend on librccl.  Same conventions as
 * kge_hip.h: device pointers, a hipStream_t (void*), no allocation, no synchronisation; returns 0, a
 * negative KGE_E* code, or 1000 + ncclResult_t when RCCL reports an error.
 */
#ifndef KGE_HIP_COLL_H
#define KGE_HIP_COLL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *kge_comm_t;   /* ncclComm_t */
#define KGE_UNIQUE_ID_BYTES 128

/* communicator: rank 0 calls kge_comm_unique_id and hands the 128 bytes to the other processes (any channel:
 * a file, MPI, a socket); then every rank calls kge_comm_init with the same id.  One communicator per GPU. */
int kge_comm_unique_id(void *id128);
int kge_comm_init(kge_comm_t *comm, int world, int rank, const void *id128);
int kge_comm_destroy(kge_comm_t comm);

/* local (B, n_per) fp32, row-major, n_per = ceil(N / world) (the last shard zero-padded by the caller)
 *   -> gathered (world, B, n_per) rank-major [scratch] -> full (B, N) row-major, leading dimension ld_full >= N.
 * full == NULL skips the re-layout (the caller ranks on the gathered tiles itself). */
int kge_allgather_scores(kge_comm_t comm, int world, const float *local, float *gathered, float *full, int64_t ld_full,
                         int64_t B, int64_t n_per, int64_t N, void *stream);

/* The score exchange that SCALES (r04): every rank ranks only its own 1/world of the queries.  A tile of `world * m`
 * queries (m = ceil(rows of the tile / world), the caller pads) is scored by every rank against ITS n_per candidates:
 *   local (world * m, n_per) fp32 row-major; rows [j*m, (j+1)*m) are the queries rank j will rank
 *     -> recv (world, m, n_per) rank-major on rank j: tile p = what rank p scored for rank j's m queries, i.e. the
 *        scores of global candidates [p*n_per, (p+1)*n_per).
 * Bytes on the fabric per rank: (world-1)/world of ONE local tile -- 1/world of kge_allgather_scores' -- and the ranking
 * work is split world ways instead of repeated on every rank.  The tile rank entry of kge_hip.h [kge_filtered_rank_from_tiles] ranks straight
 * from `recv` (no re-layout); each rank writes only its queries' columns of the zero-initialised (4, n) result matrix
 * and ONE kge_allreduce_ranks at the end of the evaluation completes it everywhere.  The own block is a device copy --
 * or stays where it is (recv_own = 0) and is ranked from `local` (kge_filtered_rank_from_tiles: own / own_rank). */
int kge_alltoall_scores(kge_comm_t comm, int world, int rank, const float *local, float *recv, int64_t m,
                        int64_t n_per, int recv_own /* 1: also copy the own block into recv[rank]; 0: leave it in `local` */,
                        void *stream);
/* in-place SUM of n int64 (the (4, n_facts) rank matrix: every column written by exactly one rank, 0 elsewhere) */
int kge_allreduce_ranks(kge_comm_t comm, int64_t *ranks, int64_t n, void *stream);

/* in-place SUM over the ranks of n int32 partial counts (the (3, B) block of kge_lp_count_ge / kge_lp_filter_sub) */
int kge_allreduce_counts(kge_comm_t comm, int32_t *counts, int64_t n, void *stream);

/* in-place SUM of n floats: the true scores (owner shard holds the value, the others 0: x + 0 is exact)
 * and the owner-built query rows of row-sharded tables (kge_lp_prep_sharded) */
int kge_allreduce_sum_f32(kge_comm_t comm, float *x, int64_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* KGE_HIP_COLL_H */
