/*
 * kge_hip.h -- C ABI of libkge_hip.so, the MI355X (gfx950) engine behind the
 * torchkge link-prediction / scoring hot path.
 *
 * The reference (torchkge v0.17.7) is pure Python and has no FFI; its drop-in
 * boundary is the Python class API.  This header is the
 * C-ABI a binding for that API sits on: plain pointers and sizes, no torch
 * types.  Every entry point
 *   - takes DEVICE pointers (fp32 row-major tables, int64 index vectors),
 *   - launches on the given hipStream_t (passed as void*) and returns without
 *     synchronising,
 *   - never allocates, frees or retains caller memory,
 *   - returns 0 on success, a negative KGE_E* code for argument errors, or a
 *     positive hipError_t if a launch failed.
 *
 * Reference interface each entry replaces (paths relative to the reference):
 *   kge_score_triples            Model.scoring_function
 *                                  models/translation.py:69-81 (TransE), :183-206 (TransH),
 *                                  :538-568 (TransD); models/bilinear.py:188-199 (DistMult),
 *                                  :460-473 (ComplEx)
 *   kge_score_triples_bwd        autograd of the above (Model.forward under Trainer,
 *                                  models/interfaces.py:65-82, utils/training.py:156-167)
 *   kge_lp_prep                  Model.inference_prepare_candidates
 *                                  translation.py:105-125, :234-258, :603-627;
 *                                  bilinear.py:247-267, :530-556
 *   kge_row_sqnorm/kge_row_dot/  the per-entity part of evaluate_projections
 *   kge_lp_scores (as a GEMM)      translation.py:260-284, :629-652 (never builds (R,N,d))
 *   kge_lp_scores                Model.inference_scoring_function (entity candidates)
 *                                  models/interfaces.py:240-260; bilinear.py:224-240, :501-522
 *   kge_lp_scores_batched        same, for an arbitrary materialised (B,N,d) candidate tensor
 *   kge_get_rank                 utils/operations.py:37-61
 *   kge_filter_lookup +          utils/modeling.py:53-102 (get_true_targets / filter_scores)
 *   kge_filter_scores
 *   kge_filtered_rank_from_scores  get_rank(scores) + get_rank(filter_scores(scores))
 *   kge_filtered_rank_from_tiles   evaluation.py:292-300 (_from_tiles: on the rank-major score tiles of the
 *                                  entity-sharded path's score all-to-all, kge_hip_coll.h)
 *   kge_lp_pair_scores,          the fused form of evaluation.py:290-300 that never
 *   kge_lp_count_ge,               materialises the (B,N) score matrix
 *   kge_lp_filter_sub,
 *   kge_rank_finalize
 *   kge_lp_split_rows / _count /   the same rank count through the certified f16 MFMA prefilter + exact recheck
 *   _recheck, kge_lp_hi_rows,      (three products per k16 unit, or ONE on planar hi operands: kge_split_args.level);
 *   kge_lp_query_pipeline          kge_lp_sad_* for TransE-L1
 *   kge_filter_index_build,      the dict-of-sets filter structures of data_structures.py:386-397 as a device CSR, and
 *   kge_filter_plan_build,         the evaluator's static per-batch plans, built by rocPRIM sorts / scans
 *   kge_column_plan_build / _emit
 *   kge_topk                     the sort + top_k slice of inference.py:148-150, :243-245
 *   kge_corrupt_scatter          BernoulliNegativeSampler.corrupt_batch / Uniform...
 *                                  sampling.py:313-325 and :206-221 (the integer scatter)
 *   kge_gather_rows              nn.Embedding lookups of the above
 *   kge_normalize_rows           Model.normalize_parameters (translation.py:83-90 etc.)
 */
#ifndef KGE_HIP_H
#define KGE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *kge_stream_t; /* hipStream_t */

#define KGE_EINVAL (-1)   /* bad argument (null pointer, bad size / kind) */
#define KGE_EALIGN (-2)   /* pointer or leading dimension not aligned as required */
#define KGE_EUNSUPPORTED (-3) /* valid arguments outside what the kernel handles (the caller takes its general path) */

/* model kinds (kge_score_triples, kge_lp_prep) */
enum {
    KGE_TRANSE_L1 = 0,
    KGE_TRANSE_L2 = 1,
    KGE_TRANSH = 2,
    KGE_TRANSD = 3,
    KGE_DISTMULT = 4,
    KGE_COMPLEX = 5
};

/* which entity is replaced by the candidates */
enum {
    KGE_SIDE_TAIL = 0,   /* candidates replace the tail: query = f(h, r)            */
    KGE_SIDE_HEAD = 1,   /* candidates replace the head: query = f(t, r)            */
    KGE_SIDE_PROJ_H = 2, /* kge_lp_prep only: Q0 = projected head, no relation term */
    KGE_SIDE_PROJ_T = 3, /* kge_lp_prep only: Q0 = projected tail, no relation term */
    KGE_SIDE_BOTH = 4    /* kge_lp_prep / kge_lp_query_pipeline: both sides of the B facts as 2B queries, [0,B) the
                          * tail-side queries f(h,r), [B,2B) the head-side queries f(t,r); outputs have 2B rows */
};

/* kge_ewise ops (query-side elementwise algebra of inference_scoring_function) */
enum {
    KGE_EW_ADD = 0,      /* a + b         */
    KGE_EW_SUB = 1,      /* a - b         */
    KGE_EW_MUL = 2,      /* a * b         */
    KGE_EW_MULSUB = 3,   /* a*b - c*d     */
    KGE_EW_MULADD = 4    /* a*b + c*d     */
};

/* all-candidates scorer modes (kge_lp_desc.mode) */
enum {
    KGE_LP_DOT = 0,        /* s = A0.T0 (+ A1.T1)              bilinear, fp32 MFMA            */
    KGE_LP_L2_EXPAND = 1,  /* s = -max(qn+en-2*A0.T0, 0)       TransE L2 as fp32 MFMA GEMM     */
    KGE_LP_L1_DIRECT = 2,  /* s = -sum_k |A0 - T0 (+a*Wq)|     broadcast-subtract, fp32 VALU   */
    KGE_LP_L2_DIRECT = 3,  /* s = -sum_k (A0 - T0 (+a*Wq))^2   broadcast-subtract, fp32 VALU   */
    KGE_LP_L2_PROJH = 4,   /* TransH: ||u - e + x w||^2 expanded around u.e, x = X[r_i,c]   fp32 MFMA + gather */
    KGE_LP_L2_PROJD = 5    /* TransD: ||u - e' - y_c w||^2 expanded around u.e', g = X[r_i,c] fp32 MFMA + gather */
};
/* the MFMA (GEMM-shaped) modes: everything but the two DIRECT ones */
#define KGE_LP_IS_MFMA(mode) ((mode) <= KGE_LP_L2_EXPAND || (mode) >= KGE_LP_L2_PROJH)

/*
 * Descriptor of one all-candidates scoring problem: B queries against the N
 * table rows [c_base, c_base+N) of the (possibly sharded) entity table.
 *   score(i, c) for local candidate c in [0,N):
 *     DOT        : chain(A0[i], T0[c], K0) then chain(A1[i], T1[c], K1)
 *     L2_EXPAND  : -fmaxf(fmaf(-2, chain(A0[i],T0[c],K0), qn[i] + en[c]), 0)
 *     L2_PROJH   : v = fmaf(-2, chain(A0[i],T0[c],K0), qn[i] + en[c]); x = X[r_idx[i], c];
 *                  v = fmaf(x, fmaf(x, z_i, p_i), v); s = -fmaxf(v, 0)
 *                  (TransH candidates e_c - x w_i with x = e_c.w_{r_i}: p_i = 2 u_i.w_i, z_i = ||w_i||^2 - 2)
 *     L2_PROJD   : v as above; g = X[r_idx[i], c]; y = yc[c];
 *                  v = fmaf(y, fmaf(y, z_i, fmaf(2, g, p_i)), v); s = -fmaxf(v, 0)
 *                  (TransD candidates e'_c + y_c w_i, g = e'_c.w_{r_i}: p_i = -2 u_i.w_i, z_i = ||w_i||^2)
 *                  with (p_i, z_i) = Wq[i*ldw + {0,1}] and X[r, c] = scal[r*scal_ld + c]
 *     L1/L2_DIRECT: diff_k = A0[i,k] - T0[c,k]; if Wq: diff_k = fmaf(a, Wq[i,k], diff_k)
 *                   with a = scal[c*scal_ld + (scal_ld > 1 ? r_idx[i] : 0)];
 *                   L2: acc = fmaf(diff_k, diff_k, acc), k ascending;
 *                   L1: acc += (|diff_k| + |diff_k+1|) + (|diff_k+2| + |diff_k+3|) per aligned group of four k
 *                       (absent k >= K0 count as 0);  s = -acc
 *   chain(x, y, K): ONE accumulator, acc = fmaf(x[k], y[k], acc), k visiting
 *   the 8-blocks of [0,K) in ascending order and, inside each 8-block, the
 *   offsets 0,4,1,5,2,6,3,7 (absent k >= K skipped) -- exactly the sequence in
 *   which the tile kernel feeds v_mfma_f32_32x32x2_f32 (an fp32 fmaf chain) --
 *   so every kernel that scores a pair (tile kernels, pair kernel, filter
 *   kernel) produces bit-identical scores, and oracle/kge_oracle.c reproduces
 *   them bit for bit on the CPU.  L2_DIRECT accumulates in plain ascending k, L1_DIRECT
 *   one add per aligned 4-group of k (the group summed as a tree: the same instruction
 *   count, a quarter of the roundings at full magnitude).
 */
typedef struct kge_lp_desc {
    int32_t mode;
    int32_t K0, K1;            /* inner dims of segment 0 / 1 (K1 = 0: unused) */
    int32_t reserved;
    int64_t B, N;              /* queries, local candidates */
    int64_t c_base;            /* global id of local candidate 0 */
    const float *A0; int64_t lda0;   /* (B,K0) query matrix */
    const float *T0; int64_t ldt0;   /* (N,K0) candidate table shard */
    const float *A1; int64_t lda1;   /* (B,K1) or NULL */
    const float *T1; int64_t ldt1;   /* (N,K1) or NULL */
    const float *qn;           /* (B) ||A0[i]||^2   (L2_EXPAND) */
    const float *en;           /* (N) ||T0[c]||^2   (L2_EXPAND) */
    const float *Wq; int64_t ldw;    /* DIRECT: (B,K0) per-query direction or NULL; PROJ: (B,2) = (p_i, z_i) */
    const float *scal; int64_t scal_ld; /* DIRECT + Wq: (N,scal_ld) per-candidate scalars; PROJ: X (n_rel, scal_ld >= N) */
    const int64_t *r_idx;      /* (B) DIRECT: column of scal used by query i (when scal_ld > 1); PROJ: row of X */
    const float *yc;           /* (N) per-candidate scalar y_c (L2_PROJD) */
} kge_lp_desc;

/* ---- K1: fused gather + normalise + score (scoring_function) ------------- */
/* tables: TransE/DistMult {E,R}; TransH {E,R,W}; TransD {E,R,Ep,Rp}; ComplEx {Ere,Eim,Rre,Rim}.
 * d_ent = row length of entity tables, d_rel = row length of relation tables
 * (equal except TransD, which needs d_ent >= d_rel).  Tables are contiguous. */
int kge_score_triples(int kind, const float *t0, const float *t1, const float *t2, const float *t3,
                      int d_ent, int d_rel, const int64_t *h, const int64_t *t, const int64_t *r,
                      int64_t B, float *out, kge_stream_t stream);

/* backward of d(sum_i go[i]*score_i)/d(table).
 * rows == NULL: added into g0..g3 (same shapes as t0..t3, caller-zeroed or accumulating) with one
 *   fp32 atomic per (triple, dimension).
 * rows != NULL: no atomics; triple i stores its gradient rows at rows[(stream*B + i)*rows_ld ...]
 *   (rows_ld >= d_ent), streams = (target table, index):
 *     TransE / DistMult: 0 (g0,h) 1 (g0,t) 2 (g1,r)        TransH: 0 (g0,h) 1 (g0,t) 2 (g1,r) 3 (g2,r)
 *     ComplEx: 0 (g0,h) 1 (g0,t) 2 (g1,h) 3 (g1,t) 4 (g2,r) 5 (g3,r)
 *     TransD:  0 (g0,h) 1 (g0,t) 2 (g2,h) 3 (g2,t) 4 (g1,r) 5 (g3,r)
 *   and the caller reduces them per target table with kge_segment_sum_rows. */
int kge_score_triples_bwd(int kind, const float *t0, const float *t1, const float *t2,
                          const float *t3, int d_ent, int d_rel, const int64_t *h,
                          const int64_t *t, const int64_t *r, int64_t B, const float *go,
                          float *g0, float *g1, float *g2, float *g3, float *rows, int64_t rows_ld,
                          kge_stream_t stream);
/* Row j of `rows` (j in [0, n0+n1)) belongs to target row key(j) = j < n0 ? k0[j] : k1[j-n0]; perm lists
 * the rows grouped by key.  out[key] += sum of its rows: runs are summed in registers per 32-entry chunk
 * of perm and flushed with one atomic row-add each. */
int kge_segment_sum_rows(const float *rows, int64_t ld, int d, const int64_t *k0, int64_t n0, const int64_t *k1,
                         int64_t n1, const int64_t *perm, float *out, int64_t out_ld, kge_stream_t stream);
/* counting sort of the small-integer keys [k0 | k1] (ids < n_keys): kge_key_hist fills hist (caller-zeroed,
 * n_keys int32), the caller turns it into exclusive offsets (int64), kge_key_scatter writes perm (cursor:
 * n_keys int32, caller-zeroed).  Order inside a key group is unspecified. */
int kge_key_hist(const int64_t *k0, int64_t n0, const int64_t *k1, int64_t n1, int32_t *hist, kge_stream_t stream);
int kge_key_scatter(const int64_t *k0, int64_t n0, const int64_t *k1, int64_t n1, const int64_t *offsets,
                    int32_t *cursor, int64_t *perm, kge_stream_t stream);

/* The same ordering by a device radix sort over key_bits bits of (id, position) pairs (ids < 2^key_bits, n0 + n1 < 2^31):
 * perm[j] = position in [k0 | k1] of the j-th id in ascending order, stable.  ws: kge_key_sort_ws_bytes(n0 + n1, key_bits)
 * bytes of device scratch.  Replaces the hist / cumsum / scatter triple in the backward of Model.scoring_function
 * (torchkge/models/interfaces.py:65-82 through autograd) for large batches. */
int64_t kge_key_sort_ws_bytes(int64_t n, int key_bits);
int kge_key_sort(const int64_t *k0, int64_t n0, const int64_t *k1, int64_t n1, int key_bits, int64_t *perm, void *ws,
                 int64_t ws_bytes, kge_stream_t stream);

/* ---- link-prediction query preparation ----------------------------------- */
/* Fills the query-side operands of a kge_lp_desc for one side:
 *   TransE   Q0 = E[h]+R[r] | E[t]-R[r];  qn = chain ||Q0||^2 (optional)
 *   TransH   Q0 = p_r(h)+R[r] | p_r(t)-R[r], p_r(e) = E[e]-(E[e].W[r])W[r];  Wq = W[r]
 *   TransD   Q0 = (s_h Rp[r]+E[h,:dr])+R[r] | (s_t Rp[r]+E[t,:dr])-R[r], s_e = Ep[e].E[e]; Wq = Rp[r]
 *   DistMult Q0 = E[h]*R[r] | R[r]*E[t]
 *   ComplEx  Q0 = re_h re_r - im_h im_r | re_r re_t + im_r im_t
 *            Q1 = re_h im_r + im_h re_r | re_r im_t - im_r re_t
 * side KGE_SIDE_PROJ_H / _PROJ_T: Q0 = the (projected) head / tail embedding alone
 * (E[e]; p_r(e) for TransH/TransD; ComplEx: Q0 = re, Q1 = im), Wq still filled.
 * Q0/Q1/Wq are (B, d_rel) contiguous. */
int kge_lp_prep(int kind, int side, const float *t0, const float *t1, const float *t2,
                const float *t3, int d_ent, int d_rel, const int64_t *h, const int64_t *t,
                const int64_t *r, int64_t B, float *Q0, float *Q1, float *qn, float *Wq,
                kge_stream_t stream);

/* The same with ROW-SHARDED entity tables (one shard per GPU): t0 (and t2 of TransD, t1 of
 * ComplEx) hold only the rows [ent_lo, ent_lo + ent_n) of the entity tables, h / t stay GLOBAL ids.  The rank
 * that owns the query's entity writes the row, every other rank writes zeros: an all-reduce SUM of Q0 (/Q1)
 * over the shards (x + 0 is exact) gives every rank the full query matrix.  Relation tables are replicated,
 * so Wq is complete on every rank.  qn (if requested) is only meaningful after that sum.  ent_n < 0: whole tables. */
int kge_lp_prep_sharded(int kind, int side, const float *t0, const float *t1, const float *t2,
                        const float *t3, int d_ent, int d_rel, const int64_t *h, const int64_t *t,
                        const int64_t *r, int64_t B, int64_t ent_lo, int64_t ent_n, float *Q0, float *Q1,
                        float *qn, float *Wq, kge_stream_t stream);
/* The same launch for the projection models (KGE_TRANSH / KGE_TRANSD, unsharded tables or replicas) with the PLANAR f16 hi
 * operand of the query rows riding along (r05): Qh = what kge_lp_hi_rows(Q0, is_query = 1, aug_mode 2) builds in a launch of
 * its own -- [Bp][hi_units_p][32 bytes], scale 2^12, augmentation columns 1, 1, rows [nq, Bp) zero -- and q_dn2[i] =
 * ||q_i - hi(q_i)||^2 (optional).  Qh = NULL: exactly kge_lp_prep_sharded.  Wq may be NULL for these kinds. */
int kge_lp_prep_hi(int kind, int side, const float *t0, const float *t1, const float *t2, const float *t3, int d_ent,
                   int d_rel, const int64_t *h, const int64_t *t, const int64_t *r, int64_t B, int64_t ent_lo,
                   int64_t ent_n, float *Q0, float *Q1, float *qn, float *Wq, void *Qh, int hi_units_p, int64_t Bp,
                   float *q_dn2, kge_stream_t stream);

/* Relation candidates of the projection models (relation prediction, `entities=False`;
 * TransH translation.py:252-256, TransD :621-626, scored as interfaces.py:261-272):
 *   out[i*ldo + rho] = -|| p_rho(h_i) + R[rho] - p_rho(t_i) ||^2  for every relation rho < n_rel,
 * p_rho as in kge_lp_prep (raw tables; Wt = W for KGE_TRANSH, Rp for KGE_TRANSD; Ep only TransD).
 * Replaces two (b, n_rel, d) gathers from the reference's (n_rel, n_ent, d) projection cache. */
int kge_relation_scores_proj(int kind, const float *E, const float *R, const float *Wt, const float *Ep,
                             int d_ent, int d_rel, const int64_t *h, const int64_t *t, int64_t B,
                             int64_t n_rel, float *out, int64_t ldo, kge_stream_t stream);

/* Per-query scalars of the projection modes (KGE_LP_L2_PROJH / _PROJD) in ONE launch: qn[i] = ||Q[i]||^2 and
 * pz[i] = (scale * (Q[i] . W[r_idx[i]]), ||W[r_idx[i]]||^2 + z_add) -- kge_lp_desc.Wq of those modes (TransH: scale 2,
 * z_add -2, W = the normal vectors; TransD: scale -2, z_add 0, W = the relation projection vectors) -- by the chains of
 * kge_row_sqnorm / kge_row_dot (same bits as the six launches it replaces); *qmax_io = max(*qmax_io, max qn).
 * KGE_EUNSUPPORTED unless K % 4 == 0, ldq % 4 == 0, ldw % 4 == 0 and Q, W are 16-byte aligned. */
int kge_proj_query_stats(const float *Q, int64_t ldq, const float *W, int64_t ldw, const int64_t *r_idx, int64_t rows,
                         int K, float scale, float z_add, float *qn, float *pz, float *qmax_io,
                         int32_t *zero_i32 /* optional (ABI 32): zero_n int32 zeroed by this launch -- the batch's rank
                         counters, as kge_lp_query_pipeline does */, int64_t zero_n, kge_stream_t stream);
/* out = op(a, b[, c, d]) elementwise over n floats (separate mul / add roundings,
 * as the reference's (re_h * re_r - im_h * im_r) etc., bilinear.py:514-522). */
int kge_ewise(int op, const float *a, const float *b, const float *c, const float *d, int64_t n,
              float *out, kge_stream_t stream);

/* out[i] = chain_k X[i,k]^2; when max_io != NULL also *max_io = max(*max_io, max_i out[i])
 * (device scalar, atomically -- the evaluator's "was the norm expansion safe" guard) */
int kge_row_sqnorm(const float *X, int64_t ld, int64_t rows, int K, float *out, float *max_io,
                   kge_stream_t stream);
/* The same norms in ANY summation order (a few ulps of a K-term sum away from the chain): for values that only bound an
 * error or fix an operand scale -- the DOT modes of the split prefilter -- never for a value that enters a score. */
int kge_row_sqnorm_any_order(const float *X, int64_t ld, int64_t rows, int K, float *out, float *max_io,
                             kge_stream_t stream);
/* out[i] = scale * chain_k X[i,k]*Y[i,k] */
int kge_row_dot(const float *X, const float *Y, int64_t ld, int64_t rows, int K, float scale,
                float *out, kge_stream_t stream);
/* out[i,:] = X[idx[i],:]  (rows of length K, table ld) */
int kge_gather_rows(const float *X, int64_t ld, const int64_t *idx, int64_t rows, int K,
                    float *out, kge_stream_t stream);
/* X[i,:] /= max(||X[i,:]||_2, 1e-12)   (torch.nn.functional.normalize, p=2, dim=1) */
int kge_normalize_rows(float *X, int64_t ld, int64_t rows, int K, kge_stream_t stream);

/* ---- all-candidates scoring ----------------------------------------------- */
/* out[i*ldo + c] = score(i,c), i<B, c<N (local candidates). */
int kge_lp_scores(const kge_lp_desc *d, float *out, int64_t ldo, kge_stream_t stream);

/* out[p] = score(qi[p], ci[p] - c_base) or 0 if ci[p] outside the shard.
 * qi == NULL means qi[p] = p.  ci holds GLOBAL candidate ids. */
int kge_lp_pair_scores(const kge_lp_desc *d, const int64_t *qi, const int64_t *ci, int64_t P,
                       float *out, kge_stream_t stream);

/* raw_count[i] += #{c in shard : score(i,c) >= s_true[i]}  (int32 atomics; the
 * caller zeroes raw_count).  No score is written to memory. */
int kge_lp_count_ge(const kge_lp_desc *d, const float *s_true, int32_t *raw_count,
                    kge_stream_t stream);
/* The same counts for a plain KGE_LP_L2_DIRECT problem (no rank-1 term, 16-byte aligned operands, K0 % 4 == 0) swept over
 * query COLUMNS (see kge_split_args.col_q): column r is scored with query row rep[r]; columns [0, n_single_p) carry one
 * query each (col_q), the next n_multi_p up to kge_lp_split_group_sets() (members); raw_count stays indexed by query. */
int kge_lp_count_ge_cols(const kge_lp_desc *d, const float *s_true, int32_t *raw_count, const int64_t *rep,
                         const int32_t *col_q, int64_t n_single_p, const int32_t *members, int64_t n_multi_p,
                         kge_stream_t stream);

/* per query i with filter segment targets[seg_lo[i]:seg_hi[i]) (GLOBAL ids):
 *   sub[i]   = sum over c in segment, c != true_idx[i], c in shard of
 *              [score(i,c) >= s_true[i]] - [-inf >= s_true[i]]
 *   found[i] = 1 if true_idx[i] occurs in the segment AND lies in this shard. */
int kge_lp_filter_sub(const kge_lp_desc *d, const float *s_true, const int64_t *true_idx,
                      const int64_t *seg_lo, const int64_t *seg_hi, const int32_t *targets,
                      int32_t *sub, int32_t *found, kge_stream_t stream);

/* The same sub / found for a whole batch of link-prediction queries, load-balanced for heavy-tailed
 * filter lists.  PRECONDITION: queries whose segments start at the same position have identical
 * query rows (true for link prediction: the key (h, r) / (t, r) fixes both the filter list,
 * utils/modeling.py:78, and the query vector) -- each distinct list is then scored once, all
 * (list, target) pairs flattened over the grid, and every query only compares its true score with
 * its list's scores; total scoring work <= n_targets whatever the skew.  n_targets = length of
 * `targets`; ws = kge_lp_filter_sub_ws_bytes(B, n_targets) bytes of scratch. */
int64_t kge_lp_filter_sub_ws_bytes(int64_t B, int64_t n_targets);
int kge_lp_filter_sub_grouped(const kge_lp_desc *d, const float *s_true, const int64_t *true_idx,
                              const int64_t *seg_lo, const int64_t *seg_hi, const int32_t *targets,
                              int64_t n_targets, int32_t *sub, int32_t *found, void *ws, int64_t ws_bytes,
                              kge_stream_t stream);

/* The same with a PLAN: the grouping of a batch (which query scores which list, the flattened work
 * offsets) depends only on the test facts and the filter index, not on the model, so an evaluator builds it
 * once per batch and reuses it for every evaluate() call.
 *   woff[B+1]: exclusive prefix sum over the queries of (segment length if the query is the FIRST of the batch
 *              with that segment, else 0); n_pairs = woff[B] (host copy: sizes the launch)
 *   long_q[n_long]: the queries whose segment is longer than 512 entries (one 256-thread block compares
 *              each; may be NULL when n_long == 0: every query is then taken by a wavefront)
 *   fs: n_targets floats of scratch (scores per target position). */
int kge_lp_filter_sub_planned(const kge_lp_desc *d, const float *s_true, const int64_t *true_idx,
                              const int64_t *seg_lo, const int64_t *seg_hi, const int32_t *targets,
                              int64_t n_targets, const int64_t *woff, int64_t n_pairs,
                              const int64_t *long_q, int64_t n_long, float *fs, int32_t *sub, int32_t *found,
                              kge_stream_t stream);

/* rank[i] = raw[i]; filt_rank[i] = found[i] ? raw[i] - sub[i] : raw[i]  (int64 out) */
int kge_rank_finalize(const int32_t *raw, const int32_t *sub, const int32_t *found, int64_t B,
                      int64_t *rank, int64_t *filt_rank, kge_stream_t stream);
/* The same for a 2B-query batch (tail-side queries first), written straight into a (4, ld) result
 * matrix whose rows are [head raw, tail raw, head filtered, tail filtered] (the four rank vectors of
 * evaluation.py:294-300), at columns off .. off + B - 1 (or pos[off .. off + B - 1]). */
int kge_rank_finalize_both(const int32_t *raw, const int32_t *sub, const int32_t *found, int64_t B,
                           int64_t *out, int64_t ld, int64_t off, const int64_t *pos /* optional: fact j of the
                           evaluation goes to column pos[j] (facts processed in another order, e.g. sorted by relation) */,
                           float *guard /* optional, with flags: 8 device floats [max ||q||^2, max ||e||^2, overflow,
                           .., .., .., re-scored pairs, ..] */,
                           float *flags /* optional, 3 floats: flags[0] = guard[0] + guard[1], flags[1] = guard[2], flags[2] =
                           guard[6] (the evaluation's guard decisions and the split prefilter's re-scored pair count, written
                           behind the ranks by the same launch) */,
                           int zero_guard /* with flags: the 8 guard floats are zeroed after they were read -- the next
                           evaluation finds them clean and needs no fill of its own */,
                           int64_t *const *out_indirect /* optional (ABI 31): the result matrix of THIS launch is *out_indirect,
                           read on the device when the launch runs -- a replayed hipGraph writes where the host pointed it
                           before the replay; `out` is then ignored and the flags go behind the four rows (out + 4 ld).
                           Meant for PINNED HOST memory (kge_host_device_pointer): the ranks reach the host without a copy
                           of their own */,
                           kge_stream_t stream);
/* *dev = the device-visible address of pinned host memory (hipHostMalloc / hipHostRegister; torch's pinned tensors);
 * KGE_EINVAL when it is not mapped into the device's address space. */
int kge_host_device_pointer(void *host, void **dev);
/* dst[0 .. n) = src[0 .. n), dst = *dst_indirect read on the device when the launch runs: the packed (4 n + 2) int64 result of
 * an evaluation whose finalize launches scatter (pos != NULL) leaves for pinned host memory in one coalesced pass inside the
 * captured graph. */
int kge_copy_i64_indirect(const int64_t *src, int64_t n, int64_t *const *dst_indirect, kge_stream_t stream);

/* generic: every query has its own candidate matrix cand[i] (N,K) at
 * cand + i*stride_b (stride_b = 0: shared), rows at stride_n.
 * mode DOT / L1_DIRECT / L2_DIRECT; score = f(q[i], cand[i,c]). */
int kge_lp_scores_batched(int mode, const float *q, int64_t ldq, const float *cand,
                          int64_t stride_b, int64_t stride_n, int64_t B, int64_t N, int K,
                          float *out, int64_t ldo, kge_stream_t stream);

/* ---- rank / filter on a materialised score matrix ------------------------- */
int kge_get_rank(const float *scores, int64_t ld, const int64_t *true_idx, int64_t B, int64_t N,
                 int low_values, int64_t *rank, kge_stream_t stream);

/* binary-search key1[i]*n_key2 + key2[i] in the sorted key array of a filter
 * index; seg_lo/seg_hi = its target segment, or 0/0 if the key is absent. */
int kge_filter_lookup(const int64_t *keys, int64_t n_keys, const int64_t *offsets,
                      const int64_t *key1, const int64_t *key2, int64_t n_key2, int64_t B,
                      int64_t *seg_lo, int64_t *seg_hi, kge_stream_t stream);
/* Both sides of B facts in one launch, for a 2B-query batch (tail-side queries first):
 *   i <  B: key (h[i], r[i]) in the tail index,      true_idx[i]     = t[i]
 *   i >= B: key (t[i-B], r[i-B]) in the head index,  true_idx[i]     = h[i-B], segment + targets_base_h
 * (the caller keeps the two target arrays concatenated: head targets start at targets_base_h). */
int kge_filter_lookup_both(const int64_t *keys_t, int64_t n_keys_t, const int64_t *offsets_t,
                           const int64_t *keys_h, int64_t n_keys_h, const int64_t *offsets_h,
                           int64_t targets_base_h, const int64_t *h, const int64_t *t, const int64_t *r,
                           int64_t n_key2, int64_t B, int64_t *seg_lo, int64_t *seg_hi, int64_t *true_idx,
                           kge_stream_t stream);

/* in place: scores[i, c] = -inf for c in segment i, c != true_idx[i]; rows whose
 * segment is empty or does not contain true_idx[i] are left untouched.
 * true_idx == NULL: every listed target is masked (get_true_targets with
 * true_idx=None, utils/modeling.py:83-84, used by inference.py:146, :241). */
int kge_filter_scores(float *scores, int64_t ld, const int64_t *true_idx, const int64_t *seg_lo,
                      const int64_t *seg_hi, const int32_t *targets, int64_t B, int64_t N,
                      kge_stream_t stream);

/* rank and filtered rank from a materialised (B,N) matrix in one pass. */
int kge_filtered_rank_from_scores(const float *scores, int64_t ld, const int64_t *true_idx,
                                  const int64_t *seg_lo, const int64_t *seg_hi,
                                  const int32_t *targets, int64_t B, int64_t N, int64_t *rank,
                                  int64_t *filt_rank, kge_stream_t stream);

/* The same for score rows that arrive as `world` RANK-MAJOR tiles -- the receive buffer of kge_alltoall_scores
 * (kge_hip_coll.h): tile p = (m, per) fp32 holds the scores of GLOBAL candidates [p*per, min(N, (p+1)*per)), row i of
 * every tile belongs to query q_first + i of a 2B-query batch (tail-side queries first).  true_idx / seg_lo / seg_hi
 * point at the first of the `rows` (<= m) queries; the true score is read from the tiles like the reference reads it
 * from the score matrix (evaluation.py:291-300).  No (B, N) re-layout: every tile row is streamed where it lies.
 * Ranks go straight into the (4, ld) result matrix of kge_rank_finalize_both (same off / pos meaning).
 * own != NULL: tile own_rank is read from `own` (m, per) -- the block of the caller's local score tile that holds its
 * own queries -- instead of from `tiles`, so the exchange never has to copy it (kge_alltoall_scores with recv_own = 0). */
int kge_filtered_rank_from_tiles(const float *tiles, int64_t m, int64_t per, int world, int64_t N,
                                 const int64_t *true_idx, const int64_t *seg_lo, const int64_t *seg_hi,
                                 const int32_t *targets, int64_t rows, int64_t q_first, int64_t B,
                                 int64_t *out, int64_t ld, int64_t off, const int64_t *pos,
                                 const float *own, int own_rank, kge_stream_t stream);

/* ---- device-side construction of the evaluator's static integer structures (index_build.hip) ----
 * rocPRIM radix sorts / scans on the caller's stream + flag and scatter kernels: no ATen sort / unique.  Every builder
 * writes its element counts to a device array; the host reads them once (the only sync) to size the views. */
/* maxima of three id arrays (out: 3 uint64 device scalars, caller-zeroed) */
int kge_i64_max3(const int64_t *a, const int64_t *b, const int64_t *c, int64_t n, int64_t *out, kge_stream_t stream);
/* The filter index: index[(key1_j, key2_j)] contains values_j (duplicates collapse), as a sorted-key CSR -- what the
 * reference keeps as dict_of_heads / dict_of_tails (data_structures.py:386-397).  key1 < n_key1, key2 < n_key2 <= key2_span,
 * values < n_values.  keys (capacity n, int64: key1 * key2_span + key2, ascending), offsets (n + 1), targets (n, int32,
 * ascending inside a key), counts[0] = number of keys, counts[1] = number of targets.  KGE_EUNSUPPORTED when
 * bits(n_key1 * n_key2) + bits(n_values) > 64 (one 64-bit radix sort carries key and value). */
int64_t kge_filter_index_ws_bytes(int64_t n);
int kge_filter_index_build(const int64_t *key1, const int64_t *key2, const int64_t *values, int64_t n,
                           int64_t n_key1, int64_t n_key2, int64_t n_values, int64_t key2_span,
                           int64_t *keys, int64_t *offsets, int32_t *targets, int64_t *counts, void *ws,
                           int64_t ws_bytes, kge_stream_t stream);
/* The plan of kge_lp_filter_sub_planned from a batch's filter segments: woff (n + 1), long_q (capacity n: the queries whose
 * segment is longer than long_len, ascending), counts[0] = n_pairs = woff[n], counts[1] = n_long. */
int64_t kge_filter_plan_ws_bytes(int64_t n, int64_t n_targets);
int kge_filter_plan_build(const int64_t *seg_lo, const int64_t *seg_hi, int64_t n, int64_t n_targets,
                          int64_t long_len, int64_t *woff, int64_t *long_q, int64_t *counts, void *ws,
                          int64_t ws_bytes, kge_stream_t stream);
/* The query COLUMNS of a both-sides batch (kge_split_args.col_q / members): queries that share their key -- (h, r) on the
 * tail side, (t, r) on the head side -- share the query row.  Keys in ascending order (relation_major: r-major), a key with m
 * queries gives ceil(m / sets) columns of up to `sets` queries; columns with ONE query come first (in key order), the
 * grouped ones follow in order of decreasing size.  _build: phase A, counts = [columns, single-query columns, distinct
 * keys]; the host pads the two column counts to the query panel and calls _emit on the SAME workspace:
 * col_q (max(n_single_p, 1) int32, -1 = padding), members (max(n_multi_p, 1) * sets int32, -1 = unused), qs_row (2B int32:
 * the column a query's split row is written to, -1 for all but the first query of a column), col_of_q (2B int64),
 * rep (max(n_single_p + n_multi_p, 1) int64: the query that provides a column's row). */
int64_t kge_column_plan_ws_bytes(int64_t n_queries, int64_t n_ent, int64_t n_rel);
int kge_column_plan_build(const int64_t *h, const int64_t *t, const int64_t *r, int64_t B, int64_t n_ent,
                          int64_t n_rel, int sets, int relation_major, int64_t *counts, void *ws,
                          int64_t ws_bytes, kge_stream_t stream);
int kge_column_plan_emit(int64_t B, int64_t n_ent, int64_t n_rel, int sets, int64_t n_single_p,
                         int64_t n_multi_p, int32_t *col_q, int32_t *members, int32_t *qs_row,
                         int64_t *col_of_q, int64_t *rep, void *ws, int64_t ws_bytes, kge_stream_t stream);

/* top-k per row in the order (score descending, index ascending); replaces the
 * full `scores.sort(descending=True)` + slice of EntityInference /
 * RelationInference (inference.py:148-150, :243-245).  out_idx/out_val: (B,k). */
int kge_topk(const float *scores, int64_t ld, int64_t B, int64_t N, int k, int64_t *out_idx,
             float *out_val, kge_stream_t stream);

/* Top-k over a candidate set processed TILE BY TILE (and shard by shard): `scores` (B, C) holds the scores of candidates
 * [c_base, c_base + C) -- scratch, modified in place: the known targets of row i's filter segment that fall into the
 * tile are masked first when `targets` != NULL (filter_scores with true_idx = None, inference.py:146, :241).  Writes the
 * tile's k best of every row as (score, GLOBAL id) into columns [col_off, col_off + k) of the (B, ldo) outputs.
 * `ids_in` != NULL: merge mode -- column c stands for candidate ids_in[i*ld_ids + c] (< 0: padding, never selected);
 * c_base is ignored.  Replaces the (b, N) score matrix + full sort of EntityInference.evaluate (inference.py:216-250)
 * by O(b * C) scratch; per-shard lists of a row-sharded model merge the same way. */
int kge_topk_chunk(float *scores, int64_t ld, int64_t B, int64_t C, int64_t c_base, int k,
                   const int64_t *seg_lo, const int64_t *seg_hi, const int32_t *targets,
                   const int64_t *ids_in, int64_t ld_ids, int64_t *out_idx, float *out_val, int64_t ldo,
                   int64_t col_off, kge_stream_t stream);

/* ---- negative sampling ----------------------------------------------------- */
/* neg_heads/neg_tails (B*n_neg): position j (batch element j % B): mask[j] != 0
 * -> head := draws_h[#ones before j], tail kept; else tail := draws_t[#zeros
 * before j], head kept.  ws: int32 workspace of kge_corrupt_ws_elems(B*n_neg). */
int64_t kge_corrupt_ws_elems(int64_t n);
int kge_corrupt_scatter(const int64_t *heads, const int64_t *tails, const uint8_t *mask,
                        const int64_t *draws_h, const int64_t *draws_t, int64_t B, int64_t n_neg,
                        int64_t *neg_heads, int64_t *neg_tails, int32_t *ws, kge_stream_t stream);

/* library / build info */
/* ---- certified f16-split prefilter of the fused rank count ------------------------
 * (torchkge_amd/csrc/lp_split_mfma.hip has the error analysis.)  kge_lp_split_count + kge_lp_split_recheck leave in raw_count exactly
 * what kge_lp_count_ge leaves there: >= 99.9% of the (query, candidate) pairs are
 * decided by an f16 hi/lo-split MFMA product with a rigorous error band, the pairs
 * inside the band are re-scored by the exact fp32 chain.
 *
 * A split operand is [rows_p][units_p] cells of 64 bytes (see the .hip file);
 * kge_lp_split_units / kge_lp_split_rows_padded give its dimensions.
 * Modes: KGE_LP_L2_EXPAND (TransE-L2), KGE_LP_DOT (DistMult, ComplEx: K0 + K1 columns) and the
 * projection modes KGE_LP_L2_PROJH/_PROJD (TransH, TransD: split operands as for L2_EXPAND; desc.scal
 * must then be 16-byte aligned with scal_ld % 4 == 0 and readable up to the 256-padded candidate
 * edge, desc.yc likewise).
 * *overflow is set to 1.0f if more than cap pairs (or more than 2048 in one 256 x 192
 * tile) fell inside the band -- raw_count is then invalid and the caller must redo the
 * count with kge_lp_count_ge.  The squared-norm maxima are the device scalars that
 * kge_row_sqnorm accumulates through max_io. */
typedef struct kge_split_args {
    const void *Qs, *Es;          /* split operands (kge_lp_split_rows) */
    const float *qn0, *qn1;       /* KGE_LP_DOT: ||q_i||^2 per K-segment (qn1 NULL when K1 == 0); L2_EXPAND: desc.qn is used */
    const float *qmax0, *qmax1;   /* KGE_LP_DOT: device scalars >= max_i qn0 / qn1 (they fix the query operand's scale) */
    const float *emax0, *emax1;   /* device scalars >= max_c ||e_c||^2 per K-segment (emax1 NULL when K1 == 0) */
    const float *xabsmax, *yabsmax; /* L2_PROJH/_PROJD: device scalars >= max |X[r,c]| (xabsmax NULL: the per-query bound
                                     * ||w_i|| max||e|| is used) and >= max |yc[c]| (PROJD), see kge_absmax */
    int32_t accum_model;          /* 0: accumulation error bounded for ANY fp32 adder (2 * 2^-24 per product); 1: the
                                   * measured behaviour of gfx950, valid only if kge_mfma_f16_selftest() returned 1 */
    float eps_scale;              /* multiplies the error band (1.0 = the proven bound; tests shrink it) */
    int32_t thr_ready;            /* 1: thr is already filled and *list_count zeroed (kge_lp_query_pipeline) */
    const float *q_cell_ss;       /* optional: the queries' cell sums (kge_lp_split_rows) and ...                      */
    const float *e2pref;          /* ... the candidates' prefix squared-norm maxima (kge_lp_split_prefix_max): the band  */
                                  /* then charges each k16 unit with its own magnitude bound (NULL: full ||q|| ||e||)   */
    float *thr;                   /* scratch: 4 * kge_lp_split_rows_padded(B, 1) floats */
    int32_t *list;                /* scratch: cap x 2 int32 (query, local candidate) */
    int32_t cap;
    int32_t *list_count;          /* device int32 */
    float *overflow;              /* device float, set to 1.0f on overflow (see above) */
    /* COLUMNS instead of queries (optional; plain thresholds only).  Queries of a link-prediction batch that share
     * their key -- (h, r) on the tail side, (t, r) on the head side -- share the query ROW and differ only in the true
     * entity, i.e. in their thresholds: Qs then holds one row per column, n_single_p rows whose column carries ONE
     * query (col_q[column] = query id, < 0: padding) followed by n_multi_p rows whose column carries up to
     * kge_lp_split_group_sets() queries (members[column * sets + j], < 0: unused); both counts are multiples of the
     * query panel (kge_lp_split_rows_padded).  The matrix-core sweep runs once per column, the compare once per
     * query; thr, raw_count and the uncertain-pair list stay indexed by QUERY.  Both NULL: column == query. */
    const int32_t *col_q;
    int64_t n_single_p;
    const int32_t *members;
    int64_t n_multi_p;
    /* with columns, q_cell_ss holds one column per COLUMN: query i reads column q_cell_ss_index[i], row stride q_cell_ss_ld */
    const int64_t *q_cell_ss_index;
    int64_t q_cell_ss_ld;
    /* ONE-PRODUCT first level (r04).  level = 1: Qs / Es are PLANAR hi operands (kge_lp_hi_rows: 32 bytes per k16 unit,
     * f16 hi parts only, kge_lp_hi_units(K) units per row) and the sweep runs ONE MFMA product per unit -- a third of the
     * matrix work, half the operand bytes.  The error band then carries the operands' measured f16 residuals:
     * q_dn2[i] = ||q_i - hi(q_i)||^2 (read at q_dn2_index[i] when given: query columns) and de2max = device scalar
     * >= max_c ||e_c - hi(e_c)||^2; it is ~8x wider than the three-product band (4.7e-4 of ||q|| max||e|| at K = 200), so
     * this level pays when the true entities sit in the sparse upper tail of the scores (a fitted model).  Every MFMA
     * mode; counts stay exact (kge_lp_split_recheck).  level = 0: the three-product sweep. */
    int32_t level;
    const float *q_dn2;
    const int64_t *q_dn2_index;
    const float *de2max;
    /* FREE-RUNNING one-product sweep (r05, level = 1 only).  es_frag = 1: Es is the FRAGMENT-MAJOR candidate table of
     * kge_lp_hi_rows_frag -- [rows_padded / 32][kge_lp_hi_units(K)][64][16 bytes], the 1-KiB block of (32-row group, k16
     * unit) holding chunk (row % 32) + 32 * (k / 8 % 2): the A operand of v_mfma_f32_32x32x16_f16 in lane order.  The
     * count kernel then keeps a 96-query panel resident in LDS, every wavefront reads the fragments of ITS 64 candidate
     * rows straight from global memory into registers and runs without block-wide barriers (lp_hi_stream.hip); same
     * thresholds, same counts, same list.  Qs stays the planar operand.  K must satisfy kge_lp_hi_stream_supported(K).
     * ABI 29 (r06): (1) rows of 33 / 65 k16 units (K = 497..512, 1009..1024 -- ComplEx d = 512) run on the CHUNKED-panel form of
     * the same kernel (lp_hi_chunk.hip: the query panel streamed through a two-slot LDS ring in chunks of 11 / 13 units,
     * 128-query panels; KGE_LP_DOT / KGE_LP_L2_EXPAND, one global list); (2) columns: col_q, and -- for the plain-threshold
     * modes on rows of <= 32 units -- GROUPED columns too (members / n_multi_p: a second launch whose epilogue compares a
     * column's accumulators with the thresholds of each member; not together with region_count).  The projection modes and
     * the chunked form sweep single-query columns only (members must be NULL: KGE_EINVAL otherwise). */
    int32_t es_frag;
    /* es_frag = 1, optional: true_idx[i] = GLOBAL id of the entity whose exact score IS s_true[i] (the evaluator's true
     * entity).  That pair is counted and can never be taken back by the recheck, so the sweep does not list it. */
    const int64_t *true_idx;
    /* optional: the block maxima kge_lp_table_prep_l2 left instead of its atomics (2 * tp_blocks floats); the threshold
     * kernel folds them into *emax0 and *de2max (written through the const pointers) before using the two scalars.
     * Only with thr_ready = 0 and a one-segment problem. */
    const float *tp_block_max;
    int32_t tp_blocks;
    /* KGE_LP_DOT on level 1: 1 = the query operand Qs was built with PER-QUERY scales (kge_lp_dot_query_pipeline: row i scaled
     * by the power of two that fits ||q_i||, not the batch maximum); qn0 then holds the total squared norm per query, qn1 /
     * qmax0 / qmax1 are not read.  Matters only when the thresholds are recomputed (thr_ready = 0). */
    int32_t q_scale_per_query;
    /* es_frag = 1, col_q = NULL, optional (r05): kge_lp_split_regions(B) int32, ZEROED by the caller when thr_ready = 1 (the
     * threshold kernel zeroes them with *list_count otherwise) -- the sweep then leaves its uncertain pairs
     * in REGIONS of `list` (cap / regions entries each), one per 32 consecutive queries, counts here; *list_count is not
     * touched.  Follow with kge_lp_split_recheck_regions (which re-scores a region with its 32 query rows resident in LDS:
     * half the row fetches of kge_lp_split_recheck).  Only when kge_lp_split_regions_supported(d). */
    int32_t *region_count;
} kge_split_args;
int kge_lp_split_group_sets(void);

int kge_lp_split_units(int K, int with_aug);
int64_t kge_lp_split_rows_padded(int64_t rows, int is_query);
/* [X0 | X1] (K1 may be 0) -> split operand with one extra column K0+K1:
 *   aug_mode 1: aug[row] * aug_mul            (L2 candidates: aug = ||e||^2, aug_mul = -0.5)
 *   aug_mode 2: aug_mul                       (L2 queries: 1.0)
 *   aug_mode 3: guard column of DOT queries   (aug = ||q||^2; meets the -65504 of padding candidates)
 *   aug_mode 4: 0                             (DOT candidates)
 * norm2max0/1: device scalars with the squared-norm maxima that fix the power-of-two scale
 * (NULL: the fixed 2^12 of the norm-guarded L2 mode). */
int kge_lp_split_rows(const float *X0, int64_t ld0, int K0, const float *X1, int64_t ld1, int K1, int64_t rows,
                      int is_query, int aug_mode, const float *aug, float aug_mul, const float *norm2max0,
                      const float *norm2max1, void *out, float *cell_ss,
                      const int64_t *row_index /* optional: output row r is built from source row row_index[r] of X0 / X1 /
                                                  aug (query columns: the row of the column's first query) */,
                      kge_stream_t stream);
/* cell_ss (optional, [units_p][rows_padded] floats): per k16 cell the sum of squares of its data values.
 * kge_lp_split_prefix_max folds the cell sums of a CANDIDATE operand into e2pref[u] = max over rows of the squared
 * norm of the first (u+1)*16 columns (units_p floats, zeroed by the caller).  With the queries' cell sums in
 * kge_split_args.q_cell_ss the error band uses || q[:k] || * sqrt(e2pref) as the accumulator's magnitude in unit u
 * (Cauchy-Schwarz on the prefix) instead of || q || || e || throughout: about half the accumulation term. */
int kge_lp_split_prefix_max(const float *cell_ss, int64_t rows, int is_query, int units_p, float *e2pref,
                            kge_stream_t stream);
/* PLANAR hi operand of the one-product level (kge_split_args.level = 1): [rows_padded][kge_lp_hi_units(K0 + K1)][32 bytes],
 * the f16 hi parts of [X0 | X1] at the same scale as kge_lp_split_rows, plus TWO augmentation columns K, K + 1 (aug_mode
 * as above; mode 1 puts hi and lo of aug * aug_mul there, mode 2 aug_mul twice, mode 3 the guard column at K).
 * dn2 (optional, rows floats): ||x - hi(x)||^2 per row, unscaled;  dn2max (optional device scalar): its maximum folded in. */
int kge_lp_hi_units(int K);
int kge_lp_hi_rows(const float *X0, int64_t ld0, int K0, const float *X1, int64_t ld1, int K1, int64_t rows,
                   int is_query, int aug_mode, const float *aug, float aug_mul, const float *norm2max0,
                   const float *norm2max1, void *out, float *dn2, float *dn2max, const int64_t *row_index,
                   kge_stream_t stream);
/* The CANDIDATE operand of the free-running one-product sweep (kge_split_args.es_frag = 1): the values of
 * kge_lp_hi_rows(is_query = 0) in FRAGMENT-MAJOR order, [rows_padded / 32][kge_lp_hi_units(K0 + K1)][64][16 bytes]
 * (same byte count).  kge_lp_hi_stream_supported(K): 1 if that sweep handles K columns (else use the planar table). */
int kge_lp_hi_rows_frag(const float *X0, int64_t ld0, int K0, const float *X1, int64_t ld1, int K1, int64_t rows,
                        int aug_mode, const float *aug, float aug_mul, const float *norm2max0,
                        const float *norm2max1, void *out, float *dn2, float *dn2max, kge_stream_t stream);
int kge_lp_hi_stream_supported(int K);
/* Candidate-table preparation of the L2 one-product sweep in ONE pass over the table: en[row] = ||X[row]||^2 by
 * kge_row_sqnorm's sequential chain (same bits), *en_max_io = max(*en_max_io, max en), out = the fragment-major hi table
 * of kge_lp_hi_rows_frag(aug_mode 1, aug = en, aug_mul = -0.5), *dn2max_io = max(*dn2max_io, max_row ||x - hi(x)||^2).
 * KGE_EUNSUPPORTED unless K % 4 == 0, ld % 4 == 0 and X is 16-byte aligned (then: the separate entry points). */
int kge_lp_table_prep_blocks(int64_t rows);
int kge_lp_table_prep_l2(const float *X, int64_t ld, int64_t rows, int K, float *en, float *en_max_io, void *out,
                         float *dn2max_io,
                         float *block_max /* optional, 2 * kge_lp_table_prep_blocks(rows) floats: the blocks leave their two
                                             maxima there INSTEAD of folding them into the scalars (hundreds of same-address
                                             atomics serialise); pass it on to kge_lp_query_pipeline, which reduces them */,
                         kge_stream_t stream);
int kge_lp_split_count(const kge_lp_desc *d, const kge_split_args *a, const float *s_true, int32_t *raw_count,
                       kge_stream_t stream);
/* The exact recheck of a list cut into regions (kge_split_args.region_count): same decrements of raw_count as
 * kge_lp_split_recheck; *list_stat += pairs re-scored, *list_count += pairs re-scored (both optional). */
int kge_lp_split_regions(int64_t B);
int kge_lp_split_regions_supported(const kge_lp_desc *d);
int kge_lp_split_recheck_regions(const kge_lp_desc *d, const float *s_true, const int32_t *list, int32_t cap,
                                 const int32_t *region_count, int32_t *raw_count, float *list_stat, int32_t *list_count,
                                 kge_stream_t stream);
/* 1 if v_mfma_f32_32x32x16_f16 on the current device accumulates as the tighter error model assumes (two
 * passes of acc + 8 products, addends truncated 24 bits below the largest, one RNE rounding), 0 if not.
 * Launches a one-wave kernel on the null stream and synchronises: call once, outside any capture. */
int kge_mfma_f16_selftest(void);
/* DistMult / ComplEx query side of one batch on the ONE-PRODUCT level in one launch (r05) -- what kge_lp_prep,
 * kge_lp_pair_scores (true scores), two kge_row_sqnorm_any_order passes, kge_lp_hi_rows(is_query) and the threshold kernel
 * of kge_lp_split_count do separately (bilinear.py:247-267, :530-556 + evaluation.py:290-300): Q0 (and Q1 = the Im half,
 * ComplEx: E1 / R1 / Q1 / emax1 non-NULL) bit-identical to kge_lp_prep, s_true bit-identical to kge_lp_pair_scores,
 * qn = ||q_i||^2 (a bound: any summation order), Qh = the planar hi operand with PER-QUERY power-of-two scales, thr the
 * thresholds that go with them, *list_count = 0, *overflow = 1 on non-finite norms; zero_n int32 at zero_i32 zeroed.
 * emax0 / emax1 / de2max: device scalars of the candidate table (kge_row_sqnorm_any_order maxima, kge_lp_hi_rows[_frag]'s
 * residual maximum), final when the launch runs.  Follow with kge_lp_split_count(level = 1, thr_ready = 1,
 * q_scale_per_query = 1).  KGE_EINVAL unless d % 8 == 0 and the tables are 16-byte aligned (then: the separate kernels). */
int kge_lp_dot_query_pipeline(int side, const float *E0, const float *E1, const float *R0, const float *R1, int d,
                              const int64_t *h, const int64_t *t, const int64_t *r, int64_t B, const float *emax0,
                              const float *emax1, const float *de2max, float *qmax_io, int accum_model, float eps_scale,
                              float *Q0, float *Q1, float *qn, float *s_true, void *Qh, float *thr,
                              float *q_dn2 /* optional */, int32_t *list_count, float *overflow, int32_t *zero_i32,
                              int64_t zero_n,
                              const float *dn_block_max /* optional: kge_lp_dot_table_prep's residual block maxima; the kernel
                                                           folds them into *de2max (written through the const pointer) */,
                              int dn_blocks,
                              const float *nm_block_max /* optional (ABI 30): kge_lp_dot_table_prep_fused's squared-norm block
                                                           maxima [2][nm_blocks]; folded into *emax0 / *emax1 */,
                              int nm_blocks,
                              float *prev_nmax /* optional (ABI 30), 2 device floats: receives the folded maxima for the NEXT
                                                  kge_lp_dot_table_prep_fused; with nm_block_max they are read first -- the
                                                  maxima the table was scaled by -- and *overflow = 2 (not the list's 1: run the
                                                  same path again) if the new ones ask for another power-of-two scale */,
                              kge_stream_t stream);
/* Candidate side of a DistMult / ComplEx problem on the one-product level in TWO launches (r05) -- what two
 * kge_row_sqnorm_any_order passes, a zero-fill and kge_lp_hi_rows[_frag](aug_mode 4) do in four, with the table read twice
 * instead of three times and no same-address atomics: (1) the squared-norm maxima of [X0 | X1] per block, (2) the hi table
 * `out` (frag = 1: fragment-major, else planar), whose blocks fold (1) into *norm2max0_io / *norm2max1_io (values already
 * there take part: other shards, an earlier call) and leave their residual maxima in dn_block_max
 * [kge_lp_dot_table_prep_blocks(rows, 1)] -- hand those to kge_lp_dot_query_pipeline.  ws: 2 *
 * kge_lp_dot_table_prep_blocks(rows, 0) floats. */
int kge_lp_dot_table_prep_blocks(int64_t rows, int which);
int kge_lp_dot_table_prep(const float *X0, int64_t ld0, int K0, const float *X1, int64_t ld1, int K1, int64_t rows, int frag,
                          float *norm2max0_io, float *norm2max1_io, void *out, float *dn_block_max, float *ws,
                          kge_stream_t stream);
/* The same in ONE launch and ONE pass over the table (ABI 30; fragment-major output only): the scale is the one the maxima
 * of a PREVIOUS evaluation ask for (prev_nmax, 2 device floats kept by kge_lp_dot_query_pipeline) -- any power of two under
 * which nothing overflows is valid, the error band is built from the residuals measured in this pass -- and this pass's
 * squared-norm maxima come out per block (nm_block_max[2][kge_lp_dot_table_prep_blocks(rows, 1)]) for the query pipeline
 * to fold and to check against the scale that was used.  A table that has outgrown its scale sets the pipeline's
 * *overflow to 2 (the caller runs the same two calls again: prev_nmax holds the new maxima by then).
 * KGE_EINVAL unless rows are float4-readable (K0, K1, ld0, ld1 % 4 == 0, 16-byte aligned bases). */
int kge_lp_dot_table_prep_fused(const float *X0, int64_t ld0, int K0, const float *X1, int64_t ld1, int K1, int64_t rows,
                                const float *prev_nmax, void *out, float *dn_block_max, float *nm_block_max,
                                kge_stream_t stream);
/* TransE-L2 query side of one batch in ONE launch -- what kge_lp_prep, kge_row_sqnorm (queries),
 * kge_lp_pair_scores (true scores), kge_lp_split_rows (queries) and the threshold kernel of
 * kge_lp_split_count do separately, with bit-identical outputs: Q (B,d), qn (B), s_true (B), Qs, thr
 * (2 * rows_padded(B,1) floats), *list_count = 0; *qmax_io = max(*qmax_io, max qn).  en = ||E[c]||^2 over
 * the WHOLE entity table (no shard), emax its device-side maximum.  Follow with kge_lp_split_count
 * (thr_ready = 1).  side = KGE_SIDE_BOTH: both sides of the B facts as ONE batch of 2B queries (tail-side
 * queries first; every output has 2B rows) -- the ranks of evaluation.py:290-300 from one count launch. */
int kge_lp_query_pipeline(int side, const float *E, const float *R, int d, const int64_t *h, const int64_t *t,
                          const int64_t *r, int64_t B, const float *en, const float *emax, float *qmax_io,
                          int accum_model, float eps_scale, float *Q, float *qn, float *s_true, void *Qs,
                          float *thr, int32_t *list_count, const float *e2pref /* optional, see above */,
                          const int32_t *qs_row /* optional: row of Qs that receives query i's split cells, < 0: none
                                                   (columns, see kge_split_args.col_q); NULL: row i */,
                          int level /* 1: Qs is the PLANAR hi operand and thr the thresholds of the one-product level */,
                          const float *de2max /* level 1: device scalar >= max_c ||e_c - hi(e_c)||^2 (kge_lp_hi_rows) */,
                          float *q_dn2 /* level 1, optional out: ||q_i - hi(q_i)||^2 per query (kge_split_args.q_dn2) */,
                          const float *tp_block_max /* optional: the block maxima of kge_lp_table_prep_l2; the kernel then
                                                       folds them into *emax and *de2max (written through the const
                                                       pointers) before using the two scalars */,
                          int tp_blocks,
                          int32_t *zero_i32 /* optional: zero_n int32 zeroed by this launch (the batch's rank counters) */,
                          int64_t zero_n, kge_stream_t stream);
/* *max_io = max(*max_io, max_i |x[i]|) -- device scalar, zero it first */
int kge_absmax(const float *x, int64_t n, float *max_io, kge_stream_t stream);
int kge_lp_split_recheck(const kge_lp_desc *d, const float *s_true, const int32_t *list, int32_t cap,
                         const int32_t *list_count, int32_t *raw_count,
                         float *list_stat /* optional device float: the number of re-scored pairs is added to it */,
                         kge_stream_t stream);

/* ---- certified integer prefilter of the fused rank count, TransE-L1 (torchkge_amd/csrc/lp_l1_sad.hip) ----------
 * kge_lp_sad_count + kge_lp_sad_recheck leave in raw_count exactly what kge_lp_count_ge leaves there for a plain
 * KGE_LP_L1_DIRECT problem (no rank-1 term; interfaces.py:253-260 with l1_dissimilarity): the sum of absolute
 * differences is evaluated on 16-bit fixed-point copies of the operands (v_sad_u16, exact integer accumulation)
 * with a proven error band, the pairs inside the band are re-scored by the exact fp32 chain.
 * Operands: kge_lp_sad_rows turns fp32 rows into (rows, kge_lp_sad_cols_padded(K)) uint16, X = rint(x * s) + 32768,
 * s = 32700 / (*emax + *rmax): device scalars with max |x| of the entity table and of the relation table (every
 * query element is e +- r; see kge_absmax).  *overflow is set to 1.0f if more than cap pairs (or more than 2048 in
 * one 128 x 128 tile) fell inside the band, or if the maxima are not finite / zero -- raw_count is then invalid and
 * the caller redoes the count with kge_lp_count_ge. */
typedef struct kge_sad_args {
    const void *Qi, *Ei;          /* fixed-point operands (kge_lp_sad_rows) */
    const float *emax, *rmax;     /* the device scalars the operands were scaled with */
    float eps_scale;              /* multiplies the error band (1.0 = the proven bound; tests shrink it) */
    int32_t *thr;                 /* scratch: 2 * B int32 */
    int32_t *list;                /* scratch: cap x 2 int32 (query, local candidate) */
    int32_t cap;
    int32_t *list_count;          /* device int32 */
    float *overflow;              /* device float, set to 1.0f on overflow (see above) */
    /* COLUMNS instead of queries (optional, as kge_split_args.col_q / members): Qi holds one row per distinct query row of
     * the batch -- n_single_p rows whose column carries one query (col_q[column], < 0: padding), then n_multi_p rows
     * whose column carries up to kge_lp_split_group_sets() queries (members[column * sets + j]); thr, raw_count and the
     * pair list stay indexed by QUERY.  Both NULL: row == query. */
    const int32_t *col_q;
    int64_t n_single_p;
    const int32_t *members;
    int64_t n_multi_p;
} kge_sad_args;
int64_t kge_lp_sad_cols_padded(int K);
int kge_lp_sad_rows(const float *X, int64_t ld, int64_t rows, int K, const float *emax, const float *rmax,
                    void *out, const int64_t *row_index /* optional: output row r <- source row row_index[r] */,
                    kge_stream_t stream);
int kge_lp_sad_count(const kge_lp_desc *d, const kge_sad_args *a, const float *s_true, int32_t *raw_count,
                     kge_stream_t stream);
/* exact re-scoring of a (query, local candidate) pair list for plain KGE_LP_L1_DIRECT / _L2_DIRECT problems:
 * raw_count[q] -= 1 for every listed pair whose score is below s_true[q] */
int kge_lp_sad_recheck(const kge_lp_desc *d, const float *s_true, const int32_t *list, int32_t cap,
                       const int32_t *list_count, int32_t *raw_count, kge_stream_t stream);

int kge_abi_version(void);
const char *kge_build_arch(void);

#ifdef __cplusplus
}
#endif
#endif /* KGE_HIP_H */
