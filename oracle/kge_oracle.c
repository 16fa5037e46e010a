/*
 * CPU ORACLE (plain C) for the torchkge link-prediction hot path.
 *
 * TEST INFRASTRUCTURE ONLY -- never linked into, loaded by, or called from the
 * product (torchkge_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load the shared object built from this file.
 *
 * Two groups of functions:
 *  (1) integer semantics of the reference, restated from
 *        torchkge/utils/operations.py:37-61    (get_rank)
 *        torchkge/utils/modeling.py:53-102     (get_true_targets/filter_scores)
 *        torchkge/sampling.py:313-325          (corrupt_batch index scatter)
 *      -> the HIP kernels must match these BIT-EXACTLY.
 *  (2) the exact fp32 arithmetic the HIP all-candidates kernels commit to
 *      (a single-accumulator fmaf chain in ascending k, which is what
 *      v_mfma_f32_32x32x2_f32 computes), so the GPU score matrix can be
 *      checked bit-for-bit on the CPU and then compared (tol 1e-5) with the
 *      reference-order restatement in kge_oracle.py.
 *
 * Parity status: pinned through tests/test_oracle_golden.py (known-answer
 * vectors of the reference's tests/test_utils.py and fixtures generated from
 * the reference itself).
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

/* ---- (1) integer semantics ------------------------------------------------ */

/* get_rank: rank_i = #{c : data[i,c] >= data[i,true_i]} (or <= when low_values).
 * NaN never counts.  utils/operations.py:56-61 */
void orc_get_rank(const float *data, const int64_t *true_idx, int64_t B, int64_t N,
                  int low_values, int64_t *rank_out)
{
    for (int64_t i = 0; i < B; ++i) {
        const float *row = data + i * N;
        const float tv = row[true_idx[i]];
        int64_t cnt = 0;
        if (low_values) { for (int64_t c = 0; c < N; ++c) cnt += (row[c] <= tv); }
        else            { for (int64_t c = 0; c < N; ++c) cnt += (row[c] >= tv); }
        rank_out[i] = cnt;
    }
}

/* filter_scores through a CSR view of dictionary[(key1_i,key2_i)]:
 * seg_i = [off[i], off[i+1]) lists the set; has_key[i]==0 <=> KeyError (row
 * untouched).  If true_i is not in the set, set.remove raises KeyError which
 * the reference swallows -> row untouched as well.  utils/modeling.py:78-102 */
void orc_filter_scores(float *scores, int64_t B, int64_t N, const int64_t *true_idx,
                       const uint8_t *has_key, const int64_t *off, const int64_t *tgt)
{
    for (int64_t i = 0; i < B; ++i) {
        if (!has_key[i]) continue;
        int found = 0;
        for (int64_t j = off[i]; j < off[i + 1]; ++j) found |= (tgt[j] == true_idx[i]);
        if (!found) continue;
        for (int64_t j = off[i]; j < off[i + 1]; ++j)
            if (tgt[j] != true_idx[i]) scores[i * N + tgt[j]] = -INFINITY;
    }
}

/* raw + filtered rank without building the filtered matrix:
 * filt = raw - sum_{c in F_i \ {true}} ([s_c >= s_true] - [-inf >= s_true]) */
void orc_filtered_rank(const float *scores, int64_t B, int64_t N, const int64_t *true_idx,
                       const uint8_t *has_key, const int64_t *off, const int64_t *tgt,
                       int64_t *rank_out, int64_t *filt_out)
{
    for (int64_t i = 0; i < B; ++i) {
        const float *row = scores + i * N;
        const float tv = row[true_idx[i]];
        int64_t raw = 0;
        for (int64_t c = 0; c < N; ++c) raw += (row[c] >= tv);
        rank_out[i] = raw;
        int64_t sub = 0; int found = 0;
        if (has_key[i]) {
            for (int64_t j = off[i]; j < off[i + 1]; ++j) {
                if (tgt[j] == true_idx[i]) { found = 1; continue; }
                sub += (int64_t)(row[tgt[j]] >= tv) - (int64_t)(-INFINITY >= tv);
            }
        }
        filt_out[i] = found ? raw - sub : raw;
    }
}

/* corrupt_batch scatter: positions j in [0, B*n_neg); mask[j]!=0 -> head
 * replaced by the next unused draws_h entry, else tail replaced by the next
 * unused draws_t entry.  sampling.py:313-325 */
void orc_corrupt_scatter(const int64_t *heads, const int64_t *tails, const uint8_t *mask,
                         const int64_t *draws_h, const int64_t *draws_t,
                         int64_t B, int64_t n_neg, int64_t *neg_heads, int64_t *neg_tails)
{
    int64_t ph = 0, pt = 0;
    for (int64_t j = 0; j < B * n_neg; ++j) {
        const int64_t b = j % B;
        if (mask[j]) { neg_heads[j] = draws_h[ph++]; neg_tails[j] = tails[b]; }
        else         { neg_heads[j] = heads[b];      neg_tails[j] = draws_t[pt++]; }
    }
}

/* ---- (2) fp32 arithmetic contract of the HIP all-candidates kernels ------- */

/* dot chain: one accumulator, acc = fmaf(a[k], t[k], acc), segment 0 then 1;
 * k runs over 8-blocks in ascending order and, inside an 8-block, in the order
 * 0,4,1,5,2,6,3,7 -- the order in which the gfx950 kernel feeds
 * v_mfma_f32_32x32x2_f32 (include/kge_hip.h, kge_lp_desc). */
static float chain_seg(const float *a, const float *t, int64_t K, float acc)
{
    for (int64_t kb = 0; kb < K; kb += 8)
        for (int j = 0; j < 4; ++j) {
            const int64_t k0 = kb + j, k1 = kb + 4 + j;
            if (k0 < K) acc = fmaf(a[k0], t[k0], acc);
            if (k1 < K) acc = fmaf(a[k1], t[k1], acc);
        }
    return acc;
}
static float chain_dot(const float *a0, const float *t0, int64_t K0,
                       const float *a1, const float *t1, int64_t K1)
{
    float acc = chain_seg(a0, t0, K0, 0.0f);
    if (K1 > 0) acc = chain_seg(a1, t1, K1, acc);
    return acc;
}

/* S[i,c] = dot(A0[i],T0[c]) (+ dot(A1[i],T1[c]))              epilogue 0
 * S[i,c] = -max(fmaf(-2, dot, qn[i]+en[c]), 0)                 epilogue 1 */
void orc_lp_gemm_chain(const float *A0, int64_t lda0, const float *T0, int64_t ldt0, int64_t K0,
                       const float *A1, int64_t lda1, const float *T1, int64_t ldt1, int64_t K1,
                       int64_t B, int64_t N, int epilogue, const float *qn, const float *en,
                       float *out)
{
    for (int64_t i = 0; i < B; ++i)
        for (int64_t c = 0; c < N; ++c) {
            float dot = chain_dot(A0 + i * lda0, T0 + c * ldt0, K0,
                                  K1 ? A1 + i * lda1 : 0, K1 ? T1 + c * ldt1 : 0, K1);
            float s = dot;
            if (epilogue == 1) {
                float d2 = fmaf(-2.0f, dot, qn[i] + en[c]);
                s = -fmaxf(d2, 0.0f);
            }
            out[i * N + c] = s;
        }
}

/* The projection modes of kge_lp_desc (include/kge_hip.h: KGE_LP_L2_PROJH = 4, KGE_LP_L2_PROJD = 5):
 *   v = fmaf(-2, chain(A[i], T[c]), qn[i] + en[c]);  x = X[r_idx[i]*ldx + c];  (p, z) = pz[2i], pz[2i+1]
 *   PROJH (TransH, translation.py:183-284 with candidates e_c - x w):  v = fmaf(x, fmaf(x, z, p), v)
 *   PROJD (TransD, translation.py:538-652 with candidates e'_c + y_c w): v = fmaf(y, fmaf(y, z, fmaf(2, x, p)), v)
 *   s = -max(v, 0) */
void orc_lp_proj_chain(int mode, const float *A, int64_t lda, const float *T, int64_t ldt, int64_t K,
                       int64_t B, int64_t N, const float *qn, const float *en, const float *X, int64_t ldx,
                       const int64_t *r_idx, const float *yc, const float *pz, float *out)
{
    for (int64_t i = 0; i < B; ++i)
        for (int64_t c = 0; c < N; ++c) {
            float dot = chain_dot(A + i * lda, T + c * ldt, K, 0, 0, 0);
            float v = fmaf(-2.0f, dot, qn[i] + en[c]);
            float x = X[r_idx[i] * ldx + c], p = pz[2 * i], z = pz[2 * i + 1];
            if (mode == 4) v = fmaf(x, fmaf(x, z, p), v);
            else v = fmaf(yc[c], fmaf(yc[c], z, fmaf(2.0f, x, p)), v);
            out[i * N + c] = -fmaxf(v, 0.0f);
        }
}

/* squared row norms with the same chain: n[i] = sum_k x[i,k]^2 (fmaf chain). */
void orc_row_sqnorm_chain(const float *X, int64_t ld, int64_t rows, int64_t K, float *out)
{
    for (int64_t i = 0; i < rows; ++i) {
        float acc = 0.0f;
        for (int64_t k = 0; k < K; ++k) acc = fmaf(X[i * ld + k], X[i * ld + k], acc);
        out[i] = acc;
    }
}

/* direct translational all-candidates (broadcast-subtract + L_p reduction):
 *   diff_k = (q[i,k] - e[c,k])            [+ a(i,c) * w[i,k]  when w != NULL]
 *   p==2: acc = fmaf(diff_k, diff_k, acc), k ascending
 *   p==1: one add per aligned group of four k:  acc += (|d0| + |d1|) + (|d2| + |d3|)   (absent k: 0)
 *         -- the HIP kernels' L1 contract since r03: as many instructions as the plain chain, a quarter of
 *         its roundings at full magnitude (sigma of a d=200 score 4.4e-6 -> 2.2e-6)
 *   S[i,c] = -acc ;  a(i,c) = scal[c*scal_ld + (scal_ld>1 ? r_idx[i] : 0)] */
void orc_lp_direct_chain(const float *Q, int64_t ldq, const float *T, int64_t ldt, int64_t K,
                         const float *Wq, int64_t ldw, const float *scal, int64_t scal_ld,
                         const int64_t *r_idx, int64_t B, int64_t N, int p, float *out)
{
    for (int64_t i = 0; i < B; ++i)
        for (int64_t c = 0; c < N; ++c) {
            float a = 0.0f;
            if (Wq) a = scal[c * scal_ld + (scal_ld > 1 ? r_idx[i] : 0)];
            float acc = 0.0f;
            if (p == 1) {
                for (int64_t k = 0; k < K; k += 4) {
                    float m[4];
                    for (int e = 0; e < 4; ++e) {
                        float diff = 0.0f;
                        if (k + e < K) {
                            diff = Q[i * ldq + k + e] - T[c * ldt + k + e];
                            if (Wq) diff = fmaf(a, Wq[i * ldw + k + e], diff);
                        }
                        m[e] = fabsf(diff);
                    }
                    acc = acc + ((m[0] + m[1]) + (m[2] + m[3]));
                }
            } else {
                for (int64_t k = 0; k < K; ++k) {
                    float diff = Q[i * ldq + k] - T[c * ldt + k];
                    if (Wq) diff = fmaf(a, Wq[i * ldw + k], diff);
                    acc = fmaf(diff, diff, acc);
                }
            }
            out[i * N + c] = -acc;
        }
}
