/* Oracle (test infrastructure): fp32 scores as the fused-multiply-add chain the
 * evaluation kernel documents, train-mask, and top-K by (score desc, id asc).
 *
 * Restates evalue/evaluator.py:54-55 (U @ I.T, S[row,col] = -1e10) and
 * evalue/metric.py:114 (topk, largest, sorted) of the reference with two points
 * fixed that the reference leaves to its BLAS / torch.topk: the order in which
 * the D products of one score are accumulated (k = 0,32,1,33,...,31,63 inside
 * each block of 64, one fmaf per product, starting from +0) and the order of
 * equal scores (lower item id first).  Built by oracle/Makefile with gcc.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

float oracle_score_one(const float *u, const float *it, int D) {
  float acc = 0.0f;
  for (int c = 0; c < D / 64; ++c)
    for (int m = 0; m < 32; ++m) {
      acc = fmaf(it[64 * c + m], u[64 * c + m], acc);
      acc = fmaf(it[64 * c + 32 + m], u[64 * c + 32 + m], acc);
    }
  return acc;
}

/* dense score matrix S[n_users][n_items] in chain order (small cases) */
void oracle_scores(const float *U, int n_users, const float *I, int n_items, int D, float *S) {
  for (int u = 0; u < n_users; ++u)
    for (int i = 0; i < n_items; ++i)
      S[(int64_t)u * n_items + i] = oracle_score_one(U + (int64_t)u * D, I + (int64_t)i * D, D);
}

static int better(float sa, int ia, float sb, int ib) { return sa > sb || (sa == sb && ia < ib); }

/* users [u_lo, u_hi): masked top-K, ids offset by id_offset; out_* indexed from u_lo */
void oracle_score_topk(const float *U, int u_lo, int u_hi, const float *I, int n_items, int D,
                       const int32_t *mask_ptr, const int32_t *mask_items, int K, int id_offset,
                       float *out_s, int32_t *out_id) {
  float *row = (float *)malloc(sizeof(float) * (size_t)n_items);
  for (int u = u_lo; u < u_hi; ++u) {
    for (int i = 0; i < n_items; ++i)
      row[i] = oracle_score_one(U + (int64_t)u * D, I + (int64_t)i * D, D);
    if (mask_ptr)
      for (int p = mask_ptr[u]; p < mask_ptr[u + 1]; ++p) row[mask_items[p]] = -1e10f;
    float *bs = out_s + (int64_t)(u - u_lo) * K;
    int32_t *bi = out_id + (int64_t)(u - u_lo) * K;
    int n = 0;
    for (int i = 0; i < n_items; ++i) {
      if (n == K && !better(row[i], i, bs[K - 1], bi[K - 1] - id_offset)) continue;
      int j = n < K ? n++ : K - 1;
      while (j > 0 && better(row[i], i, bs[j - 1], bi[j - 1] - id_offset)) {
        bs[j] = bs[j - 1];
        bi[j] = bi[j - 1];
        --j;
      }
      bs[j] = row[i];
      bi[j] = i + id_offset;
    }
  }
  free(row);
}
