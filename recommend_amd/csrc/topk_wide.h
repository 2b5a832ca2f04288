// Internal interface between eval.hip (entry points of the evaluation) and knn.hip (the fused
// score + top-K kernel for wide rows).
#pragma once
#include "common.h"

namespace rk {

// floats + ints of partial lists wide_topk needs behind `ps` (entries of 8 bytes)
size_t wide_part_entries(int32_t n_a, int32_t n_b, int32_t K);

// out[a] = the K best <A[a], B[b]> over b by (score desc, id asc), rows F floats wide (F % 64 == 0);
// mask_ptr / mask_items (may be NULL): CSR of the pairs that score -1e10; ids are returned + id_offset.
int wide_topk(const float *A, int32_t n_a, const float *B, int32_t n_b, int32_t F, const int32_t *mask_ptr,
              const int32_t *mask_items, int32_t K, int32_t id_offset, float *out_scores, int32_t *out_ids,
              float *ps, hipStream_t st);

}  // namespace rk
