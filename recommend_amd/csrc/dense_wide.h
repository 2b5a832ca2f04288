// Internal interface between dense.hip (entry points, 64-wide kernels) and dense_wide.hip
// (128-wide kernels).
#pragma once
#include "common.h"

namespace rk {

int dense128_fwd(const float *X1, const float *W1, const float *X2, const float *W2, const float *b1,
                 const float *b2, const float *alpha, float p_drop, uint64_t seed, const uint64_t *seed_step,
                 const float *residual, int64_t n_rows, float *Y, hipStream_t st);
size_t dense128_bwd_ws_bytes();
int dense128_bwd(const float *X, const float *W, const float *b, const float *alpha, float p_drop, uint64_t seed,
                 const uint64_t *seed_step, const float *G, int64_t n_rows, float *dX, float *dW, float *db,
                 float *dalpha, int accumulate, void *ws, hipStream_t st);

// the backward pass in two halves (see rk_dense_bwd_partials_f32 / rk_dense_reduce_jobs_d_f32)
size_t dense128_slab_bytes();
int dense128_bwd_partials(const float *X, const float *W, const float *b, const float *alpha, float p_drop, uint64_t seed,
                          const uint64_t *seed_step, const float *G, int64_t n_rows, float *dX, float *slab,
                          int32_t *n_part_out, hipStream_t st);
int dense128_reduce_jobs(const rk_dense_reduce_job *jobs, int32_t n_jobs, hipStream_t st);

}  // namespace rk
