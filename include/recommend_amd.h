/*
 * recommend_amd.h -- C ABI of librecommend_amd.so (MI355X / gfx950).
 *
 * Drop-in boundary for the data-parallel hot path of the multi-modal graph
 * recommender: bipartite-graph propagation (CSR/CSC SpMM, forward/backward),
 * BPR gather-dot loss with negative sampling, and full-ranking evaluation
 * (score, train-mask, top-K, ranking metrics).
 *
 * The reference is pure Python and has no FFI of its own; every entry point
 * below names the reference function (file:line under the reference tree)
 * whose arithmetic it replaces.  The Python host classes in recommend_amd/
 * keep the reference's call protocols and bind these symbols with ctypes.
 *
 * Conventions
 *  - plain pointers and sizes only; no framework types cross this boundary;
 *  - every function returns RK_OK (0) or a negative RK_ERR_* code and never
 *    throws; rk_last_error() returns a thread-local message for the last
 *    failure on the calling thread;
 *  - pointers named *_host are host memory, everything else is device memory
 *    owned by the caller; the library allocates no device memory;
 *  - `stream` is a hipStream_t passed as void*; all device work is enqueued
 *    asynchronously on it and the functions do not synchronise;
 *  - dense matrices are row-major fp32, 16-byte aligned rows; graph indices
 *    are int32 on the device (node count and nnz < 2^31); batch ids are
 *    int64 as the reference's batch dict holds them.
 */
#ifndef RECOMMEND_AMD_H
#define RECOMMEND_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RK_OK 0
#define RK_ERR_ARG (-1)         /* bad pointer / size / unsupported shape   */
#define RK_ERR_HIP (-2)         /* a HIP runtime call or launch failed      */
#define RK_ERR_UNSUPPORTED (-3) /* valid request outside the built kernels  */
#define RK_ERR_NODEV (-4)       /* no gfx950 device visible                 */

#define RK_ABI_VERSION 1

/* ---- library ---------------------------------------------------------- */
int rk_abi_version(void);
const char *rk_last_error(void);
/* number of visible HIP devices (0 on a CPU-only host, never an error) */
int rk_device_count(void);

/* ---- G1: edge list and CSR/CSC construction (host, bit-exact) ---------- */
/* Replaces GraphConstructor.build_graph, model/test/graph_constructor.py:51-117
 * (edge order :80-98).  For interaction j (dataset order) emits edge 2j
 * (u -> i+U) and edge 2j+1 (i+U -> u), both with weight rating[j]; then, if
 * add_self_loops, edge 2E+n (n -> n) for n = 0..N-1 with weight
 * max(0, max_j rating[j]).  nnz = 2E (+N).  ratings_host may be NULL (all 1). */
int rk_build_coo_host(const int64_t *users_host, const int64_t *items_host,
                      const float *ratings_host, int64_t n_inter,
                      int32_t n_users, int32_t n_items, int32_t add_self_loops,
                      int32_t *src_host, int32_t *dst_host, float *w_host);

/* Stable counting sort of the COO by `key` (dst for the forward CSR, src for
 * the transposed/CSC form used by the backward pass), ties by edge id.
 * ptr has n_nodes+1 entries; idx[p] = other[eid[p]]; eid[p] = COO edge id.
 * This ordering is the bit-exact CSR contract (dgl.graph + update_all read the
 * same COO, graph_constructor.py:101-105). */
int rk_csr_from_coo_host(const int32_t *key_host, const int32_t *other_host,
                         int64_t nnz, int32_t n_nodes, int32_t *ptr_host,
                         int32_t *idx_host, int32_t *eid_host);

/* Row binning for the SpMM kernels (see rk_spmm_plan below): rows with more
 * than long_thresh edges are cut into chunks of 512 edges, one workgroup each;
 * rows longer than one chunk get one partial-row slot per chunk.  First call
 * with the six array pointers NULL to get the counts, then with buffers of
 * n_chunks (x4), n_multi and n_multi+1 entries. */
int rk_spmm_plan_host(const int32_t *ptr_host, int32_t n_rows,
                      int32_t long_thresh, int32_t *n_chunks_out,
                      int32_t *n_multi_out, int32_t *n_slots_out,
                      int32_t *chunk_row_host, int32_t *chunk_begin_host,
                      int32_t *chunk_end_host, int32_t *chunk_slot_host,
                      int32_t *multi_rows_host, int32_t *multi_slot_ptr_host);

/* ---- G2: symmetric normalisation (device) ------------------------------ */
/* Replaces GraphConstructor._normalize_graph/_norm, graph_constructor.py:158-187:
 * in_w[v] = sum over incoming edges of w (edge-id order), clamp(min=1e-12),
 * sqrt, then w_e <- w_e / (in_w[src_e] * in_w[dst_e]).  Inputs: forward CSR
 * (rows = dst) with its eid permutation and the COO-ordered raw weights.
 * Outputs: sqrt_in_w[n_nodes] and the COO-ordered normalised weights. */
int rk_sym_normalize_f32(const int32_t *rowptr, const int32_t *col,
                         const int32_t *eid, const float *w_coo,
                         int32_t n_nodes, float *sqrt_in_w, float *w_norm_coo,
                         void *stream);

/* The two halves of rk_sym_normalize_f32 for a graph whose rows are spread over several GPUs
 * (every GPU holds the destination rows [row_offset, row_offset + n_rows) of the CSR, columns
 * global, eid / w_coo local to its own edge list):
 *   rk_in_weight_sqrt_f32: sqrt_in_w_rows[v] = sqrt(max(sum of the row's w, 1e-12)),
 *     graph_constructor.py:177-178, for the local rows;
 *   rk_sym_scale_rows_f32: w_norm[e] = w[e] / (sqrt_in_w[col] * sqrt_in_w[row_offset + v]),
 *     graph_constructor.py:186, with sqrt_in_w the vector over ALL nodes (the local pieces
 *     gathered by the caller).  The transposed block (rows = sources) takes the same call. */
int rk_in_weight_sqrt_f32(const int32_t *rowptr, const int32_t *eid,
                          const float *w_coo, int32_t n_rows,
                          float *sqrt_in_w_rows, void *stream);
int rk_sym_scale_rows_f32(const int32_t *rowptr, const int32_t *col,
                          const int32_t *eid, const float *w_coo, int32_t n_rows,
                          int32_t row_offset, const float *sqrt_in_w,
                          float *w_norm_coo, void *stream);

/* Degree normalisation of the live models (model/sgrec/graph.py:123-135,
 * model/mig/mgdcf.py:106-126): w_e = deg_in[src]^-0.5 * deg_in[dst]^-0.5 with
 * deg_in counted from the CSR (rows = dst). */
int rk_deg_normalize_f32(const int32_t *rowptr, const int32_t *col,
                         const int32_t *eid, int32_t n_nodes,
                         float *w_norm_coo, void *stream);

/* Feature-similarity edge weights (device).  Replaces GraphConstructor._compute_edge_weights,
 * model/test/graph_constructor.py:137-156, for the COO edge list (src, dst):
 *   RK_EDGE_SIM_COSINE       clamp(<x_s, x_d> / ((|x_s| + 1e-12)(|x_d| + 1e-12)), 0, 1)   (:144-147)
 *   RK_EDGE_SIM_SIGMOID_DOT  sigmoid(<x_s, x_d>)                                          (:148-151)
 * X holds the feature rows of nodes [row_lo, row_lo + n_rows) (F floats each, F % 4 == 0); every
 * other node has all-zero features, as the reference's zero padding of a user- or item-side
 * feature to all nodes makes them (:130-135).  fp32 fma chains per lane and a fixed reduction
 * tree: bitwise reproducible, within 1e-6 of a float64 evaluation. */
#define RK_EDGE_SIM_COSINE 0
#define RK_EDGE_SIM_SIGMOID_DOT 1
int rk_edge_feature_sim_f32(const int32_t *src, const int32_t *dst, int64_t nnz,
                            const float *X, int32_t row_lo, int32_t n_rows,
                            int32_t F, int32_t mode, float *out, void *stream);

/* out[p] = src[idx[p]] (permutes COO-ordered edge values into CSR/CSC order) */
int rk_gather_f32(const float *src, const int32_t *idx, int64_t n, float *out,
                  void *stream);

/* ---- G3: propagation h <- A h (device) ---------------------------------- */
typedef struct rk_spmm_plan {
  int32_t long_thresh;            /* rows with more edges go to the long tier */
  int32_t n_chunks;               /* long-tier workgroups                     */
  const int32_t *chunk_row;       /* [n_chunks] destination row               */
  const int32_t *chunk_begin;     /* [n_chunks] first edge of the chunk       */
  const int32_t *chunk_end;       /* [n_chunks] one past its last edge        */
  const int32_t *chunk_slot;      /* [n_chunks] partial slot, -1 = whole row  */
  int32_t n_multi;                /* rows cut into more than one chunk        */
  const int32_t *multi_rows;      /* [n_multi] their row ids                  */
  const int32_t *multi_slot_ptr;  /* [n_multi+1] slot range of each           */
  int32_t n_slots;
  float *partial;                 /* [n_slots * D] scratch, caller-owned      */
  int32_t *arrive;                /* [n_multi] arrival counters, zeroed ONCE by the
                                   * caller; every launch leaves them zero.  Launches
                                   * that share a plan must not overlap.            */
} rk_spmm_plan;

/* Replaces update_all(fn.u_mul_e('h','weight','m'), fn.sum('m','h')),
 * graph_constructor.py:194,205 (also model/sgrec/gcn.py:84, model/mig/mgdcf.py:152):
 *   out[v,:] = sum_{p in row v} val[p] * h[col[p],:]
 * and, when h0 != NULL, the fused epilogue of aggrate (graph_constructor.py:195)
 * and MGDCF (mgdcf.py:152-153):  out = alpha*h0 + beta*(A h); with h0 == NULL the result
 * is beta*(A h) (alpha unused).
 * The backward pass is the same call on the CSC arrays (d h = beta * A^T d out).
 * plan may be NULL (every row handled by one lane group).  D % 4 == 0. */
int rk_spmm_csr_f32(const int32_t *rowptr, const int32_t *col, const float *val,
                    int32_t n_rows, const float *h, int32_t D, float *out,
                    const float *h0, float alpha, float beta,
                    const rk_spmm_plan *plan, void *stream);

/* ---- W1: modality projection (device, fp32 MFMA) ------------------------ */
/* Replaces nn.Linear(feat_dim, emb_dim) applied to the zero-padded feature
 * matrix, model/Main_Model.py:65-74,154: user rows of the padded input are all
 * zero, so Y[u,:] = b for u < n_users and Y[n_users+i,:] = X[i,:] W^T + b.
 * X: n_items x F, W: D x F (nn.Linear layout), b: D or NULL, Y: (U+I) x D. */
int rk_proj_fwd_f32(const float *X, int32_t n_items, int32_t F, const float *W,
                    const float *b, int32_t D, int32_t n_users, float *Y,
                    void *stream);
/* The same projection with a workspace (rk_proj_fwd_ws_bytes() bytes, 16-byte aligned; 0 for
 * shapes it has no use for): for D = 64 and F a multiple of 32 the weights are split once per call
 * into three bf16 parts in the order the matrix instructions read them, and the product runs on
 * the bf16 matrix pipe with 256-row workgroup tiles (the error of a product is that of an fp32 dot
 * product, as for the node-wise layers below).  Other shapes: rk_proj_fwd_f32. */
size_t rk_proj_fwd_ws_bytes(int32_t n_items, int32_t F, int32_t D);
int rk_proj_fwd_ws_f32(const float *X, int32_t n_items, int32_t F, const float *W,
                       const float *b, int32_t D, int32_t n_users, float *Y,
                       void *ws, size_t ws_bytes, void *stream);
size_t rk_proj_bwd_ws_bytes(int32_t n_items, int32_t n_users, int32_t F,
                            int32_t D);
/* dW = dY[item rows]^T X (D x F), db = column sums of dY over all U+I rows.
 * ws must hold rk_proj_bwd_ws_bytes() bytes.  db may be NULL. */
int rk_proj_bwd_f32(const float *X, int32_t n_items, int32_t F, const float *dY,
                    int32_t D, int32_t n_users, float *dW, float *db, void *ws,
                    size_t ws_bytes, void *stream);

/* ---- W2: node-wise D -> D layers around the hops, D = 64 or 128 (device; fp32 results) ---- */
/* Replaces dropout(active(f(h))) with f = Linear, active = PReLU (one slope) in
 * agg_func (graph_constructor.py:204), Combine (Combin_Layer.py:26-30: the sum of
 * two Linears is the X2/W2 pair), MGCN's embedding branch (out_Layer.py:36: R is
 * the residual added after the dropout) and final_projection + dropout
 * (Main_Model.py:162-163):
 *   Y = dropout(prelu(X1 W1^T [+ X2 W2^T] + b1 [+ b2], alpha), p) [+ R]
 * alpha == NULL: no activation.  p_drop == 0: no dropout.  Dropout keeps element
 * (row, col) iff a 16-bit field of a counter hash of (seed, row, col) is >=
 * round(p * 65536); survivors are scaled by 1/(1-p) (same distribution as
 * nn.Dropout, different random stream).  seed_step (device pointer, may be NULL):
 * a step counter mixed into the seed when the kernel runs, so that a launch
 * replayed from a captured HIP graph draws a new mask every step; forward and
 * backward of one layer must see the same seed and the same counter value.
 * W: D x K (nn.Linear layout).  Built for K = D = 64 and K = D = 128.
 * Arithmetic (both widths): every fp32 operand is split into three bf16 parts and the six
 * leading partial products are summed in fp32 on the bf16 matrix pipe: the error
 * of a product is that of an fp32 dot product (<= 1e-6 of sum |x_k w_k|); operands
 * beyond bf16's largest finite value (3.39e38) yield NaN. */
int rk_dense_fwd_f32(const float *X1, const float *W1, const float *X2,
                     const float *W2, const float *b1, const float *b2,
                     const float *alpha, float p_drop, uint64_t seed,
                     const uint64_t *seed_step, const float *residual,
                     int64_t n_rows, int32_t K, int32_t D, float *Y, void *stream);
size_t rk_dense_bwd_ws_bytes(int64_t n_rows, int32_t K, int32_t D);
/* Backward of the single-input layer for upstream gradient G (the residual's
 * gradient is G itself): recomputes Z = X W^T + b and the dropout mask, writes
 * dX (may be NULL), dW, db (may be NULL), dalpha (NULL iff alpha is NULL).  A
 * two-input linear layer is differentiated with one call per input. */
int rk_dense_bwd_f32(const float *X, const float *W, const float *b,
                     const float *alpha, float p_drop, uint64_t seed,
                     const uint64_t *seed_step, const float *G, int64_t n_rows,
                     int32_t K, int32_t D, float *dX, float *dW, float *db,
                     float *dalpha, void *ws, size_t ws_bytes, void *stream);

/* The same with `accumulate`: if non-zero, dW, db and dalpha are ADDED to what the outputs hold
 * (the same Linear applied k times in agg_func, graph_constructor.py:203-205: one gradient buffer
 * for all k applications); dX is always overwritten. */
int rk_dense_bwd_accum_f32(const float *X, const float *W, const float *b,
                           const float *alpha, float p_drop, uint64_t seed,
                           const uint64_t *seed_step, const float *G,
                           int64_t n_rows, int32_t K, int32_t D, float *dX,
                           float *dW, float *db, float *dalpha,
                           int32_t accumulate, void *ws, size_t ws_bytes,
                           void *stream);

/* The backward pass in two halves, for callers that differentiate many layers in a row (one
 * MGCN branch: 13 calls): rk_dense_bwd_partials_f32 is rk_dense_bwd_f32 without its closing
 * reduction -- it leaves *n_part_out_host per-workgroup partial sums of dW / db / dalpha in `slab`
 * (rk_dense_bwd_slab_bytes; 0 for widths without the two-half form) -- and
 * rk_dense_reduce_jobs_d_f32 (rk_dense_reduce_jobs_f32 for D = 64) then reduces any number of slabs of
 * layers of width D in one launch.  A job lists the slabs whose sums go, added in list order,
 * to one set of outputs (one slab per application of that weight): the results are bit for
 * bit those of rk_dense_bwd_f32 / rk_dense_bwd_accum_f32 called layer by layer.  db and dalpha
 * may be NULL.  jobs_host is a HOST array. */
#define RK_DENSE_REDUCE_MAX_SLABS 8
typedef struct rk_dense_reduce_job {
  const float *slabs[RK_DENSE_REDUCE_MAX_SLABS];
  int32_t n_slabs;
  int32_t n_part;          /* partials per slab, as rk_dense_bwd_partials_f32 reported */
  float *dW, *db, *dalpha;
} rk_dense_reduce_job;
size_t rk_dense_bwd_slab_bytes(int64_t n_rows, int32_t K, int32_t D);
int rk_dense_bwd_partials_f32(const float *X, const float *W, const float *b,
                              const float *alpha, float p_drop, uint64_t seed,
                              const uint64_t *seed_step, const float *G,
                              int64_t n_rows, int32_t K, int32_t D, float *dX,
                              float *slab, size_t slab_bytes,
                              int32_t *n_part_out_host, void *stream);
int rk_dense_reduce_jobs_f32(const rk_dense_reduce_job *jobs_host, int32_t n_jobs,
                             void *stream);
int rk_dense_reduce_jobs_d_f32(const rk_dense_reduce_job *jobs_host, int32_t n_jobs,
                               int32_t D, void *stream);

/* ---- W2: modality-weighted sum (device) ---------------------------------- */
/* Replaces torch.sum(torch.stack(modal_features) * softmax(alphas)[:, None, None], 0),
 * model/test/out_Layer.py:52-54, and its autograd: out = sum_m softmax(alphas)_m F_m
 * over M <= 4 matrices of n_elem floats each (n_elem % 4 == 0).  F_host / dF_host are
 * HOST arrays of M device pointers (a dF entry may be NULL).  Backward writes
 * dF_m = softmax(alphas)_m * G and dalpha[M] (gradient w.r.t. the raw alphas). */
int rk_modal_sum_fwd_f32(const float *const *F_host, int32_t M,
                         const float *alphas, int64_t n_elem, float *out,
                         void *stream);
size_t rk_modal_sum_bwd_ws_bytes(void);
int rk_modal_sum_bwd_f32(const float *const *F_host, float *const *dF_host,
                         int32_t M, const float *alphas, const float *G,
                         int64_t n_elem, float *dalpha, void *ws, size_t ws_bytes,
                         void *stream);

/* ---- optimizer half of the training step (device) ------------------------- */
/* Replaces torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm) followed by
 * torch.optim.Adam.step(), train/graph_trainer.py:151-156, for a list of fp32 tensors in two
 * launches (two per 48 tensors).  tensors_host: HOST array of descriptors (device pointers
 * inside); they travel to the kernels in the launch arguments, so the call neither copies nor
 * allocates and can be recorded in a HIP graph.  ws: rk_adam_ws_bytes() bytes of scratch.
 * step: device counter, incremented by the call (bias correction uses the new value); lr: device
 * scalar.  max_norm <= 0: no clipping.  norm_out (may be NULL): the gradients' total 2-norm
 * before clipping.  Update rule and clip factor as torch computes them:
 *   coef = min(1, max_norm / (norm + 1e-6));  g = coef * grad + weight_decay * p
 *   m = m + (g - m)(1 - beta1);  v = beta2 v + (1 - beta2) g^2
 *   p = p - lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)                   */
typedef struct rk_adam_tensor {
  float *param;
  const float *grad;
  float *exp_avg;
  float *exp_avg_sq;
  int64_t n;
} rk_adam_tensor;
size_t rk_adam_ws_bytes(const rk_adam_tensor *tensors_host, int32_t n_tensors);
int rk_adam_clip_step_f32(const rk_adam_tensor *tensors_host, int32_t n_tensors,
                          int64_t *step, const float *lr, float beta1, float beta2,
                          float eps, float weight_decay, float max_norm,
                          float *norm_out, void *ws, size_t ws_bytes, void *stream);
/* The same step for parameters spread over several GPUs (the reference trains in one process:
 * train/graph_trainer.py:133-165 has no counterpart of the exchange).  tensors_host[0 ..
 * n_replicated) hold gradients that are whole and equal on every rank, the tensors behind them
 * are row shards.  rk_adam_sqnorm_f32 writes every chunk's sum of squares into ws, increments
 * `step` and leaves the sum over the SHARDS' chunks in shard_sq_out[0] (device; 0 without
 * shards).  The caller sums that float over the ranks (one all-reduce) and passes it to
 * rk_adam_update_f32 as shard_sq (may be NULL when n_replicated == n_tensors), which clips by the
 * norm over replicated + all shards and updates -- the same arithmetic as rk_adam_clip_step_f32,
 * which is these two calls with every tensor replicated.  Same ws for both calls. */
int rk_adam_sqnorm_f32(const rk_adam_tensor *tensors_host, int32_t n_tensors, int32_t n_replicated,
                       int64_t *step, float *shard_sq_out, void *ws, size_t ws_bytes, void *stream);
int rk_adam_update_f32(const rk_adam_tensor *tensors_host, int32_t n_tensors, int32_t n_replicated,
                       const int64_t *step, const float *lr, float beta1, float beta2, float eps,
                       float weight_decay, float max_norm, const float *shard_sq, float *norm_out,
                       const void *ws, size_t ws_bytes, void *stream);

/* ---- N: negative sampling (host RNG, bit-exact) ------------------------- */
/* Replaces GraphDataLoader._generate_negative_samples,
 * data/graph_data_loader.py:355-397, whose random.sample / random.choices calls
 * draw from CPython's process-global MT19937.  state is 625 words (624 + index).
 * rk_mt_seed follows random.seed(int) for 0 <= seed < 2^64. */
int rk_mt_seed_host(uint32_t *state625_host, uint64_t seed);
/* One batch: for each user in order, the n = n_items - |pos(u)| candidates in the
 * order list(all_items - pos(u)) has in CPython: ascending ids while set.difference
 * copies and removes (len(all_items) >> 2 > |pos(u)|), otherwise the slot order of
 * the freshly built hash table, which is emulated.  Draws R picks exactly as
 * random.sample(cands, R) does (random.choices when n < R) and writes, for every
 * pick, its rank among the candidates in ascending id order (the form the loss
 * kernel resolves on the device) and/or the item id (either output may be NULL).
 * pos_ptr/pos_items: per-user sorted, unique positives over the full interaction
 * table (CSR, host). */
int rk_neg_sample_host(uint32_t *state625_host, const int64_t *users_host,
                       int64_t B, int32_t R, const int32_t *pos_ptr_host,
                       const int32_t *pos_items_host, int32_t n_items,
                       int32_t *rank_out_host, int64_t *neg_out_host);
/* The same with the emulated candidate orders of some users supplied by the caller:
 * a user with |pos(u)| >= n_items >> 2 costs a pass over all item ids per sample, and
 * such users recur in batch after batch.  cached_users: ascending user ids;
 * cached_order[cached_ptr[i] .. cached_ptr[i+1]) = what rk_neg_order_host writes for
 * cached_users[i] (n_items - |pos| ids).  Users not in the cache are handled as above;
 * same random stream, same outputs. */
int rk_neg_order_host(const int32_t *pos_host, int32_t len, int32_t n_items,
                      int32_t *order_out_host);
int rk_neg_sample_cached_host(uint32_t *state625_host, const int64_t *users_host,
                              int64_t B, int32_t R, const int32_t *pos_ptr_host,
                              const int32_t *pos_items_host, int32_t n_items,
                              const int64_t *cached_users_host, const int64_t *cached_ptr_host,
                              const int32_t *cached_order_host, int32_t n_cached,
                              int32_t *rank_out_host, int64_t *neg_out_host);

/* ---- L1: BPR gather-dot loss (device) ----------------------------------- */
/* Replaces GraphTrainer._loss_func, train/graph_trainer.py:255-270, and
 * bpr_loss, evalue/loss.py:7-28:
 *   x[b,r] = <U[u_b], I[i_b]> - <U[u_b], I[n_{b,r}]>
 *   loss = mean(-logsigmoid(x))            (mask == NULL)
 *        = sum(-logsigmoid(x)*m)/max(sum m,1)   (mask != NULL)
 * If neg_rank != NULL the negatives are resolved inside the kernel as the
 * rank-th smallest item id not in pos(u) (device CSR pos_ptr/pos_items) and
 * written to neg (the sampling of graph_data_loader.py:378-387 with the host
 * RNG ranks from rk_neg_sample_host); otherwise neg is an input.
 * x_out: B*R saved differences for the backward pass.
 * loss_out[0] = loss, loss_out[1] = denominator used.
 * ws: at least rk_bpr_ws_bytes(B, R) bytes of scratch. */
size_t rk_bpr_ws_bytes(int64_t B, int32_t R);
int rk_bpr_fwd_f32(const float *U, const float *I, int32_t D,
                   const int64_t *users, const int64_t *pos, int64_t *neg,
                   int64_t B, int32_t R, const float *mask,
                   const int32_t *neg_rank, const int32_t *pos_ptr,
                   const int32_t *pos_items, int32_t n_items, float *x_out,
                   float *loss_out, void *ws, size_t ws_bytes, void *stream);
/* Gradient of the loss above w.r.t. U and I rows, scaled by *gout (device
 * scalar), added into dU/dI (caller zero-fills).  Up to 32768 contributions
 * (B * (1 + R)) every row is summed by one lane group in sample order: the result
 * is bit-reproducible.  Larger batches accumulate with float atomics. */
int rk_bpr_bwd_f32(const float *U, const float *I, int32_t D,
                   const int64_t *users, const int64_t *pos, const int64_t *neg,
                   int64_t B, int32_t R, const float *mask, const float *x,
                   const float *loss_out, const float *gout, float *dU,
                   float *dI, void *stream);

/* rk_bpr_fwd_f32 for callers that cannot allocate device memory or hand over a
 * workspace (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc).  It has no counterpart in the reference.  Same arguments and
 * the same result bits as rk_bpr_fwd_f32 without ws / ws_bytes: the partial sums
 * of the workgroups go to a device array built into the library (32 KiB per GPU,
 * held until the process ends), whose address is looked up once per device on the
 * first call there.  B <= rk_bpr_fwd_noalloc_max_batch() (65,536 samples), otherwise
 * RK_ERR_UNSUPPORTED.  No hipMalloc / hipFree happens in the call.  All calls on one
 * GPU share the array: two calls in flight on different streams of one GPU race
 * (calls on one stream are ordered and safe; nothing needs clearing between them).
 * The FIRST call on a device looks the array's address up (hipGetDevice +
 * hipGetSymbolAddress under a mutex, which may load the code object): make it outside
 * any stream capture -- one eager call per device is the warm-up; every later call
 * enqueues two kernels and nothing else, and can be recorded in a HIP graph. */
int64_t rk_bpr_fwd_noalloc_max_batch(void);
int rk_bpr_fwd_noalloc_f32(const float *U, const float *I, int32_t D,
                           const int64_t *users, const int64_t *pos, int64_t *neg,
                           int64_t B, int32_t R, const float *mask,
                           const int32_t *neg_rank, const int32_t *pos_ptr,
                           const int32_t *pos_items, int32_t n_items,
                           float *x_out, float *loss_out, void *stream);

/* ---- L2: the other pairwise losses over the same gather-dot --------------- */
/* Same arguments as rk_bpr_fwd_f32 / rk_bpr_bwd_f32 plus `mode`:
 *   RK_LOSS_BPR          mean(-logsigmoid(x))                 (= rk_bpr_*)
 *   RK_LOSS_LOG_SIGMOID  mean(-log(sigmoid(x))), the two-step form of mmgcn_loss,
 *                        evalue/loss.py:153-173 (= model/Main_Model.py:672-692), and of the
 *                        SGrec loss, model/Main_Model.py:902-928
 *   RK_LOSS_INFO_BPR     mean_b( logsumexp([pos_b, neg_b,*]) - pos_b ), the softmax
 *                        cross-entropy with the positive as class 0 of
 *                        compute_info_bpr_loss, evalue/loss.py:95-117 (mask must be NULL;
 *                        x_out then holds the negatives' log-probabilities). */
#define RK_LOSS_BPR 0
#define RK_LOSS_LOG_SIGMOID 1
#define RK_LOSS_INFO_BPR 2
int rk_rank_loss_fwd_f32(const float *U, const float *I, int32_t D,
                         const int64_t *users, const int64_t *pos, int64_t *neg,
                         int64_t B, int32_t R, const float *mask,
                         const int32_t *neg_rank, const int32_t *pos_ptr,
                         const int32_t *pos_items, int32_t n_items, int32_t mode,
                         float *x_out, float *loss_out, void *ws, size_t ws_bytes,
                         void *stream);
int rk_rank_loss_bwd_f32(const float *U, const float *I, int32_t D,
                         const int64_t *users, const int64_t *pos,
                         const int64_t *neg, int64_t B, int32_t R,
                         const float *mask, const float *x, const float *loss_out,
                         const float *gout, int32_t mode, float *dU, float *dI,
                         void *stream);

/* Score-level form: bpr_loss(pos_scores[B], neg_scores[B,R], mask),
 * evalue/loss.py:7-28, for callers that computed the scores themselves.
 * ws as rk_bpr_ws_bytes(B, R). */
int rk_bpr_scores_fwd_f32(const float *pos, const float *neg, int64_t B,
                          int32_t R, const float *mask, float *x_out,
                          float *loss_out, void *ws, size_t ws_bytes,
                          void *stream);
int rk_bpr_scores_bwd_f32(const float *x, const float *mask,
                          const float *loss_out, const float *gout, int64_t B,
                          int32_t R, float *dpos, float *dneg, void *stream);

/* ---- E1: fused score + train-mask + top-K (device, fp32 MFMA) ----------- */
/* Replaces torch.matmul(U, I.T); S[row_id, col_id] = -1e10; torch.topk(S, k),
 * evalue/evaluator.py:54-55,98-99 and evalue/metric.py:114, without writing S.
 * Score of (u,i) is the fp32 fma chain over the embedding index in the order
 * 0,32,1,33,...,31,63 within each block of 64 (D % 64 == 0), starting from 0.
 * Ranking: score descending, ties by lower item id.  mask_ptr/mask_items: CSR of
 * the training interactions per user (items ascending, local to this item
 * range), may be NULL.  Item ids written are local index + item_id_offset.
 * out_scores/out_ids: n_users x K, sorted.  1 <= K <= n_items (evalue/metric.py:9-10).  K <= 32 is one
 * per-lane list; a longer ranking (D = 64 or 128 only) is produced 32 places at a time by the
 * exact kernel, every page restricted to the items that rank strictly after the last entry
 * of the page before (one pass over the items per page).
 *
 * All three entry points return the same bits; they differ in how they get there.
 * rk_score_topk_exact_f32 ranks every score on the exact kernel (D = 64 or 128: user rows resident
 * in registers; any other multiple of 64, e.g. the reference's default emb_dim 256: the kernel of
 * rk_score_topk_wide_f32, which walks the width in chunks, with the mask).
 * rk_score_topk_filtered_f32 (K <= 28, D = 64 or 128; otherwise it is the exact one) first ranks
 * by fp16 scores (both tables copied to fp16, one fp16 MFMA per 16 k), picks candidates per user
 * -- the K+8 best by sorted lists in one pass over the scores (long item tables that one slice
 * per user tile covers), or every item at or above a per-user threshold, the (K+8)-th largest of
 * the maxima of disjoint strata of the items (64 per user, or 32 per item slice when a table is cut
 * into several), in two passes (short tables, item shards, small user blocks); RK_FILTER_FORM=lists|threshold in the environment, read at every call, forces a
 * form --, re-scores the candidates with the exact chain, and runs the exact kernel only for
 * users whose result that procedure cannot certify (the fp16 bound on what lies outside the
 * candidates within the fp16 error of the K-th exact score: ties at the K-th place, non-finite
 * embeddings, more than 64 candidates) and for the (at most 512) users with the longest training
 * lists, which the filter passes do not mask.
 * rk_score_topk_f32 picks between the two by size (the filter pays off from 2^27 scores).
 * All take the same workspace. */
size_t rk_score_topk_ws_bytes(int32_t n_users, int32_t n_items, int32_t D,
                              int32_t K);
int rk_score_topk_f32(const float *Uemb, int32_t n_users, const float *Iemb,
                      int32_t n_items, int32_t D, const int32_t *mask_ptr,
                      const int32_t *mask_items, int32_t K,
                      int32_t item_id_offset, float *out_scores,
                      int32_t *out_ids, void *ws, size_t ws_bytes,
                      void *stream);
int rk_score_topk_exact_f32(const float *Uemb, int32_t n_users,
                            const float *Iemb, int32_t n_items, int32_t D,
                            const int32_t *mask_ptr, const int32_t *mask_items,
                            int32_t K, int32_t item_id_offset, float *out_scores,
                            int32_t *out_ids, void *ws, size_t ws_bytes,
                            void *stream);
int rk_score_topk_filtered_f32(const float *Uemb, int32_t n_users,
                               const float *Iemb, int32_t n_items, int32_t D,
                               const int32_t *mask_ptr,
                               const int32_t *mask_items, int32_t K,
                               int32_t item_id_offset, float *out_scores,
                               int32_t *out_ids, void *ws, size_t ws_bytes,
                               void *stream);
/* Diagnostic, no counterpart in the reference: how many users the LAST
 * rk_score_topk_filtered_f32 call that used this workspace (same sizes) left to the exact
 * kernel -- uncertified results plus the users with the longest training lists.  Waits for
 * `stream`, then copies one counter to *open_users (host).  The results do not depend on it;
 * the time of a call does: a filter whose thresholds are too weak still returns the exact
 * kernel's bits, only slowly, and this number is how the tests notice.  0 is written if the
 * call took the exact path (K > 28, widths other than 64/128). */
int rk_score_topk_open_users(int32_t n_users, int32_t n_items, int32_t D,
                             int32_t K, const void *ws, size_t ws_bytes,
                             void *stream, int32_t *open_users);
/* Merge P candidate lists per user (each Kin long, in any order, P * Kin <= 1024;
 * padding entries have id < 0) into the best K by (score desc, id asc).  Used for
 * the item slices inside one GPU and for the per-GPU lists of a user slice after
 * the exchange between item-sharded ranks. */
int rk_topk_merge_f32(const float *scores, const int32_t *ids, int32_t n_users,
                      int32_t P, int32_t Kin, int32_t K, float *out_scores,
                      int32_t *out_ids, void *stream);

/* ---- M: top-K of an existing score matrix (device) ------------------------ */
/* Replaces torch.topk(scores, k, dim=1, largest=True, sorted=True) of
 * evaluate_all_at_k, evalue/metric.py:114, for callers that hand over the dense
 * scores[n_rows, n_items] (row stride ld floats).  Ranking: score descending, ties by
 * lower column id.  out_scores/out_ids: n_rows x K.  1 <= K <= n_items (above 32: pages
 * of 32 places, one scan of the scores per page). */
size_t rk_rows_topk_ws_bytes(int32_t n_rows, int32_t K);
int rk_rows_topk_f32(const float *scores, int32_t n_rows, int32_t n_items,
                     int64_t ld, int32_t K, float *out_scores, int32_t *out_ids,
                     void *ws, size_t ws_bytes, void *stream);

/* ---- (f).4: similarity + top-K over wide feature rows (device) ------------- */
/* Replaces the dense pipeline of the item-item kNN graph, model/sgrec/graph.py:
 * 100-105 (score = normalized_feat @ normalized_feat.T; torch.topk(score, k)):
 * out[a] = the K largest <A[a, :], B[b, :]> over b (score descending, ties by
 * lower b) for A[n_a, F], B[n_b, F] row-major fp32, F a multiple of 64 (pad with
 * zero columns), without holding any block of the score matrix.  Scores are exact
 * fp32 products accumulated in fp32 along the width.  No mask: a row of A that is
 * also a row of B ranks itself, as the reference's graph does.
 * 1 <= K <= min(n_b, 32).  ws: rk_score_topk_wide_ws_bytes(). */
size_t rk_score_topk_wide_ws_bytes(int32_t n_a, int32_t n_b, int32_t K);
int rk_score_topk_wide_f32(const float *A, int32_t n_a, const float *B,
                           int32_t n_b, int32_t F, int32_t K, float *out_scores,
                           int32_t *out_ids, void *ws, size_t ws_bytes,
                           void *stream);

/* ---- M: ranking metrics (device) ---------------------------------------- */
/* Replaces evaluate_all_at_k, evalue/metric.py:106-164, for every k in k_list
 * at once from one sorted top-Kmax id list per user and the CSR of the target
 * (validation/test) items per user (sorted).  discounts[j] = 1/log2(j+2).  Up to 8 values of k,
 * each in [1, id_stride] (up to 128 the ranked ids of a user sit in registers; longer rankings are
 * walked from memory, one place after the other).
 * out[ki*6 + m], m = precision, recall, hit_rate, ndcg, map, mrr: means over
 * all n_users rows.  sums_out (may be NULL): the same as double sums before
 * the division (for cross-GPU reduction).  ws: rk_rank_metrics_ws_bytes(). */
size_t rk_rank_metrics_ws_bytes(int32_t n_users, int32_t nk);
int rk_rank_metrics_f32(const int32_t *topk_ids, int32_t n_users,
                        int32_t id_stride, const int32_t *tgt_ptr,
                        const int32_t *tgt_items, const int32_t *k_list_host,
                        int32_t nk, const float *discounts, float *out,
                        double *sums_out, int64_t mean_denominator, void *ws,
                        size_t ws_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RECOMMEND_AMD_H */
