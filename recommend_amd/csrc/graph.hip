// Graph normalisation kernels (gfx950): incoming-weight sums, symmetric and
// degree normalisation of the edge values, and the COO -> CSR value permutation.
// One-time set-up work, one thread per destination row.
#include "common.h"

namespace {

using rk::kWave;

// ------------------------------------------------------------- G2 ---------
// Rows up to kLongRow edges: one thread per row, edges summed in CSR (= edge id) order.  Longer
// rows (popular items reach 10^5 edges) would serialise a whole launch behind one thread, so a
// second launch gives each of them a wavefront: lane-strided partial sums folded by a fixed
// shuffle tree.  Both orders give the same bits whenever the sum is exact in fp32 (integer
// ratings, the case the reference's data produces); otherwise they differ in the last place.
constexpr int32_t kLongRow = 512;

__global__ void in_weight_sqrt_kernel(const int32_t *__restrict__ rowptr,
                                      const int32_t *__restrict__ eid,
                                      const float *__restrict__ w_coo, int32_t n,
                                      float *__restrict__ sq) {
  const int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  const int32_t lo = rowptr[v], hi = rowptr[v + 1];
  if (hi - lo > kLongRow) return;
  float s = 0.0f;
  for (int32_t p = lo; p < hi; ++p) s += w_coo[eid[p]];
  // torch.clamp(min=1e-12).sqrt(), graph_constructor.py:178
  sq[v] = sqrtf(fmaxf(s, 1e-12f));
}

__global__ __launch_bounds__(256) void in_weight_sqrt_long_kernel(const int32_t *__restrict__ rowptr,
                                                                  const int32_t *__restrict__ eid,
                                                                  const float *__restrict__ w_coo, int32_t n,
                                                                  float *__restrict__ sq) {
  const int32_t v = blockIdx.x * 4 + threadIdx.x / kWave;
  const int lane = threadIdx.x % kWave;
  if (v >= n) return;
  const int32_t lo = rowptr[v], hi = rowptr[v + 1];
  if (hi - lo <= kLongRow) return;
  float s = 0.0f;
  for (int32_t p = lo + lane; p < hi; p += kWave) s += w_coo[eid[p]];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, kWave);
  if (lane == 0) sq[v] = sqrtf(fmaxf(s, 1e-12f));
}

// w / (in_w[src] * in_w[dst]), graph_constructor.py:186: independent per edge
__global__ void sym_norm_kernel(const int32_t *__restrict__ rowptr,
                                const int32_t *__restrict__ col,
                                const int32_t *__restrict__ eid,
                                const float *__restrict__ w_coo,
                                const float *__restrict__ sq, int32_t n, int32_t row_offset,
                                float *__restrict__ w_norm) {
  const int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  const int32_t lo = rowptr[v], hi = rowptr[v + 1];
  if (hi - lo > kLongRow) return;
  const float sv = sq[v + row_offset];
  for (int32_t p = lo; p < hi; ++p) {
    const int32_t e = eid[p];
    w_norm[e] = w_coo[e] / (sq[col[p]] * sv);
  }
}

__global__ __launch_bounds__(256) void sym_norm_long_kernel(const int32_t *__restrict__ rowptr,
                                                            const int32_t *__restrict__ col,
                                                            const int32_t *__restrict__ eid,
                                                            const float *__restrict__ w_coo,
                                                            const float *__restrict__ sq, int32_t n,
                                                            int32_t row_offset, float *__restrict__ w_norm) {
  const int32_t v = blockIdx.x * 4 + threadIdx.x / kWave;
  const int lane = threadIdx.x % kWave;
  if (v >= n) return;
  const int32_t lo = rowptr[v], hi = rowptr[v + 1];
  if (hi - lo <= kLongRow) return;
  const float sv = sq[v + row_offset];
  for (int32_t p = lo + lane; p < hi; p += kWave) {
    const int32_t e = eid[p];
    w_norm[e] = w_coo[e] / (sq[col[p]] * sv);
  }
}

// ------------------------------------------------- feature-similarity edge weights -----
// One wavefront per edge: the dot product of the two endpoints' feature rows and both squared
// norms in one sweep (float4 per lane, lane-strided, a fixed shuffle tree: bitwise reproducible).
// Nodes outside [row_lo, row_lo + n_rows) carry all-zero features (the zero padding of
// graph_constructor.py:130-135): their rows are never read, dot and norm are 0.
// mode 0: clamp(dot / ((|a| + 1e-12) (|b| + 1e-12)), 0, 1); mode 1: sigmoid(dot).
__global__ __launch_bounds__(256) void edge_sim_kernel(const int32_t *__restrict__ src,
                                                       const int32_t *__restrict__ dst, int64_t nnz,
                                                       const float *__restrict__ X, int32_t row_lo, int32_t n_rows,
                                                       int32_t F, int mode, float *__restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 4 + threadIdx.x / kWave;
  const int lane = threadIdx.x % kWave;
  if (e >= nnz) return;
  const int32_t a = src[e] - row_lo, b = dst[e] - row_lo;
  const bool in_a = a >= 0 && a < n_rows, in_b = b >= 0 && b < n_rows;
  float dot = 0.0f, na = 0.0f, nb = 0.0f;
  if (in_a && in_b) {
    const float4 *pa = reinterpret_cast<const float4 *>(X + (int64_t)a * F);
    const float4 *pb = reinterpret_cast<const float4 *>(X + (int64_t)b * F);
    for (int q = lane; q < F / 4; q += kWave) {
      const float4 x = pa[q], y = pb[q];
      dot = fmaf(x.x, y.x, dot); dot = fmaf(x.y, y.y, dot); dot = fmaf(x.z, y.z, dot); dot = fmaf(x.w, y.w, dot);
      na = fmaf(x.x, x.x, na); na = fmaf(x.y, x.y, na); na = fmaf(x.z, x.z, na); na = fmaf(x.w, x.w, na);
      nb = fmaf(y.x, y.x, nb); nb = fmaf(y.y, y.y, nb); nb = fmaf(y.z, y.z, nb); nb = fmaf(y.w, y.w, nb);
    }
  } else if (mode == 0 && (in_a || in_b)) {   // only the norm of the endpoint that has features (dot = 0 anyway)
    const float4 *p = reinterpret_cast<const float4 *>(X + (int64_t)(in_a ? a : b) * F);
    for (int q = lane; q < F / 4; q += kWave) {
      const float4 x = p[q];
      na = fmaf(x.x, x.x, na); na = fmaf(x.y, x.y, na); na = fmaf(x.z, x.z, na); na = fmaf(x.w, x.w, na);
    }
    if (!in_a) { nb = na; na = 0.0f; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    dot += __shfl_xor(dot, off, kWave);
    na += __shfl_xor(na, off, kWave);
    nb += __shfl_xor(nb, off, kWave);
  }
  if (lane != 0) return;
  float w;
  if (mode == 0) {
    const float sim = dot / ((sqrtf(na) + 1e-12f) * (sqrtf(nb) + 1e-12f));
    w = fminf(fmaxf(sim, 0.0f), 1.0f);
    if (sim != sim) w = sim;      // torch.clamp keeps a NaN
  } else {
    w = 1.0f / (1.0f + expf(-dot));
  }
  out[e] = w;
}

__global__ void deg_pow_kernel(const int32_t *__restrict__ rowptr, int32_t n,
                               float *__restrict__ dinv) {
  const int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  // in_degrees().float().pow(-0.5), model/sgrec/graph.py:126-128
  dinv[v] = powf((float)(rowptr[v + 1] - rowptr[v]), -0.5f);
}

// one wavefront per destination row, lanes over its edges (independent per edge)
__global__ __launch_bounds__(256) void deg_norm_kernel(const int32_t *__restrict__ rowptr,
                                                       const int32_t *__restrict__ col,
                                                       const int32_t *__restrict__ eid,
                                                       int32_t n, float *__restrict__ w_norm) {
  const int32_t v = blockIdx.x * 4 + threadIdx.x / kWave;
  const int lane = threadIdx.x % kWave;
  if (v >= n) return;
  const int32_t lo = rowptr[v], hi = rowptr[v + 1];
  const float dv = powf((float)(hi - lo), -0.5f);
  for (int32_t p = lo + lane; p < hi; p += kWave) {
    const int32_t u = col[p];
    const float du = powf((float)(rowptr[u + 1] - rowptr[u]), -0.5f);
    w_norm[eid[p]] = du * dv;
  }
}

__global__ void gather_kernel(const float *__restrict__ src,
                              const int32_t *__restrict__ idx, int64_t n,
                              float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = src[idx[i]];
}

int launch_in_weight_sqrt(const int32_t *rowptr, const int32_t *eid, const float *w_coo, int32_t n_rows,
                          float *sq_rows, hipStream_t st) {
  in_weight_sqrt_kernel<<<(int)rk::ceil_div(n_rows, 256), 256, 0, st>>>(rowptr, eid, w_coo, n_rows, sq_rows);
  RK_CHECK_LAUNCH("in_weight_sqrt_kernel");
  in_weight_sqrt_long_kernel<<<(int)rk::ceil_div(n_rows, 4), 256, 0, st>>>(rowptr, eid, w_coo, n_rows, sq_rows);
  RK_CHECK_LAUNCH("in_weight_sqrt_long_kernel");
  return RK_OK;
}

int launch_sym_scale(const int32_t *rowptr, const int32_t *col, const int32_t *eid, const float *w_coo,
                     const float *sq, int32_t n_rows, int32_t row_offset, float *w_norm, hipStream_t st) {
  sym_norm_kernel<<<(int)rk::ceil_div(n_rows, 256), 256, 0, st>>>(rowptr, col, eid, w_coo, sq, n_rows, row_offset, w_norm);
  RK_CHECK_LAUNCH("sym_norm_kernel");
  sym_norm_long_kernel<<<(int)rk::ceil_div(n_rows, 4), 256, 0, st>>>(rowptr, col, eid, w_coo, sq, n_rows, row_offset,
                                                                  w_norm);
  RK_CHECK_LAUNCH("sym_norm_long_kernel");
  return RK_OK;
}

}  // namespace

extern "C" {

int rk_in_weight_sqrt_f32(const int32_t *rowptr, const int32_t *eid, const float *w_coo, int32_t n_rows,
                          float *sqrt_in_w_rows, void *stream) {
  RK_CHECK_ARG(n_rows >= 0, "negative row count");
  if (n_rows == 0) return RK_OK;
  RK_CHECK_ARG(rowptr && eid && w_coo && sqrt_in_w_rows, "null pointer");
  return launch_in_weight_sqrt(rowptr, eid, w_coo, n_rows, sqrt_in_w_rows, (hipStream_t)stream);
}

int rk_sym_scale_rows_f32(const int32_t *rowptr, const int32_t *col, const int32_t *eid, const float *w_coo,
                          int32_t n_rows, int32_t row_offset, const float *sqrt_in_w, float *w_norm_coo,
                          void *stream) {
  RK_CHECK_ARG(n_rows >= 0 && row_offset >= 0, "negative row count or offset");
  if (n_rows == 0) return RK_OK;
  RK_CHECK_ARG(rowptr && col && eid && w_coo && sqrt_in_w && w_norm_coo, "null pointer");
  return launch_sym_scale(rowptr, col, eid, w_coo, sqrt_in_w, n_rows, row_offset, w_norm_coo, (hipStream_t)stream);
}

int rk_sym_normalize_f32(const int32_t *rowptr, const int32_t *col,
                         const int32_t *eid, const float *w_coo, int32_t n_nodes,
                         float *sqrt_in_w, float *w_norm_coo, void *stream) {
  RK_CHECK_ARG(rowptr && col && eid && w_coo && sqrt_in_w && w_norm_coo, "null pointer");
  RK_CHECK_ARG(n_nodes >= 0, "negative node count");
  if (n_nodes == 0) return RK_OK;
  hipStream_t st = (hipStream_t)stream;
  const int rc = launch_in_weight_sqrt(rowptr, eid, w_coo, n_nodes, sqrt_in_w, st);
  if (rc != RK_OK) return rc;
  return launch_sym_scale(rowptr, col, eid, w_coo, sqrt_in_w, n_nodes, 0, w_norm_coo, st);
}

int rk_edge_feature_sim_f32(const int32_t *src, const int32_t *dst, int64_t nnz, const float *X, int32_t row_lo,
                            int32_t n_rows, int32_t F, int32_t mode, float *out, void *stream) {
  RK_CHECK_ARG(nnz >= 0 && n_rows >= 0 && row_lo >= 0, "negative size");
  RK_CHECK_ARG(mode == RK_EDGE_SIM_COSINE || mode == RK_EDGE_SIM_SIGMOID_DOT, "unknown similarity mode");
  RK_CHECK_ARG(F > 0 && F % 4 == 0, "feature width must be a positive multiple of 4");
  if (nnz == 0) return RK_OK;
  RK_CHECK_ARG(src && dst && out && (n_rows == 0 || X), "null pointer");
  RK_CHECK_ARG(rk::aligned16(X), "feature rows must be 16-byte aligned");
  edge_sim_kernel<<<(unsigned)rk::ceil_div(nnz, 4), 256, 0, (hipStream_t)stream>>>(src, dst, nnz, X, row_lo, n_rows, F,
                                                                                    mode, out);
  RK_CHECK_LAUNCH("edge_sim_kernel");
  return RK_OK;
}

int rk_deg_normalize_f32(const int32_t *rowptr, const int32_t *col,
                         const int32_t *eid, int32_t n_nodes, float *w_norm_coo,
                         void *stream) {
  RK_CHECK_ARG(rowptr && col && eid && w_norm_coo, "null pointer");
  RK_CHECK_ARG(n_nodes >= 0, "negative node count");
  if (n_nodes == 0) return RK_OK;
  deg_norm_kernel<<<(int)rk::ceil_div(n_nodes, 4), 256, 0, (hipStream_t)stream>>>(
      rowptr, col, eid, n_nodes, w_norm_coo);
  RK_CHECK_LAUNCH("deg_norm_kernel");
  return RK_OK;
}

int rk_gather_f32(const float *src, const int32_t *idx, int64_t n, float *out,
                  void *stream) {
  RK_CHECK_ARG(n >= 0, "negative length");
  if (n == 0) return RK_OK;
  RK_CHECK_ARG(src && idx && out, "null pointer");
  gather_kernel<<<(unsigned)rk::ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(src, idx, n, out);
  RK_CHECK_LAUNCH("gather_kernel");
  return RK_OK;
}

}  // extern "C"
