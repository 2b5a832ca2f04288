// Graph normalisation and CSR SpMM kernels (gfx950).
//
// SpMM layout: one lane group of LPR = D/4 lanes owns one destination row and
// every lane keeps a float4 slice of the 4*LPR-wide embedding row, so each
// gathered source row is one fully coalesced D*4-byte read.  Edge (col, val)
// pairs are loaded LPR at a time by the group (one coalesced read) and then
// broadcast lane to lane with shuffles.  Rows longer than plan->long_thresh
// are cut into fixed chunks that whole wavefronts reduce into a scratch slab;
// a second small launch sums the chunk partials in chunk order, so the result
// is deterministic for any row-length distribution.
#include "common.h"

namespace {

using rk::kWave;

// ------------------------------------------------------------- G2 ---------
__global__ void in_weight_sqrt_kernel(const int32_t *__restrict__ rowptr,
                                      const int32_t *__restrict__ eid,
                                      const float *__restrict__ w_coo, int32_t n,
                                      float *__restrict__ sq) {
  const int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  float s = 0.0f;
  for (int32_t p = rowptr[v]; p < rowptr[v + 1]; ++p) s += w_coo[eid[p]];
  // torch.clamp(min=1e-12).sqrt(), graph_constructor.py:178
  sq[v] = sqrtf(fmaxf(s, 1e-12f));
}

__global__ void sym_norm_kernel(const int32_t *__restrict__ rowptr,
                                const int32_t *__restrict__ col,
                                const int32_t *__restrict__ eid,
                                const float *__restrict__ w_coo,
                                const float *__restrict__ sq, int32_t n,
                                float *__restrict__ w_norm) {
  const int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  const float sv = sq[v];
  for (int32_t p = rowptr[v]; p < rowptr[v + 1]; ++p) {
    const int32_t e = eid[p];
    // w / (in_w[src] * in_w[dst]), graph_constructor.py:186
    w_norm[e] = w_coo[e] / (sq[col[p]] * sv);
  }
}

__global__ void deg_pow_kernel(const int32_t *__restrict__ rowptr, int32_t n,
                               float *__restrict__ dinv) {
  const int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  // in_degrees().float().pow(-0.5), model/sgrec/graph.py:126-128
  dinv[v] = powf((float)(rowptr[v + 1] - rowptr[v]), -0.5f);
}

__global__ void deg_norm_kernel(const int32_t *__restrict__ rowptr,
                                const int32_t *__restrict__ col,
                                const int32_t *__restrict__ eid,
                                int32_t n, float *__restrict__ w_norm) {
  const int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  const float dv = powf((float)(rowptr[v + 1] - rowptr[v]), -0.5f);
  for (int32_t p = rowptr[v]; p < rowptr[v + 1]; ++p) {
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

// ------------------------------------------------------------- G3 ---------
__device__ __forceinline__ void fma4(float4 &a, float v, const float4 &x) {
  a.x = fmaf(v, x.x, a.x);
  a.y = fmaf(v, x.y, a.y);
  a.z = fmaf(v, x.z, a.z);
  a.w = fmaf(v, x.w, a.w);
}

// Accumulate edges [b, e) of one row into acc; the LPR lanes of the group call
// this together.  `sub` is the lane's index inside its group.
template <int LPR>
__device__ __forceinline__ void accumulate_edges(const int32_t *__restrict__ col,
                                                 const float *__restrict__ val,
                                                 const float4 *__restrict__ h4,
                                                 int32_t b, int32_t e, int sub,
                                                 float4 &acc) {
  for (int32_t e0 = b; e0 < e; e0 += LPR) {
    const int32_t mine = e0 + sub;
    int32_t c_l = 0;
    float v_l = 0.0f;
    if (mine < e) {
      c_l = col[mine];
      v_l = val[mine];
    }
    const int cnt = min(LPR, e - e0);
    int j = 0;
    for (; j + 4 <= cnt; j += 4) {
      const int32_t c0 = __shfl(c_l, j, LPR), c1 = __shfl(c_l, j + 1, LPR);
      const int32_t c2 = __shfl(c_l, j + 2, LPR), c3 = __shfl(c_l, j + 3, LPR);
      const float4 x0 = h4[(int64_t)c0 * LPR + sub];
      const float4 x1 = h4[(int64_t)c1 * LPR + sub];
      const float4 x2 = h4[(int64_t)c2 * LPR + sub];
      const float4 x3 = h4[(int64_t)c3 * LPR + sub];
      fma4(acc, __shfl(v_l, j, LPR), x0);
      fma4(acc, __shfl(v_l, j + 1, LPR), x1);
      fma4(acc, __shfl(v_l, j + 2, LPR), x2);
      fma4(acc, __shfl(v_l, j + 3, LPR), x3);
    }
    for (; j < cnt; ++j) {
      const int32_t c0 = __shfl(c_l, j, LPR);
      const float4 x0 = h4[(int64_t)c0 * LPR + sub];
      fma4(acc, __shfl(v_l, j, LPR), x0);
    }
  }
}

template <int LPR, bool AXPBY>
__device__ __forceinline__ void store_row(float4 *__restrict__ out4,
                                          const float4 *__restrict__ h04,
                                          int64_t row, int sub, float4 acc,
                                          float alpha, float beta) {
  if (AXPBY) {
    const float4 z = h04[row * LPR + sub];
    acc.x = alpha * z.x + beta * acc.x;
    acc.y = alpha * z.y + beta * acc.y;
    acc.z = alpha * z.z + beta * acc.z;
    acc.w = alpha * z.w + beta * acc.w;
  }
  out4[row * LPR + sub] = acc;
}

// Blocks [0, n_row_blocks) own 256/LPR consecutive rows each; blocks after
// that own one long-row chunk per wavefront.
template <int LPR, bool AXPBY>
__global__ __launch_bounds__(256) void spmm_rows_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, int32_t n_rows,
    const float4 *__restrict__ h4, float4 *__restrict__ out4,
    const float4 *__restrict__ h04, float alpha, float beta,
    int32_t long_thresh, int32_t n_row_blocks,
    const int32_t *__restrict__ chunk_begin,
    const int32_t *__restrict__ chunk_end, int32_t n_chunks,
    float4 *__restrict__ partial4) {
  constexpr int GROUPS = 256 / LPR;  // rows per block
  const int tid = threadIdx.x;
  const int sub = tid % LPR;
  if ((int32_t)blockIdx.x < n_row_blocks) {
    const int64_t row = (int64_t)blockIdx.x * GROUPS + tid / LPR;
    if (row >= n_rows) return;
    const int32_t b = rowptr[row], e = rowptr[row + 1];
    if (e - b > long_thresh) return;  // written by spmm_long_reduce_kernel
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    accumulate_edges<LPR>(col, val, h4, b, e, sub, acc);
    store_row<LPR, AXPBY>(out4, h04, row, sub, acc, alpha, beta);
    return;
  }
  // long-row chunk: one wavefront per chunk, its 64/LPR groups stride the chunk
  constexpr int GPW = kWave / LPR;  // groups per wave
  const int wave = tid / kWave;
  const int32_t chunk = ((int32_t)blockIdx.x - n_row_blocks) * (256 / kWave) + wave;
  if (chunk >= n_chunks) return;
  const int lane = tid % kWave;
  const int g = lane / LPR;
  const int32_t cb = chunk_begin[chunk], ce = chunk_end[chunk];
  // group g takes edges [cb + g*span, cb + (g+1)*span)
  const int32_t span = ((ce - cb + GPW - 1) / GPW + LPR - 1) / LPR * LPR;
  const int32_t b = min(cb + g * span, ce), e = min(b + span, ce);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  accumulate_edges<LPR>(col, val, h4, b, e, sub, acc);
  // fold the groups of the wave in a fixed order
#pragma unroll
  for (int off = LPR; off < kWave; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off, kWave);
    acc.y += __shfl_xor(acc.y, off, kWave);
    acc.z += __shfl_xor(acc.z, off, kWave);
    acc.w += __shfl_xor(acc.w, off, kWave);
  }
  if (g == 0) partial4[(int64_t)chunk * LPR + sub] = acc;
}

template <int LPR, bool AXPBY>
__global__ __launch_bounds__(256) void spmm_long_reduce_kernel(
    const int32_t *__restrict__ long_rows,
    const int32_t *__restrict__ long_chunk_ptr, int32_t n_long,
    const float4 *__restrict__ partial4, float4 *__restrict__ out4,
    const float4 *__restrict__ h04, float alpha, float beta) {
  constexpr int GROUPS = 256 / LPR;
  const int sub = threadIdx.x % LPR;
  const int32_t i = blockIdx.x * GROUPS + threadIdx.x / LPR;
  if (i >= n_long) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int32_t c = long_chunk_ptr[i]; c < long_chunk_ptr[i + 1]; ++c) {
    const float4 p = partial4[(int64_t)c * LPR + sub];
    acc.x += p.x;
    acc.y += p.y;
    acc.z += p.z;
    acc.w += p.w;
  }
  store_row<LPR, AXPBY>(out4, h04, long_rows[i], sub, acc, alpha, beta);
}

template <int LPR, bool AXPBY>
int launch_spmm(const int32_t *rowptr, const int32_t *col, const float *val,
                int32_t n_rows, const float *h, float *out, const float *h0,
                float alpha, float beta, const rk_spmm_plan *plan,
                hipStream_t st) {
  constexpr int GROUPS = 256 / LPR;
  const int32_t n_row_blocks = (int32_t)rk::ceil_div(n_rows, GROUPS);
  const bool has_long = plan && plan->n_long > 0;
  const int32_t n_chunks = has_long ? plan->n_chunks : 0;
  const int32_t chunk_blocks = (int32_t)rk::ceil_div(n_chunks, 256 / kWave);
  const int32_t thresh = has_long ? plan->long_thresh : INT32_MAX;
  if (n_row_blocks + chunk_blocks == 0) return RK_OK;
  spmm_rows_kernel<LPR, AXPBY><<<n_row_blocks + chunk_blocks, 256, 0, st>>>(
      rowptr, col, val, n_rows, reinterpret_cast<const float4 *>(h),
      reinterpret_cast<float4 *>(out), reinterpret_cast<const float4 *>(h0),
      alpha, beta, thresh, n_row_blocks, has_long ? plan->chunk_begin : nullptr,
      has_long ? plan->chunk_end : nullptr, n_chunks,
      has_long ? reinterpret_cast<float4 *>(plan->partial) : nullptr);
  RK_CHECK_LAUNCH("spmm_rows_kernel");
  if (has_long) {
    spmm_long_reduce_kernel<LPR, AXPBY>
        <<<(int32_t)rk::ceil_div(plan->n_long, GROUPS), 256, 0, st>>>(
            plan->long_rows, plan->long_chunk_ptr, plan->n_long,
            reinterpret_cast<const float4 *>(plan->partial),
            reinterpret_cast<float4 *>(out),
            reinterpret_cast<const float4 *>(h0), alpha, beta);
    RK_CHECK_LAUNCH("spmm_long_reduce_kernel");
  }
  return RK_OK;
}

template <int LPR>
int dispatch_axpby(const int32_t *rowptr, const int32_t *col, const float *val,
                   int32_t n_rows, const float *h, float *out, const float *h0,
                   float alpha, float beta, const rk_spmm_plan *plan,
                   hipStream_t st) {
  return h0 ? launch_spmm<LPR, true>(rowptr, col, val, n_rows, h, out, h0, alpha, beta, plan, st)
            : launch_spmm<LPR, false>(rowptr, col, val, n_rows, h, out, h0, alpha, beta, plan, st);
}

}  // namespace

extern "C" {

int rk_sym_normalize_f32(const int32_t *rowptr, const int32_t *col,
                         const int32_t *eid, const float *w_coo, int32_t n_nodes,
                         float *sqrt_in_w, float *w_norm_coo, void *stream) {
  RK_CHECK_ARG(rowptr && col && eid && w_coo && sqrt_in_w && w_norm_coo, "null pointer");
  RK_CHECK_ARG(n_nodes >= 0, "negative node count");
  if (n_nodes == 0) return RK_OK;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (int)rk::ceil_div(n_nodes, 256);
  in_weight_sqrt_kernel<<<blocks, 256, 0, st>>>(rowptr, eid, w_coo, n_nodes, sqrt_in_w);
  RK_CHECK_LAUNCH("in_weight_sqrt_kernel");
  sym_norm_kernel<<<blocks, 256, 0, st>>>(rowptr, col, eid, w_coo, sqrt_in_w, n_nodes, w_norm_coo);
  RK_CHECK_LAUNCH("sym_norm_kernel");
  return RK_OK;
}

int rk_deg_normalize_f32(const int32_t *rowptr, const int32_t *col,
                         const int32_t *eid, int32_t n_nodes, float *w_norm_coo,
                         void *stream) {
  RK_CHECK_ARG(rowptr && col && eid && w_norm_coo, "null pointer");
  RK_CHECK_ARG(n_nodes >= 0, "negative node count");
  if (n_nodes == 0) return RK_OK;
  deg_norm_kernel<<<(int)rk::ceil_div(n_nodes, 256), 256, 0, (hipStream_t)stream>>>(
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

int rk_spmm_csr_f32(const int32_t *rowptr, const int32_t *col, const float *val,
                    int32_t n_rows, const float *h, int32_t D, float *out,
                    const float *h0, float alpha, float beta,
                    const rk_spmm_plan *plan, void *stream) {
  RK_CHECK_ARG(n_rows >= 0, "negative row count");
  if (n_rows == 0) return RK_OK;
  RK_CHECK_ARG(rowptr && col && val && h && out, "null pointer");
  RK_CHECK_ARG(out != h, "out must not alias h");
  RK_CHECK_ARG(rk::aligned16(h) && rk::aligned16(out) && rk::aligned16(h0),
               "embedding matrices must be 16-byte aligned");
  if (plan && plan->n_long > 0) {
    RK_CHECK_ARG(plan->long_rows && plan->long_chunk_ptr && plan->chunk_begin &&
                     plan->chunk_end && plan->partial && plan->n_chunks > 0 &&
                     plan->long_thresh > 0,
                 "incomplete spmm plan");
    RK_CHECK_ARG(rk::aligned16(plan->partial), "plan->partial must be 16-byte aligned");
  }
  hipStream_t st = (hipStream_t)stream;
  switch (D) {
    case 16: return dispatch_axpby<4>(rowptr, col, val, n_rows, h, out, h0, alpha, beta, plan, st);
    case 32: return dispatch_axpby<8>(rowptr, col, val, n_rows, h, out, h0, alpha, beta, plan, st);
    case 64: return dispatch_axpby<16>(rowptr, col, val, n_rows, h, out, h0, alpha, beta, plan, st);
    case 128: return dispatch_axpby<32>(rowptr, col, val, n_rows, h, out, h0, alpha, beta, plan, st);
    case 256: return dispatch_axpby<64>(rowptr, col, val, n_rows, h, out, h0, alpha, beta, plan, st);
    default:
      return rk::fail(RK_ERR_UNSUPPORTED,
                      "rk_spmm_csr_f32: embedding width %d not built (16/32/64/128/256)", D);
  }
}

}  // extern "C"
