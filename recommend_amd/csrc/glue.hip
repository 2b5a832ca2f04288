// Modality-weighted sum of the per-modality node features and its backward (gfx950):
//   out = sum_m softmax(alphas)_m * F_m          (model/test/out_Layer.py:52-54)
// One pass over the M feature matrices instead of a stack, M multiplies and a reduction;
// the backward pass writes dF_m = s_m * G and the M dot products <G, F_m> that the softmax
// gradient needs in the same sweep (per-workgroup partials added in workgroup order).
#include "common.h"

namespace {

using rk::kWave;
constexpr int kMaxModal = 4;
constexpr int kWsumBlocks = 1024;

struct ModalPtrs {
  const float *f[kMaxModal];
};
struct ModalOut {
  float *d[kMaxModal];
};

// softmax over the M raw weights, in the order torch.softmax evaluates it (max-shifted)
__device__ __forceinline__ void softmax_small(const float *__restrict__ alphas, int M, float (&s)[kMaxModal]) {
  float a[kMaxModal];
#pragma unroll
  for (int m = 0; m < kMaxModal; ++m) a[m] = m < M ? alphas[m] : -INFINITY;
  const float mx = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]));
  float sum = 0.0f;
#pragma unroll
  for (int m = 0; m < kMaxModal; ++m) {
    s[m] = m < M ? expf(a[m] - mx) : 0.0f;
    sum += s[m];
  }
#pragma unroll
  for (int m = 0; m < kMaxModal; ++m) s[m] /= sum;
}

__global__ __launch_bounds__(256) void wsum_fwd_kernel(ModalPtrs F, int M, const float *__restrict__ alphas,
                                                       int64_t n4, float *__restrict__ out) {
  float s[kMaxModal];
  softmax_small(alphas, M, s);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int m = 0; m < kMaxModal; ++m) {   // (static indices keep the pointer table in registers)
      if (m >= M) break;
      const float4 v = reinterpret_cast<const float4 *>(F.f[m])[i];
      // F_0 * s_0, then + F_m * s_m: the order of the reference's sum over the stacked dimension
      acc.x = m == 0 ? v.x * s[0] : acc.x + v.x * s[m];
      acc.y = m == 0 ? v.y * s[0] : acc.y + v.y * s[m];
      acc.z = m == 0 ? v.z * s[0] : acc.z + v.z * s[m];
      acc.w = m == 0 ? v.w * s[0] : acc.w + v.w * s[m];
    }
    reinterpret_cast<float4 *>(out)[i] = acc;
  }
}

__global__ __launch_bounds__(256) void wsum_bwd_kernel(ModalPtrs F, ModalOut dF, int M,
                                                       const float *__restrict__ alphas,
                                                       const float *__restrict__ G, int64_t n4,
                                                       float *__restrict__ part) {
  __shared__ float red[4][kMaxModal];
  float s[kMaxModal];
  softmax_small(alphas, M, s);
  float dot[kMaxModal] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 g = reinterpret_cast<const float4 *>(G)[i];
#pragma unroll
    for (int m = 0; m < kMaxModal; ++m) {
      if (m >= M) break;
      const float4 v = reinterpret_cast<const float4 *>(F.f[m])[i];
      dot[m] += (g.x * v.x + g.y * v.y) + (g.z * v.z + g.w * v.w);
      if (dF.d[m]) reinterpret_cast<float4 *>(dF.d[m])[i] = make_float4(g.x * s[m], g.y * s[m], g.z * s[m], g.w * s[m]);
    }
  }
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  for (int m = 0; m < kMaxModal; ++m) {
    float d = dot[m];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) d += __shfl_xor(d, off, kWave);
    if (lane == 0) red[wave][m] = d;
  }
  __syncthreads();
  if (threadIdx.x < kMaxModal)
    part[blockIdx.x * kMaxModal + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ds_m = sum of the partials (workgroup order); d alpha_j = s_j * (ds_j - sum_m s_m ds_m)
__global__ void wsum_dalpha_kernel(const float *__restrict__ part, int n_part, int M,
                                   const float *__restrict__ alphas, float *__restrict__ dalpha) {
  const int lane = threadIdx.x;
  float d[kMaxModal] = {0.f, 0.f, 0.f, 0.f};
  for (int p = lane; p < n_part; p += kWave)
    for (int m = 0; m < kMaxModal; ++m) d[m] += part[p * kMaxModal + m];
  for (int m = 0; m < kMaxModal; ++m) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) d[m] += __shfl_xor(d[m], off, kWave);
  }
  if (lane == 0) {
    float s[kMaxModal];
    softmax_small(alphas, M, s);
    float mean = 0.0f;
    for (int m = 0; m < M; ++m) mean += s[m] * d[m];
    for (int m = 0; m < M; ++m) dalpha[m] = s[m] * (d[m] - mean);
  }
}

int grid_for(int64_t n4) { return (int)std::min<int64_t>(rk::ceil_div(n4, 256), kWsumBlocks); }

}  // namespace

extern "C" {

int rk_modal_sum_fwd_f32(const float *const *F_host, int32_t M, const float *alphas, int64_t n_elem,
                         float *out, void *stream) {
  RK_CHECK_ARG(M >= 1 && M <= kMaxModal, "between 1 and 4 modalities");
  RK_CHECK_ARG(n_elem >= 0 && n_elem % 4 == 0, "element count must be a multiple of 4");
  if (n_elem == 0) return RK_OK;
  RK_CHECK_ARG(F_host && alphas && out && rk::aligned16(out), "null or unaligned pointer");
  ModalPtrs P{};
  for (int m = 0; m < M; ++m) {
    RK_CHECK_ARG(F_host[m] && rk::aligned16(F_host[m]), "null or unaligned feature matrix");
    P.f[m] = F_host[m];
  }
  wsum_fwd_kernel<<<grid_for(n_elem / 4), 256, 0, (hipStream_t)stream>>>(P, M, alphas, n_elem / 4, out);
  RK_CHECK_LAUNCH("wsum_fwd_kernel");
  return RK_OK;
}

size_t rk_modal_sum_bwd_ws_bytes(void) { return (size_t)kWsumBlocks * kMaxModal * sizeof(float); }

int rk_modal_sum_bwd_f32(const float *const *F_host, float *const *dF_host, int32_t M, const float *alphas,
                         const float *G, int64_t n_elem, float *dalpha, void *ws, size_t ws_bytes,
                         void *stream) {
  RK_CHECK_ARG(M >= 1 && M <= kMaxModal, "between 1 and 4 modalities");
  RK_CHECK_ARG(n_elem > 0 && n_elem % 4 == 0, "element count must be a positive multiple of 4");
  RK_CHECK_ARG(F_host && dF_host && alphas && G && dalpha && ws && rk::aligned16(G), "null or unaligned pointer");
  RK_CHECK_ARG(ws_bytes >= rk_modal_sum_bwd_ws_bytes(), "workspace too small");
  ModalPtrs P{};
  ModalOut O{};
  for (int m = 0; m < M; ++m) {
    RK_CHECK_ARG(F_host[m] && rk::aligned16(F_host[m]) && rk::aligned16(dF_host[m]), "null or unaligned matrix");
    P.f[m] = F_host[m];
    O.d[m] = dF_host[m];   // may be NULL: that modality needs no gradient
  }
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_for(n_elem / 4);
  wsum_bwd_kernel<<<grid, 256, 0, st>>>(P, O, M, alphas, G, n_elem / 4, (float *)ws);
  RK_CHECK_LAUNCH("wsum_bwd_kernel");
  wsum_dalpha_kernel<<<1, kWave, 0, st>>>((const float *)ws, grid, M, alphas, dalpha);
  RK_CHECK_LAUNCH("wsum_dalpha_kernel");
  return RK_OK;
}

}  // extern "C"
