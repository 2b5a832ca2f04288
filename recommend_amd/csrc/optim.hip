// Adam with global-norm gradient clipping over a list of parameter tensors in two launches
// (gfx950): the optimizer half of the reference's training step, train/graph_trainer.py:151-156
// (clip_grad_norm_(parameters, 1.0) followed by torch.optim.Adam.step()).
//
// The tensors are described by a host array of descriptors, which travels to the kernels BY VALUE
// in their arguments (up to kMaxTensors per launch; more tensors take more launches of the same
// two kernels): nothing is copied or allocated, so the call can be recorded in a HIP graph.  The
// tensors are cut into chunks of kChunk elements, one workgroup each.  Launch 1 writes every chunk's sum of squared
// gradient elements; launch 2 adds those partial sums in chunk order (every workgroup for itself,
// the same order and hence the same bits everywhere), derives the clip factor and applies Adam's
// update to its chunk.  No float atomics; the step count lives in device memory so that a launch
// replayed from a captured HIP graph advances it.
#include <algorithm>

#include "common.h"

namespace {

constexpr int kChunk = 4096;   // elements per workgroup
constexpr int kMaxTensors = 48;

struct AdamArgs {
  rk_adam_tensor t[kMaxTensors];
  int32_t chunk_start[kMaxTensors + 1];   // first chunk (= workgroup of this launch) of every tensor
  int32_t n;
};

// the tensor and element range of this workgroup's chunk
__device__ __forceinline__ rk_adam_tensor my_chunk(const AdamArgs &a, int64_t &lo, int64_t &hi) {
  int i = 0;
  while (i + 1 < a.n && a.chunk_start[i + 1] <= (int32_t)blockIdx.x) ++i;
  const rk_adam_tensor t = a.t[i];
  lo = (int64_t)((int32_t)blockIdx.x - a.chunk_start[i]) * kChunk;
  hi = lo + kChunk < t.n ? lo + kChunk : t.n;
  return t;
}

__global__ __launch_bounds__(256) void adam_sqnorm_kernel(AdamArgs args, float *__restrict__ partial,
                                                          int64_t *__restrict__ step, int bump_step) {
  __shared__ float red[4];
  int64_t lo, hi;
  const rk_adam_tensor t = my_chunk(args, lo, hi);
  float s0 = 0.0f, s1 = 0.0f;
  int64_t i = lo + threadIdx.x;
  for (; i + 256 < hi; i += 512) {
    const float a = t.grad[i], b = t.grad[i + 256];
    s0 = fmaf(a, a, s0);
    s1 = fmaf(b, b, s1);
  }
  if (i < hi) {
    const float a = t.grad[i];
    s0 = fmaf(a, a, s0);
  }
  float s = s0 + s1;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, rk::kWave);
  if (threadIdx.x % rk::kWave == 0) red[threadIdx.x / rk::kWave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    if (blockIdx.x == 0 && bump_step) step[0] += 1;   // read by the update launches that follow on the stream
  }
}

__global__ __launch_bounds__(256) void adam_update_kernel(AdamArgs args, int32_t n_chunks,
                                                          const float *__restrict__ partial,
                                                          const int64_t *__restrict__ step, const float *__restrict__ lr_dev,
                                                          float beta1, float beta2, float eps, float weight_decay,
                                                          float max_norm, float *__restrict__ norm_out) {
  __shared__ float red[4];
  // total squared norm: lane-strided partial sums in a fixed order, then a fixed tree
  float s = 0.0f;
  for (int32_t c = threadIdx.x; c < n_chunks; c += 256) s += partial[c];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, rk::kWave);
  if (threadIdx.x % rk::kWave == 0) red[threadIdx.x / rk::kWave] = s;
  __syncthreads();
  const float total = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
  if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) norm_out[0] = total;
  // clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
  const float coef = max_norm > 0.0f ? fminf(max_norm / (total + 1e-6f), 1.0f) : 1.0f;
  const float t_step = (float)step[0];
  const float bc1 = 1.0f - powf(beta1, t_step), bc2 = 1.0f - powf(beta2, t_step);
  const float step_size = lr_dev[0] / bc1, bc2_sqrt = sqrtf(bc2);

  int64_t lo, hi;
  const rk_adam_tensor t = my_chunk(args, lo, hi);
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    float p = t.param[i];
    float g = t.grad[i] * coef;
    if (weight_decay != 0.0f) g = fmaf(weight_decay, p, g);
    // torch.optim.Adam (single-tensor form): lerp of the first moment, addcmul of the second,
    // denom = sqrt(v) / sqrt(bc2) + eps, param -= step_size * m / denom
    const float m = t.exp_avg[i] + (g - t.exp_avg[i]) * (1.0f - beta1);
    const float v = t.exp_avg_sq[i] * beta2 + (1.0f - beta2) * g * g;
    t.exp_avg[i] = m;
    t.exp_avg_sq[i] = v;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    t.param[i] = p - step_size * (m / denom);
  }
}

}  // namespace

extern "C" {

size_t rk_adam_ws_bytes(const rk_adam_tensor *tensors_host, int32_t n_tensors) {
  if (!tensors_host || n_tensors <= 0) return 0;
  int64_t chunks = 0;
  for (int32_t i = 0; i < n_tensors; ++i) chunks += rk::ceil_div(tensors_host[i].n, kChunk);
  return (size_t)(chunks + 1) * sizeof(float);
}

int rk_adam_clip_step_f32(const rk_adam_tensor *tensors_host, int32_t n_tensors, int64_t *step, const float *lr,
                          float beta1, float beta2, float eps, float weight_decay, float max_norm, float *norm_out,
                          void *ws, size_t ws_bytes, void *stream) {
  RK_CHECK_ARG(n_tensors >= 0, "negative tensor count");
  if (n_tensors == 0) return RK_OK;
  RK_CHECK_ARG(tensors_host && step && lr && ws, "null pointer");
  RK_CHECK_ARG(beta1 >= 0.0f && beta1 < 1.0f && beta2 >= 0.0f && beta2 < 1.0f && eps >= 0.0f, "bad Adam constants");
  RK_CHECK_ARG(ws_bytes >= rk_adam_ws_bytes(tensors_host, n_tensors), "workspace too small");
  for (int32_t i = 0; i < n_tensors; ++i) {
    const rk_adam_tensor &t = tensors_host[i];
    RK_CHECK_ARG(t.n > 0 && t.param && t.grad && t.exp_avg && t.exp_avg_sq, "tensor %d: null pointer or empty", i);
  }
  hipStream_t st = (hipStream_t)stream;
  float *partial = (float *)ws;
  // launch groups of up to kMaxTensors tensors; all partial sums first, then all updates
  int64_t total_chunks = 0;
  for (int32_t i = 0; i < n_tensors; ++i) total_chunks += rk::ceil_div(tensors_host[i].n, kChunk);
  RK_CHECK_ARG(total_chunks < INT32_MAX, "too many elements");
  for (int pass = 0; pass < 2; ++pass) {
    int32_t chunk_base = 0;
    for (int32_t g0 = 0; g0 < n_tensors; g0 += kMaxTensors) {
      AdamArgs args;
      args.n = std::min(kMaxTensors, n_tensors - g0);
      int32_t c = 0;
      for (int i = 0; i < args.n; ++i) {
        args.t[i] = tensors_host[g0 + i];
        args.chunk_start[i] = c;
        c += (int32_t)rk::ceil_div(args.t[i].n, kChunk);
      }
      args.chunk_start[args.n] = c;
      if (pass == 0) {
        adam_sqnorm_kernel<<<c, 256, 0, st>>>(args, partial + chunk_base, step, g0 == 0);
        RK_CHECK_LAUNCH("adam_sqnorm_kernel");
      } else {
        adam_update_kernel<<<c, 256, 0, st>>>(args, (int32_t)total_chunks, partial, step, lr, beta1, beta2, eps,
                                              weight_decay, max_norm, norm_out);
        RK_CHECK_LAUNCH("adam_update_kernel");
      }
      chunk_base += c;
    }
  }
  return RK_OK;
}

}  // extern "C"
