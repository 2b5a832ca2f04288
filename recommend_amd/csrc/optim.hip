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
//
// Several GPUs (rk_adam_sqnorm_f32 + rk_adam_update_f32): the first n_replicated tensors hold
// whole gradients that are equal on every rank; the tensors behind them are row shards, whose
// squared norms are partial.  The first call leaves the sum over the shards' chunks in one device
// float; the caller all-reduces that float over the ranks and hands it to the second call, whose
// workgroups add it to the replicated chunks' sum: every rank derives the same clip factor.
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

// sum of partial[first .. first + count) in a fixed order: the squared norm of the row shards on this rank
__global__ __launch_bounds__(256) void adam_shard_sum_kernel(const float *__restrict__ partial, int32_t first,
                                                             int32_t count, float *__restrict__ out) {
  __shared__ float red[4];
  float s = 0.0f;
  for (int32_t c = threadIdx.x; c < count; c += 256) s += partial[first + c];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, rk::kWave);
  if (threadIdx.x % rk::kWave == 0) red[threadIdx.x / rk::kWave] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

// n_chunks: the chunks whose partial sums every workgroup adds itself (all of them on one GPU, those of
// the replicated tensors otherwise); shard_sq (may be NULL): the shards' squared norm summed over ranks
__global__ __launch_bounds__(256) void adam_update_kernel(AdamArgs args, int32_t n_chunks,
                                                          const float *__restrict__ partial,
                                                          const float *__restrict__ shard_sq,
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
  float sq = (red[0] + red[1]) + (red[2] + red[3]);
  if (shard_sq) sq += shard_sq[0];
  const float total = sqrtf(sq);
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

namespace {

int check_tensors(const rk_adam_tensor *tensors_host, int32_t n_tensors, int32_t n_replicated, const void *ws,
                  size_t ws_bytes, int64_t *total_chunks, int64_t *replicated_chunks) {
  RK_CHECK_ARG(tensors_host && ws, "null pointer");
  RK_CHECK_ARG(n_replicated >= 0 && n_replicated <= n_tensors, "n_replicated outside [0, n_tensors]");
  RK_CHECK_ARG(ws_bytes >= rk_adam_ws_bytes(tensors_host, n_tensors), "workspace too small");
  int64_t total = 0, rep = 0;
  for (int32_t i = 0; i < n_tensors; ++i) {
    const rk_adam_tensor &t = tensors_host[i];
    RK_CHECK_ARG(t.n > 0 && t.param && t.grad && t.exp_avg && t.exp_avg_sq, "tensor %d: null pointer or empty", i);
    total += rk::ceil_div(t.n, kChunk);
    if (i < n_replicated) rep = total;
  }
  RK_CHECK_ARG(total < INT32_MAX, "too many elements");
  *total_chunks = total;
  *replicated_chunks = rep;
  return RK_OK;
}

// pass 0: every chunk's sum of squares (and the step bump); pass 1: the updates
int launch_pass(int pass, const rk_adam_tensor *tensors_host, int32_t n_tensors, int64_t *step, const float *lr,
                float beta1, float beta2, float eps, float weight_decay, float max_norm, int32_t sum_chunks,
                const float *shard_sq, float *norm_out, float *partial, hipStream_t st) {
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
      adam_update_kernel<<<c, 256, 0, st>>>(args, sum_chunks, partial, shard_sq, step, lr, beta1, beta2, eps,
                                            weight_decay, max_norm, norm_out);
      RK_CHECK_LAUNCH("adam_update_kernel");
    }
    chunk_base += c;
  }
  return RK_OK;
}

}  // namespace

int rk_adam_sqnorm_f32(const rk_adam_tensor *tensors_host, int32_t n_tensors, int32_t n_replicated, int64_t *step,
                       float *shard_sq_out, void *ws, size_t ws_bytes, void *stream) {
  RK_CHECK_ARG(n_tensors >= 0, "negative tensor count");
  if (n_tensors == 0) return RK_OK;
  RK_CHECK_ARG(step && shard_sq_out, "null pointer");
  int64_t total = 0, rep = 0;
  if (int rc = check_tensors(tensors_host, n_tensors, n_replicated, ws, ws_bytes, &total, &rep)) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (int rc = launch_pass(0, tensors_host, n_tensors, step, nullptr, 0, 0, 0, 0, 0, 0, nullptr, nullptr, (float *)ws, st))
    return rc;
  adam_shard_sum_kernel<<<1, 256, 0, st>>>((const float *)ws, (int32_t)rep, (int32_t)(total - rep), shard_sq_out);
  RK_CHECK_LAUNCH("adam_shard_sum_kernel");
  return RK_OK;
}

int rk_adam_update_f32(const rk_adam_tensor *tensors_host, int32_t n_tensors, int32_t n_replicated, const int64_t *step,
                       const float *lr, float beta1, float beta2, float eps, float weight_decay, float max_norm,
                       const float *shard_sq, float *norm_out, const void *ws, size_t ws_bytes, void *stream) {
  RK_CHECK_ARG(n_tensors >= 0, "negative tensor count");
  if (n_tensors == 0) return RK_OK;
  RK_CHECK_ARG(step && lr, "null pointer");
  RK_CHECK_ARG(beta1 >= 0.0f && beta1 < 1.0f && beta2 >= 0.0f && beta2 < 1.0f && eps >= 0.0f, "bad Adam constants");
  int64_t total = 0, rep = 0;
  if (int rc = check_tensors(tensors_host, n_tensors, n_replicated, ws, ws_bytes, &total, &rep)) return rc;
  RK_CHECK_ARG(shard_sq || rep == total, "row-sharded tensors need the all-reduced squared norm of the shards");
  return launch_pass(1, tensors_host, n_tensors, const_cast<int64_t *>(step), lr, beta1, beta2, eps, weight_decay,
                     max_norm, (int32_t)rep, rep == total ? nullptr : shard_sq, norm_out,
                     const_cast<float *>((const float *)ws), (hipStream_t)stream);
}

int rk_adam_clip_step_f32(const rk_adam_tensor *tensors_host, int32_t n_tensors, int64_t *step, const float *lr,
                          float beta1, float beta2, float eps, float weight_decay, float max_norm, float *norm_out,
                          void *ws, size_t ws_bytes, void *stream) {
  RK_CHECK_ARG(n_tensors >= 0, "negative tensor count");
  if (n_tensors == 0) return RK_OK;
  RK_CHECK_ARG(step && lr, "null pointer");
  RK_CHECK_ARG(beta1 >= 0.0f && beta1 < 1.0f && beta2 >= 0.0f && beta2 < 1.0f && eps >= 0.0f, "bad Adam constants");
  int64_t total = 0, rep = 0;
  if (int rc = check_tensors(tensors_host, n_tensors, n_tensors, ws, ws_bytes, &total, &rep)) return rc;
  hipStream_t st = (hipStream_t)stream;
  // all partial sums first, then all updates
  if (int rc = launch_pass(0, tensors_host, n_tensors, step, nullptr, 0, 0, 0, 0, 0, 0, nullptr, nullptr, (float *)ws, st))
    return rc;
  return launch_pass(1, tensors_host, n_tensors, step, lr, beta1, beta2, eps, weight_decay, max_norm, (int32_t)total,
                     nullptr, norm_out, (float *)ws, st);
}

}  // extern "C"
