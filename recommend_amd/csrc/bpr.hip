// Pairwise ranking losses on gathered dot products, forward and backward, with the
// negative-item lookup folded into the forward kernel (gfx950).  The per-sample term is
// selected by `mode`: BPR -logsigmoid(x) (evalue/loss.py:7-28), -log(sigmoid(x)) as
// mmgcn_loss / the SGrec loss spell it (evalue/loss.py:153-173, Main_Model.py:902-928), or
// the softmax cross-entropy over [pos, negatives] of compute_info_bpr_loss
// (evalue/loss.py:95-117).
//
// One group of 16 lanes owns one training sample; every lane walks the
// embedding row in float4 steps (stride 16 float4), so D = 64 is exactly one
// coalesced 256-byte read per gathered row.  The per-sample losses are reduced
// in a fixed order (group -> block -> final block), never with float atomics,
// so the forward value is run-to-run deterministic.
#include <mutex>

#include "common.h"

namespace {

constexpr int GL = 16;           // lanes per sample
constexpr int SPB = 256 / GL;    // samples per block

__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int off = GL / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, GL);
  return v;
}

__device__ __forceinline__ float dot_rows(const float4 *__restrict__ a,
                                          const float4 *__restrict__ b, int d4,
                                          int sub) {
  float s = 0.0f;
  for (int q = sub; q < d4; q += GL) {
    const float4 x = a[q], y = b[q];
    s = fmaf(x.x, y.x, s);
    s = fmaf(x.y, y.y, s);
    s = fmaf(x.z, y.z, s);
    s = fmaf(x.w, y.w, s);
  }
  return group_sum(s);
}

// rank-th smallest id not in the sorted list pos[0..len)
__device__ __forceinline__ int64_t kth_missing(const int32_t *__restrict__ pos,
                                               int32_t len, int32_t rank) {
  int32_t lo = 0, hi = len;
  while (lo < hi) {
    const int32_t mid = (lo + hi) >> 1;
    if (pos[mid] - mid <= rank)
      lo = mid + 1;
    else
      hi = mid;
  }
  return (int64_t)rank + lo;
}

// -logsigmoid(x) = max(-x, 0) + log1p(exp(-|x|))
__device__ __forceinline__ float neg_logsigmoid(float x) {
  return fmaxf(-x, 0.0f) + log1pf(expf(-fabsf(x)));
}

constexpr int kModeLogSigmoid = 0;     // RK_LOSS_BPR
constexpr int kModeLogOfSigmoid = 1;   // RK_LOSS_LOG_SIGMOID
constexpr int kModeSoftmaxCE = 2;      // RK_LOSS_INFO_BPR

// per-pair term of modes 0 and 1; mode 1 keeps the reference's two-step form, which
// saturates (sigmoid underflows to 0, the log gives inf) where logsigmoid does not
__device__ __forceinline__ float pair_term(float x, int mode) {
  if (mode == kModeLogOfSigmoid) return -logf(1.0f / (1.0f + expf(-x)));
  return neg_logsigmoid(x);
}

__global__ __launch_bounds__(256) void bpr_fwd_kernel(
    const float *__restrict__ U, const float *__restrict__ I, int32_t D,
    const int64_t *__restrict__ users, const int64_t *__restrict__ pos,
    int64_t *__restrict__ neg, int64_t B, int32_t R,
    const float *__restrict__ mask, const int32_t *__restrict__ neg_rank,
    const int32_t *__restrict__ pos_ptr, const int32_t *__restrict__ pos_items, int mode,
    float *__restrict__ x_out, float *__restrict__ part) {
  __shared__ float s_loss[SPB], s_den[SPB];
  const int sub = threadIdx.x % GL, grp = threadIdx.x / GL;
  const int64_t s = (int64_t)blockIdx.x * SPB + grp;
  const int d4 = D / 4;
  float loss = 0.0f, den = 0.0f;
  if (s < B) {
    const int64_t u = users[s];
    const float4 *pu = reinterpret_cast<const float4 *>(U + u * D);
    const float4 *pi = reinterpret_cast<const float4 *>(I + pos[s] * D);
    const float ps = dot_rows(pu, pi, d4, sub);
    float mx = ps, se = 1.0f;   // running max and sum of exp(. - max) over [pos, negatives]
    for (int32_t r = 0; r < R; ++r) {
      int64_t n;
      if (neg_rank) {
        const int32_t rank = neg_rank[s * R + r];
        // rank < 0: the "no candidate" dummy item 0 (graph_data_loader.py:383)
        n = rank < 0 ? 0
                     : kth_missing(pos_items + pos_ptr[u], pos_ptr[u + 1] - pos_ptr[u], rank);
        if (sub == 0) neg[s * R + r] = n;
      } else {
        n = neg[s * R + r];
      }
      const float4 *ni = reinterpret_cast<const float4 *>(I + n * D);
      const float x = ps - dot_rows(pu, ni, d4, sub);
      const float m = mask ? mask[s * R + r] : 1.0f;
      if (sub == 0) x_out[s * R + r] = x;
      if (mode == kModeSoftmaxCE) {
        const float ns = ps - x;
        const float nm = fmaxf(mx, ns);
        se = se * expf(mx - nm) + expf(ns - nm);
        mx = nm;
      } else {
        loss += pair_term(x, mode) * m;
        den += m;
      }
    }
    if (mode == kModeSoftmaxCE) {
      // cross entropy with the positive as class 0: logsumexp - pos; the backward pass needs the
      // negatives' log-probabilities, which replace the saved differences
      const float lse = mx + logf(se);
      loss = lse - ps;
      den = 1.0f;
      if (sub == 0)
        for (int32_t r = 0; r < R; ++r) x_out[s * R + r] = (ps - x_out[s * R + r]) - lse;
    }
  }
  if (sub == 0) {
    s_loss[grp] = loss;
    s_den[grp] = den;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float l = 0.0f, d = 0.0f;
    for (int g = 0; g < SPB; ++g) {
      l += s_loss[g];
      d += s_den[g];
    }
    part[2 * blockIdx.x] = l;
    part[2 * blockIdx.x + 1] = d;
  }
}

__global__ void bpr_final_kernel(const float *__restrict__ part, int32_t n_part,
                                 int use_mask, float *__restrict__ loss_out) {
  // one wavefront, fixed order: lane strides, then a shuffle tree
  float l = 0.0f, d = 0.0f;
  for (int32_t i = threadIdx.x; i < n_part; i += rk::kWave) {
    l += part[2 * i];
    d += part[2 * i + 1];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    l += __shfl_xor(l, off, rk::kWave);
    d += __shfl_xor(d, off, rk::kWave);
  }
  if (threadIdx.x == 0) {
    // bpr_loss: loss.mean()  |  loss.sum() / mask.sum().clamp_min(1.0)
    const float denom = use_mask ? fmaxf(d, 1.0f) : d;
    loss_out[0] = l / denom;
    loss_out[1] = denom;
  }
}

__device__ __forceinline__ void atomic_axpy(float *__restrict__ dst,
                                            const float4 *__restrict__ src,
                                            float a, int d4, int sub) {
  for (int q = sub; q < d4; q += GL) {
    const float4 v = src[q];
    atomicAdd(dst + 4 * q + 0, a * v.x);
    atomicAdd(dst + 4 * q + 1, a * v.y);
    atomicAdd(dst + 4 * q + 2, a * v.z);
    atomicAdd(dst + 4 * q + 3, a * v.w);
  }
}

__global__ __launch_bounds__(256) void bpr_bwd_kernel(
    const float *__restrict__ U, const float *__restrict__ I, int32_t D,
    const int64_t *__restrict__ users, const int64_t *__restrict__ pos,
    const int64_t *__restrict__ neg, int64_t B, int32_t R,
    const float *__restrict__ mask, const float *__restrict__ x,
    const float *__restrict__ loss_out, const float *__restrict__ gout, int mode,
    float *__restrict__ dU, float *__restrict__ dI) {
  const int sub = threadIdx.x % GL;
  const int64_t s = (int64_t)blockIdx.x * SPB + threadIdx.x / GL;
  if (s >= B) return;
  const int d4 = D / 4;
  const float scale = gout[0] / loss_out[1];
  const int64_t u = users[s], p = pos[s];
  const float4 *pu = reinterpret_cast<const float4 *>(U + u * D);
  const float4 *pi = reinterpret_cast<const float4 *>(I + p * D);
  float gsum = 0.0f;
  for (int32_t r = 0; r < R; ++r) {
    const float xv = x[s * R + r];
    const float m = mask ? mask[s * R + r] : 1.0f;
    // d(-logsigmoid(x))/dx = -sigmoid(-x) = -1/(1+exp(x)); for the cross entropy the saved value
    // is log p_r and d loss / d neg_r = p_r, d loss / d pos = -sum_r p_r
    const float g = mode == kModeSoftmaxCE ? -scale * expf(xv) : -scale * m / (1.0f + expf(xv));
    const int64_t n = neg[s * R + r];
    const float4 *ni = reinterpret_cast<const float4 *>(I + n * D);
    // x = <pu,pi> - <pu,ni>:  d/dni = -g*pu ; d/dpu += g*(pi - ni)
    atomic_axpy(dI + n * D, pu, -g, d4, sub);
    atomic_axpy(dU + u * D, ni, -g, d4, sub);
    gsum += g;
  }
  atomic_axpy(dI + p * D, pu, gsum, d4, sub);
  atomic_axpy(dU + u * D, pi, gsum, d4, sub);
}

// ---- bit-reproducible backward ---------------------------------------------------------------
// Every row of dU / dI is written by ONE lane group, which adds the row's contributions in index
// order in registers: contribution c of the user side is sample c (key users[c]); of the item side
// c < B is the positive of sample c (key pos[c]), c >= B the negative c - B (key neg[c - B]).  The
// group of contribution c scans all keys 16 at a time; if an earlier contribution has its key it
// leaves (that one owns the row), otherwise it accumulates every match as the scan passes it.  No
// atomics, no sorting, no scratch; n^2 / 16 compare steps of a lane group for n contributions
// (n = 512 at the Elec preset: 32 steps per group), which is why rk_rank_loss_bwd_f32 switches to
// the atomic kernel above kDetMaxContrib contributions.  Up to kLdsKeys contributions every
// workgroup first copies the keys to LDS and scans them there (the scan is a chain of dependent
// key loads: out of L2 it was 95 us for 2048 contributions, the Clothing preset's batch).
constexpr int64_t kDetMaxContrib = 32768;
constexpr int kLdsKeys = 4096;   // 32 KB of int64 keys
constexpr int kMaxQ = 4;   // float4 per lane and pass: rows up to 256 floats in one pass, wider rows in several

// d loss / d x of negative r of one sample (x_row / mask_row: the sample's R saved values), as in
// bpr_bwd_kernel
__device__ __forceinline__ float bwd_coef(const float *__restrict__ x_row, const float *__restrict__ mask_row,
                                          int32_t r, float scale, int mode) {
  const float xv = x_row[r];
  const float m = mask_row ? mask_row[r] : 1.0f;
  return mode == kModeSoftmaxCE ? -scale * expf(xv) : -scale * m / (1.0f + expf(xv));
}

// A row with many contributions (a very active user draws dozens of samples of a batch) used to be a
// chain of dependent loads -- key, sample index, item index, row -- one contribution after the
// other: 95 us for the 2048 contributions of the Clothing preset.  Now the scan only COLLECTS the
// matching contributions (kPend at a time); then lane j of the group expands contribution j into
// (row, coefficient) pairs in LDS, all lanes at once, and the rows are fetched four at a time before
// they are added -- in the same order as before, so the sums are the same bits.
constexpr int kPend = 16;   // contributions collected per flush: one per lane of a group
constexpr int kWork = 64;   // (row, coefficient) pairs per flush; a contribution of the user side is R + 1 pairs

template <bool ITEMS, bool LDS_KEYS>
__global__ __launch_bounds__(256) void bpr_bwd_det_kernel(
    const float *__restrict__ U, const float *__restrict__ I, int32_t D,
    const int64_t *__restrict__ users, const int64_t *__restrict__ pos,
    const int64_t *__restrict__ neg, int64_t B, int32_t R,
    const float *__restrict__ mask, const float *__restrict__ x,
    const float *__restrict__ loss_out, const float *__restrict__ gout, int mode,
    float *__restrict__ dst) {
  __shared__ int64_t keys[LDS_KEYS ? kLdsKeys : 1];
  __shared__ int32_t pend[SPB][kPend];
  __shared__ int64_t wrow[SPB][kWork];
  __shared__ float wcoef[SPB][kWork];
  const int sub = threadIdx.x % GL, grp = threadIdx.x / GL;
  const int64_t n = ITEMS ? B * (1 + R) : B;
  const int64_t c = (int64_t)blockIdx.x * SPB + grp;
  auto gkey = [&](int64_t i) -> int64_t { return !ITEMS ? users[i] : (i < B ? pos[i] : neg[i - B]); };
  if constexpr (LDS_KEYS) {
    for (int64_t i = threadIdx.x; i < n; i += 256) keys[i] = gkey(i);
    __syncthreads();
  }
  if (c >= n) return;                       // whole lane groups leave; nothing below synchronises wider
  auto key = [&](int64_t i) -> int64_t { return LDS_KEYS ? keys[LDS_KEYS ? i : 0] : gkey(i); };
  const int64_t my = key(c);
  const int d4 = D / 4;
  const float scale = gout[0] / loss_out[1];
  const int gshift = (threadIdx.x % rk::kWave) / GL * GL;   // this group's bits in a wavefront ballot
  const int per = ITEMS ? 1 : R + 1;        // rows one contribution adds (the host keeps per <= kWork)
  const int pcap = min(kPend, kWork / per); // contributions per flush
  const float *__restrict__ src = ITEMS ? U : I;

  for (int q0 = 0; q0 < d4; q0 += GL * kMaxQ) {
    float4 acc[kMaxQ];
#pragma unroll
    for (int t = 0; t < kMaxQ; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    int np = 0;                             // collected contributions (uniform over the group)
    auto flush = [&]() {
      // lane j expands contribution j: (row offset, coefficient) pairs in the order they are added
      if (sub < np) {
        const int64_t ci = pend[grp][sub];
        if (!ITEMS) {
          // d/dU[u] of sample ci: sum_r (-g_r) * I[neg_r] + (sum_r g_r) * I[pos]
          float gsum = 0.0f;
          for (int32_t r = 0; r < R; ++r) {
            const float g = bwd_coef(x + ci * R, mask ? mask + ci * R : nullptr, r, scale, mode);
            wrow[grp][sub * per + r] = neg[ci * R + r] * D;
            wcoef[grp][sub * per + r] = -g;
            gsum += g;
          }
          wrow[grp][sub * per + R] = pos[ci] * D;
          wcoef[grp][sub * per + R] = gsum;
        } else if (ci < B) {
          float gsum = 0.0f;
          for (int32_t r = 0; r < R; ++r) gsum += bwd_coef(x + ci * R, mask ? mask + ci * R : nullptr, r, scale, mode);
          wrow[grp][sub] = users[ci] * D;
          wcoef[grp][sub] = gsum;
        } else {
          const int64_t sr = ci - B, smp = sr / R;
          const float g = bwd_coef(x + smp * R, mask ? mask + smp * R : nullptr, (int32_t)(sr - smp * R), scale, mode);
          wrow[grp][sub] = users[smp] * D;
          wcoef[grp][sub] = -g;
        }
      }
      // (the group is 16 lanes of one wavefront: its LDS operations complete in program order)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int nw = np * per;
      for (int w0 = 0; w0 < nw; w0 += 4) {
        float4 v[4][kMaxQ];
        float co[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {       // four rows in flight; past the end the last pair is read again, unused
          const int jj = min(w0 + j, nw - 1);
          const float4 *r4 = reinterpret_cast<const float4 *>(src + wrow[grp][jj]);
          co[j] = wcoef[grp][jj];
#pragma unroll
          for (int t = 0; t < kMaxQ; ++t) v[j][t] = r4[min(q0 + t * GL + sub, d4 - 1)];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (w0 + j < nw) {
#pragma unroll
            for (int t = 0; t < kMaxQ; ++t) {
              if (q0 + t * GL + sub < d4) {
                acc[t].x = fmaf(co[j], v[j][t].x, acc[t].x);
                acc[t].y = fmaf(co[j], v[j][t].y, acc[t].y);
                acc[t].z = fmaf(co[j], v[j][t].z, acc[t].z);
                acc[t].w = fmaf(co[j], v[j][t].w, acc[t].w);
              }
            }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();      // the pairs have been read before the next flush overwrites them
      np = 0;
    };
    for (int64_t c0 = 0; c0 < n; c0 += GL) {
      const int64_t cc = c0 + sub;
      const bool hit = cc < n && key(cc) == my;
      uint32_t bits = (uint32_t)((__ballot(hit) >> gshift) & 0xFFFFu);
      if (c0 < c) {
        const int64_t below = c - c0;       // indices c0 .. c0 + below - 1 come before c
        if (bits & (below >= GL ? 0xFFFFu : ((1u << below) - 1u))) return;   // an earlier contribution owns the row
        if (below < GL) bits &= ~((1u << below) - 1u);
      }
      while (bits) {
        const int b = __builtin_ctz(bits);
        bits &= bits - 1;
        if (sub == 0) pend[grp][np] = (int32_t)(c0 + b);
        if (++np == pcap) flush();
      }
    }
    if (np) flush();
    float4 *out4 = reinterpret_cast<float4 *>(dst + my * D);
#pragma unroll
    for (int t = 0; t < kMaxQ; ++t) {
      const int q = q0 + t * GL + sub;
      if (q < d4) {
        float4 o = out4[q];
        o.x += acc[t].x; o.y += acc[t].y; o.z += acc[t].z; o.w += acc[t].w;
        out4[q] = o;
      }
    }
  }
}

// score-level form of the same loss (bpr_loss called on precomputed scores)
__global__ __launch_bounds__(256) void bpr_scores_fwd_kernel(
    const float *__restrict__ pos, const float *__restrict__ neg, int64_t B, int32_t R,
    const float *__restrict__ mask, float *__restrict__ x_out, float *__restrict__ part) {
  __shared__ float s_loss[4], s_den[4];
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float loss = 0.0f, den = 0.0f;
  if (s < B) {
    const float ps = pos[s];
    for (int32_t r = 0; r < R; ++r) {
      const float x = ps - neg[s * R + r];
      const float m = mask ? mask[s * R + r] : 1.0f;
      x_out[s * R + r] = x;
      loss += neg_logsigmoid(x) * m;
      den += m;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    loss += __shfl_xor(loss, off, rk::kWave);
    den += __shfl_xor(den, off, rk::kWave);
  }
  if (threadIdx.x % rk::kWave == 0) {
    s_loss[threadIdx.x / rk::kWave] = loss;
    s_den[threadIdx.x / rk::kWave] = den;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = ((s_loss[0] + s_loss[1]) + s_loss[2]) + s_loss[3];
    part[2 * blockIdx.x + 1] = ((s_den[0] + s_den[1]) + s_den[2]) + s_den[3];
  }
}

__global__ __launch_bounds__(256) void bpr_scores_bwd_kernel(
    const float *__restrict__ x, const float *__restrict__ mask,
    const float *__restrict__ loss_out, const float *__restrict__ gout, int64_t B, int32_t R,
    float *__restrict__ dpos, float *__restrict__ dneg) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= B) return;
  const float scale = gout[0] / loss_out[1];
  float gsum = 0.0f;
  for (int32_t r = 0; r < R; ++r) {
    const float m = mask ? mask[s * R + r] : 1.0f;
    const float g = -scale * m / (1.0f + expf(x[s * R + r]));
    dneg[s * R + r] = -g;
    gsum += g;
  }
  dpos[s] = gsum;
}

}  // namespace

// ---- the forward pass for callers that cannot allocate -------------------------------------
// The per-workgroup partial sums live in a file-scope device array instead of a caller
// workspace.  kNoAllocMaxBatch bounds the batch; each GPU the library runs on holds its own
// copy of the array (kNoAllocMaxBatch / SPB * 2 + 2 floats = 32 KiB + 8 B) for the life of the
// process.  `scratch` is a host-side stand-in, not a device address: host code only ever uses
// the per-device pointer that hipGetSymbolAddress returned.
constexpr int64_t kNoAllocMaxBatch = 65536;
constexpr int kNoAllocMaxDevices = 64;
static __device__ __align__(256) float scratch[2 * (kNoAllocMaxBatch / SPB) + 2];
static float *g_scratch_ptr[kNoAllocMaxDevices];   // per device, filled on the first call there
static std::mutex g_scratch_mutex;

static int scratch_for_current_device(float **out) {
  int dev = -1;
  RK_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kNoAllocMaxDevices)
    return rk::fail(RK_ERR_UNSUPPORTED, "device ordinal %d beyond the %d this entry point tracks", dev,
                    kNoAllocMaxDevices);
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    void *p = nullptr;
    RK_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return rk::fail(RK_ERR_HIP, "hipGetSymbolAddress returned a null scratch address");
    g_scratch_ptr[dev] = static_cast<float *>(p);
  }
  *out = g_scratch_ptr[dev];
  return RK_OK;
}

extern "C" {

int rk_bpr_scores_fwd_f32(const float *pos, const float *neg, int64_t B, int32_t R,
                          const float *mask, float *x_out, float *loss_out, void *ws,
                          size_t ws_bytes, void *stream) {
  RK_CHECK_ARG(B > 0 && R > 0, "empty score tensors");
  RK_CHECK_ARG(pos && neg && x_out && loss_out && ws, "null pointer");
  const int32_t nb = (int32_t)rk::ceil_div(B, 256);
  RK_CHECK_ARG(ws_bytes >= (size_t)(2 * nb) * sizeof(float), "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  bpr_scores_fwd_kernel<<<nb, 256, 0, st>>>(pos, neg, B, R, mask, x_out, (float *)ws);
  RK_CHECK_LAUNCH("bpr_scores_fwd_kernel");
  bpr_final_kernel<<<1, rk::kWave, 0, st>>>((const float *)ws, nb, mask != nullptr, loss_out);
  RK_CHECK_LAUNCH("bpr_final_kernel");
  return RK_OK;
}

int rk_bpr_scores_bwd_f32(const float *x, const float *mask, const float *loss_out,
                          const float *gout, int64_t B, int32_t R, float *dpos, float *dneg,
                          void *stream) {
  RK_CHECK_ARG(B > 0 && R > 0, "empty score tensors");
  RK_CHECK_ARG(x && loss_out && gout && dpos && dneg, "null pointer");
  bpr_scores_bwd_kernel<<<(int32_t)rk::ceil_div(B, 256), 256, 0, (hipStream_t)stream>>>(
      x, mask, loss_out, gout, B, R, dpos, dneg);
  RK_CHECK_LAUNCH("bpr_scores_bwd_kernel");
  return RK_OK;
}

size_t rk_bpr_ws_bytes(int64_t B, int32_t R) {
  (void)R;
  if (B < 0) return 0;
  return (size_t)(2 * rk::ceil_div(B, SPB) + 2) * sizeof(float);
}

int rk_rank_loss_fwd_f32(const float *U, const float *I, int32_t D,
                         const int64_t *users, const int64_t *pos, int64_t *neg,
                         int64_t B, int32_t R, const float *mask,
                         const int32_t *neg_rank, const int32_t *pos_ptr,
                         const int32_t *pos_items, int32_t n_items, int32_t mode, float *x_out,
                         float *loss_out, void *ws, size_t ws_bytes, void *stream) {
  RK_CHECK_ARG(mode >= 0 && mode <= 2, "unknown loss mode");
  RK_CHECK_ARG(mode != kModeSoftmaxCE || mask == nullptr, "the cross-entropy form takes no mask");
  // train/graph_trainer.py:262-263 raises on an empty batch
  RK_CHECK_ARG(B > 0 && R > 0, "batch error, recive null user or item or neg_items");
  RK_CHECK_ARG(U && I && users && pos && neg && x_out && loss_out && ws, "null pointer");
  RK_CHECK_ARG(D > 0 && D % 4 == 0, "embedding width must be a multiple of 4");
  RK_CHECK_ARG(rk::aligned16(U) && rk::aligned16(I), "embeddings must be 16-byte aligned");
  RK_CHECK_ARG(ws_bytes >= rk_bpr_ws_bytes(B, R), "workspace too small");
  if (neg_rank) RK_CHECK_ARG(pos_ptr && pos_items && n_items > 0, "sampling needs the user->positives CSR");
  hipStream_t st = (hipStream_t)stream;
  const int32_t nb = (int32_t)rk::ceil_div(B, SPB);
  float *part = (float *)ws;
  bpr_fwd_kernel<<<nb, 256, 0, st>>>(U, I, D, users, pos, neg, B, R, mask, neg_rank, pos_ptr,
                                      pos_items, mode, x_out, part);
  RK_CHECK_LAUNCH("bpr_fwd_kernel");
  bpr_final_kernel<<<1, rk::kWave, 0, st>>>(part, nb, mask != nullptr, loss_out);
  RK_CHECK_LAUNCH("bpr_final_kernel");
  return RK_OK;
}

int rk_bpr_fwd_f32(const float *U, const float *I, int32_t D,
                   const int64_t *users, const int64_t *pos, int64_t *neg,
                   int64_t B, int32_t R, const float *mask,
                   const int32_t *neg_rank, const int32_t *pos_ptr,
                   const int32_t *pos_items, int32_t n_items, float *x_out,
                   float *loss_out, void *ws, size_t ws_bytes, void *stream) {
  return rk_rank_loss_fwd_f32(U, I, D, users, pos, neg, B, R, mask, neg_rank, pos_ptr, pos_items,
                              n_items, kModeLogSigmoid, x_out, loss_out, ws, ws_bytes, stream);
}

int64_t rk_bpr_fwd_noalloc_max_batch(void) { return kNoAllocMaxBatch; }

int rk_bpr_fwd_noalloc_f32(const float *U, const float *I, int32_t D, const int64_t *users,
                           const int64_t *pos, int64_t *neg, int64_t B, int32_t R,
                           const float *mask, const int32_t *neg_rank, const int32_t *pos_ptr,
                           const int32_t *pos_items, int32_t n_items, float *x_out,
                           float *loss_out, void *stream) {
  if (B > kNoAllocMaxBatch)
    return rk::fail(RK_ERR_UNSUPPORTED,
                    "rk_bpr_fwd_noalloc_f32: batch of %lld samples exceeds the built-in scratch (%lld); "
                    "use rk_bpr_fwd_f32 with a caller workspace",
                    (long long)B, (long long)kNoAllocMaxBatch);
  float *part = nullptr;
  const int rc = scratch_for_current_device(&part);
  if (rc != RK_OK) return rc;
  // every partial the final reduction reads is written by this call's forward kernel first, so
  // nothing left in the array by an earlier (larger) call is ever read
  return rk_rank_loss_fwd_f32(U, I, D, users, pos, neg, B, R, mask, neg_rank, pos_ptr, pos_items,
                              n_items, kModeLogSigmoid, x_out, loss_out, part,
                              sizeof(float) * (2 * (kNoAllocMaxBatch / SPB) + 2), stream);
}

int rk_rank_loss_bwd_f32(const float *U, const float *I, int32_t D,
                         const int64_t *users, const int64_t *pos, const int64_t *neg,
                         int64_t B, int32_t R, const float *mask, const float *x,
                         const float *loss_out, const float *gout, int32_t mode, float *dU,
                         float *dI, void *stream) {
  RK_CHECK_ARG(mode >= 0 && mode <= 2, "unknown loss mode");
  RK_CHECK_ARG(B > 0 && R > 0, "empty batch");
  RK_CHECK_ARG(U && I && users && pos && neg && x && loss_out && gout && dU && dI, "null pointer");
  RK_CHECK_ARG(D > 0 && D % 4 == 0, "embedding width must be a multiple of 4");
  hipStream_t st = (hipStream_t)stream;
  if (B * (1 + (int64_t)R) > kDetMaxContrib || R + 1 > kWork) {
    // very large batches (or more negatives per sample than one flush of the deterministic kernel
    // holds): float atomics (sums of three or more contributions to one row may then differ in the
    // last bits from run to run)
    bpr_bwd_kernel<<<(int32_t)rk::ceil_div(B, SPB), 256, 0, st>>>(U, I, D, users, pos, neg, B, R, mask, x, loss_out,
                                                                   gout, mode, dU, dI);
    RK_CHECK_LAUNCH("bpr_bwd_kernel");
    return RK_OK;
  }
  const int64_t n_items_side = B * (1 + (int64_t)R);
  if (B <= kLdsKeys)
    bpr_bwd_det_kernel<false, true><<<(int32_t)rk::ceil_div(B, SPB), 256, 0, st>>>(U, I, D, users, pos, neg, B, R, mask,
                                                                                  x, loss_out, gout, mode, dU);
  else
    bpr_bwd_det_kernel<false, false><<<(int32_t)rk::ceil_div(B, SPB), 256, 0, st>>>(U, I, D, users, pos, neg, B, R, mask,
                                                                                   x, loss_out, gout, mode, dU);
  RK_CHECK_LAUNCH("bpr_bwd_det_kernel<users>");
  if (n_items_side <= kLdsKeys)
    bpr_bwd_det_kernel<true, true><<<(int32_t)rk::ceil_div(n_items_side, SPB), 256, 0, st>>>(
        U, I, D, users, pos, neg, B, R, mask, x, loss_out, gout, mode, dI);
  else
    bpr_bwd_det_kernel<true, false><<<(int32_t)rk::ceil_div(n_items_side, SPB), 256, 0, st>>>(
        U, I, D, users, pos, neg, B, R, mask, x, loss_out, gout, mode, dI);
  RK_CHECK_LAUNCH("bpr_bwd_det_kernel<items>");
  return RK_OK;
}

int rk_bpr_bwd_f32(const float *U, const float *I, int32_t D,
                   const int64_t *users, const int64_t *pos, const int64_t *neg,
                   int64_t B, int32_t R, const float *mask, const float *x,
                   const float *loss_out, const float *gout, float *dU,
                   float *dI, void *stream) {
  return rk_rank_loss_bwd_f32(U, I, D, users, pos, neg, B, R, mask, x, loss_out, gout,
                              kModeLogSigmoid, dU, dI, stream);
}

}  // extern "C"
