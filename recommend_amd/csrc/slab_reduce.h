// Deterministic reduction of per-workgroup partial results ("slabs").
//
// slab[p * stride + e], p = 0..n_part-1, is summed over p for every element e in
// a fixed order: the partial range is cut into SEG contiguous segments; inside a
// segment four lanes per element stride the partials and are folded through LDS;
// the segment sums are then added in segment order.  No float atomics, so the
// result does not depend on scheduling.  Used for weight/bias gradients.
#pragma once
#include "common.h"

namespace rk {
namespace {

constexpr int kSlabMaxSeg = 16;

__global__ __launch_bounds__(256) void slab_reduce_stage1(const float *__restrict__ slab,
                                                          int32_t n_part, int64_t stride,
                                                          int64_t n_elem, int32_t n_seg,
                                                          float *__restrict__ tmp) {
  __shared__ float red[4][64];
  const int tid = threadIdx.x, q = tid >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + (tid & 63);
  const int32_t seg = blockIdx.y;
  const int32_t lo = (int32_t)((int64_t)n_part * seg / n_seg);
  const int32_t hi = (int32_t)((int64_t)n_part * (seg + 1) / n_seg);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < n_elem) {
    int32_t p = lo + q;
    for (; p + 12 < hi; p += 16) {
      s0 += slab[(int64_t)p * stride + e];
      s1 += slab[(int64_t)(p + 4) * stride + e];
      s2 += slab[(int64_t)(p + 8) * stride + e];
      s3 += slab[(int64_t)(p + 12) * stride + e];
    }
    for (; p < hi; p += 4) s0 += slab[(int64_t)p * stride + e];
  }
  red[q][tid & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (q == 0 && e < n_elem)
    tmp[(int64_t)seg * n_elem + e] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

__global__ __launch_bounds__(256) void slab_reduce_stage2(const float *__restrict__ tmp,
                                                          int32_t n_seg, int64_t n_elem,
                                                          float *__restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n_elem) return;
  float s = 0.0f;
  for (int32_t k = 0; k < n_seg; ++k) s += tmp[(int64_t)k * n_elem + e];
  out[e] = s;
}

inline int slab_segments(int32_t n_part) {
  int s = n_part / 32;
  return s < 1 ? 1 : (s > kSlabMaxSeg ? kSlabMaxSeg : s);
}

// scratch floats needed besides the slab itself
inline size_t slab_tmp_floats(int32_t n_part, int64_t n_elem) {
  return (size_t)slab_segments(n_part) * (size_t)n_elem;
}

// out[e] = sum_p slab[p*stride + e]; tmp holds slab_tmp_floats(n_part, n_elem) floats
inline int slab_reduce(const float *slab, int32_t n_part, int64_t stride, int64_t n_elem, float *tmp,
                       float *out, hipStream_t st) {
  if (n_elem <= 0) return RK_OK;
  const int n_seg = slab_segments(n_part);
  dim3 grid((unsigned)ceil_div(n_elem, 64), (unsigned)n_seg);
  slab_reduce_stage1<<<grid, 256, 0, st>>>(slab, n_part, stride, n_elem, n_seg, n_seg == 1 ? out : tmp);
  RK_CHECK_LAUNCH("slab_reduce_stage1");
  if (n_seg > 1) {
    slab_reduce_stage2<<<(unsigned)ceil_div(n_elem, 256), 256, 0, st>>>(tmp, n_seg, n_elem, out);
    RK_CHECK_LAUNCH("slab_reduce_stage2");
  }
  return RK_OK;
}

// Column sums of a rows x D matrix, stage 1: workgroup b sums rows [b*kColsumRows, ...) into
// part[b*D + c]; finish with slab_reduce(part, n_blocks, D, D, ...).
constexpr int kColsumRows = 256;   // (1024 left three quarters of the CUs idle on a 62 k-row matrix)

__global__ __launch_bounds__(256) void colsum_stage1(const float *__restrict__ M, int64_t rows,
                                                     int32_t D, float *__restrict__ part) {
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * kColsumRows;
  const int64_t r1 = r0 + kColsumRows < rows ? r0 + kColsumRows : rows;
  const int32_t G = D / 4;   // 16-byte column groups per row
  if (D % 4 == 0 && G <= 256 && 256 % G == 0 && ((reinterpret_cast<uintptr_t>(M) | reinterpret_cast<uintptr_t>(part)) & 15) == 0) {
    // G lanes read one row as float4, 256 / G rows per pass, eight passes in flight: the kernel has
    // one workgroup per 1024 rows and lives on the loads it keeps outstanding
    __shared__ float4 red4[256];
    const int32_t g = tid % G, rr = tid / G, RP = 256 / G;
    float4 acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 *src = reinterpret_cast<const float4 *>(M) + g;
    int64_t r = r0 + rr;
    for (; r + 7 * RP < r1; r += 8 * RP) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(r + (int64_t)u * RP) * G];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc[u].x += v[u].x;
        acc[u].y += v[u].y;
        acc[u].z += v[u].z;
        acc[u].w += v[u].w;
      }
    }
    for (; r < r1; r += RP) {
      const float4 v = src[r * G];
      acc[0].x += v.x;
      acc[0].y += v.y;
      acc[0].z += v.z;
      acc[0].w += v.w;
    }
    float4 t = acc[0];
#pragma unroll
    for (int u = 1; u < 8; ++u) {
      t.x += acc[u].x;
      t.y += acc[u].y;
      t.z += acc[u].z;
      t.w += acc[u].w;
    }
    red4[tid] = t;
    __syncthreads();
    if (rr == 0) {
      for (int k = 1; k < RP; ++k) {   // fixed order
        const float4 o = red4[k * G + g];
        t.x += o.x;
        t.y += o.y;
        t.z += o.z;
        t.w += o.w;
      }
      reinterpret_cast<float4 *>(part + (int64_t)blockIdx.x * D)[g] = t;
    }
    return;
  }
  __shared__ float red[4][64];
  const int q = tid >> 6;
  for (int32_t c0 = 0; c0 < D; c0 += 64) {
    const int32_t c = c0 + (tid & 63);
    float s0 = 0.f, s1 = 0.f;
    if (c < D) {
      int64_t r = r0 + q;
      for (; r + 4 < r1; r += 8) {
        s0 += M[r * D + c];
        s1 += M[(r + 4) * D + c];
      }
      for (; r < r1; r += 4) s0 += M[r * D + c];
    }
    __syncthreads();
    red[q][tid & 63] = s0 + s1;
    __syncthreads();
    if (q == 0 && c < D)
      part[(int64_t)blockIdx.x * D + c] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  }
}

}  // namespace
}  // namespace rk
