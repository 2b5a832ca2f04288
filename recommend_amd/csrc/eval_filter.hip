// Candidate filter for the full-ranking evaluation (gfx950).
//
// The exact scores are fp32 fma chains on v_mfma_f32_32x32x2_f32, which runs at the vector-ALU
// rate and shares the issue port with the selection code.  Most of that work is spent on items
// that are nowhere near a user's top K.  The filter computes every score to 16 bits instead: each
// embedding is split into two bf16 numbers, x ~ h + m, and u.i ~ u_h.i_h + u_h.i_m + u_m.i_h on
// v_mfma_f32_32x32x16_bf16 (three matrix-pipe instructions of 32 cycles per 16 k, against
// eight of 64 cycles).  The same fused kernel keeps the K+4.. best filter scores per user; those
// candidates are then scored exactly (the documented fp32 chain, bit for bit), ranked by
// (score desc, id asc), and CERTIFIED: with |filter - exact| <= eps for every item, no item
// outside the candidates can reach the K-th exact score if (smallest candidate filter score) + eps
// is below it.  Users that fail the test (ties at the boundary, pathological data) are ranked by
// the exact kernel afterwards; it skips wavefronts whose users are all certified.  The result is
// therefore the exact kernel's result in every case.
#include "topk_filter.h"
#include "topk_kernel.h"

namespace {

// round-to-nearest-even bf16 of a finite float, as its 16-bit pattern
__device__ __forceinline__ uint32_t bf16_bits(float x) {
  const uint32_t u = __float_as_uint(x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

// one thread per 8 consecutive elements of a row: 16 B of high parts and 16 B of remainders
__global__ __launch_bounds__(256) void split_rows_kernel(const float *__restrict__ X, int64_t n_rows, int D,
                                                         float *__restrict__ split, float *__restrict__ norms,
                                                         float *__restrict__ max_norm) {
  const int per_row = D / 8;                               // 8 (D = 64) or 16 (D = 128) threads per row
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = t / per_row;
  const int seg = (int)(t % per_row);
  float ss = 0.0f;
  if (row < n_rows) {
    const float4 a = *reinterpret_cast<const float4 *>(X + row * D + 8 * seg);
    const float4 b = *reinterpret_cast<const float4 *>(X + row * D + 8 * seg + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t h[8], m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      h[j] = bf16_bits(v[j]);
      m[j] = bf16_bits(v[j] - __uint_as_float(h[j] << 16));   // the subtraction is exact in fp32
      ss = fmaf(v[j], v[j], ss);
    }
    uint4 hp, mp;
    hp.x = h[0] | (h[1] << 16); hp.y = h[2] | (h[3] << 16); hp.z = h[4] | (h[5] << 16); hp.w = h[6] | (h[7] << 16);
    mp.x = m[0] | (m[1] << 16); mp.y = m[2] | (m[3] << 16); mp.z = m[4] | (m[5] << 16); mp.w = m[6] | (m[7] << 16);
    float *out = split + row * D;                          // D bf16 = D/2 floats per plane
    *reinterpret_cast<uint4 *>(out + 4 * seg) = hp;
    *reinterpret_cast<uint4 *>(out + D / 2 + 4 * seg) = mp;
  }
  if (norms == nullptr && max_norm == nullptr) return;
  // the per_row threads of a row sit in adjacent lanes (per_row divides 64)
  for (int off = per_row / 2; off > 0; off >>= 1) ss += __shfl_xor(ss, off, rk::kWave);
  if (row < n_rows && seg == 0) {
    const float nrm = sqrtf(ss);
    if (norms) norms[row] = nrm;
    // non-negative floats order like their bit patterns
    if (max_norm) atomicMax(reinterpret_cast<unsigned int *>(max_norm), __float_as_uint(nrm));
  }
}

// the documented exact score: one fmaf per product, k = 0,32,1,33,...,31,63 inside each block of
// 64, starting from +0 (what v_mfma_f32_32x32x2_f32 computes with the exact kernel's operands)
__device__ __forceinline__ float exact_score(const float *__restrict__ u, const float *__restrict__ it, int D) {
  float acc = 0.0f;
  for (int c = 0; c < D / 64; ++c)
    for (int m = 0; m < 32; ++m) {
      acc = fmaf(it[64 * c + m], u[64 * c + m], acc);
      acc = fmaf(it[64 * c + 32 + m], u[64 * c + 32 + m], acc);
    }
  return acc;
}

__device__ __forceinline__ bool ranks_before(float sa, int32_t ia, float sb, int32_t ib) {
  const bool va = ia >= 0, vb = ib >= 0;
  if (va != vb) return va;
  if (!va) return false;
  return sa > sb || (sa == sb && ia < ib);
}

// one wavefront per user, one lane per candidate (kc <= 32)
__global__ __launch_bounds__(256) void refine_kernel(
    const float *__restrict__ Uemb, const float *__restrict__ Iemb, int32_t n_users, int32_t n_items, int D,
    const float *__restrict__ cand_s, const int32_t *__restrict__ cand_id, int kc, int K,
    const float *__restrict__ unorm, const float *__restrict__ imax, int32_t item_id_offset,
    float *__restrict__ out_s, int32_t *__restrict__ out_id, int32_t *__restrict__ done,
    int32_t *__restrict__ open_count, int32_t *__restrict__ open_list, int32_t open_cap) {
  const int wave = threadIdx.x / rk::kWave, lane = threadIdx.x % rk::kWave;
  const int32_t user = blockIdx.x * 4 + wave;
  if (user >= n_users) return;
  int32_t id = -1;
  float approx = -INFINITY;
  if (lane < kc) {
    id = cand_id[(int64_t)user * kc + lane];
    approx = cand_s[(int64_t)user * kc + lane];
  }
  const bool valid = id >= 0;
  float s = -INFINITY;
  if (valid) s = approx == kMasked ? kMasked : exact_score(Uemb + (int64_t)user * D, Iemb + (int64_t)id * D, D);
  // rank among the candidates by (score desc, id asc); ids are distinct
  int rank = 0;
  for (int l = 0; l < kc; ++l) {
    const float sl = __shfl(s, l, rk::kWave);
    const int32_t il = __shfl(id, l, rk::kWave);
    rank += ranks_before(sl, il, s, id) ? 1 : 0;
  }
  if (valid && rank < K) {
    out_s[(int64_t)user * K + rank] = s;
    out_id[(int64_t)user * K + rank] = id + item_id_offset;
  }
  // certification
  const int n_valid = __popcll(__ballot(valid));
  float a_min = valid ? approx : INFINITY;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) a_min = fminf(a_min, __shfl_xor(a_min, off, rk::kWave));
  const uint64_t kth = __ballot(valid && rank == K - 1);
  const float tau = kth ? __shfl(s, __ffsll((long long)kth) - 1, rk::kWave) : INFINITY;
  if (lane == 0) {
    // |filter - exact| <= 2^-12 |u| max|i|: the split leaves 2^-18 of each factor and the dropped
    // u_m.i_m term is of the same size (3 * 2^-18 together), 3D fp32 additions of the filter add at
    // most 2^-15.4 and the D roundings of the exact chain 2^-17 (D = 128): 2^-14.5, and a five-fold
    // margin on top.  The absolute term covers flushed denormals.  Non-finite embeddings make the
    // norms, hence eps, infinite or NaN: nothing is certified and the exact kernel ranks everyone.
    const float eps = 0x1p-12f * unorm[user] * imax[0] + 1e-30f;
    const bool ok = kc >= n_items || n_valid < kc || a_min + eps < tau;
    done[user] = ok ? 1 : 0;
    if (!ok) {   // left to the exact kernel: listed (in any order) while the list has room
      const int32_t pos = atomicAdd(open_count, 1);
      if (pos < open_cap) open_list[pos] = user;
    }
  }
}

template <int KC>
int launch_filter(int DC, int grid, hipStream_t st, const float *Us, int32_t n_users, const float *Is,
                  int32_t n_items, const int32_t *mask_ptr, const int32_t *mask_items, int32_t nsplit,
                  float *ps, int32_t *pi) {
  switch (DC) {
    case 1:
      score_topk_kernel<KC, 1, true><<<grid, 256, 0, st>>>(Us, n_users, Is, n_items, mask_ptr, mask_items,
                                                            nsplit, 0, nullptr, nullptr, nullptr, 0, ps, pi);
      break;
    case 2:
      score_topk_kernel<KC, 2, true><<<grid, 256, 0, st>>>(Us, n_users, Is, n_items, mask_ptr, mask_items,
                                                            nsplit, 0, nullptr, nullptr, nullptr, 0, ps, pi);
      break;
    default:
      return rk::fail(RK_ERR_UNSUPPORTED, "top-K filter: embedding width %d not built (64/128)", 64 * DC);
  }
  RK_CHECK_LAUNCH("score_topk_kernel (filter)");
  return RK_OK;
}

}  // namespace

namespace rk {

int split_rows(const float *X, int64_t n_rows, int D, float *split, float *norms, float *max_norm,
               hipStream_t st) {
  if (n_rows == 0) return RK_OK;
  const int64_t threads = n_rows * (D / 8);
  split_rows_kernel<<<(unsigned)ceil_div(threads, 256), 256, 0, st>>>(X, n_rows, D, split, norms, max_norm);
  RK_CHECK_LAUNCH("split_rows_kernel");
  return RK_OK;
}

int launch_filter_topk(int kc, int DC, int grid, hipStream_t st, const float *Usplit, int32_t n_users,
                       const float *Isplit, int32_t n_items, const int32_t *mask_ptr,
                       const int32_t *mask_items, int32_t nsplit, float *ps, int32_t *pi) {
  switch (kc) {
    case 8: return launch_filter<8>(DC, grid, st, Usplit, n_users, Isplit, n_items, mask_ptr, mask_items, nsplit, ps, pi);
    case 12: return launch_filter<12>(DC, grid, st, Usplit, n_users, Isplit, n_items, mask_ptr, mask_items, nsplit, ps, pi);
    case 16: return launch_filter<16>(DC, grid, st, Usplit, n_users, Isplit, n_items, mask_ptr, mask_items, nsplit, ps, pi);
    case 20: return launch_filter<20>(DC, grid, st, Usplit, n_users, Isplit, n_items, mask_ptr, mask_items, nsplit, ps, pi);
    case 24: return launch_filter<24>(DC, grid, st, Usplit, n_users, Isplit, n_items, mask_ptr, mask_items, nsplit, ps, pi);
    case 28: return launch_filter<28>(DC, grid, st, Usplit, n_users, Isplit, n_items, mask_ptr, mask_items, nsplit, ps, pi);
    case 32: return launch_filter<32>(DC, grid, st, Usplit, n_users, Isplit, n_items, mask_ptr, mask_items, nsplit, ps, pi);
    default: return fail(RK_ERR_UNSUPPORTED, "top-K filter: candidate list length %d not built", kc);
  }
}

int refine_candidates(const float *Uemb, const float *Iemb, int32_t n_users, int32_t n_items, int D,
                      const float *cand_s, const int32_t *cand_id, int kc, int K, const float *unorm,
                      const float *imax, int32_t item_id_offset, float *out_s, int32_t *out_id,
                      int32_t *done, int32_t *open_count, int32_t *open_list, int32_t open_cap,
                      hipStream_t st) {
  refine_kernel<<<(unsigned)ceil_div(n_users, 4), 256, 0, st>>>(Uemb, Iemb, n_users, n_items, D, cand_s, cand_id, kc, K,
                                                                 unorm, imax, item_id_offset, out_s, out_id, done,
                                                                 open_count, open_list, open_cap);
  RK_CHECK_LAUNCH("refine_kernel");
  return RK_OK;
}

}  // namespace rk
