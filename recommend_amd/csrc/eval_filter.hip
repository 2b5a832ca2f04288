// Candidate filter for the full-ranking evaluation (gfx950).
//
// The exact scores are fp32 fma chains on v_mfma_f32_32x32x2_f32, which runs at the vector-ALU
// rate and shares the issue port with the selection code.  Most of that work is spent on items
// that are nowhere near a user's top K.  The filter computes every score in fp16 instead: both
// tables are copied to fp16 (scaled by powers of two into its range: every user row by its own
// exponent, the item table by one), and u.i comes from v_mfma_f32_32x32x16_f16 -- one
// matrix-pipe instruction of 32 cycles per 16 k against eight of 64 cycles.  The same fused kernel
// keeps the K+8 best filter scores per user; those candidates are then scored exactly (the
// documented fp32 chain, bit for bit), ranked by (score desc, id asc), and CERTIFIED: with
// |filter - exact| <= eps for every item, no item outside the candidates can reach the K-th exact
// score if (smallest candidate filter score) + eps is below it.  Users that fail the test (ties at
// the boundary, pathological data) are ranked by the exact kernel afterwards.  The result is
// therefore the exact kernel's result in every case.
#include <hip/hip_fp16.h>

#include <cstdlib>

#include "topk_filter.h"
#include "topk_kernel.h"
#include "topk_threshold.h"

namespace {

// exponent e with |x| * 2^-e in [0.5, 1) (0 for x = 0, inf, nan): rows scaled by 2^-e use the
// top of the fp16 range without overflowing
__device__ __forceinline__ int scale_exponent(float maxabs) {
  if (!(maxabs > 0.0f) || !(maxabs < INFINITY)) return 0;
  int e;
  (void)frexpf(maxabs, &e);
  return e;
}

// per-row statistics: one thread per 8 consecutive elements, the D/8 threads of a row in adjacent
// lanes.  norms[row] = |x|, row_maxabs[row] = max |x_k| (may be NULL); stats_part (may be NULL;
// 2 x kStatSlots, zeroed by the caller): partial maxima of the norms and of |x_k| over the table,
// spread over many slots because thousands of wavefronts hitting one address serialise
// (non-negative floats order like their bit patterns)
__global__ __launch_bounds__(256) void row_stats_kernel(const float *__restrict__ X, int64_t n_rows, int D,
                                                        float *__restrict__ norms,
                                                        float *__restrict__ row_maxabs,
                                                        unsigned int *__restrict__ stats_part,
                                                        __half *__restrict__ half_out, int32_t *__restrict__ row_exp) {
  const int per_row = D / 8;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = t / per_row;
  const int seg = (int)(t % per_row);
  float ss = 0.0f, mx = 0.0f;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (row < n_rows) {
    const float4 a = *reinterpret_cast<const float4 *>(X + row * D + 8 * seg);
    const float4 b = *reinterpret_cast<const float4 *>(X + row * D + 8 * seg + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ss = fmaf(v[j], v[j], ss);
      // a NaN must not get lost in the maximum: it makes the norm NaN already, which is what
      // disables certification
      mx = fmaxf(mx, fabsf(v[j]));
    }
  }
  for (int off = per_row / 2; off > 0; off >>= 1) {
    ss += __shfl_xor(ss, off, rk::kWave);
    mx = fmaxf(mx, __shfl_xor(mx, off, rk::kWave));
  }
  const float nrm = sqrtf(ss);
  if (row < n_rows && seg == 0) {
    if (norms) norms[row] = nrm;
    if (row_maxabs) row_maxabs[row] = mx;
  }
  // rows scaled by their OWN power of two (half_out given): every lane of the row's group holds the
  // row's maximum now and the eight values it loaded, so the fp16 row leaves from here -- what
  // to_half_rows_kernel does for a table that shares one exponent, which has to wait for the
  // table's statistics
  if (half_out != nullptr && row < n_rows) {
    const int e = scale_exponent(mx);
    if (row_exp && seg == 0) row_exp[row] = e;
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x0 = ldexpf(v[2 * j], -e), x1 = ldexpf(v[2 * j + 1], -e);
      w[j] = (uint32_t)__half_as_ushort(__float2half_rn(x0)) | ((uint32_t)__half_as_ushort(__float2half_rn(x1)) << 16);
    }
    *reinterpret_cast<uint4 *>(half_out + row * D + 8 * seg) = make_uint4(w[0], w[1], w[2], w[3]);
  }
  if (stats_part) {
    // one pair of atomics per wavefront; the maxima are taken on the bit patterns, so that a NaN
    // (above every number as an unsigned pattern) survives and reaches the error bound
    unsigned int bn = row < n_rows ? __float_as_uint(nrm) & 0x7FFFFFFFu : 0u;
    unsigned int bm = row < n_rows ? __float_as_uint(mx) : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      bn = max(bn, (unsigned int)__shfl_xor((int)bn, off, rk::kWave));
      bm = max(bm, (unsigned int)__shfl_xor((int)bm, off, rk::kWave));
    }
    if (threadIdx.x % rk::kWave == 0) {
      const int slot = (blockIdx.x * 4 + threadIdx.x / rk::kWave) % rk::kStatSlots;
      atomicMax(stats_part + slot, bn);
      atomicMax(stats_part + rk::kStatSlots + slot, bm);
    }
  }
}

// table_stats[0] = largest norm, table_stats[1] = largest |x_k|
__global__ __launch_bounds__(rk::kStatSlots) void table_stats_kernel(const unsigned int *__restrict__ stats_part,
                                                                      float *__restrict__ table_stats) {
  __shared__ unsigned int red[2][rk::kStatSlots / rk::kWave];
  unsigned int bn = stats_part[threadIdx.x], bm = stats_part[rk::kStatSlots + threadIdx.x];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    bn = max(bn, (unsigned int)__shfl_xor((int)bn, off, rk::kWave));
    bm = max(bm, (unsigned int)__shfl_xor((int)bm, off, rk::kWave));
  }
  if (threadIdx.x % rk::kWave == 0) {
    red[0][threadIdx.x / rk::kWave] = bn;
    red[1][threadIdx.x / rk::kWave] = bm;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    unsigned int b = 0;
    for (int i = 0; i < rk::kStatSlots / rk::kWave; ++i) b = max(b, red[threadIdx.x][i]);
    table_stats[threadIdx.x] = __uint_as_float(b);
  }
}

// X[n_rows][D] fp32 -> fp16 rows of x * 2^-e, round to nearest even, e = the table's exponent
// (table_stats[1]); rows scaled by their own exponent are converted by row_stats_kernel.  One
// thread per 8 elements = 16 bytes of output.
__global__ __launch_bounds__(256) void to_half_rows_kernel(const float *__restrict__ X, int64_t n_rows, int D,
                                                           const float *__restrict__ table_stats,
                                                           __half *__restrict__ out) {
  const int per_row = D / 8;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = t / per_row;
  const int seg = (int)(t % per_row);
  if (row >= n_rows) return;
  const int e = scale_exponent(table_stats[1]);
  const float4 a = *reinterpret_cast<const float4 *>(X + row * D + 8 * seg);
  const float4 b = *reinterpret_cast<const float4 *>(X + row * D + 8 * seg + 4);
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  uint32_t w[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // ldexp on the value, not a multiplication by 2^-e: e runs from -148 (subnormal rows) to 128,
    // where the factor itself is not an fp32 number; the result, at most 1 in magnitude, always is
    const float x0 = ldexpf(v[2 * j], -e), x1 = ldexpf(v[2 * j + 1], -e);
    w[j] = (uint32_t)__half_as_ushort(__float2half_rn(x0)) | ((uint32_t)__half_as_ushort(__float2half_rn(x1)) << 16);
  }
  *reinterpret_cast<uint4 *>(out + row * D + 8 * seg) = make_uint4(w[0], w[1], w[2], w[3]);
}

// The documented exact score of one candidate per lane: one fmaf per product, k = 0,32,1,33,...,
// 31,63 inside each block of 64, starting from +0 (what v_mfma_f32_32x32x2_f32 computes with the
// exact kernel's operands).  The chain runs in one lane, but the item rows do not come in by
// lane: a lane reading its own 4 D-byte row touches a cache line of its own in every load
// instruction, 64 lines per instruction, and the texture addresser takes one line per cycle
// (counters at 192k users x ~35 candidates: it is busy 95 % of the kernel, 118 M line accesses =
// 461 k per CU against 512 k cycles).  Here D / 4 lanes read one row side by side (two lines), 256 /
// D rows per instruction, eight instructions = a batch of 2048 / D rows, into `rows` (this
// wavefront's own 2048 / D rows of D + 4 floats: the padding spreads the lanes' 16-byte reads over
// the banks); then the lanes of the batch run the chain on their rows out of LDS.
// id < 0: no candidate in this lane (the result is -inf); candidates sit in lanes 0..n-1.
// u: this lane's user row, in LDS.
// This pays where every lane of the wavefront holds a candidate (refine_kernel: 266 -> 184 us at
// 192k users x 28).  Where one wavefront serves one user with ~35 candidates (refine_wide_kernel)
// the LDS buffer halves the wavefronts in flight and the kernel turns latency-bound: measured
// 244 us with exact_score below against 292-313 us staged, in batches of 32 or 16 rows.
constexpr int kStageLoads = 8, kStageBatches = 2;   // load instructions per batch; batches whose loads travel together
__device__ __forceinline__ float staged_exact_score(const float *__restrict__ u, const float *__restrict__ Iemb,
                                                    int D, int32_t id, int n, float *__restrict__ rows) {
  const int lane = threadIdx.x % rk::kWave;
  const int lpr = D / 4, rpi = rk::kWave / lpr, batch = kStageLoads * rpi, stride = D + 4;
  const int seg = lane % lpr, sub = lane / lpr;
  float s = -INFINITY;
  // The loads of kStageBatches batches are issued before the first is waited for: one memory round
  // trip per round, the batches then take their turns in the (small) LDS buffer.
  for (int b0 = 0; b0 < n; b0 += kStageBatches * batch) {
    float4 v[kStageBatches][kStageLoads];
    bool have[kStageBatches][kStageLoads];
#pragma unroll
    for (int h = 0; h < kStageBatches; ++h) {
      const int here = min(batch, n - b0 - h * batch);       // (<= 0: no such batch)
#pragma unroll
      for (int j = 0; j < kStageLoads; ++j) {
        const int c = j * rpi + sub;                           // the row of the batch this lane helps with
        const int32_t cid = __shfl(id, min(b0 + h * batch + c, (int)rk::kWave - 1));
        have[h][j] = c < here && cid >= 0;
        v[h][j] = have[h][j] ? *reinterpret_cast<const float4 *>(Iemb + (int64_t)cid * D + 4 * seg) : make_float4(0, 0, 0, 0);
      }
    }
#pragma unroll
    for (int h = 0; h < kStageBatches; ++h) {
      const int from = b0 + h * batch, here = min(batch, n - from);
      if (here <= 0) break;                                    // wave-uniform
#pragma unroll
      for (int j = 0; j < kStageLoads; ++j)
        if (have[h][j]) *reinterpret_cast<float4 *>(rows + (j * rpi + sub) * stride + 4 * seg) = v[h][j];
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wavefront's LDS writes are done
      if (lane >= from && lane < from + here && id >= 0) {
        const float *it = rows + (lane - from) * stride;
        float acc = 0.0f;
        for (int c = 0; c < D / 64; ++c)
#pragma unroll
          for (int m = 0; m < 32; m += 4) {
            const float4 ia = *reinterpret_cast<const float4 *>(it + 64 * c + m);
            const float4 ib = *reinterpret_cast<const float4 *>(it + 64 * c + 32 + m);
            const float4 ua = *reinterpret_cast<const float4 *>(u + 64 * c + m);
            const float4 ub = *reinterpret_cast<const float4 *>(u + 64 * c + 32 + m);
            acc = fmaf(ia.x, ua.x, acc); acc = fmaf(ib.x, ub.x, acc);
            acc = fmaf(ia.y, ua.y, acc); acc = fmaf(ib.y, ub.y, acc);
            acc = fmaf(ia.z, ua.z, acc); acc = fmaf(ib.z, ub.z, acc);
            acc = fmaf(ia.w, ua.w, acc); acc = fmaf(ib.w, ub.w, acc);
          }
        s = acc;
      }
      __builtin_amdgcn_wave_barrier();      // (the next batch overwrites the rows)
    }
  }
  return s;
}
// floats of one wavefront's row buffer; LDS of a 4-wavefront workgroup that stages rows this way
__host__ __device__ inline int staged_rows_floats(int D) { return kStageLoads * (256 / D) * (D + 4); }
inline size_t staged_rows_bytes(int D) { return (size_t)4 * staged_rows_floats(D) * sizeof(float); }

// The same chain for a lane that reads its own item row: the 16 loads of a block of 64 are issued
// together (loads issued eight at a time between the fmas made four round trips of the gather).
// u: the user's row through a wave-uniform pointer (scalar loads).
__device__ __forceinline__ float exact_score(const float *__restrict__ u, const float *__restrict__ it, int D) {
  float acc = 0.0f;
  for (int c = 0; c < D / 64; ++c) {
    float4 r[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) r[j] = *reinterpret_cast<const float4 *>(it + 64 * c + 4 * j);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const float4 ia = r[m], ib = r[8 + m];
      const float4 ua = *reinterpret_cast<const float4 *>(u + 64 * c + 4 * m);
      const float4 ub = *reinterpret_cast<const float4 *>(u + 64 * c + 32 + 4 * m);
      acc = fmaf(ia.x, ua.x, acc); acc = fmaf(ib.x, ub.x, acc);
      acc = fmaf(ia.y, ua.y, acc); acc = fmaf(ib.y, ub.y, acc);
      acc = fmaf(ia.z, ua.z, acc); acc = fmaf(ib.z, ub.z, acc);
      acc = fmaf(ia.w, ua.w, acc); acc = fmaf(ib.w, ub.w, acc);
    }
  }
  return acc;
}

// (score desc, id asc) as ONE unsigned order: the float's bits made monotone (-0 and +0 tie, as
// they compare) above the complemented id; 0 for "no entry", which nothing real ranks behind.
__device__ __forceinline__ unsigned long long rank_key(float s, int32_t id) {
  if (id < 0) return 0ull;
  uint32_t u = __float_as_uint(s == 0.0f ? 0.0f : s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)u << 32) | (uint32_t)(0x7FFFFFFF - id);
}

__device__ __forceinline__ bool ranks_before(float sa, int32_t ia, float sb, int32_t ib) {
  const bool va = ia >= 0, vb = ib >= 0;
  if (va != vb) return va;
  if (!va) return false;
  return sa > sb || (sa == sb && ia < ib);
}

// order of two entries of the candidate lists as the list merge defines it: real entries by (score
// desc, id asc), padding (id < 0) behind them in list position order
__device__ __forceinline__ bool list_beats(float sa, int32_t ia, int ja, float sb, int32_t ib, int jb) {
  const bool va = ia >= 0, vb = ib >= 0;
  if (va != vb) return va;
  if (!va) return ja < jb;
  return sa > sb || (sa == sb && (ia < ib || (ia == ib && ja < jb)));
}

constexpr int kRefineMaxD = 128;   // widest embedding the filter is built for
// half a wavefront per user, one lane per candidate (kc <= 32).  P = 1: cand_* holds one list of kc
// per user.  P = 2: it holds the sorted lists of two item slices per user ([user][2][kc]), and the
// half-wavefront first merges them to the best kc by filter score (every entry's rank = its place in
// its own list + a bisection of the other list through lane shuffles; the same order and padding
// rules as the list merge kernel) instead of a merge launch writing that list and this one reading it.
__global__ __launch_bounds__(256) void refine_kernel(
    const float *__restrict__ Uemb, const float *__restrict__ Iemb, int32_t n_users, int32_t n_items, int D,
    const float *__restrict__ cand_s, const int32_t *__restrict__ cand_id, int P, int kc, int K,
    const int32_t *__restrict__ mask_ptr, const int32_t *__restrict__ heavy_thresh, const float *__restrict__ unorm, const int32_t *__restrict__ uexp, const float *__restrict__ istats,
    int32_t item_id_offset, float *__restrict__ out_s, int32_t *__restrict__ out_id, int32_t *__restrict__ done,
    int32_t *__restrict__ open_count, int32_t *__restrict__ open_list, int32_t open_cap) {
  const int wave = threadIdx.x / rk::kWave, lane = threadIdx.x % rk::kWave;
  const int sub = lane >> 5, l32 = lane & 31;
  if ((blockIdx.x * 4 + wave) * 2 >= n_users) return;   // wave-uniform
  const int32_t user = (blockIdx.x * 4 + wave) * 2 + sub;
  const bool user_ok = user < n_users;
  int32_t id = -1;
  float approx = -INFINITY;
  if (P == 1) {
    if (user_ok && l32 < kc) {
      id = cand_id[(int64_t)user * kc + l32];
      approx = cand_s[(int64_t)user * kc + l32];
    }
  } else {
    __shared__ float m_s[8][32];
    __shared__ int32_t m_id[8][32];
    float sa = -INFINITY, sb = -INFINITY;
    int32_t ia = -1, ib = -1;
    if (user_ok && l32 < kc) {
      const int64_t base = (int64_t)user * 2 * kc + l32;
      sa = cand_s[base];
      ia = cand_id[base];
      sb = cand_s[base + kc];
      ib = cand_id[base + kc];
    }
    // entries of the other list that come before this lane's entry: bisection with a fixed trip
    // count (kc <= 32), every lane shuffling in every round
    int lo_a = 0, hi_a = kc, lo_b = 0, hi_b = kc;
#pragma unroll
    for (int it = 0; it < 6; ++it) {
      const int mid_a = (lo_a + hi_a) >> 1, mid_b = (lo_b + hi_b) >> 1;
      const float sbm = __shfl(sb, mid_a & 31, 32), sam = __shfl(sa, mid_b & 31, 32);
      const int32_t ibm = __shfl(ib, mid_a & 31, 32), iam = __shfl(ia, mid_b & 31, 32);
      if (lo_a < hi_a) {
        if (list_beats(sbm, ibm, kc + mid_a, sa, ia, l32)) lo_a = mid_a + 1; else hi_a = mid_a;
      }
      if (lo_b < hi_b) {
        if (list_beats(sam, iam, mid_b, sb, ib, kc + l32)) lo_b = mid_b + 1; else hi_b = mid_b;
      }
    }
    const int hw = wave * 2 + sub, rank_a = l32 + lo_a, rank_b = l32 + lo_b;
    if (l32 < kc) {
      if (rank_a < kc) {
        m_s[hw][rank_a] = ia >= 0 ? sa : -INFINITY;
        m_id[hw][rank_a] = ia;
      }
      if (rank_b < kc) {
        m_s[hw][rank_b] = ib >= 0 ? sb : -INFINITY;
        m_id[hw][rank_b] = ib;
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wavefront's LDS writes are done
    if (user_ok && l32 < kc) {
      approx = m_s[hw][l32];
      id = m_id[hw][l32];
    }
  }
  const bool valid = id >= 0;
  const bool masked = approx == kMasked;   // (a product of fp16 rows is at most D in magnitude)
  // the user's row: loaded once by the half-wavefront (a float4 per lane) into LDS, read from there
  // by every lane at the same address (each lane loading it for itself doubled the kernel's L1 traffic)
  __shared__ float4 urow[8][kRefineMaxD / 4];
  for (int j = l32; j < D / 4; j += 32)
    urow[wave * 2 + sub][j] = user_ok ? *reinterpret_cast<const float4 *>(Uemb + (int64_t)user * D + 4 * j) : make_float4(0, 0, 0, 0);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0): this wavefront's LDS writes are done
  extern __shared__ float4 staged_rows[];
  float s = staged_exact_score(reinterpret_cast<const float *>(urow[wave * 2 + sub]), Iemb, D, valid && !masked ? id : -1,
                               rk::kWave, reinterpret_cast<float *>(staged_rows) + wave * staged_rows_floats(D));
  if (valid && masked) s = kMasked;
  // rank among the candidates by (score desc, id asc); ids are distinct
  const unsigned long long key = rank_key(s, id);
  int rank = 0;
  // (all 32 lanes of the half, unrolled: lanes past kc hold key 0, and the shuffles of a loop with
  // a run-time trip count are waited for one pair at a time)
#pragma unroll
  for (int l = 0; l < 32; ++l) {
    const uint32_t hi = __shfl((uint32_t)(key >> 32), l, 32), lo = __shfl((uint32_t)key, l, 32);
    rank += (((unsigned long long)hi << 32) | lo) > key ? 1 : 0;
  }
  if (valid && rank < K) {
    out_s[(int64_t)user * K + rank] = s;
    out_id[(int64_t)user * K + rank] = id + item_id_offset;
  }
  // certification, in the units of the exact scores: the filter worked on u * 2^-eu and i * 2^-ei
  const int e_sum = (user_ok ? uexp[user] : 0) + scale_exponent(istats[1]);
  const float back = ldexpf(1.0f, e_sum);           // inf / 0 at absurd magnitudes: nothing certified
  const int n_valid = __popc((uint32_t)(__ballot(valid) >> (32 * sub)));
  // (an exact score that is NaN has no place in the order above: the exact kernel, in which it
  // never ranks, settles such a user)
  const bool has_nan = (uint32_t)(__ballot(valid && s != s) >> (32 * sub)) != 0;
  float a_min = valid ? (masked ? kMasked : approx * back) : INFINITY;
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) a_min = fminf(a_min, __shfl_xor(a_min, off, 32));
  const uint32_t kth = (uint32_t)(__ballot(valid && rank == K - 1) >> (32 * sub));
  const float tau_l = __shfl(s, kth ? __ffs((int)kth) - 1 : 0, 32);
  const float tau = kth ? tau_l : INFINITY;
  if (l32 == 0 && user_ok) {
    // |filter - exact| <= eps.  Rounding an element to fp16 changes it by at most 2^-11 of itself
    // (normal range) or 2^-25 (below 2^-14, after scaling): |du.i| + |u.di| + |du.di| <=
    // (2^-10 + 2^-22) |u| |i| + 2D 2^-25 2^(eu+ei); the fp32 accumulations of the filter and of the
    // exact chain add at most 2D 2^-24 |u| |i| = 2^-16 (D = 128).  1.25 * 2^-10 and D * 2^-22 bound
    // the two groups with a margin (the matrix unit honours fp16 subnormal inputs: checked with
    // tools/mfma_denorm_probe.hip).  Non-finite embeddings make a norm, hence eps, infinite or NaN:
    // nothing is certified then and the exact kernel ranks everyone.
    const float eps = 1.25f * 0x1p-10f * unorm[user] * istats[0] + (float)D * 0x1p-22f * back + 1e-30f;
    // Every candidate slot must be filled (kc is at most n_items here): a filter score that is
    // NaN never ranks, and a user with such scores has to be ranked by the exact kernel.
    // The K-th score must also lie above the mask score: the filter ranks masked items below
    // every real score of its (rescaled) units, the exact kernel ranks them at -1e10, above real
    // scores that are lower still.
    // (users with very long training lists were not masked by the filter pass: kHeavyUsers)
    const bool heavy = heavy_thresh != nullptr && mask_ptr[user + 1] - mask_ptr[user] > heavy_thresh[0];
    const bool ok = !heavy && n_valid == kc && !has_nan && (kc >= n_items || (a_min + eps < tau && tau > kMasked));
    done[user] = ok ? 1 : 0;
    if (!ok) {   // left to the exact kernel: listed (in any order) while the list has room
      const int32_t pos = atomicAdd(open_count, 1);
      if (pos < open_cap) open_list[pos] = user;
    }
  }
}

// The threshold filter's counterpart of refine_kernel: one wavefront per user, one lane per
// candidate.  The candidates are ALL items whose filter score is >= thr[user] (cnt per item slice,
// at most kCandCap taking part), in no particular order: every one is scored exactly, the best K by
// (score desc, id asc) are written, and done[user] = 1 iff no other item can rank among them: every
// item outside has a filter score below thr, hence an exact score below thr + eps.
__global__ __launch_bounds__(256) void refine_wide_kernel(
    const float *__restrict__ Uemb, const float *__restrict__ Iemb, int32_t n_users, int32_t n_items, int D,
    const uintx2 *__restrict__ cand, const int32_t *__restrict__ cnt, int nsplit, const float *__restrict__ thr, int K,
    const int32_t *__restrict__ mask_ptr, const int32_t *__restrict__ heavy_thresh, const float *__restrict__ unorm,
    const int32_t *__restrict__ uexp, const float *__restrict__ istats, int32_t item_id_offset,
    float *__restrict__ out_s, int32_t *__restrict__ out_id, int32_t *__restrict__ done,
    int32_t *__restrict__ open_count, int32_t *__restrict__ open_list, int32_t open_cap) {
  const int wave = threadIdx.x / rk::kWave, lane = threadIdx.x % rk::kWave;
  const int32_t user = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);   // (uniform: the user's row comes by scalar loads)
  if (user >= n_users) return;
  int32_t total = 0;
  bool over = false;
  int64_t my_place = -1;
  for (int sl = 0; sl < nsplit; ++sl) {
    int32_t c = cnt[(int64_t)user * nsplit + sl];
    if (c > kCandCap) {
      over = true;
      c = kCandCap;
    }
    if (lane >= total && lane < total + c) my_place = ((int64_t)user * nsplit + sl) * kCandCap + (lane - total);
    total += c;
  }
  if (total > rk::kWave) over = true;
  const bool valid = my_place >= 0;
  int32_t id = -1;
  float approx = -INFINITY;
  if (valid) {
    const uintx2 e = cand[my_place];
    approx = __uint_as_float(e.x);
    id = (int32_t)e.y;
  }
  const bool masked = approx == kMasked;   // (a product of fp16 rows is at most D in magnitude)
  const int n = __builtin_amdgcn_readfirstlane(min(total, (int32_t)rk::kWave));
  float s = -INFINITY;
  if (valid) s = masked ? kMasked : exact_score(Uemb + (int64_t)user * D, Iemb + (int64_t)id * D, D);
  // rank among the candidates by (score desc, id asc); ids are distinct
  const unsigned long long key = rank_key(s, valid ? id : -1);
  int rank = 0;
  for (int l = 0; l < n; ++l) {
    const uint32_t hi = __builtin_amdgcn_readlane((int)(key >> 32), l), lo = __builtin_amdgcn_readlane((int)key, l);
    rank += (((unsigned long long)hi << 32) | lo) > key ? 1 : 0;
  }
  if (valid && rank < K) {
    out_s[(int64_t)user * K + rank] = s;
    out_id[(int64_t)user * K + rank] = id + item_id_offset;
  }
  // certification, in the units of the exact scores: the filter worked on u * 2^-eu and i * 2^-ei
  const int e_sum = uexp[user] + scale_exponent(istats[1]);
  const float back = ldexpf(1.0f, e_sum);           // inf / 0 at absurd magnitudes: nothing certified
  const int n_valid = __popcll(__ballot(valid));
  // (an exact score that is NaN or -inf has no place in the order above: the exact kernel, in
  // which neither ever ranks, settles such a user)
  const bool has_nan = __ballot(valid && !(s > -INFINITY)) != 0;
  const uint64_t kth = __ballot(valid && rank == K - 1);
  const float tau_l = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), kth ? __ffsll((long long)kth) - 1 : 0));
  const float tau = kth ? tau_l : INFINITY;
  if (lane == 0) {
    // |filter - exact| <= eps, as in refine_kernel
    const float eps = 1.25f * 0x1p-10f * unorm[user] * istats[0] + (float)D * 0x1p-22f * back + 1e-30f;
    const float t = thr[user];
    // every item outside the candidates has a filter score below t (masked ones the mask score)
    const float bound = t == kMasked ? kMasked : t * back;
    // The K-th score must also lie above the mask score: the filter ranks masked items below every
    // real score of its (rescaled) units, the exact kernel ranks them at -1e10, above real scores
    // that are lower still.  (users with very long training lists were not masked: kHeavyUsers)
    const bool heavy = heavy_thresh != nullptr && mask_ptr[user + 1] - mask_ptr[user] > heavy_thresh[0];
    const bool ok = !heavy && !over && !has_nan && n_valid >= K &&
                    (total >= n_items || (bound + eps < tau && tau > kMasked));
    done[user] = ok ? 1 : 0;
    if (!ok) {   // left to the exact kernel: listed (in any order) while the list has room
      const int32_t pos = atomicAdd(open_count, 1);
      if (pos < open_cap) open_list[pos] = user;
    }
  }
}

// hist[b]: users whose training list has bit length b (length in [2^(b-1), 2^b))
__global__ __launch_bounds__(256) void mask_hist_kernel(const int32_t *__restrict__ mask_ptr, int32_t n_users,
                                                        int32_t *__restrict__ hist) {
  __shared__ int32_t h[32];
  if (threadIdx.x < 32) h[threadIdx.x] = 0;
  __syncthreads();
  const int32_t u = blockIdx.x * 256 + threadIdx.x;
  if (u < n_users) {
    const int32_t len = mask_ptr[u + 1] - mask_ptr[u];
    if (len > 0) atomicAdd(&h[32 - __clz(len)], 1);
  }
  __syncthreads();
  if (threadIdx.x < 32 && h[threadIdx.x] != 0) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
// heavy[32] = the smallest 2^(b-1) - 1 with at most kHeavyUsers users of bit length >= b, raised to `floor_len`
__global__ void heavy_thresh_kernel(int32_t *__restrict__ heavy, int32_t floor_len) {
  int32_t above = 0, b = 32;
  while (b > 1 && above + heavy[b - 1] <= kHeavyUsers) {
    --b;
    above += heavy[b];
  }
  // bit lengths >= b are heavy: lengths > 2^(b-1) - 1
  const int32_t t = b >= 32 ? INT32_MAX : (int32_t)((1u << (b - 1)) - 1u);
  heavy[32] = max(t, floor_len);
}

template <int KC>
int launch_filter(int DC, int grid, hipStream_t st, const float *Us, int32_t n_users, const float *Is,
                  int32_t n_items, const int32_t *mask_ptr, const int32_t *mask_items, int32_t nsplit,
                  float *ps, int32_t *pi, const int32_t *heavy_thresh) {
  switch (DC) {
    case 1:
      score_topk_kernel<KC, 1, true><<<grid, 256, 0, st>>>(Us, n_users, Is, n_items, mask_ptr, mask_items,
                                                            nsplit, 0, nullptr, nullptr, nullptr, 0, ps, pi, heavy_thresh);
      break;
    case 2:
      score_topk_kernel<KC, 2, true><<<grid, 256, 0, st>>>(Us, n_users, Is, n_items, mask_ptr, mask_items,
                                                            nsplit, 0, nullptr, nullptr, nullptr, 0, ps, pi, heavy_thresh);
      break;
    default:
      return rk::fail(RK_ERR_UNSUPPORTED, "top-K filter: embedding width %d not built (64/128)", 64 * DC);
  }
  RK_CHECK_LAUNCH("score_topk_kernel (filter)");
  return RK_OK;
}

}  // namespace

#ifdef RK_STAMP
extern "C" int rk_debug_stamps(unsigned long long *out, int reset) {
  if (reset) { unsigned long long z[10] = {0,0,0,0,0,0,0,0,0,0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), 10 * sizeof(unsigned long long));
}
#endif

namespace rk {

int half_rows(const float *X, int64_t n_rows, int D, bool per_row_scale, void *out, float *norms,
              float *row_maxabs, int32_t *row_exp, unsigned int *stats_part, float *table_stats,
              hipStream_t st) {
  if (n_rows == 0) return RK_OK;
  const unsigned grid = (unsigned)ceil_div(n_rows * (D / 8), 256);
  if (per_row_scale) {   // statistics and conversion in one pass over the rows
    row_stats_kernel<<<grid, 256, 0, st>>>(X, n_rows, D, norms, row_maxabs, nullptr, (__half *)out, row_exp);
    RK_CHECK_LAUNCH("row_stats_kernel");
    return RK_OK;
  }
  row_stats_kernel<<<grid, 256, 0, st>>>(X, n_rows, D, norms, nullptr, stats_part, nullptr, nullptr);
  RK_CHECK_LAUNCH("row_stats_kernel");
  table_stats_kernel<<<1, kStatSlots, 0, st>>>(stats_part, table_stats);
  RK_CHECK_LAUNCH("table_stats_kernel");
  to_half_rows_kernel<<<grid, 256, 0, st>>>(X, n_rows, D, table_stats, (__half *)out);
  RK_CHECK_LAUNCH("to_half_rows_kernel");
  return RK_OK;
}

int heavy_threshold(const int32_t *mask_ptr, int32_t n_users, int32_t n_items, int32_t *heavy, hipStream_t st) {
  mask_hist_kernel<<<(unsigned)ceil_div(n_users, 256), 256, 0, st>>>(mask_ptr, n_users, heavy);
  RK_CHECK_LAUNCH("mask_hist_kernel");
  heavy_thresh_kernel<<<1, 1, 0, st>>>(heavy, std::max(ceil_div(n_items, TI) / 4, (int64_t)32));
  RK_CHECK_LAUNCH("heavy_thresh_kernel");
  return RK_OK;
}

int launch_filter_topk(int kc, int DC, int grid, hipStream_t st, const float *Usplit, int32_t n_users,
                       const float *Isplit, int32_t n_items, const int32_t *mask_ptr,
                       const int32_t *mask_items, int32_t nsplit, float *ps, int32_t *pi,
                       const int32_t *heavy_thresh) {
  switch (kc) {
    case 8: return launch_filter<8>(DC, grid, st, Usplit, n_users, Isplit, n_items, mask_ptr, mask_items, nsplit, ps, pi, heavy_thresh);
    case 12: return launch_filter<12>(DC, grid, st, Usplit, n_users, Isplit, n_items, mask_ptr, mask_items, nsplit, ps, pi, heavy_thresh);
    case 16: return launch_filter<16>(DC, grid, st, Usplit, n_users, Isplit, n_items, mask_ptr, mask_items, nsplit, ps, pi, heavy_thresh);
    case 20: return launch_filter<20>(DC, grid, st, Usplit, n_users, Isplit, n_items, mask_ptr, mask_items, nsplit, ps, pi, heavy_thresh);
    case 24: return launch_filter<24>(DC, grid, st, Usplit, n_users, Isplit, n_items, mask_ptr, mask_items, nsplit, ps, pi, heavy_thresh);
    case 28: return launch_filter<28>(DC, grid, st, Usplit, n_users, Isplit, n_items, mask_ptr, mask_items, nsplit, ps, pi, heavy_thresh);
    case 32: return launch_filter<32>(DC, grid, st, Usplit, n_users, Isplit, n_items, mask_ptr, mask_items, nsplit, ps, pi, heavy_thresh);
    default: return fail(RK_ERR_UNSUPPORTED, "top-K filter: candidate list length %d not built", kc);
  }
}

int refine_candidates(const float *Uemb, const float *Iemb, int32_t n_users, int32_t n_items, int D,
                      const float *cand_s, const int32_t *cand_id, int P, int kc, int K, const int32_t *mask_ptr,
                      const int32_t *heavy_thresh, const float *unorm,
                      const int32_t *uexp, const float *istats, int32_t item_id_offset, float *out_s,
                      int32_t *out_id, int32_t *done, int32_t *open_count, int32_t *open_list,
                      int32_t open_cap, hipStream_t st) {
  RK_CHECK_ARG(D % 64 == 0 && D <= kRefineMaxD, "top-K filter: embedding width %d not built (64/128)", D);
  refine_kernel<<<(unsigned)ceil_div(n_users, 8), 256, staged_rows_bytes(D), st>>>(Uemb, Iemb, n_users, n_items, D, cand_s, cand_id, P, kc, K,
                                                                 mask_ptr, heavy_thresh, unorm, uexp, istats, item_id_offset, out_s, out_id, done,
                                                                 open_count, open_list, open_cap);
  RK_CHECK_LAUNCH("refine_kernel");
  return RK_OK;
}

int threshold_candidates(int kc, int D, int32_t nsplit, hipStream_t st, const float *Uhalf, int32_t n_users,
                         const float *Ihalf, int32_t n_items, const int32_t *mask_ptr, const int32_t *mask_items,
                         const int32_t *heavy_thresh, float *strata, float *thr, void *cand, int32_t *cnt) {
  const int grid = (int)ceil_div(n_users, UPB) * nsplit;
  RK_CHECK_ARG(nsplit >= 1 && nsplit <= 16, "top-K filter: %d item slices (1..16)", nsplit);
  const int tgrid = (int)ceil_div(n_users, 4 * (64 / nsplit));   // 64 / nsplit users per wavefront
  if (D != 64 && D != 128) return fail(RK_ERR_UNSUPPORTED, "top-K filter: embedding width %d not built (64/128)", D);
  RK_CHECK_ARG(threshold_cand_bytes(n_users, nsplit) < ((size_t)1 << 32), "top-K filter: candidate places beyond 4 GiB");
  // two or more slices: 32 strata per slice are enough (64 and more per user)
  const bool half_strata = nsplit >= 2;
  const int32_t n_strata = half_strata ? kStrata / 2 : kStrata;
  if (D == 64 && half_strata)
    strata_max_kernel<1, 3, true><<<grid, 256, 0, st>>>(Uhalf, n_users, Ihalf, n_items, mask_ptr, mask_items, nsplit, heavy_thresh, strata);
  else if (D == 64)
    strata_max_kernel<1, 3><<<grid, 256, 0, st>>>(Uhalf, n_users, Ihalf, n_items, mask_ptr, mask_items, nsplit, heavy_thresh, strata);
  else if (half_strata)
    strata_max_kernel<2, 2, true><<<grid, 256, 0, st>>>(Uhalf, n_users, Ihalf, n_items, mask_ptr, mask_items, nsplit, heavy_thresh, strata);
  else
    strata_max_kernel<2, 2><<<grid, 256, 0, st>>>(Uhalf, n_users, Ihalf, n_items, mask_ptr, mask_items, nsplit, heavy_thresh, strata);
  RK_CHECK_LAUNCH("strata_max_kernel");
  switch (kc) {
    case 8: strata_threshold_kernel<8><<<tgrid, 256, 0, st>>>(strata, n_users, nsplit, n_strata, thr); break;
    case 12: strata_threshold_kernel<12><<<tgrid, 256, 0, st>>>(strata, n_users, nsplit, n_strata, thr); break;
    case 16: strata_threshold_kernel<16><<<tgrid, 256, 0, st>>>(strata, n_users, nsplit, n_strata, thr); break;
    case 20: strata_threshold_kernel<20><<<tgrid, 256, 0, st>>>(strata, n_users, nsplit, n_strata, thr); break;
    case 24: strata_threshold_kernel<24><<<tgrid, 256, 0, st>>>(strata, n_users, nsplit, n_strata, thr); break;
    case 28: strata_threshold_kernel<28><<<tgrid, 256, 0, st>>>(strata, n_users, nsplit, n_strata, thr); break;
    case 32: strata_threshold_kernel<32><<<tgrid, 256, 0, st>>>(strata, n_users, nsplit, n_strata, thr); break;
    default: return fail(RK_ERR_UNSUPPORTED, "top-K filter: candidate list length %d not built", kc);
  }
  RK_CHECK_LAUNCH("strata_threshold_kernel");
  if (D == 64)
    collect_kernel<1, 3><<<grid, 256, 0, st>>>(Uhalf, n_users, Ihalf, n_items, mask_ptr, mask_items, nsplit, heavy_thresh, thr,
                                               (uintx2 *)cand, cnt);
  else
    collect_kernel<2, 2><<<grid, 256, 0, st>>>(Uhalf, n_users, Ihalf, n_items, mask_ptr, mask_items, nsplit, heavy_thresh, thr,
                                               (uintx2 *)cand, cnt);
  RK_CHECK_LAUNCH("collect_kernel");
  return RK_OK;
}

int refine_threshold_candidates(const float *Uemb, const float *Iemb, int32_t n_users, int32_t n_items, int D,
                                const void *cand, const int32_t *cnt, int32_t nsplit, const float *thr, int K,
                                const int32_t *mask_ptr, const int32_t *heavy_thresh, const float *unorm,
                                const int32_t *uexp, const float *istats, int32_t item_id_offset, float *out_s,
                                int32_t *out_id, int32_t *done, int32_t *open_count, int32_t *open_list,
                                int32_t open_cap, hipStream_t st) {
  refine_wide_kernel<<<(unsigned)ceil_div(n_users, 4), 256, 0, st>>>(
      Uemb, Iemb, n_users, n_items, D, (const uintx2 *)cand, cnt, nsplit, thr, K, mask_ptr, heavy_thresh, unorm, uexp,
      istats, item_id_offset, out_s, out_id, done, open_count, open_list, open_cap);
  RK_CHECK_LAUNCH("refine_wide_kernel");
  return RK_OK;
}

size_t threshold_strata_bytes(int32_t n_users, int32_t nsplit) { return (size_t)n_users * nsplit * kStrata * sizeof(float); }
size_t threshold_cand_bytes(int32_t n_users, int32_t nsplit) { return (size_t)n_users * nsplit * kCandCap * 8; }

}  // namespace rk
