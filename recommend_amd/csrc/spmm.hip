// CSR SpMM  out = A h  (optionally alpha*h0 + beta*(A h))  for gfx950.
//
// Layout.  One lane group of LPR = D/4 lanes owns one destination row; every
// lane keeps a float4 slice of the D-wide row, so a gathered source row is one
// coalesced D*4-byte read (256 B at D = 64).  The group loads LPR (col, val)
// pairs with one coalesced read and broadcasts them lane to lane (DPP
// row_newbcast at LPR = 16, shuffles otherwise); gathers are issued four at a
// time.  A row's tail batch is padded to a multiple of four by re-reading its
// last edge with weight 0, so there is no serial remainder loop.
//
// Row-length skew.  Rows with more than plan->long_thresh edges are handled by
// whole workgroups: a chunk of up to kChunk edges per workgroup, its 256/LPR
// groups each taking a contiguous slice, folded through LDS in a fixed order.
// A row that is a single chunk is stored directly; longer rows store one
// partial row per chunk, and the workgroup that finishes a row's last outstanding
// chunk (an arrival counter per such row; partials are stored and re-read with
// agent-scope accesses that go through to memory, so no cache write-back or
// invalidate is involved) adds the partials in slot order, whichever chunk that
// workgroup happened to hold.  Long chunks are scheduled first.  No float
// atomics anywhere: the result is bitwise reproducible.
#include "common.h"

namespace {

using rk::kWave;

__device__ __forceinline__ void fma4(float4 &a, float v, const float4 &x) {
  a.x = fmaf(v, x.x, a.x);
  a.y = fmaf(v, x.y, a.y);
  a.z = fmaf(v, x.z, a.z);
  a.w = fmaf(v, x.w, a.w);
}

__device__ __forceinline__ void add4(float4 &a, const float4 &b) {
  a.x += b.x;
  a.y += b.y;
  a.z += b.z;
  a.w += b.w;
}

// value of lane J of the caller's LPR-lane group
template <int LPR, int J>
__device__ __forceinline__ int bcast_i(int v) {
  if constexpr (LPR == 16) {
    // DPP row_newbcast:J -- lane J of each row of 16 lanes, no LDS traffic
    return __builtin_amdgcn_update_dpp(0, v, 0x150 + J, 0xF, 0xF, false);
  } else {
    return __shfl(v, J, LPR);
  }
}
template <int LPR, int J>
__device__ __forceinline__ float bcast_f(float v) {
  return __builtin_bit_cast(float, bcast_i<LPR, J>(__builtin_bit_cast(int, v)));
}

template <int LPR, int Q>
__device__ __forceinline__ void gather4(const float4 *__restrict__ h4, int32_t c_l, float v_l,
                                        int sub, float4 &acc) {
  const int32_t c0 = bcast_i<LPR, Q>(c_l), c1 = bcast_i<LPR, Q + 1>(c_l);
  const int32_t c2 = bcast_i<LPR, Q + 2>(c_l), c3 = bcast_i<LPR, Q + 3>(c_l);
  const float4 x0 = h4[(int64_t)c0 * LPR + sub];
  const float4 x1 = h4[(int64_t)c1 * LPR + sub];
  const float4 x2 = h4[(int64_t)c2 * LPR + sub];
  const float4 x3 = h4[(int64_t)c3 * LPR + sub];
  fma4(acc, bcast_f<LPR, Q>(v_l), x0);
  fma4(acc, bcast_f<LPR, Q + 1>(v_l), x1);
  fma4(acc, bcast_f<LPR, Q + 2>(v_l), x2);
  fma4(acc, bcast_f<LPR, Q + 3>(v_l), x3);
}

template <int LPR, int Q>
__device__ __forceinline__ void batch_from(const float4 *__restrict__ h4, int32_t c_l, float v_l,
                                           int sub, int cnt, float4 &acc) {
  if constexpr (Q < LPR) {
    if (Q < cnt) {
      gather4<LPR, Q>(h4, c_l, v_l, sub, acc);
      batch_from<LPR, Q + 4>(h4, c_l, v_l, sub, cnt, acc);
    }
  }
}

// Accumulate edges [b, e) into acc; called by all LPR lanes of a group together.
template <int LPR>
__device__ __forceinline__ void accumulate_edges(const int32_t *__restrict__ col,
                                                 const float *__restrict__ val,
                                                 const float4 *__restrict__ h4, int32_t b,
                                                 int32_t e, int sub, float4 &acc) {
  for (int32_t e0 = b; e0 < e; e0 += LPR) {
    const int cnt = min(LPR, e - e0);
    // lanes past the end of the row re-read its last edge with weight 0
    const int32_t mine = e0 + min(sub, cnt - 1);
    const int32_t c_l = col[mine];
    const float v_l = sub < cnt ? val[mine] : 0.0f;
    batch_from<LPR, 0>(h4, c_l, v_l, sub, cnt, acc);
  }
}

// epilogue of a finished row: kEpiNone out = A h; kEpiAxpby out = alpha*h0 + beta*(A h);
// kEpiScale out = beta*(A h) (the gradient of the axpby form with respect to h)
constexpr int kEpiNone = 0, kEpiAxpby = 1, kEpiScale = 2;
template <int LPR, int AXPBY>
__device__ __forceinline__ void store_row(float4 *__restrict__ out4, const float4 *__restrict__ h04,
                                          int64_t row, int sub, float4 acc, float alpha, float beta) {
  if (AXPBY == kEpiAxpby) {
    const float4 z = h04[row * LPR + sub];
    acc.x = alpha * z.x + beta * acc.x;
    acc.y = alpha * z.y + beta * acc.y;
    acc.z = alpha * z.z + beta * acc.z;
    acc.w = alpha * z.w + beta * acc.w;
  } else if (AXPBY == kEpiScale) {
    acc.x *= beta;
    acc.y *= beta;
    acc.z *= beta;
    acc.w *= beta;
  }
  out4[row * LPR + sub] = acc;
}

// A float4 as two 8-byte agent-scope relaxed atomic accesses: the stores are written through to
// memory and the loads are served from there, whichever CU and XCD the two sides run on.
__device__ __forceinline__ void store_through(float4 *p, const float4 &v) {
  unsigned long long *q = reinterpret_cast<unsigned long long *>(p);
  const unsigned long long lo = (unsigned long long)__float_as_uint(v.x) | ((unsigned long long)__float_as_uint(v.y) << 32);
  const unsigned long long hi = (unsigned long long)__float_as_uint(v.z) | ((unsigned long long)__float_as_uint(v.w) << 32);
  __hip_atomic_store(q, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 load_through(const float4 *p) {
  unsigned long long *q = reinterpret_cast<unsigned long long *>(const_cast<float4 *>(p));
  const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)),
                     __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32)));
}

// Edges per long-tier workgroup (rk_spmm_plan_host cuts the rows).  A chunk is a chain of dependent
// gather rounds (its 16 groups take four edges at a time), and on a small graph the longest chunk IS
// the hop: 2048 edges ran 20.3 / 24.4 / 42.9 / 83.3 us per hop at the Baby / Sports / Clothing / Elec
// shapes, 512 edges 13.1 / 18.7 / 34.0 / 81.0, 256 edges 12.7 / 19.7 / 34.5 / 82.4, 128 edges
// 14.1 / 21.7 / 36.6 / 101.7 (more partial rows to store and fold).
constexpr int kChunk = 512;

struct PlanArgs {
  int32_t long_thresh, n_chunks, n_multi;
  const int32_t *chunk_row, *chunk_begin, *chunk_end, *chunk_slot, *multi_slot_ptr;
  float4 *partial4;
  int32_t *arrive;
};

template <int LPR, int AXPBY>
__global__ __launch_bounds__(256) void spmm_rows_kernel(
    const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const float *__restrict__ val, int32_t n_rows, const float4 *__restrict__ h4,
    float4 *__restrict__ out4, const float4 *__restrict__ h04, float alpha, float beta,
    PlanArgs pl) {
  constexpr int GROUPS = 256 / LPR;
  __shared__ float4 red[GROUPS][LPR];
  const int tid = threadIdx.x;
  const int sub = tid % LPR, grp = tid / LPR;
  if ((int32_t)blockIdx.x >= pl.n_chunks) {
    // short tier: GROUPS consecutive rows per workgroup
    const int64_t row = (int64_t)((int32_t)blockIdx.x - pl.n_chunks) * GROUPS + grp;
    if (row >= n_rows) return;
    const int32_t b = rowptr[row], e = rowptr[row + 1];
    if (e - b > pl.long_thresh) return;  // owned by the long tier
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    accumulate_edges<LPR>(col, val, h4, b, e, sub, acc);
    store_row<LPR, AXPBY>(out4, h04, row, sub, acc, alpha, beta);
    return;
  }
  // long tier: this workgroup reduces one chunk of one row
  const int32_t chunk = blockIdx.x;
  const int32_t cb = pl.chunk_begin[chunk], ce = pl.chunk_end[chunk];
  const int32_t span = ((ce - cb + GROUPS - 1) / GROUPS + LPR - 1) / LPR * LPR;
  const int32_t b = min(cb + grp * span, ce), e = min(b + span, ce);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  accumulate_edges<LPR>(col, val, h4, b, e, sub, acc);
  red[grp][sub] = acc;
  __syncthreads();
  const int32_t slot = pl.chunk_slot[chunk];    // uniform over the workgroup
  if (grp == 0) {
    float4 s = red[0][sub];
#pragma unroll
    for (int g = 1; g < GROUPS; ++g) add4(s, red[g][sub]);
    if (slot < 0)
      store_row<LPR, AXPBY>(out4, h04, pl.chunk_row[chunk], sub, s, alpha, beta);
    else
      store_through(&pl.partial4[(int64_t)slot * LPR + sub], s);
  }
  if (slot < 0) return;
  // One chunk of a row cut into several.  The partial went out with write-through stores; once
  // every wavefront has drained its stores one thread counts this chunk in.  Whoever draws the
  // last ticket of the row sums the row's slots: it resets the counter for the next launch, and
  // one wavefront reads the slots back (loads that bypass this CU's L1 and this XCD's view of
  // other XCDs' lines) and adds them exactly as a separate launch over finished partials would
  // (GPW lane groups striding the slots, then a fixed shuffle tree), so the bits do not depend on
  // which chunk came last.  A release/acquire fence pair would do as well, but at agent scope
  // the release writes back the whole L2 -- with the output rows of the hop in it -- once per chunk.
  // HARDWARE-CONTRACT CODE, not HIP memory-model code: the ordering "partial stores reach memory,
  // then the ticket is drawn" rests on (a) agent-scope relaxed atomic stores being written through
  // to memory on gfx950 and counted by vmcnt until they are, (b) this s_waitcnt vmcnt(0) in every
  // storing wavefront, (c) the workgroup barrier in front of the one thread that draws the ticket.
  // The HIP memory model would ask for a release fence here; see the comment above for its cost.
  __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0)
  asm volatile("" ::: "memory");                // (compiler barrier only: no instruction)
  __syncthreads();                              // also: everybody is done reading red[]
  int32_t *flag = reinterpret_cast<int32_t *>(&red[0][0]);
  if (tid == 0) {
    int32_t lo = 0, hi = pl.n_multi;            // multi_slot_ptr[lo] <= slot < multi_slot_ptr[lo + 1]
    while (hi - lo > 1) {
      const int32_t mid = (lo + hi) >> 1;
      if (pl.multi_slot_ptr[mid] <= slot)
        lo = mid;
      else
        hi = mid;
    }
    const int32_t n_row_chunks = pl.multi_slot_ptr[lo + 1] - pl.multi_slot_ptr[lo];
    const int32_t ticket = __hip_atomic_fetch_add(&pl.arrive[lo], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = ticket == n_row_chunks - 1;
    if (last) __hip_atomic_store(&pl.arrive[lo], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    flag[0] = last ? lo : -1;
  }
  __syncthreads();
  const int32_t mi = flag[0];
  if (mi < 0 || tid >= kWave) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // no instruction: keeps the loads below the ticket
  {
    constexpr int GPW = kWave / LPR;
    const int g = tid / LPR;                    // tid < kWave here
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int32_t sl = pl.multi_slot_ptr[mi] + g; sl < pl.multi_slot_ptr[mi + 1]; sl += GPW)
      add4(acc, load_through(&pl.partial4[(int64_t)sl * LPR + sub]));
#pragma unroll
    for (int off = LPR; off < kWave; off <<= 1) {
      acc.x += __shfl_xor(acc.x, off, kWave);
      acc.y += __shfl_xor(acc.y, off, kWave);
      acc.z += __shfl_xor(acc.z, off, kWave);
      acc.w += __shfl_xor(acc.w, off, kWave);
    }
    if (g == 0) store_row<LPR, AXPBY>(out4, h04, pl.chunk_row[chunk], sub, acc, alpha, beta);
  }
}

template <int LPR, int AXPBY>
int launch_spmm(const int32_t *rowptr, const int32_t *col, const float *val, int32_t n_rows,
                const float *h, float *out, const float *h0, float alpha, float beta,
                const rk_spmm_plan *plan, hipStream_t st) {
  constexpr int GROUPS = 256 / LPR;
  const bool has_long = plan && plan->n_chunks > 0;
  PlanArgs pl;
  pl.long_thresh = has_long ? plan->long_thresh : INT32_MAX;
  pl.n_chunks = has_long ? plan->n_chunks : 0;
  pl.chunk_row = has_long ? plan->chunk_row : nullptr;
  pl.chunk_begin = has_long ? plan->chunk_begin : nullptr;
  pl.chunk_end = has_long ? plan->chunk_end : nullptr;
  pl.chunk_slot = has_long ? plan->chunk_slot : nullptr;
  pl.partial4 = has_long ? reinterpret_cast<float4 *>(plan->partial) : nullptr;
  pl.n_multi = has_long ? plan->n_multi : 0;
  pl.multi_slot_ptr = has_long ? plan->multi_slot_ptr : nullptr;
  pl.arrive = has_long ? plan->arrive : nullptr;
  const int64_t blocks = rk::ceil_div(n_rows, GROUPS) + pl.n_chunks;
  spmm_rows_kernel<LPR, AXPBY><<<(unsigned)blocks, 256, 0, st>>>(
      rowptr, col, val, n_rows, reinterpret_cast<const float4 *>(h),
      reinterpret_cast<float4 *>(out), reinterpret_cast<const float4 *>(h0), alpha, beta, pl);
  RK_CHECK_LAUNCH("spmm_rows_kernel");
  return RK_OK;
}

template <int LPR>
int dispatch_axpby(const int32_t *rowptr, const int32_t *col, const float *val, int32_t n_rows,
                   const float *h, float *out, const float *h0, float alpha, float beta,
                   const rk_spmm_plan *plan, hipStream_t st) {
  if (h0) return launch_spmm<LPR, kEpiAxpby>(rowptr, col, val, n_rows, h, out, h0, alpha, beta, plan, st);
  if (beta != 1.0f) return launch_spmm<LPR, kEpiScale>(rowptr, col, val, n_rows, h, out, h0, alpha, beta, plan, st);
  return launch_spmm<LPR, kEpiNone>(rowptr, col, val, n_rows, h, out, h0, alpha, beta, plan, st);
}

}  // namespace

extern "C" {

int rk_spmm_plan_host(const int32_t *ptr, int32_t n_rows, int32_t long_thresh,
                      int32_t *n_chunks_out, int32_t *n_multi_out, int32_t *n_slots_out,
                      int32_t *chunk_row, int32_t *chunk_begin, int32_t *chunk_end,
                      int32_t *chunk_slot, int32_t *multi_rows, int32_t *multi_slot_ptr) {
  RK_CHECK_ARG(ptr && n_rows >= 0 && long_thresh > 0, "bad plan arguments");
  RK_CHECK_ARG(n_chunks_out && n_multi_out && n_slots_out, "null count outputs");
  const bool fill = chunk_row && chunk_begin && chunk_end && chunk_slot && multi_rows && multi_slot_ptr;
  int32_t nc = 0, nm = 0, ns = 0;
  for (int32_t r = 0; r < n_rows; ++r) {
    const int32_t b = ptr[r], e = ptr[r + 1];
    if (e - b <= long_thresh) continue;
    const bool multi = e - b > kChunk;
    if (multi && fill) {
      multi_rows[nm] = r;
      multi_slot_ptr[nm] = ns;
    }
    for (int32_t c = b; c < e; c += kChunk) {
      if (fill) {
        chunk_row[nc] = r;
        chunk_begin[nc] = c;
        chunk_end[nc] = c + kChunk < e ? c + kChunk : e;
        chunk_slot[nc] = multi ? ns : -1;
      }
      if (multi) ++ns;
      ++nc;
    }
    if (multi) ++nm;
  }
  if (fill) multi_slot_ptr[nm] = ns;
  *n_chunks_out = nc;
  *n_multi_out = nm;
  *n_slots_out = ns;
  return RK_OK;
}

int rk_spmm_csr_f32(const int32_t *rowptr, const int32_t *col, const float *val, int32_t n_rows,
                    const float *h, int32_t D, float *out, const float *h0, float alpha,
                    float beta, const rk_spmm_plan *plan, void *stream) {
  RK_CHECK_ARG(n_rows >= 0, "negative row count");
  if (n_rows == 0) return RK_OK;
  RK_CHECK_ARG(rowptr && col && val && h && out, "null pointer");
  RK_CHECK_ARG(out != h, "out must not alias h");
  RK_CHECK_ARG(rk::aligned16(h) && rk::aligned16(out) && rk::aligned16(h0),
               "embedding matrices must be 16-byte aligned");
  if (plan && plan->n_chunks > 0) {
    RK_CHECK_ARG(plan->chunk_row && plan->chunk_begin && plan->chunk_end && plan->chunk_slot &&
                     plan->long_thresh > 0,
                 "incomplete spmm plan");
    RK_CHECK_ARG(plan->n_multi == 0 ||
                     (plan->multi_rows && plan->multi_slot_ptr && plan->partial && plan->arrive &&
                      rk::aligned16(plan->partial)),
                 "spmm plan has multi-chunk rows but no 16-byte aligned partial buffer or no arrival counters");
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
