// Modality projection Y = [b ; X W^T + b] and its weight/bias gradients on the matrix cores,
// gfx950: the forward multiplies three-way bf16 splits of the fp32 operands
// (v_mfma_f32_32x32x16_bf16, fp32 accumulation, fp32-level error), the weight gradient uses the
// fp32-input MFMA (v_mfma_f32_32x32x2_f32).
//
// The reference pads the item feature matrix with one all-zero row per user
// and runs nn.Linear over all U+I rows; a zero row times W is exactly zero, so
// the user rows of the output are the bias and only the I item rows need a
// GEMM.  Forward: 128 x 64 output tile per workgroup (4 wavefronts x 32 rows),
// K streamed 32 columns at a time, two tiles in flight in registers, split into
// bf16 parts on the way into double-buffered, XOR-swizzled LDS tiles.
// Backward: dW = dY_items^T X is a GEMM whose reduction runs over the items;
// it is cut into item slices whose 64 x 128 partial tiles are summed in slice
// order by a second kernel (deterministic, no float atomics).
#include "common.h"
#include "slab_reduce.h"

namespace {

using rk::kWave;
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 64, BK = 32;

// user rows of the output: the bias, four columns per thread (D is a multiple of 4, Y 16-byte aligned)
__global__ __launch_bounds__(256) void proj_fill_users_kernel(const float *__restrict__ b, int32_t D,
                                                              int64_t n4, float *__restrict__ Y) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (b) v = *reinterpret_cast<const float4 *>(b + (i * 4) % D);
  reinterpret_cast<float4 *>(Y)[i] = v;
}

// fp32 products on the bf16 matrix pipe: x = h + m + l with h, m, l bf16 (3 x 8 significant bits =
// the whole fp32 mantissa; the subtractions are exact) and
//   x * w = hh + hm + mh + mm + hl + lh   (+ ml + lm + ll, below 2^-24 |x| |w|, dropped):
// six v_mfma_f32_32x32x16_bf16 (32 cycles each) per 16 k against eight fp32-input MFMAs of 64.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct Parts4 {
  bf16x4 h, m, l;
};
__device__ __forceinline__ Parts4 split4(const floatx4 &v) {
  const float x[4] = {v.x, v.y, v.z, v.w};
  Parts4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const __bf16 h = (__bf16)x[j];
    const float r1 = x[j] - (float)h;
    const __bf16 m = (__bf16)r1;
    o.h[j] = h;
    o.m[j] = m;
    o.l[j] = (__bf16)(r1 - (float)m);
  }
  return o;
}
__device__ __forceinline__ void mfma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], floatx16 &acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}

// the same for two accumulators that share A, alternating so that no MFMA waits for the one before
__device__ __forceinline__ void mfma6x2(const bf16x8 (&a)[3], const bf16x8 (&b0)[3], const bf16x8 (&b1)[3],
                                        floatx16 &acc0, floatx16 &acc1) {
  constexpr int ia[6] = {2, 0, 1, 1, 0, 0}, ib[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
  for (int p = 0; p < 6; ++p) {
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ia[p]], b0[ib[p]], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ia[p]], b1[ib[p]], acc1, 0, 0, 0);
  }
}

// LDS tile of one operand: three bf16 planes (h, m, l) of ROWS x 32 k, 64 bytes per row, the 16-byte
// chunk c of row r stored at chunk c ^ ((r >> 2) & 3).  A ds_read_b128 is served in groups of 16
// lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, the same + 32) over a 256-byte bank row = 4 tile
// rows: the four rows of a group that share r mod 4 differ in (r >> 2) & 3, so the MFMA operand
// reads (row per lane, chunk 2s + half) are conflict free without padding.
__device__ __forceinline__ uint32_t tile_off(int r, int chunk) {
  return (uint32_t)(r * 64 + ((chunk ^ ((r >> 2) & 3)) * 16));
}

__global__ __launch_bounds__(256, 2) void proj_fwd_kernel(const float *__restrict__ X, int32_t n_items,
                                                          int32_t F, const float *__restrict__ W,
                                                          const float *__restrict__ b, int32_t D,
                                                          float *__restrict__ Yitems) {
  constexpr int APL = BM * 64, BPL = BN * 64;   // bytes per plane
  __shared__ __attribute__((aligned(16))) char As[2][3 * APL];
  __shared__ __attribute__((aligned(16))) char Bs[2][3 * BPL];
  const int tid = threadIdx.x, wave = tid / kWave, lane = tid % kWave;
  const int col = lane & 31, half = lane >> 5;
  const int32_t m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

  // Two K tiles are on their way from memory at any time (register sets 0 / 1): one step of
  // MFMAs is shorter than the memory latency, and with a single tile in flight every step ended
  // waiting for it.  Loads are unconditional (rows past the end re-read the last row: their
  // results are never stored; columns past F re-read the last chunk and are zeroed on their way
  // into LDS) and written as inline assembly with the kernel's own s_waitcnt in front of the
  // first use: across the loop's back edge the compiler's wait-count pass otherwise waits for the
  // newest loads (vmcnt(0)) before it issues the next tile's, which serialises memory and
  // arithmetic.  These loads are the only vector-memory operations in the loop.
  floatx4 sa[2][4], sb[2][2];
  auto load = [&](int32_t k0, floatx4 (&ra)[4], floatx4 (&rb)[2]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = tid + 256 * j, r = e >> 3, c = (e & 7) * 4;
      const float *src = X + (int64_t)min(m0 + r, n_items - 1) * F + min(k0 + c, F - 4);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[j]) : "v"(src));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = tid + 256 * j, r = e >> 3, c = (e & 7) * 4;
      const float *src = W + (int64_t)min(n0 + r, D - 1) * F + min(k0 + c, F - 4);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rb[j]) : "v"(src));
    }
  };
  // all loads but the newest `newer` have landed; the registers pass through the statement so that
  // no use of them can be scheduled ahead of it
  auto landed = [&](bool newer6, floatx4 (&ra)[4], floatx4 (&rb)[2]) {
    if (newer6)
      asm volatile("s_waitcnt vmcnt(6)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(rb[0]), "+v"(rb[1]));
    else
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(rb[0]), "+v"(rb[1]));
  };
  // every value is split once per workgroup, on its way from the registers into LDS
  const floatx4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto store = [&](int buf, int32_t k0, const floatx4 (&ra)[4], const floatx4 (&rb)[2]) {
    const bool k_in = k0 + (tid & 7) * 4 < F;   // (e & 7) = (tid & 7) for every j
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = tid + 256 * j, r = e >> 3, q = e & 7;   // 4 values = half a 16-byte chunk
      const Parts4 p = split4(k_in ? ra[j] : zero4);
      char *dst = &As[buf][tile_off(r, q >> 1) + 8 * (q & 1)];
      *reinterpret_cast<bf16x4 *>(dst) = p.h;
      *reinterpret_cast<bf16x4 *>(dst + APL) = p.m;
      *reinterpret_cast<bf16x4 *>(dst + 2 * APL) = p.l;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = tid + 256 * j, r = e >> 3, q = e & 7;
      const Parts4 p = split4(k_in ? rb[j] : zero4);
      char *dst = &Bs[buf][tile_off(r, q >> 1) + 8 * (q & 1)];
      *reinterpret_cast<bf16x4 *>(dst) = p.h;
      *reinterpret_cast<bf16x4 *>(dst + BPL) = p.m;
      *reinterpret_cast<bf16x4 *>(dst + 2 * BPL) = p.l;
    }
  };

  floatx16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  // Every workgroup starts its walk over K at another column: rows are F * 4 bytes apart (16 KB at
  // F = 4096), so workgroups in step would all ask the same few memory channels at any moment.
  const int32_t nk = (F + BK - 1) / BK;
  const int32_t kofs = (int32_t)((blockIdx.x * 37u) % (uint32_t)nk);
  auto ktile = [&](int32_t kt) { return ((kt + kofs) % nk) * BK; };
  // tile j waits in register set j & 1 until it is stored into LDS buffer j & 1
  auto step = [&](int32_t kt, int cur) {   // cur = kt & 1, a constant at each call site
    if (kt + 2 < nk) load(ktile(kt + 2), sa[cur], sb[cur]);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      // step s2 covers k = 16 s2 .. 16 s2 + 15 of the tile; lane half h supplies chunk 2 s2 + h
      bf16x8 a[3], b0[3], b1[3];
      const uint32_t oa = tile_off(wave * 32 + col, 2 * s2 + half);
      const uint32_t ob0 = tile_off(col, 2 * s2 + half), ob1 = tile_off(32 + col, 2 * s2 + half);
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        a[p] = *reinterpret_cast<const bf16x8 *>(&As[cur][p * APL + oa]);
        b0[p] = *reinterpret_cast<const bf16x8 *>(&Bs[cur][p * BPL + ob0]);
        b1[p] = *reinterpret_cast<const bf16x8 *>(&Bs[cur][p * BPL + ob1]);
      }
      mfma6x2(a, b0, b1, acc[0], acc[1]);
    }
    if (kt + 1 < nk) {
      landed(kt + 2 < nk, sa[cur ^ 1], sb[cur ^ 1]);
      store(cur ^ 1, ktile(kt + 1), sa[cur ^ 1], sb[cur ^ 1]);
    }
    __syncthreads();
  };
  load(ktile(0), sa[0], sb[0]);
  if (nk > 1) load(ktile(1), sa[1], sb[1]);
  landed(nk > 1, sa[0], sb[0]);
  store(0, ktile(0), sa[0], sb[0]);
  __syncthreads();
  for (int32_t kt = 0; kt < nk; kt += 2) {
    step(kt, 0);
    if (kt + 1 < nk) step(kt + 1, 1);
  }
  // C layout: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * half
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int32_t n = n0 + 32 * t + col;
    if (n >= D) continue;
    const float bias = b ? b[n] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int32_t m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (m < n_items) Yitems[(int64_t)m * D + n] = acc[t][r] + bias;
    }
  }
}

// dW partial: workgroup (f-tile, d-tile, slice) reduces items [i_lo, i_hi).
// A[i = d][k = item] = dY[item][d], B[k = item][j = f] = X[item][f].
constexpr int WF = 128;  // f columns per workgroup (4 wavefronts x 32)
constexpr int WI = 32;   // items per LDS step

__global__ __launch_bounds__(256) void proj_dw_kernel(const float *__restrict__ X, int32_t n_items,
                                                      int32_t F, const float *__restrict__ dYitems,
                                                      int32_t D, int32_t items_per_slice,
                                                      float *__restrict__ slab) {
  __shared__ __attribute__((aligned(16))) float Ys[2][WI * 64];
  __shared__ __attribute__((aligned(16))) float Xs[2][WI * WF];
  const int tid = threadIdx.x, wave = tid / kWave, lane = tid % kWave;
  const int col = lane & 31, half = lane >> 5;
  const int32_t f0 = blockIdx.x * WF, d0 = blockIdx.y * 64;
  const int32_t i_lo = blockIdx.z * items_per_slice;
  const int32_t i_hi = min(i_lo + items_per_slice, n_items);

  float4 sy[2], sx[4];
  auto load = [&](int32_t i0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {  // 32 x 64 floats = 512 float4
      const int e = tid + 256 * j, r = e >> 4, c = (e & 15) * 4;
      sy[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i0 + r < i_hi && d0 + c < D)
        sy[j] = *reinterpret_cast<const float4 *>(dYitems + (int64_t)(i0 + r) * D + d0 + c);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // 32 x 128 floats = 1024 float4
      const int e = tid + 256 * j, r = e >> 5, c = (e & 31) * 4;
      sx[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i0 + r < i_hi && f0 + c < F)
        sx[j] = *reinterpret_cast<const float4 *>(X + (int64_t)(i0 + r) * F + f0 + c);
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = tid + 256 * j;
      *reinterpret_cast<float4 *>(&Ys[buf][(e >> 4) * 64 + (e & 15) * 4]) = sy[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = tid + 256 * j;
      *reinterpret_cast<float4 *>(&Xs[buf][(e >> 5) * WF + (e & 31) * 4]) = sx[j];
    }
  };

  floatx16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  const int32_t steps = (i_hi - i_lo + WI - 1) / WI;
  if (steps > 0) {
    load(i_lo);
    store(0);
  }
  __syncthreads();
  for (int32_t s = 0; s < steps; ++s) {
    const int buf = s & 1;
    if (s + 1 < steps) load(i_lo + (s + 1) * WI);
#pragma unroll
    for (int kk = 0; kk < WI; kk += 2) {
      const float bx = Xs[buf][(kk + half) * WF + wave * 32 + col];
      const float a0 = Ys[buf][(kk + half) * 64 + col];
      const float a1 = Ys[buf][(kk + half) * 64 + 32 + col];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bx, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bx, acc[1], 0, 0, 0);
    }
    if (s + 1 < steps) store(buf ^ 1);
    __syncthreads();
  }
  // slab[slice][d][f]
  float *out = slab + (int64_t)blockIdx.z * D * F;
  const int32_t f = f0 + wave * 32 + col;
  if (f < F) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int32_t d = d0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (d < D) out[(int64_t)d * F + f] = acc[t][r];
      }
  }
}

inline int dw_slices(int32_t n_items, int32_t F, int32_t D) {
  const int64_t tiles = rk::ceil_div(F, WF) * rk::ceil_div(D, 64);
  int64_t s = rk::ceil_div(1024, tiles);
  const int64_t max_s = std::max<int64_t>(1, n_items / 256);
  if (s > max_s) s = max_s;
  return (int)std::max<int64_t>(1, s);
}

}  // namespace

extern "C" {

int rk_proj_fwd_f32(const float *X, int32_t n_items, int32_t F, const float *W, const float *b,
                    int32_t D, int32_t n_users, float *Y, void *stream) {
  RK_CHECK_ARG(n_items >= 0 && n_users >= 0 && F > 0 && D > 0, "bad sizes");
  RK_CHECK_ARG(F % 4 == 0 && D % 4 == 0, "feature and embedding widths must be multiples of 4");
  RK_CHECK_ARG(W && Y && (n_items == 0 || X), "null pointer");
  RK_CHECK_ARG(rk::aligned16(X) && rk::aligned16(W) && rk::aligned16(Y) && rk::aligned16(b),
               "matrices and bias must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int64_t nu = (int64_t)n_users * D;
  if (nu > 0) {
    proj_fill_users_kernel<<<(unsigned)rk::ceil_div(nu / 4, 256), 256, 0, st>>>(b, D, nu / 4, Y);
    RK_CHECK_LAUNCH("proj_fill_users_kernel");
  }
  if (n_items > 0) {
    dim3 grid((unsigned)rk::ceil_div(n_items, BM), (unsigned)rk::ceil_div(D, BN));
    proj_fwd_kernel<<<grid, 256, 0, st>>>(X, n_items, F, W, b, D, Y + nu);
    RK_CHECK_LAUNCH("proj_fwd_kernel");
  }
  return RK_OK;
}

size_t rk_proj_bwd_ws_bytes(int32_t n_items, int32_t n_users, int32_t F, int32_t D) {
  if (n_items < 0 || n_users < 0 || F <= 0 || D <= 0) return 0;
  // dW slabs + segment sums; the bias partials reuse the same buffer afterwards
  const int slices = dw_slices(n_items, F, D);
  const size_t slab = ((size_t)slices * D * F + rk::slab_tmp_floats(slices, (int64_t)D * F)) * sizeof(float);
  const int64_t nb = rk::ceil_div((int64_t)n_items + n_users, rk::kColsumRows);
  const size_t bias = ((size_t)nb * D + rk::slab_tmp_floats((int32_t)nb, D)) * sizeof(float);
  return slab > bias ? slab : bias;
}

int rk_proj_bwd_f32(const float *X, int32_t n_items, int32_t F, const float *dY, int32_t D,
                    int32_t n_users, float *dW, float *db, void *ws, size_t ws_bytes,
                    void *stream) {
  RK_CHECK_ARG(n_items >= 0 && n_users >= 0 && F > 0 && D > 0, "bad sizes");
  RK_CHECK_ARG(F % 4 == 0 && D % 4 == 0, "feature and embedding widths must be multiples of 4");
  RK_CHECK_ARG(dY && dW && ws && (n_items == 0 || X), "null pointer");
  RK_CHECK_ARG(ws_bytes >= rk_proj_bwd_ws_bytes(n_items, n_users, F, D), "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int64_t nW = (int64_t)D * F;
  const int slices = dw_slices(n_items, F, D);
  float *slab = (float *)ws;
  if (n_items > 0) {
    const int32_t ips = (int32_t)(rk::ceil_div(rk::ceil_div(n_items, slices), WI) * WI);
    dim3 grid((unsigned)rk::ceil_div(F, WF), (unsigned)rk::ceil_div(D, 64), (unsigned)slices);
    proj_dw_kernel<<<grid, 256, 0, st>>>(X, n_items, F, dY + (int64_t)n_users * D, D, ips, slab);
    RK_CHECK_LAUNCH("proj_dw_kernel");
    const int rc = rk::slab_reduce(slab, slices, nW, nW, slab + (size_t)slices * nW, dW, st);
    if (rc != RK_OK) return rc;
  } else {
    RK_CHECK_HIP(hipMemsetAsync(dW, 0, nW * sizeof(float), st));
  }
  if (db) {
    const int64_t rows = (int64_t)n_users + n_items;
    const int64_t nb = rk::ceil_div(rows, rk::kColsumRows);
    if (nb == 0) {
      RK_CHECK_HIP(hipMemsetAsync(db, 0, D * sizeof(float), st));
    } else {
      rk::colsum_stage1<<<(unsigned)nb, 256, 0, st>>>(dY, rows, D, slab);
      RK_CHECK_LAUNCH("colsum_stage1");
      const int rc = rk::slab_reduce(slab, (int32_t)nb, D, D, slab + (size_t)nb * D, db, st);
      if (rc != RK_OK) return rc;
    }
  }
  return RK_OK;
}

}  // extern "C"
