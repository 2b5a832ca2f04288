// Modality projection Y = [b ; X W^T + b] and its weight/bias gradients on the matrix cores, gfx950:
// at D = 64 (feature widths that are multiples of 32; of 384 or 512 for the gradient) on three-way
// bf16 splits with tall workgroup tiles and a wavefront that only issues the DMAs
// (proj_fwd_tall_kernel, proj_dw_tall_kernel, further down); every other shape on fp32 operands
// (v_mfma_f32_32x32x2_f32: proj_fwd_kernel, proj_dw_kernel).
//
// The reference pads the item feature matrix with one all-zero row per user
// and runs nn.Linear over all U+I rows; a zero row times W is exactly zero, so
// the user rows of the output are the bias and only the I item rows need a
// GEMM.  fp32-operand forward: 128 x 64 output tile per workgroup (4 wavefronts x 32 rows),
// K streamed 32 columns at a time through double-buffered, padded LDS tiles.
// Backward: dW = dY_items^T X is a GEMM whose reduction runs over the items;
// it is cut into item slices whose 64 x 128 partial tiles are summed in slice
// order by a second kernel (deterministic, no float atomics).
#include "bf16x3.h"
#include "common.h"
#include "slab_reduce.h"

namespace {

using rk::kWave;
using rk::floatx16;
using rk::floatx4;

constexpr int BM = 128, BN = 64, BK = 32;
constexpr int LDA = BK + 4;  // padded LDS row (floats): conflict-free b128 reads

// user rows of the output: the bias, four columns per thread (D is a multiple of 4, Y 16-byte aligned)
__global__ __launch_bounds__(256) void proj_fill_users_kernel(const float *__restrict__ b, int32_t D,
                                                              int64_t n4, float *__restrict__ Y) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (b) v = *reinterpret_cast<const float4 *>(b + (i * 4) % D);
  reinterpret_cast<float4 *>(Y)[i] = v;
}

__global__ __launch_bounds__(256) void proj_fwd_kernel(const float *__restrict__ X, int32_t n_items,
                                                       int32_t F, const float *__restrict__ W,
                                                       const float *__restrict__ b, int32_t D,
                                                       float *__restrict__ Yitems) {
  __shared__ __attribute__((aligned(16))) float As[2][BM * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDA];
  const int tid = threadIdx.x, wave = tid / kWave, lane = tid % kWave;
  const int col = lane & 31, half = lane >> 5;
  const int32_t m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

  float4 sa[4], sb[2];
  auto load = [&](int32_t k0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = tid + 256 * j, r = e >> 3, c = (e & 7) * 4;
      sa[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m0 + r < n_items && k0 + c < F)
        sa[j] = *reinterpret_cast<const float4 *>(X + (int64_t)(m0 + r) * F + k0 + c);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = tid + 256 * j, r = e >> 3, c = (e & 7) * 4;
      sb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n0 + r < D && k0 + c < F)
        sb[j] = *reinterpret_cast<const float4 *>(W + (int64_t)(n0 + r) * F + k0 + c);
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = tid + 256 * j;
      *reinterpret_cast<float4 *>(&As[buf][(e >> 3) * LDA + (e & 7) * 4]) = sa[j];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = tid + 256 * j;
      *reinterpret_cast<float4 *>(&Bs[buf][(e >> 3) * LDA + (e & 7) * 4]) = sb[j];
    }
  };

  floatx16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  const int32_t nk = (F + BK - 1) / BK;
  load(0);
  store(0);
  __syncthreads();
  for (int32_t kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load((kt + 1) * BK);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // lane half h takes columns 8j+4h .. 8j+4h+3 of the K tile
      const float4 a = *reinterpret_cast<const float4 *>(&As[buf][(wave * 32 + col) * LDA + 8 * j + 4 * half]);
      const float4 b0 = *reinterpret_cast<const float4 *>(&Bs[buf][col * LDA + 8 * j + 4 * half]);
      const float4 b1 = *reinterpret_cast<const float4 *>(&Bs[buf][(32 + col) * LDA + 8 * j + 4 * half]);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc[1], 0, 0, 0);
    }
    if (kt + 1 < nk) store(buf ^ 1);
    __syncthreads();
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


// ---------------------------------------------------------------------------------------------
// Forward for D = 64 and F a multiple of 32 on the bf16 matrix pipe (three-way operand splits,
// bf16x3.h) with TALL workgroup tiles.  What bounds a projection of 4096-wide features is the rate
// at which the L1s can be filled (every workgroup streams its rows of X once and re-reads the
// whole of W): a 128-row tile re-reads 0.5 GB of weights beside the 1.03 GB of features, a 256-row
// tile half of that, and 247 workgroups are one round on 256 CUs.
//   * X: 256 rows x 32 columns per stage by LDS-DMA (pieces of 1 KB, XOR-swizzled through the
//     choice of the global piece a lane fetches: the 32-byte fragment reads of 16 consecutive
//     rows fall on 16 different bank groups), three stages; each computing wavefront reads its
//     own 64 rows.
//   * W: split ONCE per call into bf16 planes laid out in fragment order (proj_split_w_kernel), so
//     a stage's 12 KB chunk lands in LDS by DMA and is read as it lies; shared by the four
//     wavefronts (one barrier per stage, which also says that everybody is done with the buffer
//     the next DMA overwrites).
// The DMAs are inline assembly (no register destination; the kernel's own s_waitcnt vmcnt(N)
// builtin in front of the barrier is the synchronisation; M0 written one wait state ahead).
constexpr int PR = 256;                 // rows of X per workgroup
constexpr int PK = 32;                  // columns of X per stage
constexpr int PXB = PR * PK * 4;        // bytes of X per stage (8 KB per computing wavefront)
// D = 64: three stages of 32 + 12 KB; D = 128: two of 32 + 24 KB (three do not fit 160 KB)
template <int D> struct TallFwd {
  static constexpr int NCB = D / 32;                    // blocks of 32 output columns, all of them in every wavefront
  static constexpr int WB = 3 * 2 * 2 * D * 16;         // bytes of split weights per stage: [part][k16][half][col][8 bf16]
  static constexpr int PART = 2 * 2 * D * 16;           // bytes of one part of a stage
  static constexpr int ST = D == 64 ? 3 : 2;            // stages in the ring
  static constexpr int NX = PXB / 1024, NW = WB / 1024; // DMA instructions per stage
};

template <int D>
__global__ __launch_bounds__(256) void proj_split_w_kernel(const float *__restrict__ W, int32_t F,
                                                           char *__restrict__ Wp) {
  // thread -> (stage c, k16 s, half h, column): eight consecutive k of one row of W
  using T = TallFwd<D>;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n = (int64_t)(F / PK) * 2 * 2 * D;
  if (t >= n) return;
  const int col = (int)(t % D), h = (int)((t / D) & 1), s2 = (int)((t / (2 * D)) & 1);
  const int64_t c = t / (4 * D);
  const float *src = W + (int64_t)col * F + c * PK + s2 * 16 + h * 8;
  const floatx4 v0 = *reinterpret_cast<const floatx4 *>(src), v1 = *reinterpret_cast<const floatx4 *>(src + 4);
  const float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
  rk::SplitN<8> sp;
  rk::split3(x, sp);
  char *dst = Wp + c * T::WB + (size_t)((s2 * 2 + h) * (D * 16) + col * 16);
  *reinterpret_cast<rk::bf16x8 *>(dst) = sp.h[0];
  *reinterpret_cast<rk::bf16x8 *>(dst + T::PART) = sp.m[0];
  *reinterpret_cast<rk::bf16x8 *>(dst + 2 * T::PART) = sp.l[0];
}

template <int D>
__global__ __launch_bounds__(320) void proj_fwd_tall_kernel(const float *__restrict__ X, int32_t n_items, int32_t F,
                                                            const char *__restrict__ Wp, const float *__restrict__ b,
                                                            int32_t steps_per_split, float *__restrict__ Yitems) {
  // blockIdx.y: which part of the feature columns (stages st_lo ..) this workgroup reduces over.  With
  // one part the result goes to Yitems with the bias; with several, part y's partial sums go to
  // Yitems + y * n_items * D (a slab in the workspace), and proj_sum_parts_kernel adds them in order.
  using T = TallFwd<D>;
  constexpr int NCB = T::NCB, ST = T::ST, SB = PXB + T::WB;
  __shared__ __attribute__((aligned(16))) char plds[ST * SB];
  const int tid = threadIdx.x, lane = tid % kWave;
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  const int64_t row0 = (int64_t)blockIdx.x * PR;
  const int st_lo = blockIdx.y * steps_per_split;
  const int n_steps = min(steps_per_split, F / PK - st_lo);

  if (wave == 4) {
    // The producer: a fifth wavefront that only issues the DMAs.  A DMA instruction is accepted at
    // the rate memory delivers, and issue is in order: a wavefront that stages its own rows sits in
    // front of its DMAs for longer than its arithmetic takes and the memory queues run dry while it
    // computes (measured: 289 us; a bare stream of the same tiles: 183 us).  Here the four computing
    // wavefronts never wait for an issue slot, and this one never does anything else.
    // Piece i of a stage: rows 8i .. 8i+7 of the workgroup's 256, eight 16-byte pieces each; LDS
    // position p of row r receives global piece p ^ ((r >> 1) & 7).
    uint32_t voff[T::NX];
#pragma unroll
    for (int i = 0; i < T::NX; ++i) {
      const int r = 8 * i + lane / 8;
      int64_t g = row0 + r;
      g = g < n_items ? g : n_items - 1;      // rows past the end re-read the last row; they are not stored
      voff[i] = (uint32_t)((g - row0) * F * 4 + (((lane & 7) ^ ((r >> 1) & 7)) * 16));
    }
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)plds);
    const uint32_t woff = (uint32_t)(lane * 16);
    auto issue = [&](int st) {
      const uint32_t buf = lds0 + (uint32_t)((st % ST) * SB);
      const char *xb = reinterpret_cast<const char *>(X + row0 * F + (int64_t)(st_lo + st) * PK);
      const char *wb = Wp + (int64_t)(st_lo + st) * T::WB;
#pragma unroll
      for (int i = 0; i < T::NW; ++i)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     : : "v"(woff + (uint32_t)(i * 1024)), "s"(wb), "s"(buf + (uint32_t)(PXB + i * 1024)) : "memory", "m0");
#pragma unroll
      for (int i = 0; i < T::NX; ++i)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     : : "v"(voff[i]), "s"(xb), "s"(buf + (uint32_t)(i * 1024)) : "memory", "m0");
    };
    // ST - 1 stages are requested ahead of the one being computed; the in-order counter holds 63
    // operations (44 per stage at D = 64, 56 at D = 128)
    constexpr int kPerStage = T::NX + T::NW;
    static_assert(kPerStage <= 63, "a stage's DMAs fit the memory counter");
    for (int s0 = 0; s0 < ST - 1 && s0 < n_steps; ++s0) issue(s0);
    for (int st = 0; st < n_steps; ++st) {
      // stage st has landed: the only DMAs issued after it are those of stage st + 1 (three-stage ring)
      if (ST == 3 && st + 1 < n_steps)
        __builtin_amdgcn_s_waitcnt(0x0F70 | (kPerStage & 15) | ((kPerStage >> 4) << 14));
      else
        __builtin_amdgcn_s_waitcnt(0x0F70);
      __builtin_amdgcn_s_barrier();      // stage st is there, and everybody is past stage st - 1, whose buffer comes next
      if (st + ST - 1 < n_steps) issue(st + ST - 1);
    }
    return;
  }

  const int col = lane & 31, half = lane >> 5;
  floatx16 acc[2][NCB];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.0f;

  for (int st = 0; st < n_steps; ++st) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");       // (the tiles change behind the compiler's back)
    const char *xs = plds + (st % ST) * SB + wave * 8192;
    const char *ws = plds + (st % ST) * SB + PXB;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      rk::SplitN<8> ax[2];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const int r = 32 * rb + col, q = 4 * s2 + 2 * half, sw = (r >> 1) & 7;
        const floatx4 v0 = *reinterpret_cast<const floatx4 *>(xs + r * 128 + ((q ^ sw) * 16));
        const floatx4 v1 = *reinterpret_cast<const floatx4 *>(xs + r * 128 + (((q + 1) ^ sw) * 16));
        const float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        rk::split3(x, ax[rb]);
      }
#pragma unroll
      for (int cp = 0; cp < NCB; cp += 2) {      // two column blocks at a time: four accumulators in turn
        rk::SplitN<8> bw[2];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
          const char *p = ws + (s2 * 2 + half) * (D * 16) + ((cp + c2) * 32 + col) * 16;
          bw[c2].h[0] = *reinterpret_cast<const rk::bf16x8 *>(p);
          bw[c2].m[0] = *reinterpret_cast<const rk::bf16x8 *>(p + T::PART);
          bw[c2].l[0] = *reinterpret_cast<const rk::bf16x8 *>(p + 2 * T::PART);
        }
        rk::mfma_split3_2x2(ax[0], ax[1], bw[0], bw[1], acc[0][cp], acc[0][cp + 1], acc[1][cp], acc[1][cp + 1]);
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this stage's tiles have been read before the next barrier frees them
  }
  // C layout: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * half
  float *out = Yitems + (int64_t)blockIdx.y * n_items * D;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const int n = 32 * cb + col;
    const float bias = (b && gridDim.y == 1) ? b[n] : 0.0f;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = row0 + 64 * wave + 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < n_items) out[m * D + n] = acc[rb][cb][r] + bias;
      }
  }
}

// Y = bias + part 0 + part 1 + ... (in this order), four columns per thread
__global__ __launch_bounds__(256) void proj_sum_parts_kernel(const float *__restrict__ parts, int32_t n_parts, int64_t n4,
                                                             const float *__restrict__ b, int32_t D, float *__restrict__ Y) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  floatx4 v = {0.f, 0.f, 0.f, 0.f};
  if (b) v = *reinterpret_cast<const floatx4 *>(b + (i * 4) % D);
  for (int32_t p = 0; p < n_parts; ++p) v += reinterpret_cast<const floatx4 *>(parts)[(int64_t)p * n4 + i];
  reinterpret_cast<floatx4 *>(Y)[i] = v;
}

// Parts of the feature columns a forward launch is cut into: few row tiles (a small item table) would
// leave most CUs without a workgroup, and a CU's L1 fills at a fixed rate however few of them work.
inline int tall_fwd_parts(int32_t n_items, int32_t F) {
  const int64_t tiles = rk::ceil_div(n_items, PR), steps = F / PK;
  int64_t parts = 256 / tiles;                   // one workgroup per CU at most
  if (parts > steps / 8) parts = steps / 8;      // eight stages per part at least
  return (int)(parts < 1 ? 1 : parts);
}

inline bool tall_fwd_shape(int32_t F, int32_t D) { return (D == 64 || D == 128) && F % PK == 0 && F >= 2 * PK; }
inline size_t tall_fwd_wbytes(int32_t F, int32_t D) { return (size_t)(F / PK) * (D == 64 ? TallFwd<64>::WB : TallFwd<128>::WB); }

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


// ---------------------------------------------------------------------------------------------
// dW = dY_items^T X in the same form (D = 64 or 128): the reduction over the items on the bf16 matrix pipe.
// A workgroup owns all D rows of dW x FT feature columns (D = 64: FT = 512 or 384, four wavefronts x NF
// blocks of 32; D = 128: FT = 256) and one slice of the items; a stage is 16 items x FT columns of X
// (16 to 32 KB by LDS-DMA from a fifth, producer wavefront, as in the forward) plus the stage's 6 (12) KB of dY, which is
// split ONCE per call into bf16 parts laid out in fragment order (proj_split_dy_kernel; items past
// the end are zero there, so the rows of X re-read in their place add nothing).  X enters the MFMA
// as the B operand, eight ITEMS per lane at one feature: eight 4-byte LDS reads per block of 32
// features, rows 8..15 of a stage shifted by 128 bytes (through the global piece a DMA lane fetches)
// so that the two halves of a wavefront read different banks.  Per slice a 64 x F partial result;
// the slices are summed in slice order (slab_reduce): deterministic, no float atomics.
constexpr int QI = 16;                              // items per stage
constexpr int QST = 3;                              // stages in the ring
// bytes of split dY per stage: [part][d block][half][d][8 items]
constexpr int dy_stage_bytes(int D) { return 3 * (D / 32) * 2 * 32 * 16; }

template <int D>
__global__ __launch_bounds__(256) void proj_split_dy_kernel(const float *__restrict__ dY, int32_t n_items,
                                                            int64_t n_stages, char *__restrict__ Yp) {
  // thread -> (stage, d block, half, d): eight consecutive items at one column of dY
  constexpr int NDB = D / 32, PART = NDB * 2 * 32 * 16;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_stages * (2 * D)) return;
  const int d = (int)(t & 31), h = (int)((t >> 5) & 1), db = (int)((t >> 6) % NDB);
  const int64_t st = t / (2 * D);
  float x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int64_t i = st * QI + 8 * h + j;
    x[j] = i < n_items ? dY[i * D + 32 * db + d] : 0.0f;
  }
  rk::SplitN<8> sp;
  rk::split3(x, sp);
  char *dst = Yp + st * dy_stage_bytes(D) + (size_t)((db * 2 + h) * 512 + d * 16);
  *reinterpret_cast<rk::bf16x8 *>(dst) = sp.h[0];
  *reinterpret_cast<rk::bf16x8 *>(dst + PART) = sp.m[0];
  *reinterpret_cast<rk::bf16x8 *>(dst + 2 * PART) = sp.l[0];
}

template <int NDB, int NF>
__global__ __launch_bounds__(320) void proj_dw_tall_kernel(const float *__restrict__ X, int32_t n_items, int32_t F,
                                                           const char *__restrict__ Yp, int32_t stages_per_slice,
                                                           int64_t n_stages, float *__restrict__ slab) {
  constexpr int FT = 128 * NF;                       // feature columns per workgroup
  constexpr int XB = QI * FT * 4;                    // bytes of X per stage
  constexpr int QYB = dy_stage_bytes(32 * NDB), YPART = NDB * 2 * 32 * 16;
  constexpr int NX = XB / 1024, NY = QYB / 1024;     // DMA instructions per stage
  constexpr int PPRW = FT / 4;                       // 16-byte pieces per row of a stage
  constexpr int PST = QST;
  __shared__ __attribute__((aligned(16))) char qlds[PST * (XB + QYB)];
  const int tid = threadIdx.x, lane = tid % kWave;
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  const int32_t f0 = blockIdx.x * FT;
  const int64_t s_lo = (int64_t)blockIdx.y * stages_per_slice;
  const int64_t s_hi = s_lo + stages_per_slice < n_stages ? s_lo + stages_per_slice : n_stages;
  const int n_steps = (int)(s_hi - s_lo);

  if (wave == 4) {
    // the producer (see proj_fwd_tall_kernel).  DMA instruction j of a stage fills LDS bytes
    // [1024 j, 1024 j + 1024): position p of row r receives global piece p ^ (8 * (r >> 3)).
    int rowj[NX], piecej[NX];
    uint32_t voff[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int e = 64 * j + lane;
      rowj[j] = e / PPRW;
      piecej[j] = (e % PPRW) ^ (8 * (rowj[j] >> 3));
      voff[j] = (uint32_t)(rowj[j] * F * 4 + piecej[j] * 16);
    }
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)qlds);
    const uint32_t yoff = (uint32_t)(lane * 16);
    auto issue = [&](int st) {
      const uint32_t buf = lds0 + (uint32_t)((st % PST) * (XB + QYB));
      const int64_t i0 = (s_lo + st) * QI;
      const char *yb = Yp + (s_lo + st) * QYB;
#pragma unroll
      for (int j = 0; j < NY; ++j)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     : : "v"(yoff + (uint32_t)(j * 1024)), "s"(yb), "s"(buf + (uint32_t)(XB + j * 1024)) : "memory", "m0");
      if (i0 + QI <= n_items) {
        const char *xb = reinterpret_cast<const char *>(X + i0 * F + f0);
#pragma unroll
        for (int j = 0; j < NX; ++j)
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                       : : "v"(voff[j]), "s"(xb), "s"(buf + (uint32_t)(j * 1024)) : "memory", "m0");
      } else {
        // the last stage of the table: rows past the end re-read the last row (their dY parts are zero)
#pragma unroll
        for (int j = 0; j < NX; ++j) {
          int64_t g = i0 + rowj[j];
          g = g < n_items ? g : n_items - 1;
          const char *addr = reinterpret_cast<const char *>(X + g * F + f0) + piecej[j] * 16;
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                       : : "v"(addr), "s"(buf + (uint32_t)(j * 1024)) : "memory", "m0");
        }
      }
    };
    constexpr int kPerStage = NX + NY;               // 28 to 38 of the 63 operations the counter holds
    if (n_steps > 0) issue(0);
    if (n_steps > 1) issue(1);
    for (int st = 0; st < n_steps; ++st) {
      if (st + 1 < n_steps)
        __builtin_amdgcn_s_waitcnt(0x0F70 | (kPerStage & 15) | ((kPerStage >> 4) << 14));
      else
        __builtin_amdgcn_s_waitcnt(0x0F70);
      __builtin_amdgcn_s_barrier();
      if (st + 2 < n_steps) issue(st + 2);
    }
    return;
  }

  const int col = lane & 31, half = lane >> 5;
  floatx16 acc[NDB][NF];
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int fb = 0; fb < NF; ++fb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[db][fb][r] = 0.0f;

  for (int st = 0; st < n_steps; ++st) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");       // (the tiles change behind the compiler's back)
    const char *xs = qlds + (st % PST) * (XB + QYB);
    const char *ys = xs + XB;
    rk::SplitN<8> ay[NDB], bx[NF];
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
      const char *p = ys + (db * 2 + half) * 512 + col * 16;
      ay[db].h[0] = *reinterpret_cast<const rk::bf16x8 *>(p);
      ay[db].m[0] = *reinterpret_cast<const rk::bf16x8 *>(p + YPART);
      ay[db].l[0] = *reinterpret_cast<const rk::bf16x8 *>(p + 2 * YPART);
    }
#pragma unroll
    for (int fb = 0; fb < NF; ++fb) {
      const int f = wave * (32 * NF) + 32 * fb + col;
      const char *p = xs + (8 * half) * (FT * 4) + (((f >> 2) ^ (8 * half)) * 16) + (f & 3) * 4;
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = *reinterpret_cast<const float *>(p + j * (FT * 4));
      rk::split3(x, bx[fb]);
    }
    // one MFMA of each accumulator in turn (per accumulator: mfma_split3's operations and order)
#define RK_DW_TERM(PA, PB)                                                                                       \
  _Pragma("unroll") for (int fb = 0; fb < NF; ++fb) _Pragma("unroll") for (int db = 0; db < NDB; ++db)            \
    acc[db][fb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ay[db].PA[0], bx[fb].PB[0], acc[db][fb], 0, 0, 0);
    RK_DW_TERM(l, h) RK_DW_TERM(h, l) RK_DW_TERM(m, m) RK_DW_TERM(m, h) RK_DW_TERM(h, m) RK_DW_TERM(h, h)
#undef RK_DW_TERM
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this stage's tiles have been read before the next barrier frees them
  }
  // slab[slice][d][f]; C layout: column (f) = lane & 31, row (d) = (r & 3) + 8 * (r >> 2) + 4 * half
  float *out = slab + (int64_t)blockIdx.y * (32 * NDB) * F;
#pragma unroll
  for (int fb = 0; fb < NF; ++fb) {
    const int32_t f = f0 + wave * (32 * NF) + 32 * fb + col;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = 32 * db + (r & 3) + 8 * (r >> 2) + 4 * half;
        out[(int64_t)d * F + f] = acc[db][fb][r];
      }
  }
}

// feature columns per workgroup of the tall dW kernel for a shape (0: the plain kernel takes it)
inline int tall_dw_ft(int32_t F, int32_t D) {
  if (D == 128) return F % 256 == 0 ? 256 : 0;      // sixteen accumulator blocks of a 512-column tile do not fit the registers
  if (D != 64) return 0;
  return F % 512 == 0 ? 512 : (F % 384 == 0 ? 384 : 0);
}
struct TallDw {
  int ft, slices, stages_per_slice;
  int64_t n_stages;
};
inline TallDw tall_dw_plan(int32_t n_items, int32_t F, int32_t D) {
  TallDw p{};
  p.ft = tall_dw_ft(F, D);
  if (p.ft == 0 || n_items <= 0) return p;
  p.n_stages = rk::ceil_div(n_items, QI);
  const int64_t tiles = F / p.ft;
  int64_t want = rk::ceil_div(248, tiles);                       // about one workgroup per CU
  if (want > p.n_stages) want = p.n_stages;
  p.stages_per_slice = (int)rk::ceil_div(p.n_stages, want);
  p.slices = (int)rk::ceil_div(p.n_stages, p.stages_per_slice);
  return p;
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

size_t rk_proj_fwd_ws_bytes(int32_t n_items, int32_t F, int32_t D) {
  if (n_items <= 0 || F <= 0 || D <= 0 || !tall_fwd_shape(F, D)) return 0;
  const int parts = tall_fwd_parts(n_items, F);
  return tall_fwd_wbytes(F, D) + (parts > 1 ? (size_t)parts * n_items * D * sizeof(float) : 0);
}

int rk_proj_fwd_ws_f32(const float *X, int32_t n_items, int32_t F, const float *W, const float *b,
                       int32_t D, int32_t n_users, float *Y, void *ws, size_t ws_bytes, void *stream) {
  if (n_items <= 0 || F <= 0 || D <= 0 || !tall_fwd_shape(F, D))
    return rk_proj_fwd_f32(X, n_items, F, W, b, D, n_users, Y, stream);
  RK_CHECK_ARG(n_users >= 0, "bad sizes");
  RK_CHECK_ARG(X && W && Y && ws, "null pointer");
  RK_CHECK_ARG(rk::aligned16(X) && rk::aligned16(W) && rk::aligned16(Y) && rk::aligned16(b) && rk::aligned16(ws),
               "matrices, bias and workspace must be 16-byte aligned");
  RK_CHECK_ARG(ws_bytes >= rk_proj_fwd_ws_bytes(n_items, F, D), "workspace too small");
  // (the kernel addresses a workgroup's rows with 32-bit byte offsets from its first row)
  RK_CHECK_ARG((int64_t)PR * F * 4 < ((int64_t)1 << 32), "feature width too large");
  hipStream_t st = (hipStream_t)stream;
  const int64_t nu = (int64_t)n_users * D;
  if (nu > 0) {
    proj_fill_users_kernel<<<(unsigned)rk::ceil_div(nu / 4, 256), 256, 0, st>>>(b, D, nu / 4, Y);
    RK_CHECK_LAUNCH("proj_fill_users_kernel");
  }
  const int64_t nt = (int64_t)(F / PK) * 4 * D;
  if (D == 64)
    proj_split_w_kernel<64><<<(unsigned)rk::ceil_div(nt, 256), 256, 0, st>>>(W, F, (char *)ws);
  else
    proj_split_w_kernel<128><<<(unsigned)rk::ceil_div(nt, 256), 256, 0, st>>>(W, F, (char *)ws);
  RK_CHECK_LAUNCH("proj_split_w_kernel");
  const int parts = tall_fwd_parts(n_items, F);
  const int32_t sps = (int32_t)rk::ceil_div(F / PK, parts);
  const dim3 grid((unsigned)rk::ceil_div(n_items, PR), (unsigned)rk::ceil_div(F / PK, sps));
  float *slab = (float *)((char *)ws + tall_fwd_wbytes(F, D));
  float *dst = grid.y == 1 ? Y + nu : slab;
  if (D == 64)
    proj_fwd_tall_kernel<64><<<grid, 320, 0, st>>>(X, n_items, F, (const char *)ws, b, sps, dst);
  else
    proj_fwd_tall_kernel<128><<<grid, 320, 0, st>>>(X, n_items, F, (const char *)ws, b, sps, dst);
  RK_CHECK_LAUNCH("proj_fwd_tall_kernel");
  if (grid.y > 1) {
    const int64_t n4 = (int64_t)n_items * D / 4;
    proj_sum_parts_kernel<<<(unsigned)rk::ceil_div(n4, 256), 256, 0, st>>>(slab, (int32_t)grid.y, n4, b, D, Y + nu);
    RK_CHECK_LAUNCH("proj_sum_parts_kernel");
  }
  return RK_OK;
}

size_t rk_proj_bwd_ws_bytes(int32_t n_items, int32_t n_users, int32_t F, int32_t D) {
  if (n_items < 0 || n_users < 0 || F <= 0 || D <= 0) return 0;
  // dW slabs + segment sums (+ the split parts of dY for the tall kernel); the bias partials reuse the buffer afterwards
  const TallDw tp = tall_dw_plan(n_items, F, D);
  const int slices = tp.ft ? tp.slices : dw_slices(n_items, F, D);
  size_t slab = ((size_t)slices * D * F + rk::slab_tmp_floats(slices, (int64_t)D * F)) * sizeof(float);
  if (tp.ft) slab = ((slab + 255) & ~(size_t)255) + (size_t)tp.n_stages * dy_stage_bytes(D);
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
  const TallDw tp = tall_dw_plan(n_items, F, D);
  const int slices = tp.ft ? tp.slices : dw_slices(n_items, F, D);
  float *slab = (float *)ws;
  if (n_items > 0 && tp.ft) {
    RK_CHECK_ARG(rk::aligned16(X) && rk::aligned16(dY) && rk::aligned16(ws), "matrices and workspace must be 16-byte aligned");
    RK_CHECK_ARG((int64_t)QI * F * 4 < ((int64_t)1 << 32), "feature width too large");
    const size_t slab_bytes = ((size_t)slices * nW + rk::slab_tmp_floats(slices, nW)) * sizeof(float);
    char *yp = (char *)ws + ((slab_bytes + 255) & ~(size_t)255);
    const unsigned sgrid = (unsigned)rk::ceil_div(tp.n_stages * 2 * D, 256);
    if (D == 64)
      proj_split_dy_kernel<64><<<sgrid, 256, 0, st>>>(dY + (int64_t)n_users * D, n_items, tp.n_stages, yp);
    else
      proj_split_dy_kernel<128><<<sgrid, 256, 0, st>>>(dY + (int64_t)n_users * D, n_items, tp.n_stages, yp);
    RK_CHECK_LAUNCH("proj_split_dy_kernel");
    dim3 grid((unsigned)(F / tp.ft), (unsigned)slices);
    if (D == 128)
      proj_dw_tall_kernel<4, 2><<<grid, 320, 0, st>>>(X, n_items, F, yp, tp.stages_per_slice, tp.n_stages, slab);
    else if (tp.ft == 512)
      proj_dw_tall_kernel<2, 4><<<grid, 320, 0, st>>>(X, n_items, F, yp, tp.stages_per_slice, tp.n_stages, slab);
    else
      proj_dw_tall_kernel<2, 3><<<grid, 320, 0, st>>>(X, n_items, F, yp, tp.stages_per_slice, tp.n_stages, slab);
    RK_CHECK_LAUNCH("proj_dw_tall_kernel");
    const int rc = rk::slab_reduce(slab, slices, nW, nW, slab + (size_t)slices * nW, dW, st);
    if (rc != RK_OK) return rc;
  } else if (n_items > 0) {
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
