// The node-wise layer  Y = dropout(prelu(X1 W1^T [+ X2 W2^T] + b1 [+ b2])) [+ R]  and its backward
// for 128-wide embeddings (gfx950): the same operation as dense.hip computes for 64-wide ones, for
// the reference's emb_dim = 128 presets (config.py:201) and the large synthetic configuration.
//
// Arithmetic: v_mfma_f32_32x32x2_f32, i.e. fp32 operands and an fp32 fma chain (exact fp32
// products, no operand splitting).  At this width the layer has four times the arithmetic per byte
// of the 64-wide one and the fp32-input MFMA rate (157 TFLOP/s) is level with the memory time.
//
// Structure.  A workgroup of four wavefronts walks tiles of 64 rows (persistent grid).  The tile of
// X (64 x 128 fp32, rows padded to 132 floats: conflict-free 16-byte row reads and 4-byte column
// reads) is staged through LDS, the next tile's rows travelling in registers while this one is
// multiplied.  Wavefront w owns output columns 32w..32w+31; its 64 weight values per lane
// (W[n][64*half + s]) stay in registers for the whole launch.  MFMA operand maps (32x32x2): lane l
// supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; result register r of lane l is
// C[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].  The reduction index a lane supplies in step s is
// 64*(l>>5) + s: A and B only have to agree on it.
//
// Backward, per tile: Z = X W^T + b recomputed (activation only), dZ = mask * scale * G through
// the PReLU derivative, written to LDS; dX = dZ W (wavefront w: columns k = 32w..32w+31, W^T
// fragments in registers); dW[n][k] += sum_rows dZ[row][n] X[row][k] (wavefront w: n = 32w..32w+31,
// all four k tiles; the accumulators live in registers across the launch); db, d(alpha) likewise.
// Per-workgroup partials are added in workgroup order by a second launch: no float atomics.
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "dense_wide.h"

namespace {

using rk::kWave;
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int DW = 128;        // layer width (in = out)
constexpr int TR = 64;         // rows per workgroup tile
constexpr int LDP = DW + 4;    // padded LDS row (floats)
constexpr int PARTW = DW * DW + DW + 8;   // per-workgroup partial: dW, db, d(alpha) (+pad)
constexpr int NREDW = DW * DW + DW + 1;

__device__ __forceinline__ int crow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// the dropout hash of dense.hip: one hash per 2 x 2 block of (row, column), four 16-bit fields
struct DropQuad {
  uint32_t lo, hi;
};
__device__ __forceinline__ DropQuad drop_quad(uint32_t seed_hi, uint32_t row_pair, uint32_t col_term) {
  uint32_t x = row_pair * 0x9E3779B1u + col_term;   // col_term = (col >> 1) * 0x85EBCA77 ^ seed_lo
  x ^= x >> 15;
  x *= 0x2C1B3C6Du;
  x ^= x >> 12;
  x *= 0x297A2D39u;
  x ^= x >> 15;
  uint32_t y = (x ^ seed_hi) * 0x85EBCA6Bu;
  y ^= y >> 13;
  return DropQuad{x, y};
}
__device__ __forceinline__ bool kept(uint32_t word, int col_odd, uint32_t thresh16) {
  return ((word >> (16 * col_odd)) & 0xFFFFu) >= thresh16;
}

// rows [row0, row0 + 64) of M (128 floats each) into registers: thread t takes the 16-byte chunks
// t, t + 256, ... of the tile (a row is 32 chunks); rows past the end re-read the last row
struct TileRegs {
  floatx4 v[8];
};
__device__ __forceinline__ void load_tile(const float *__restrict__ M, int64_t row0, int64_t n_rows, int tid,
                                          TileRegs &t) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int f = tid + 256 * j;
    int64_t row = row0 + (f >> 5);
    row = row < n_rows ? row : n_rows - 1;
    t.v[j] = *reinterpret_cast<const floatx4 *>(M + row * DW + 4 * (f & 31));
  }
}
__device__ __forceinline__ void store_tile(float *lds, int tid, const TileRegs &t) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int f = tid + 256 * j;
    *reinterpret_cast<floatx4 *>(lds + (f >> 5) * LDP + 4 * (f & 31)) = t.v[j];
  }
}

// acc[rt] += tile rows (32*rt + i) times the lane's 64 fragment values (reduction index 64*half + s)
__device__ __forceinline__ void mma_rows(const float *tile, int i, int half, const float (&frag)[64],
                                         floatx16 (&acc)[2]) {
  const float *r0 = tile + i * LDP + 64 * half;
  const float *r1 = r0 + 32 * LDP;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const float4 a0 = *reinterpret_cast<const float4 *>(r0 + 4 * q);
    const float4 a1 = *reinterpret_cast<const float4 *>(r1 + 4 * q);
    const float x0[4] = {a0.x, a0.y, a0.z, a0.w}, x1[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[e], frag[4 * q + e], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[e], frag[4 * q + e], acc[1], 0, 0, 0);
    }
  }
}

__device__ __forceinline__ void zero(floatx16 (&acc)[2]) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
}

template <bool DUAL, bool HAS_R>
__global__ __launch_bounds__(256, 1) void dense128_fwd_kernel(
    const float *__restrict__ X1, const float *__restrict__ W1, const float *__restrict__ X2,
    const float *__restrict__ W2, const float *__restrict__ b1, const float *__restrict__ b2,
    const float *__restrict__ alpha, float p_drop, uint64_t seed, const uint64_t *__restrict__ seed_step,
    const float *__restrict__ R, int64_t n_rows, float *__restrict__ Y) {
  __shared__ __attribute__((aligned(16))) float xa[2][TR * LDP];
  __shared__ __attribute__((aligned(16))) float xb[DUAL ? 2 : 1][DUAL ? TR * LDP : 4];
  const int tid = threadIdx.x, wave = tid / kWave, lane = tid % kWave;
  const int i = lane & 31, half = lane >> 5;
  const int n = 32 * wave + i;                       // this lane's output column
  const int64_t n_tiles = (n_rows + TR - 1) / TR;
  const uint64_t sd = seed + (seed_step ? seed_step[0] * 0x9E3779B97F4A7C15ull : 0ull);
  const uint32_t seed_lo = (uint32_t)sd, seed_hi = (uint32_t)(sd >> 32);

  float w1[64], w2[DUAL ? 64 : 1];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const float4 v = *reinterpret_cast<const float4 *>(W1 + n * DW + 64 * half + 4 * q);
    w1[4 * q] = v.x, w1[4 * q + 1] = v.y, w1[4 * q + 2] = v.z, w1[4 * q + 3] = v.w;
    if constexpr (DUAL) {
      const float4 u = *reinterpret_cast<const float4 *>(W2 + n * DW + 64 * half + 4 * q);
      w2[DUAL ? 4 * q : 0] = u.x, w2[DUAL ? 4 * q + 1 : 0] = u.y, w2[DUAL ? 4 * q + 2 : 0] = u.z,
                       w2[DUAL ? 4 * q + 3 : 0] = u.w;
    }
  }
  float bias = b1 ? b1[n] : 0.0f;
  if (DUAL && b2) bias += b2[n];
  const float al = alpha ? alpha[0] : 1.0f;
  const bool drop = p_drop > 0.0f;
  const uint32_t thresh = (uint32_t)(p_drop * 65536.0f + 0.5f);
  const float scale = drop ? 1.0f / (1.0f - p_drop) : 1.0f;
  const uint32_t col_term = (uint32_t)(n >> 1) * 0x85EBCA77u ^ seed_lo;

  int64_t tile = blockIdx.x;
  if (tile >= n_tiles) return;   // uniform over the workgroup
  TileRegs ta, tb;
  load_tile(X1, tile * TR, n_rows, tid, ta);
  store_tile(xa[0], tid, ta);
  if constexpr (DUAL) {
    load_tile(X2, tile * TR, n_rows, tid, tb);
    store_tile(xb[0], tid, tb);
  }
  __syncthreads();
  int cur = 0;
  for (; tile < n_tiles; tile += gridDim.x, cur ^= 1) {
    const int64_t row0 = tile * TR;
    const int64_t next = tile + gridDim.x;
    if (next < n_tiles) {   // the next tile's rows travel while this one is multiplied
      load_tile(X1, next * TR, n_rows, tid, ta);
      if constexpr (DUAL) load_tile(X2, next * TR, n_rows, tid, tb);
    }
    floatx16 acc[2];
    zero(acc);
    mma_rows(xa[cur], i, half, w1, acc);
    if constexpr (DUAL) mma_rows(xb[DUAL ? cur : 0], i, half, w2, acc);
    // epilogue in accumulator layout: lane = column n, registers = rows
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int64_t row = row0 + 32 * rt + crow(r, half);       // even; row + 1 shares the hash
        DropQuad dq{0u, 0u};
        if (drop) dq = drop_quad(seed_hi, (uint32_t)(row >> 1), col_term);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float z = acc[rt][r + e] + bias;
          z = z > 0.0f ? z : al * z;
          if (drop) z = kept(e ? dq.hi : dq.lo, n & 1, thresh) ? z * scale : 0.0f;
          if (row + e < n_rows) {
            if constexpr (HAS_R) z += R[(row + e) * DW + n];
            Y[(row + e) * DW + n] = z;
          }
        }
      }
    if (next < n_tiles) {
      store_tile(xa[cur ^ 1], tid, ta);
      if constexpr (DUAL) store_tile(xb[DUAL ? cur ^ 1 : 0], tid, tb);
    }
    __syncthreads();
  }
}

template <bool ACT>
__global__ __launch_bounds__(256, 1) void dense128_bwd_kernel(
    const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ b,
    const float *__restrict__ alpha, float p_drop, uint64_t seed, const uint64_t *__restrict__ seed_step,
    const float *__restrict__ G, int64_t n_rows, float *__restrict__ dX, float *__restrict__ part) {
  __shared__ __attribute__((aligned(16))) float xs[TR * LDP];
  __shared__ __attribute__((aligned(16))) float gs[TR * LDP];   // G, then dZ in place
  __shared__ float red_b[4][2];
  const int tid = threadIdx.x, wave = tid / kWave, lane = tid % kWave;
  const int i = lane & 31, half = lane >> 5;
  const int n = 32 * wave + i;
  const int64_t n_tiles = (n_rows + TR - 1) / TR;
  const uint64_t sd = seed + (seed_step ? seed_step[0] * 0x9E3779B97F4A7C15ull : 0ull);
  const uint32_t seed_lo = (uint32_t)sd, seed_hi = (uint32_t)(sd >> 32);

  // W[n][64*half + s]: B operand of Z = X W^T;  W[64*half + s][32*wave + i]: B operand of dX = dZ W
  float wz[ACT ? 64 : 1], wx[64];
#pragma unroll
  for (int s = 0; s < 64; ++s) {
    if constexpr (ACT) wz[ACT ? s : 0] = W[n * DW + 64 * half + s];
    wx[s] = dX ? W[(64 * half + s) * DW + n] : 0.0f;
  }
  const float bias = (ACT && b) ? b[n] : 0.0f;
  const float al = ACT ? alpha[0] : 1.0f;
  const bool drop = p_drop > 0.0f;
  const uint32_t thresh = (uint32_t)(p_drop * 65536.0f + 0.5f);
  const float scale = drop ? 1.0f / (1.0f - p_drop) : 1.0f;
  const uint32_t col_term = (uint32_t)(n >> 1) * 0x85EBCA77u ^ seed_lo;

  floatx16 dw[4];   // dW[32*wave + crow(r)][32*kt + i]
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dw[kt][r] = 0.0f;
  float db_acc = 0.0f, da_acc = 0.0f;

  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row0 = tile * TR;
    {
      TileRegs tx, tg;
      load_tile(X, row0, n_rows, tid, tx);
      load_tile(G, row0, n_rows, tid, tg);
      store_tile(xs, tid, tx);
      store_tile(gs, tid, tg);
    }
    __syncthreads();
    floatx16 z[2];
    if constexpr (ACT) {
      zero(z);
      mma_rows(xs, i, half, wz, z);
    }
    // dZ in accumulator layout (lane = column n, registers = rows)
    float dz[2][16];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int lr = 32 * rt + crow(r, half);
        const int64_t row = row0 + lr;
        DropQuad dq{0u, 0u};
        if (drop) dq = drop_quad(seed_hi, (uint32_t)(row >> 1), col_term);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float g = row + e < n_rows ? gs[(lr + e) * LDP + n] : 0.0f;   // rows past the end: no gradient
          if (drop) g = kept(e ? dq.hi : dq.lo, n & 1, thresh) ? g * scale : 0.0f;
          if constexpr (ACT) {
            const float zz = z[rt][r + e] + bias;
            const bool pos = zz > 0.0f;
            da_acc += pos ? 0.0f : zz * g;
            g = pos ? g : al * g;
          }
          db_acc += g;
          dz[rt][r + e] = g;
        }
      }
    __syncthreads();   // every wavefront has read its columns of G
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) gs[(32 * rt + crow(r, half)) * LDP + n] = dz[rt][r];
    __syncthreads();   // the dZ tile is complete

    if (dX) {
      floatx16 dx[2];
      zero(dx);
      mma_rows(gs, i, half, wx, dx);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = row0 + 32 * rt + crow(r, half);
          if (row < n_rows) dX[row * DW + n] = dx[rt][r];
        }
    }
    // dW[n][k] += sum over the tile's rows of dZ[row][n] X[row][k]; the step's reduction index
    // (row) is 32*half + s
#pragma unroll 4
    for (int s = 0; s < 32; ++s) {
      const int row = 32 * half + s;
      const float a = gs[row * LDP + n];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
        dw[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xs[row * LDP + 32 * kt + i], dw[kt], 0, 0, 0);
    }
    __syncthreads();   // the tiles are free for the next iteration
  }

  float *out = part + (int64_t)blockIdx.x * PARTW;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(32 * wave + crow(r, half)) * DW + 32 * kt + i] = dw[kt][r];
  db_acc += __shfl_xor(db_acc, 32, kWave);            // lanes l and l+32: rows of the same column
  if (half == 0) out[DW * DW + n] = db_acc;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) da_acc += __shfl_xor(da_acc, off, kWave);
  if (lane == 0) red_b[wave][0] = da_acc;
  __syncthreads();
  if (tid == 0) out[DW * DW + DW] = ((red_b[0][0] + red_b[1][0]) + red_b[2][0]) + red_b[3][0];
}

// partials summed in workgroup order, 16 lanes per element (as dense.hip's reduction)
__global__ __launch_bounds__(256) void dense128_reduce_kernel(const float *__restrict__ slab, int32_t n_part,
                                                              float *__restrict__ dW, float *__restrict__ db,
                                                              float *__restrict__ dalpha, int accumulate) {
  __shared__ float red[16][17];
  const int el = threadIdx.x & 15, q = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + el;
  float s0 = 0.0f, s1 = 0.0f;
  if (idx < NREDW) {
    int32_t p = q;
    for (; p + 16 < n_part; p += 32) {
      s0 += slab[(int64_t)p * PARTW + idx];
      s1 += slab[(int64_t)(p + 16) * PARTW + idx];
    }
    if (p < n_part) s0 += slab[(int64_t)p * PARTW + idx];
  }
  red[q][el] = s0 + s1;
  __syncthreads();
  if (q != 0 || idx >= NREDW) return;
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += red[k][el];
  if (idx < DW * DW)
    dW[idx] = accumulate ? dW[idx] + s : s;
  else if (idx < DW * DW + DW) {
    if (db) db[idx - DW * DW] = accumulate ? db[idx - DW * DW] + s : s;
  } else if (dalpha)
    dalpha[0] = accumulate ? dalpha[0] + s : s;
}

constexpr int kMaxGrid = 512;
inline int grid_for(int64_t n_rows) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  return (int)std::min<int64_t>(rk::ceil_div(n_rows, TR), std::min(cus, kMaxGrid));
}

}  // namespace

namespace rk {

int dense128_fwd(const float *X1, const float *W1, const float *X2, const float *W2, const float *b1,
                 const float *b2, const float *alpha, float p_drop, uint64_t seed, const uint64_t *seed_step,
                 const float *residual, int64_t n_rows, float *Y, hipStream_t st) {
  const int grid = grid_for(n_rows);
  if (X2) {
    if (residual)
      dense128_fwd_kernel<true, true><<<grid, 256, 0, st>>>(X1, W1, X2, W2, b1, b2, alpha, p_drop, seed, seed_step,
                                                             residual, n_rows, Y);
    else
      dense128_fwd_kernel<true, false><<<grid, 256, 0, st>>>(X1, W1, X2, W2, b1, b2, alpha, p_drop, seed, seed_step,
                                                              nullptr, n_rows, Y);
  } else {
    if (residual)
      dense128_fwd_kernel<false, true><<<grid, 256, 0, st>>>(X1, W1, nullptr, nullptr, b1, nullptr, alpha, p_drop, seed,
                                                              seed_step, residual, n_rows, Y);
    else
      dense128_fwd_kernel<false, false><<<grid, 256, 0, st>>>(X1, W1, nullptr, nullptr, b1, nullptr, alpha, p_drop,
                                                               seed, seed_step, nullptr, n_rows, Y);
  }
  RK_CHECK_LAUNCH("dense128_fwd_kernel");
  return RK_OK;
}

size_t dense128_bwd_ws_bytes() { return (size_t)kMaxGrid * PARTW * sizeof(float); }

int dense128_bwd(const float *X, const float *W, const float *b, const float *alpha, float p_drop, uint64_t seed,
                 const uint64_t *seed_step, const float *G, int64_t n_rows, float *dX, float *dW, float *db,
                 float *dalpha, int accumulate, void *ws, hipStream_t st) {
  const int grid = grid_for(n_rows);
  if (alpha)
    dense128_bwd_kernel<true><<<grid, 256, 0, st>>>(X, W, b, alpha, p_drop, seed, seed_step, G, n_rows, dX, (float *)ws);
  else
    dense128_bwd_kernel<false><<<grid, 256, 0, st>>>(X, W, b, nullptr, p_drop, seed, seed_step, G, n_rows, dX,
                                                     (float *)ws);
  RK_CHECK_LAUNCH("dense128_bwd_kernel");
  dense128_reduce_kernel<<<(NREDW + 15) / 16, 256, 0, st>>>((const float *)ws, grid, dW, db, dalpha, accumulate);
  RK_CHECK_LAUNCH("dense128_reduce_kernel");
  return RK_OK;
}

}  // namespace rk
