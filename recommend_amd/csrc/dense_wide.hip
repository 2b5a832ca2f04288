// The node-wise layer  Y = dropout(prelu(X1 W1^T [+ X2 W2^T] + b1 [+ b2])) [+ R]  and its backward
// for 128-wide embeddings (gfx950): the same operation as dense.hip computes for 64-wide ones, for
// the reference's emb_dim = 128 presets (config.py:201) and the large synthetic configuration.
//
// Forward arithmetic: fp32 products as three-way bf16 operand splits on the bf16 matrix pipe (bf16x3.h:
// six v_mfma_f32_32x32x16_bf16 per 16 k, fp32 sums inside the MFMA; the three dropped terms are below
// 2^-24 |x||w|), as in the 64-wide kernels.  The fp32-input MFMA this replaces runs at the vector
// ALU's rate and made the layer compute-bound at a third of the memory rate.
//
// Forward structure.  A workgroup of four wavefronts walks tiles of 64 rows (persistent grid).
// Wavefront w owns output columns 32w..32w+31 and keeps its 128 x 32 block of W, split, in registers
// for the whole launch (96 registers).  The X tile is shared: its rows travel global -> registers
// (the next tile while this one is multiplied), are split ONCE by the thread that loaded them and
// stored as three bf16 planes in LDS (rows padded to 136 elements: conflict-free 16-byte operand
// reads), from which all four wavefronts read their A operands ready-made.  Two plane buffers, one
// barrier per tile; the two-input form runs its inputs through the same buffers one after the other.
// MFMA operand maps (32x32x16): lane l supplies A[i = l&31][k = 8*(l>>5) + j] and
// B[k = 8*(l>>5) + j][j' = l&31], j < 8; result register r of lane l is
// C[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].
//
// Backward arithmetic: v_mfma_f32_32x32x2_f32 (fp32 operands, exact fp32 products).  Lane l supplies
// A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; the reduction index a lane supplies in step s is
// 64*(l>>5) + s: A and B only have to agree on it.  The tile of X (64 x 128 fp32, rows padded to 132
// floats) is staged through LDS.
//
// Backward, per tile: Z = X W^T + b recomputed (activation only), dZ = mask * scale * G through
// the PReLU derivative, written to LDS; dX = dZ W (wavefront w: columns k = 32w..32w+31, W^T
// fragments in registers); dW[n][k] += sum_rows dZ[row][n] X[row][k] (wavefront w: n = 32w..32w+31,
// all four k tiles; the accumulators live in registers across the launch); db, d(alpha) likewise.
// Per-workgroup partials are added in workgroup order by a second launch: no float atomics.
#include <string.h>

#include <algorithm>
#include <type_traits>

#include "bf16x3.h"
#include "common.h"
#include "dense_wide.h"

namespace {

using rk::bf16x8;
using rk::floatx16;
using rk::floatx4;
using rk::kWave;
using rk::SplitN;

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

// ---- forward: bf16 planes of the X tile in LDS
constexpr int LDB = DW + 8;          // padded plane row (bf16 elements): 272 bytes
constexpr int PLANE = TR * LDB;      // elements of one plane of one tile

// the thread's eight 16-byte chunks of a tile, split and stored as 8-byte packs into the three planes
__device__ __forceinline__ void split_store(__bf16 (*planes)[PLANE], int tid, const TileRegs &t) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int f = tid + 256 * j;
    rk::u32x2 h, m, l;
    rk::split3_quad(t.v[j][0], t.v[j][1], t.v[j][2], t.v[j][3], h, m, l);
    const int off = (f >> 5) * LDB + 4 * (f & 31);
    *reinterpret_cast<rk::u32x2 *>(&planes[0][off]) = h;
    *reinterpret_cast<rk::u32x2 *>(&planes[1][off]) = m;
    *reinterpret_cast<rk::u32x2 *>(&planes[2][off]) = l;
  }
}

// the lane's block of W as B operand: element 8s + j is W[n][16s + 8*half + j]
__device__ __forceinline__ void load_w_block(const float *__restrict__ W, int n, int half, SplitN<64> &w) {
  float wrow[64];
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float4 v = *reinterpret_cast<const float4 *>(W + n * DW + 16 * s + 8 * half + 4 * q);
      wrow[8 * s + 4 * q] = v.x, wrow[8 * s + 4 * q + 1] = v.y, wrow[8 * s + 4 * q + 2] = v.z, wrow[8 * s + 4 * q + 3] = v.w;
    }
  rk::split3(wrow, w);
}

// acc[rb] += rows 32*rb + i of the tile in `planes` times the lane's block of W.  The two row blocks'
// chains are issued alternately: an MFMA that accumulates onto the result of the one just before it
// waits for that result, which halves the issue rate of a single chain.
__device__ __forceinline__ void mma_planes(const __bf16 (*planes)[PLANE], int i, int half, const SplitN<64> &w,
                                           floatx16 (&acc)[2]) {
  const int base = i * LDB + 8 * half;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    bf16x8 ah[2], am[2], al[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      ah[rb] = *reinterpret_cast<const bf16x8 *>(&planes[0][base + 32 * rb * LDB + 16 * s]);
      am[rb] = *reinterpret_cast<const bf16x8 *>(&planes[1][base + 32 * rb * LDB + 16 * s]);
      al[rb] = *reinterpret_cast<const bf16x8 *>(&planes[2][base + 32 * rb * LDB + 16 * s]);
    }
    // the small terms first, the leading one last (as rk::mfma_split3)
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[0], w.h[s], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[1], w.h[s], acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[0], w.l[s], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[1], w.l[s], acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[0], w.m[s], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[1], w.m[s], acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[0], w.h[s], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[1], w.h[s], acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[0], w.m[s], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[1], w.m[s], acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[0], w.h[s], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[1], w.h[s], acc[1], 0, 0, 0);
  }
}

template <bool DUAL, bool HAS_R>
__global__ __launch_bounds__(256, (DUAL || HAS_R) ? 1 : 2) void dense128_fwd_kernel(
    const float *__restrict__ X1, const float *__restrict__ W1, const float *__restrict__ X2,
    const float *__restrict__ W2, const float *__restrict__ b1, const float *__restrict__ b2,
    const float *__restrict__ alpha, float p_drop, uint64_t seed, const uint64_t *__restrict__ seed_step,
    const float *__restrict__ R, int64_t n_rows, float *__restrict__ Y) {
  // Plain layer: one plane buffer (51 KB) and at most 256 registers, so that TWO workgroups share a
  // CU and one multiplies while the other waits for its rows or at a barrier (with one workgroup per
  // CU a single tile of loads in flight bounded the kernel at 3 TB/s).  The two-input and residual
  // forms need the registers of a whole SIMD lane: two buffers, one barrier per input.
  constexpr int NBUF = (DUAL || HAS_R) ? 2 : 1;
  __shared__ __attribute__((aligned(16))) __bf16 planes[NBUF][3][PLANE];
  const int tid = threadIdx.x, wave = tid / kWave, lane = tid % kWave;
  const int i = lane & 31, half = lane >> 5;
  const int n = 32 * wave + i;                       // this lane's output column
  const int64_t n_tiles = (n_rows + TR - 1) / TR;
  const uint64_t sd = seed + (seed_step ? seed_step[0] * 0x9E3779B97F4A7C15ull : 0ull);
  const uint32_t seed_lo = (uint32_t)sd, seed_hi = (uint32_t)(sd >> 32);

  int64_t tile = blockIdx.x;
  if (tile >= n_tiles) return;   // uniform over the workgroup
  TileRegs tr;
  load_tile(X1, tile * TR, n_rows, tid, tr);          // in flight while the weights are split

  SplitN<64> w1, w2;
  load_w_block(W1, n, half, w1);
  if constexpr (DUAL) load_w_block(W2, n, half, w2);
  float bias = b1 ? b1[n] : 0.0f;
  if (DUAL && b2) bias += b2[n];
  const float al = alpha ? alpha[0] : 1.0f;
  const bool drop = p_drop > 0.0f;
  const uint32_t thresh = (uint32_t)(p_drop * 65536.0f + 0.5f);
  const float scale = drop ? 1.0f / (1.0f - p_drop) : 1.0f;
  const uint32_t col_term = (uint32_t)(n >> 1) * 0x85EBCA77u ^ seed_lo;

  split_store(planes[0], tid, tr);
  __syncthreads();
  int buf = 0;
  constexpr int NS = DUAL ? 2 : 1;     // inputs that pass through the plane buffers per tile
  for (; tile < n_tiles; tile += gridDim.x) {
    const int64_t row0 = tile * TR;
    const int64_t next = tile + gridDim.x;
    floatx16 acc[2];
    zero(acc);
    // residual rows in accumulator layout, requested ahead of the multiplications that hide their latency
    float res[HAS_R ? 2 : 1][HAS_R ? 16 : 1];
    if constexpr (HAS_R) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = std::min<int64_t>(row0 + 32 * rt + crow(r, half), n_rows - 1);   // (past the end: unused)
          res[HAS_R ? rt : 0][HAS_R ? r : 0] = R[row * DW + n];
        }
    }
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      // the rows that follow travel while these are multiplied: this tile's second input, or the next tile's first
      const bool more = (st + 1 < NS) || next < n_tiles;
      if (more) {
        if (DUAL && st == 0) load_tile(X2, row0, n_rows, tid, tr);
        else load_tile(X1, next * TR, n_rows, tid, tr);
      }
      mma_planes(planes[buf], i, half, (DUAL && st == 1) ? w2 : w1, acc);
      if (st == NS - 1) {
        // Epilogue in accumulator layout: lane = column n, registers = rows.  (Written with the
        // per-element row test and plain addressing on purpose: a straight-line variant per dropout /
        // full-tile case, or hoisted store offsets, cost the plain layer 20 registers more, which it
        // spills at two workgroups per CU: 390 -> 570 us on 1.5 M rows.)
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
              if constexpr (HAS_R) z += res[HAS_R ? rt : 0][HAS_R ? r + e : 0];
              if (row + e < n_rows) Y[(row + e) * DW + n] = z;
            }
          }
      }
      if constexpr (NBUF == 2) {
        // the other buffer was last read one stage ago, and every wavefront has passed the barrier since
        if (more) split_store(planes[buf ^ 1], tid, tr);
        __syncthreads();
        buf ^= 1;
      } else {
        __syncthreads();                      // everybody has read the planes
        if (more) split_store(planes[0], tid, tr);
        __syncthreads();
      }
    }
  }
}

// ---- backward
// Per tile of 64 rows, everything on the bf16 matrix pipe (three-way splits as in the forward):
//   Z  = X W^T + b        A = X rows        (row-major planes RX), B = W[n-block w]      (registers)
//   dX = dZ W             A = dZ rows       (row-major planes RZ), B = W^T[k-block w]    (registers)
//   dW += dZ^T X          A = dZ[:, n-block w] as it stands in the accumulator layout (registers),
//                         B = X columns (column-major planes XT), all four k tiles
// X is split once, by the thread that loaded it, and stored twice: row-major (pairs packed along
// k) and column-major (pairs packed along the rows).  dZ is split once, by the lane that computed
// it (lane = column n, registers = rows), feeds the dW MFMAs from there and is scattered, element by
// element, into the row-major planes the dX MFMAs read.  G is read from memory in the accumulator
// layout (coalesced 128-byte segments), no staging.
//
// The rows of a tile inside XT stand in the order the dW operands want them: lane (n, half) holds
// dZ of rows (r&3) + 8*(r>>2) + 4*half, so reduction step (rb, s2) of the lane's half runs over rows
// 32rb + 16s2 + 4half + {0..3} and + 8 + {0..3}; XT stores row 16G + 8a + 4h + b at position
// 16G + 8h + 4a + b (bits 2 and 3 of the row index swapped), which makes those eight rows one
// aligned 16-byte read.  A loading thread takes exactly such a group of eight rows (and four columns).
constexpr int XS = TR + 8;                  // padded XT row (bf16 elements): 144 bytes, conflict-free operand reads
constexpr int XT_PLANE = DW * XS;

__device__ __forceinline__ int xt_pos(int row) { return (row & ~12) | ((row & 4) << 1) | ((row & 8) >> 1); }

// rows of the thread's group g (0..7): 16*(g>>1) + 4*(g&1) + {0,1,2,3} and the same + 8
__device__ __forceinline__ int group_row(int g, int t) { return 16 * (g >> 1) + 4 * (g & 1) + 8 * (t >> 2) + (t & 3); }

__device__ __forceinline__ void load_tile_groups(const float *__restrict__ M, int64_t row0, int64_t n_rows, int tid,
                                                 TileRegs &t) {
  const int g = tid >> 5, c = tid & 31;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int64_t row = row0 + group_row(g, j);
    row = row < n_rows ? row : n_rows - 1;
    t.v[j] = *reinterpret_cast<const floatx4 *>(M + row * DW + 4 * c);
  }
}

// split the thread's 8 rows x 4 columns once (pairs along the rows) and store both layouts
template <bool WITH_RX, int RXP>
__device__ __forceinline__ void split_store_both(__bf16 (*rx)[RXP], __bf16 (*xt)[XT_PLANE], int tid, const TileRegs &t) {
  const int g = tid >> 5, c = tid & 31;
  // part p, column q, row pair u (rows 2u, 2u+1 of the group): one dword
  uint32_t pk[3][4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int u = 0; u < 4; u += 2) {
      rk::u32x2 h, m, l;
      rk::split3_quad(t.v[2 * u][q], t.v[2 * u + 1][q], t.v[2 * u + 2][q], t.v[2 * u + 3][q], h, m, l);
      pk[0][q][u] = h[0], pk[0][q][u + 1] = h[1];
      pk[1][q][u] = m[0], pk[1][q][u + 1] = m[1];
      pk[2][q][u] = l[0], pk[2][q][u + 1] = l[1];
    }
  const int p0 = xt_pos(group_row(g, 0));         // positions p0 .. p0 + 7 hold the group's rows in order
#pragma unroll
  for (int p = 0; p < 3; ++p) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      rk::u32x4 v = {pk[p][q][0], pk[p][q][1], pk[p][q][2], pk[p][q][3]};
      *reinterpret_cast<rk::u32x4 *>(&xt[p][(4 * c + q) * XS + p0]) = v;
    }
    if constexpr (WITH_RX) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        // rows 2u and 2u+1 of the group, columns 4c .. 4c+3, pairs along k
        rk::u32x2 lo, hi;
        lo[0] = (pk[p][0][u] & 0xFFFFu) | (pk[p][1][u] << 16);
        lo[1] = (pk[p][2][u] & 0xFFFFu) | (pk[p][3][u] << 16);
        hi[0] = (pk[p][0][u] >> 16) | (pk[p][1][u] & 0xFFFF0000u);
        hi[1] = (pk[p][2][u] >> 16) | (pk[p][3][u] & 0xFFFF0000u);
        *reinterpret_cast<rk::u32x2 *>(&rx[p][group_row(g, 2 * u) * LDB + 4 * c]) = lo;
        *reinterpret_cast<rk::u32x2 *>(&rx[p][group_row(g, 2 * u + 1) * LDB + 4 * c]) = hi;
      }
    }
  }
}

// the lane's block of W^T as B operand of dX = dZ W: element 8s + j is W[16s + 8*half + j][k]
__device__ __forceinline__ void load_wt_block(const float *__restrict__ W, int k, int half, SplitN<64> &w) {
  float wcol[64];
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) wcol[8 * s + j] = W[(16 * s + 8 * half + j) * DW + k];
  rk::split3(wcol, w);
}

template <bool ACT>
__global__ __launch_bounds__(256, 1) void dense128_bwd_kernel(
    const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ b,
    const float *__restrict__ alpha, float p_drop, uint64_t seed, const uint64_t *__restrict__ seed_step,
    const float *__restrict__ G, int64_t n_rows, float *__restrict__ dX, float *__restrict__ part) {
  constexpr int RXP = ACT ? PLANE : 8;
  __shared__ __attribute__((aligned(16))) __bf16 rx[3][RXP];                          // X rows (Z recompute only)
  __shared__ __attribute__((aligned(16))) __bf16 xt[3][XT_PLANE];                     // X columns
  __shared__ __attribute__((aligned(16))) __bf16 rz[3][PLANE];                        // dZ rows
  __shared__ float red_b[4][2];
  const int tid = threadIdx.x, wave = tid / kWave, lane = tid % kWave;
  const int i = lane & 31, half = lane >> 5;
  const int n = 32 * wave + i;
  const int64_t n_tiles = (n_rows + TR - 1) / TR;
  const uint64_t sd = seed + (seed_step ? seed_step[0] * 0x9E3779B97F4A7C15ull : 0ull);
  const uint32_t seed_lo = (uint32_t)sd, seed_hi = (uint32_t)(sd >> 32);

  int64_t tile = blockIdx.x;
  TileRegs tx;
  if (tile < n_tiles) load_tile_groups(X, tile * TR, n_rows, tid, tx);      // in flight while the weights are split

  SplitN<64> wz, wx;
  if constexpr (ACT) load_w_block(W, n, half, wz);
  if (dX) load_wt_block(W, n, half, wx);
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

  for (; tile < n_tiles; tile += gridDim.x) {
    const int64_t row0 = tile * TR;
    // upstream gradient in accumulator layout, straight from memory (rows past the end: no gradient);
    // uniform tile base + a 32-bit lane offset + a constant per register
    const bool full = row0 + TR <= n_rows;       // uniform; only the last tile can be partial
    const float *__restrict__ gt = G + (row0 + 4 * half) * DW + n;   // per-tile lane pointer + constants
    float dz[2][16];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (full) {
          dz[rb][r] = gt[(32 * rb + (r & 3) + 8 * (r >> 2)) * DW];
        } else {
          const int64_t row = std::min<int64_t>(row0 + 32 * rb + crow(r, half), n_rows - 1);
          dz[rb][r] = G[row * DW + n];
        }
      }
    split_store_both<ACT, RXP>(rx, xt, tid, tx);
    __syncthreads();                                   // #1: the X planes are complete
    if (tile + gridDim.x < n_tiles) load_tile_groups(X, (tile + gridDim.x) * TR, n_rows, tid, tx);

    floatx16 z[2];
    if constexpr (ACT) {
      zero(z);
      mma_planes(reinterpret_cast<const __bf16 (*)[PLANE]>(&rx[0][0]), i, half, wz, z);
    }
    if (!full) {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (row0 + 32 * rb + crow(r, half) >= n_rows) dz[rb][r] = 0.0f;
    }
    // dZ = dropout mask * scale * G through the PReLU derivative at the recomputed Z (compiled with and
    // without the mask, selected by one wave-uniform branch per tile)
    auto make_dz = [&](auto drop_tag) {
      constexpr bool kDrop = decltype(drop_tag)::value;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const int64_t row = row0 + 32 * rb + crow(r, half);      // even; row + 1 shares the hash
          DropQuad dq{0u, 0u};
          if constexpr (kDrop) dq = drop_quad(seed_hi, (uint32_t)(row >> 1), col_term);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            float g = dz[rb][r + e];
            if constexpr (kDrop) g = kept(e ? dq.hi : dq.lo, n & 1, thresh) ? g * scale : 0.0f;
            if constexpr (ACT) {
              const float zz = z[rb][r + e] + bias;
              const bool pos = zz > 0.0f;
              da_acc += pos ? 0.0f : zz * g;
              g = pos ? g : al * g;
            }
            db_acc += g;
            dz[rb][r + e] = g;
          }
        }
    };
    if (drop) make_dz(std::true_type{});
    else make_dz(std::false_type{});
    // split once: the dW operands stay in registers, the elements go to the row-major planes
    SplitN<16> dzs[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      rk::split3(dz[rb], dzs[rb]);
      if (dX) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int off = (32 * rb + crow(8 * s2 + j, half)) * LDB + n;
            rz[0][off] = dzs[rb].h[s2][j];
            rz[1][off] = dzs[rb].m[s2][j];
            rz[2][off] = dzs[rb].l[s2][j];
          }
      }
    }
    __syncthreads();                                   // #2: the dZ planes are complete
    if (dX) {
      floatx16 dx[2];
      zero(dx);
      mma_planes(rz, i, half, wx, dx);
      if (full) {
        float *__restrict__ xt_out = dX + (row0 + 4 * half) * DW + n;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int r = 0; r < 16; ++r) xt_out[(32 * rb + (r & 3) + 8 * (r >> 2)) * DW] = dx[rb][r];
      } else {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + 32 * rb + crow(r, half);
            if (row < n_rows) dX[row * DW + n] = dx[rb][r];
          }
      }
    }
    // dW[n][k] += sum over the tile's rows of dZ[row][n] X[row][k]: four independent chains (k tiles)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 xh[4], xm[4], xl[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          const int off = (32 * kt + i) * XS + 32 * rb + 16 * s2 + 8 * half;
          xh[kt] = *reinterpret_cast<const bf16x8 *>(&xt[0][off]);
          xm[kt] = *reinterpret_cast<const bf16x8 *>(&xt[1][off]);
          xl[kt] = *reinterpret_cast<const bf16x8 *>(&xt[2][off]);
        }
        const bf16x8 ah = dzs[rb].h[s2], am = dzs[rb].m[s2], al_ = dzs[rb].l[s2];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) dw[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al_, xh[kt], dw[kt], 0, 0, 0);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) dw[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xl[kt], dw[kt], 0, 0, 0);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) dw[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, xm[kt], dw[kt], 0, 0, 0);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) dw[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, xh[kt], dw[kt], 0, 0, 0);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) dw[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xm[kt], dw[kt], 0, 0, 0);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) dw[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh[kt], dw[kt], 0, 0, 0);
      }
    __syncthreads();                                   // #3: the planes are free for the next tile
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

// The same reduction for several layers in one launch (blockIdx.y = job), as dense.hip's
// dense_reduce_jobs_kernel: a job that lists several slabs adds their sums in list order, with the
// operations the accumulate form of the kernel above would have performed call by call.
constexpr int kJobsPerLaunchW = 16;
struct ReduceJobsW {
  rk_dense_reduce_job job[kJobsPerLaunchW];
};
__global__ __launch_bounds__(1024) void dense128_reduce_jobs_kernel(ReduceJobsW jobs) {
  // 64 consecutive elements x 16 strides over the partials: a wavefront reads 256 contiguous bytes of a partial
  // (the sums and their order per element are those of the 256-thread kernel above)
  __shared__ float red[16][65];
  const rk_dense_reduce_job &jb = jobs.job[blockIdx.y];
  const int el = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + el;
  float total = 0.0f;
  for (int32_t sl = 0; sl < jb.n_slabs; ++sl) {      // uniform over the workgroup
    const float *__restrict__ slab = jb.slabs[sl];
    float s0 = 0.0f, s1 = 0.0f;
    if (idx < NREDW) {
      int32_t p = q;
#pragma unroll 4      // eight loads in flight; the additions keep their order
      for (; p + 16 < jb.n_part; p += 32) {
        s0 += slab[(int64_t)p * PARTW + idx];
        s1 += slab[(int64_t)(p + 16) * PARTW + idx];
      }
      if (p < jb.n_part) s0 += slab[(int64_t)p * PARTW + idx];
    }
    __syncthreads();                                 // the previous slab's fold has been read
    red[q][el] = s0 + s1;
    __syncthreads();
    if (q == 0) {
      float s = 0.0f;
#pragma unroll
      for (int k = 0; k < 16; ++k) s += red[k][el];
      total = sl == 0 ? s : total + s;
    }
  }
  if (q != 0 || idx >= NREDW) return;
  if (idx < DW * DW)
    jb.dW[idx] = total;
  else if (idx < DW * DW + DW) {
    if (jb.db) jb.db[idx - DW * DW] = total;
  } else if (jb.dalpha)
    jb.dalpha[0] = total;
}

constexpr int kMaxGrid = 512;
inline int grid_for(int64_t n_rows, int wg_per_cu = 1) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  return (int)std::min<int64_t>(rk::ceil_div(n_rows, TR), std::min(cus * wg_per_cu, kMaxGrid * wg_per_cu));
}

}  // namespace

namespace rk {

int dense128_fwd(const float *X1, const float *W1, const float *X2, const float *W2, const float *b1,
                 const float *b2, const float *alpha, float p_drop, uint64_t seed, const uint64_t *seed_step,
                 const float *residual, int64_t n_rows, float *Y, hipStream_t st) {
  const int grid = grid_for(n_rows, (X2 || residual) ? 1 : 2);
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
size_t dense128_slab_bytes() { return (size_t)kMaxGrid * PARTW * sizeof(float); }

int dense128_bwd_partials(const float *X, const float *W, const float *b, const float *alpha, float p_drop, uint64_t seed,
                          const uint64_t *seed_step, const float *G, int64_t n_rows, float *dX, float *slab,
                          int32_t *n_part_out, hipStream_t st) {
  const int grid = grid_for(n_rows);
  if (alpha)
    dense128_bwd_kernel<true><<<grid, 256, 0, st>>>(X, W, b, alpha, p_drop, seed, seed_step, G, n_rows, dX, slab);
  else
    dense128_bwd_kernel<false><<<grid, 256, 0, st>>>(X, W, b, nullptr, p_drop, seed, seed_step, G, n_rows, dX, slab);
  RK_CHECK_LAUNCH("dense128_bwd_kernel");
  *n_part_out = grid;
  return RK_OK;
}

int dense128_reduce_jobs(const rk_dense_reduce_job *jobs, int32_t n_jobs, hipStream_t st) {
  for (int32_t j0 = 0; j0 < n_jobs; j0 += kJobsPerLaunchW) {
    const int32_t nj = std::min<int32_t>(kJobsPerLaunchW, n_jobs - j0);
    ReduceJobsW args;
    memset(&args, 0, sizeof args);
    for (int32_t j = 0; j < nj; ++j) args.job[j] = jobs[j0 + j];
    dense128_reduce_jobs_kernel<<<dim3((NREDW + 63) / 64, nj), 1024, 0, st>>>(args);
    RK_CHECK_LAUNCH("dense128_reduce_jobs_kernel");
  }
  return RK_OK;
}

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
