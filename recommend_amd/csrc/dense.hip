// Fused node-wise layer  Y = dropout(prelu(X1 W1^T [+ X2 W2^T] + b1 [+ b2])) [+ R]
// and its backward, for the 64-wide embeddings of the propagation hops
// (gfx950; fp32 results from v_mfma_f32_32x32x16_bf16 on three-way bf16 splits).
//
// These are the D x D layers wrapped around every SpMM hop and inside Combine
// (reference: graph_constructor.py:204, Combin_Layer.py:26-30, out_Layer.py:36,
// Main_Model.py:162-163).  N is the node count, so each layer is a skinny GEMM
// that streams an N x 64 matrix once.  Forward reads X (and R), writes Y.
// Backward reads X and G once, RECOMPUTES Z = X W^T + b on the matrix cores
// instead of saving it, regenerates the dropout mask from the same counter hash,
// and produces dX, dW, db and d(alpha) in the same pass; per-workgroup partial
// sums of dW/db/d(alpha) are reduced in workgroup order by a second launch (no
// float atomics, bitwise reproducible).
//
// Both kernels are persistent and every wavefront works alone on tiles of 32 rows
// (no workgroup barrier in the loop): it stages its rows (X, and G in the backward)
// by LDS-DMA into private, XOR-swizzled LDS tiles one or two tiles ahead, keeps the
// weight fragments in registers (or LDS) for the whole launch, and sends results
// through the consumed tile in LDS so that they leave as row-wise 16-byte stores.
//
// MFMA operand maps (32x32x16): lane l supplies A[i = l&31][k = 8*(l>>5) + j] and
// B[k = 8*(l>>5) + j][j' = l&31] for j < 8; the result register r of lane l is
// C[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].  The reduction index is walked so that a
// lane's operand values are contiguous in its row: A and B only have to agree on
// which index a lane's j-th value of a step stands for.
#include <string.h>

#include <type_traits>

#include "bf16x3.h"
#include "common.h"
#include "dense_wide.h"
#include "slab_reduce.h"

namespace {

using rk::kWave;
using rk::bf16x8;
using rk::floatx16;
using rk::floatx4;
using rk::mfma_split3;
using rk::split3;
using rk::Split3;
using rk::SplitN;

constexpr int DD = 64;        // layer width (in = out)
constexpr int WT = 32;        // rows per wavefront tile
constexpr int LDZ = DD + 4;   // padded row of the dZ tile: conflict-free b128 row reads

__device__ __forceinline__ int crow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// Counter-based dropout, identical in forward and backward.  One hash covers the 2 x 2 block
// (rows 2i, 2i+1) x (columns 2j, 2j+1): 64 bits = four 16-bit fields, field (row&1, col&1).
// An element is kept when its field is >= thresh16 = round(p * 65536).
struct DropQuad {
  uint32_t lo, hi;   // fields of the even row, of the odd row
};
__device__ __forceinline__ DropQuad drop_quad(uint32_t seed_lo, uint32_t seed_hi, uint32_t row_pair,
                                              uint32_t col_term) {
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

// --------------------------------------------------------------- tile staging
// One wave-instruction of LDS-DMA writes 64 x 16 B = 1 KiB of LDS linearly from per-lane global
// addresses: piece p of a tile holds rows 4p..4p+3, and slot s of row r holds the 16-byte chunk
// s ^ (r & 15) of that row, so that both the row-per-lane b128 reads and the column-per-lane b32
// reads below are (nearly) conflict free without padding.
__device__ __forceinline__ uint32_t dma_lane_off(int lane) {
  const int l_row = lane >> 4, l_slot = lane & 15;
  return (uint32_t)(l_row * 256 + ((l_slot ^ l_row) * 16));
}

__device__ __forceinline__ void stage_rows(const float *__restrict__ M, int64_t row0, int64_t n_rows,
                                           float *dst, uint32_t lane_off) {
  if (row0 + WT <= n_rows) {
    const char *base = reinterpret_cast<const char *>(M + row0 * DD);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const uint32_t off = (lane_off ^ (uint32_t)(((4 * p) & 15) * 16)) + (uint32_t)(p * 1024);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + off),
                                       (__attribute__((address_space(3))) void *)(dst + p * 256), 16, 0, 0);
    }
    return;
  }
  // last, partial tile: rows past the end re-read the last row (their results are never stored,
  // and they enter the weight gradient with a zero factor)
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const uint32_t off = lane_off ^ (uint32_t)(((4 * p) & 15) * 16);
    int64_t row = row0 + 4 * p + (int64_t)(off >> 8);
    row = row < n_rows ? row : n_rows - 1;
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(M + row * DD) + (off & 255)),
        (__attribute__((address_space(3))) void *)(dst + p * 256), 16, 0, 0);
  }
}

// The same staging with the DMA instructions written as inline assembly, for the backward kernel.
// The compiler's wait-count pass tracks LDS-DMA per LDS array; an access to an array that never
// receives a DMA (the dZ tile, the weight parts) it cannot place, and it then waits for every
// DMA in flight (vmcnt(0)), which drains the prefetch of the next tile in the middle of this one.
// Written this way the pass knows nothing of the DMAs, and the kernel's own s_waitcnt in front of
// the first read of each tile is the synchronisation.  `row0` and `dst` are wave-uniform.
// Rules relied on: one wait state between the scalar write of M0 and the DMA that reads it (the
// s_nop 0 inside the statement); completion by vmcnt in issue order (do_tile's vmcnt(0/8/16) through
// the s_waitcnt builtin); the LDS tile a DMA overwrites has been read back completely (the
// lgkmcnt(0) in front of every re-staging).  No register destination, so nothing can be moved
// ahead of those waits by the register allocator.
__device__ __forceinline__ void stage_rows_raw(const float *__restrict__ M, int64_t row0, int64_t n_rows,
                                               float *dst, uint32_t lane_off) {
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)dst);
  if (row0 + WT <= n_rows) {
    const char *base = reinterpret_cast<const char *>(M + row0 * DD);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const uint32_t off = (lane_off ^ (uint32_t)(((4 * p) & 15) * 16)) + (uint32_t)(p * 1024);
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                   : : "v"(off), "s"(base), "s"(lds0 + (uint32_t)(p * 1024)) : "memory", "m0");
    }
    return;
  }
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const uint32_t off = lane_off ^ (uint32_t)(((4 * p) & 15) * 16);
    int64_t row = row0 + 4 * p + (int64_t)(off >> 8);
    row = row < n_rows ? row : n_rows - 1;
    const char *addr = reinterpret_cast<const char *>(M + row * DD) + (off & 255);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 : : "v"(addr), "s"(lds0 + (uint32_t)(p * 1024)) : "memory", "m0");
  }
}

// row `col` of a staged tile, columns 32*half .. 32*half+31: chunk 8*half+q sits in slot
// (8*half+q) ^ (col & 15); q < 8 only touches bits the other terms leave alone
__device__ __forceinline__ uint32_t frag_lane_off(int col, int half) {
  return (uint32_t)(col * 256 + (((8 * half) ^ (col & 15)) * 16));
}
__device__ __forceinline__ void load_row_frag(const float *tile, uint32_t frag_off, float (&f)[32]) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float4 v = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(tile) +
                                                       (frag_off ^ (uint32_t)(q * 16)));
    f[4 * q + 0] = v.x;
    f[4 * q + 1] = v.y;
    f[4 * q + 2] = v.z;
    f[4 * q + 3] = v.w;
  }
}

// 32 weight values of one lane, from global memory (once per launch): row `row` of W,
// columns 32*half .. 32*half+31
__device__ __forceinline__ void load_w_row(const float *__restrict__ W, int row, int half, float (&f)[32]) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float4 v = *reinterpret_cast<const float4 *>(W + row * DD + 32 * half + 4 * q);
    f[4 * q + 0] = v.x;
    f[4 * q + 1] = v.y;
    f[4 * q + 2] = v.z;
    f[4 * q + 3] = v.w;
  }
}

// ------------------------------------------------------------------ forward
// (the plain layer is held to 256 registers per lane: two of its workgroups share a CU)
template <bool DUAL, bool HAS_R>
__global__ __launch_bounds__(256, (DUAL || HAS_R) ? 1 : 2) void dense_fwd_kernel(
    const float *__restrict__ X1, const float *__restrict__ W1, const float *__restrict__ X2,
    const float *__restrict__ W2, const float *__restrict__ b1, const float *__restrict__ b2,
    const float *__restrict__ alpha, float p_drop, uint64_t seed, const uint64_t *__restrict__ seed_step,
    const float *__restrict__ R, int64_t n_rows, float *__restrict__ Y) {
  // Two tiles per wavefront and operand, so that TWO tiles are in flight while one is multiplied:
  // with one (8 KB per wavefront, 32 KB per CU) the kernel waited for memory a third of the time.
  // Separate arrays: a DMA into one is known not to alias reads of the other.
  __shared__ __attribute__((aligned(1024))) float xa0[4][WT * DD];
  __shared__ __attribute__((aligned(1024))) float xa1[4][WT * DD];
  __shared__ __attribute__((aligned(1024))) float xb0[DUAL ? 4 : 1][DUAL ? WT * DD : 4];
  __shared__ __attribute__((aligned(1024))) float xb1[DUAL ? 4 : 1][DUAL ? WT * DD : 4];
  const int tid = threadIdx.x, wave = tid / kWave, lane = tid % kWave;
  const int col = lane & 31, half = lane >> 5;
  const int64_t n_tiles = (n_rows + WT - 1) / WT;
  // the per-call seed, advanced by a step counter in device memory when one is given (a launch
  // replayed from a captured graph then still draws a fresh mask every step)
  const uint64_t sd = seed + (seed_step ? seed_step[0] * 0x9E3779B97F4A7C15ull : 0ull);
  const uint32_t seed_lo = (uint32_t)sd, seed_hi = (uint32_t)(sd >> 32);
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  if (tile >= n_tiles) return;   // wave-uniform; there is no workgroup barrier below

  uint32_t lane_off = dma_lane_off(lane);
  uint32_t frag_off = frag_lane_off(col, half);
  stage_rows(X1, tile * WT, n_rows, xa0[wave], lane_off);
  if constexpr (DUAL) stage_rows(X2, tile * WT, n_rows, xb0[DUAL ? wave : 0], lane_off);
  if (tile + stride < n_tiles) {
    stage_rows(X1, (tile + stride) * WT, n_rows, xa1[wave], lane_off);
    if constexpr (DUAL) stage_rows(X2, (tile + stride) * WT, n_rows, xb1[DUAL ? wave : 0], lane_off);
  }

  // B operand: W[n = 32t + col][k = 32*half + m], kept in registers for the whole launch
  // (the MFMA sums over (half, j) for each step s: A and B only have to agree on which k a lane's
  // j-th value is, and both hold k = 32*half + 8s + j)
  Split3 w1[2], w2[DUAL ? 2 : 1];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float wrow[32];
    load_w_row(W1, 32 * t + col, half, wrow);
    split3(wrow, w1[t]);
    if constexpr (DUAL) {
      load_w_row(W2, 32 * t + col, half, wrow);
      split3(wrow, w2[DUAL ? t : 0]);
    }
  }
  float bias[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    bias[t] = b1 ? b1[32 * t + col] : 0.0f;
    if (DUAL && b2) bias[t] += b2[32 * t + col];
  }
  const float al = alpha ? alpha[0] : 1.0f;   // no activation = slope 1
  const bool drop = p_drop > 0.0f;
  const uint32_t thresh = (uint32_t)(p_drop * 65536.0f + 0.5f);
  const float scale = drop ? 1.0f / (1.0f - p_drop) : 1.0f;
  uint32_t col_term[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) col_term[t] = (uint32_t)((32 * t + col) >> 1) * 0x85EBCA77u ^ seed_lo;

  // one tile out of buffers (xa, xb), which then receive the tile after next
  auto process = [&](float *xa, float *xb) {
    const int64_t row0 = tile * WT;
    const int64_t next = tile + 2 * stride;
    asm volatile("" : "+v"(lane_off), "+v"(frag_off));   // keep per-piece addresses out of registers

    floatx16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // residual rows in accumulator layout, issued before the MFMAs that hide their latency
    const bool full = row0 + WT <= n_rows;   // wave-uniform: only the last tile can be partial
    float res[HAS_R ? 2 : 1][HAS_R ? 16 : 1];
    if constexpr (HAS_R) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = row0 + crow(r, half);
          res[HAS_R ? t : 0][HAS_R ? r : 0] = (full || row < n_rows) ? R[row * DD + 32 * t + col] : 0.0f;
        }
    }

    // This tile's DMA (8 pieces per operand, issued a whole tile earlier) must have landed.  Newer
    // in the in-order memory counter are the 8 stores of the tile in between -- always a full
    // tile -- and, unless the input has ended, the DMAs of the tile after next: "all but the
    // newest 8 done" covers both cases and leaves the prefetch in flight.
    __builtin_amdgcn_s_waitcnt(0x0F78);   // vmcnt(8)
    float a[32];
    load_row_frag(xa, frag_off, a);
    // The fragment is in registers.  The buffer first turns the output tile (below) and then takes
    // the DMA of the tile after next, which has two tiles' MFMAs and epilogues to land.
    Split3 as;
    split3(a, as);
    rk::mfma_split3_shared_a(as, w1[0], w1[1], acc[0], acc[1]);   // the two column blocks' chains, alternately
    if constexpr (DUAL) {
      load_row_frag(xb, frag_off, a);
      __builtin_amdgcn_s_waitcnt(0xC07F);
      if (next < n_tiles) stage_rows(X2, next * WT, n_rows, xb, lane_off);
      split3(a, as);
      rk::mfma_split3_shared_a(as, w2[0], w2[DUAL ? 1 : 0], acc[0], acc[1]);
    }

    // Epilogue in accumulator layout: lane = column 32t + col, registers = rows.  Compiled four
    // times (dropout on/off, full/partial tile) and selected by two wave-uniform branches per
    // tile, so that the per-element code is straight-line; no activation is alpha = 1.
    // The results go through the consumed X tile in LDS (plain row-major) and leave as 8 row-wise
    // 16-byte stores per lane instead of 64 dword stores: the memory counter is shared by loads,
    // DMAs and stores and holds 63 operations, so with 64 stores per tile every wait for a tile
    // was a wait for all stores before it as well.
    auto epilogue = [&](auto drop_tag, auto full_tag) {
      constexpr bool kDrop = decltype(drop_tag)::value, kFull = decltype(full_tag)::value;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int n = 32 * t + col;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          // rows crow(r) (even) and crow(r) + 1 share one dropout hash
          DropQuad dq{0u, 0u};
          if constexpr (kDrop)
            dq = drop_quad(seed_lo, seed_hi, (uint32_t)((row0 + crow(r, half)) >> 1), col_term[t]);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            float z = acc[t][r + e] + bias[t];
            z = z > 0.0f ? z : al * z;
            if constexpr (kDrop) z = kept(e ? dq.hi : dq.lo, n & 1, thresh) ? z * scale : 0.0f;
            if constexpr (HAS_R) z += res[HAS_R ? t : 0][HAS_R ? r + e : 0];
            xa[crow(r + e, half) * DD + n] = z;
          }
        }
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the tile is in LDS
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int row = 4 * p + (lane >> 4);
        const float4 v = *reinterpret_cast<const float4 *>(xa + row * DD + 4 * (lane & 15));
        if (kFull || row0 + row < n_rows) *reinterpret_cast<float4 *>(Y + (row0 + row) * DD + 4 * (lane & 15)) = v;
      }
    };
    if (drop) {
      if (full) epilogue(std::true_type{}, std::true_type{});
      else epilogue(std::true_type{}, std::false_type{});
    } else {
      if (full) epilogue(std::false_type{}, std::true_type{});
      else epilogue(std::false_type{}, std::false_type{});
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the output tile has been read back
    if (next < n_tiles) stage_rows(X1, next * WT, n_rows, xa, lane_off);
  };
  for (;;) {
    process(xa0[wave], xb0[DUAL ? wave : 0]);
    tile += stride;
    if (tile >= n_tiles) break;
    process(xa1[wave], xb1[DUAL ? wave : 0]);
    tile += stride;
    if (tile >= n_tiles) break;
  }
}

// ----------------------------------------------------------------- backward
constexpr int PART = DD * DD + DD + 8;        // per-workgroup partial: dW, db, d(alpha) (+pad)

// ACT: Y = dropout(prelu(X W^T + b)); otherwise Y = dropout(X W^T + b) (dZ = mask * G).
template <bool ACT>
__global__ __launch_bounds__(256, 1) void dense_bwd_kernel(
    const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ b,
    const float *__restrict__ alpha, float p_drop, uint64_t seed, const uint64_t *__restrict__ seed_step,
    const float *__restrict__ G, int64_t n_rows, float *__restrict__ dX, float *__restrict__ part) {
  __shared__ __attribute__((aligned(1024))) float x0[4][WT * DD];   // X tiles, double buffered:
  __shared__ __attribute__((aligned(1024))) float x1[4][WT * DD];   // the tile is read again for dW
  __shared__ __attribute__((aligned(1024))) float gt[4][WT * DD];   // G tile, loaded one tile ahead
  __shared__ __attribute__((aligned(16))) float zt[4][WT * LDZ];    // dZ tile (layout change for dX)
  __shared__ float red_b[4][DD + 1];
  __shared__ bf16x8 wz_lds[ACT ? 2 : 1][ACT ? 12 : 1][ACT ? kWave : 1];
  const int tid = threadIdx.x, lane = tid % kWave;
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);   // in scalar registers: tile addresses are uniform
  const int col = lane & 31, half = lane >> 5;
  const int64_t n_tiles = (n_rows + WT - 1) / WT;
  // the per-call seed, advanced by a step counter in device memory when one is given (a launch
  // replayed from a captured graph then still draws a fresh mask every step)
  const uint64_t sd = seed + (seed_step ? seed_step[0] * 0x9E3779B97F4A7C15ull : 0ull);
  const uint32_t seed_lo = (uint32_t)sd, seed_hi = (uint32_t)(sd >> 32);
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t tile = (int64_t)blockIdx.x * 4 + wave;

  uint32_t lane_off = dma_lane_off(lane);
  uint32_t frag_off = frag_lane_off(col, half);
  // column-per-lane read of a staged tile: X[crow(r, half)][32*kt + col]
  uint32_t xcol_off = (uint32_t)(half * 4 * 256 + (((col >> 2) ^ (4 * half)) * 16) + (col & 3) * 4);
  if (tile < n_tiles) {
    stage_rows_raw(X, tile * WT, n_rows, x0[wave], lane_off);
    stage_rows_raw(G, tile * WT, n_rows, gt[wave], lane_off);
  }

  // W[n = 32t + col][k = 32*half + m]: B operand of Z = X W^T
  // Its bf16 parts are the same for all four wavefronts and are needed once per tile: they wait
  // in LDS (vector v of lane l at [v][l]: conflict-free 16-byte reads), the registers go to the
  // operands below.
  if constexpr (ACT) {
    if (wave < 2) {
      float wrow[32];
      load_w_row(W, 32 * wave + col, half, wrow);
      Split3 ws;
      split3(wrow, ws);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        wz_lds[wave][v][lane] = ws.h[v];
        wz_lds[wave][4 + v][lane] = ws.m[v];
        wz_lds[wave][8 + v][lane] = ws.l[v];
      }
    }
    __syncthreads();
  }
  // W[n = 32*half + m][k = 32*kt + col]: B operand of dX = dZ W
  Split3 wx[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float wrow[32];
#pragma unroll
    for (int m = 0; m < 32; ++m) wrow[m] = dX ? W[(32 * half + m) * DD + 32 * t + col] : 0.0f;
    split3(wrow, wx[t]);
  }
  const float al = ACT ? alpha[0] : 1.0f;
  const bool drop = p_drop > 0.0f;
  const uint32_t thresh = (uint32_t)(p_drop * 65536.0f + 0.5f);
  const float scale = drop ? 1.0f / (1.0f - p_drop) : 1.0f;
  float bias[2];
  uint32_t col_term[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    bias[t] = (ACT && b) ? b[32 * t + col] : 0.0f;
    col_term[t] = (uint32_t)((32 * t + col) >> 1) * 0x85EBCA77u ^ seed_lo;
  }

  // dW[n = 32t + crow(r)][k = 32kt + col], accumulated over all tiles of this wavefront
  floatx16 dw[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dw[t][kt][r] = 0.0f;
  float db_acc[2] = {0.0f, 0.0f}, da_acc = 0.0f;

  // One tile: X in `cur`; the next tile's DMA goes to `nxt`.  All traffic of a tile is 8 + 8 DMA
  // instructions (X, G) and 8 row-wise 16-byte stores (dX): the memory counter is in order and
  // shared by all three, so "G of this tile has landed" is "all but the newest 16 are done" --
  // the stores of the tile before and the X tile just requested stay in flight.
  auto do_tile = [&](int64_t tl, float *cur, float *nxt, bool first) {
    const int64_t row0 = tl * WT;
    asm volatile("" : "+v"(lane_off), "+v"(frag_off), "+v"(xcol_off));
    const bool has_next = tl + stride < n_tiles;
    if (has_next) stage_rows_raw(X, (tl + stride) * WT, n_rows, nxt, lane_off);
    if (!has_next)
      __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0)
    else if (first || !dX)
      __builtin_amdgcn_s_waitcnt(0x0F78);        // vmcnt(8): no stores yet / at all
    else
      __builtin_amdgcn_s_waitcnt(0x4F70);        // vmcnt(16)
    asm volatile("" ::: "memory");

    // upstream gradient in accumulator layout (lane = column, registers = rows), read
    // column-per-lane from the staged tile: G[crow(r, half)][32t + col]
    const bool full = row0 + WT <= n_rows;   // wave-uniform: only the last tile can be partial
    float dz[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t off = (xcol_off ^ (uint32_t)((r & 3) * 16)) +
                             (uint32_t)(((r & 3) + 8 * (r >> 2)) * 256 + 8 * (t ^ ((r >> 2) & 1)) * 16);
        dz[t][r] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(gt[wave]) + off);
      }
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): G is in registers, its tile can be refilled
    asm volatile("" ::: "memory");
    if (has_next) stage_rows_raw(G, (tl + stride) * WT, n_rows, gt[wave], lane_off);
    if (!full) {
      // the staging of a partial tile repeats the last row; those rows have no gradient
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dz[t][r] = row0 + crow(r, half) < n_rows ? dz[t][r] : 0.0f;
    }

    // A tile whose upstream gradient is all zero contributes nothing to dW, db, d(alpha) and has
    // dX = 0: the loss touches a few hundred rows, and until the first hop of the backward pass
    // spreads them, the row-wise layers see exactly those rows.  (-0.0 counts as zero.)
    {
      uint32_t bits = 0;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) bits |= __float_as_uint(dz[t][r]);
      if (!__any((bits << 1) != 0u)) {
        if (dX) {
          float *dxp = dX + (row0 + (lane >> 4)) * DD + 4 * (lane & 15);
#pragma unroll
          for (int p = 0; p < 8; ++p)
            if (full || row0 + 4 * p + (lane >> 4) < n_rows)
              *reinterpret_cast<float4 *>(dxp + 4 * p * DD) = float4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        return;
      }
    }

    floatx16 z[2];
    if constexpr (ACT) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) z[t][r] = 0.0f;
      float a[32];
      load_row_frag(cur, frag_off, a);
      Split3 as;
      split3(a, as);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        Split3 wz;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          wz.h[v] = wz_lds[ACT ? t : 0][ACT ? v : 0][ACT ? lane : 0];
          wz.m[v] = wz_lds[ACT ? t : 0][ACT ? 4 + v : 0][ACT ? lane : 0];
          wz.l[v] = wz_lds[ACT ? t : 0][ACT ? 8 + v : 0][ACT ? lane : 0];
        }
        mfma_split3(as, wz, z[t]);
      }
    }
    // dZ = dropout mask * scale * G, through the PReLU derivative at the recomputed Z (compiled
    // with and without the mask, selected by one wave-uniform branch per tile)
    auto make_dz = [&](auto drop_tag) {
      constexpr bool kDrop = decltype(drop_tag)::value;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int n = 32 * t + col;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          DropQuad dq{0u, 0u};
          if constexpr (kDrop)
            dq = drop_quad(seed_lo, seed_hi, (uint32_t)((row0 + crow(r, half)) >> 1), col_term[t]);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            float gs = dz[t][r + e];
            if constexpr (kDrop) gs = kept(e ? dq.hi : dq.lo, n & 1, thresh) ? gs * scale : 0.0f;
            if constexpr (ACT) {
              const float zz = z[t][r + e] + bias[t];
              const bool pos = zz > 0.0f;
              da_acc += pos ? 0.0f : zz * gs;       // torch prelu backward: z > 0 ? g : alpha * g
              gs = pos ? gs : al * gs;
            }
            db_acc[t] += gs;
            dz[t][r + e] = gs;
            zt[wave][crow(r + e, half) * LDZ + n] = gs;
          }
        }
      }
    };
    if (drop) make_dz(std::true_type{});
    else make_dz(std::false_type{});

    // dW[n][k] += sum_row dZ[row][n] X[row][k]: A = dZ as it stands (i = n on the lanes, the
    // reduction index walks the rows crow(r, half)), B = X read column-per-lane from the tile
    SplitN<16> dzs[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) split3(dz[t], dzs[t]);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      float xc[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t off = (xcol_off ^ (uint32_t)((r & 3) * 16)) +
                             (uint32_t)(((r & 3) + 8 * (r >> 2)) * 256 + 8 * (kt ^ ((r >> 2) & 1)) * 16);
        xc[r] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(cur) + off);
      }
      SplitN<16> xs;
      split3(xc, xs);
#pragma unroll
      for (int t = 0; t < 2; ++t) mfma_split3(dzs[t], xs, dw[t][kt]);
    }

    if (dX) {
      // dX[row][k] = sum_n dZ[row][n] W[n][k]: A = dZ rows (row-per-lane from the dZ tile)
      float a[32];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 v = *reinterpret_cast<const float4 *>(&zt[wave][col * LDZ + 32 * half + 4 * q]);
        a[4 * q + 0] = v.x;
        a[4 * q + 1] = v.y;
        a[4 * q + 2] = v.z;
        a[4 * q + 3] = v.w;
      }
      Split3 as;
      split3(a, as);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        floatx16 dx;
#pragma unroll
        for (int r = 0; r < 16; ++r) dx[r] = 0.0f;
        mfma_split3(as, wx[kt], dx);
        // the X tile has been consumed: it turns the output tile around (plain row-major)
#pragma unroll
        for (int r = 0; r < 16; ++r) cur[crow(r, half) * DD + 32 * kt + col] = dx[r];
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
      asm volatile("" ::: "memory");
      const float *src = cur + (lane >> 4) * DD + 4 * (lane & 15);
      float *dxp = dX + (row0 + (lane >> 4)) * DD + 4 * (lane & 15);
      if (full) {
        floatx4 v[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) v[p] = *reinterpret_cast<const floatx4 *>(src + 4 * p * DD);
#pragma unroll
        for (int p = 0; p < 8; ++p) *reinterpret_cast<floatx4 *>(dxp + 4 * p * DD) = v[p];
      } else {
#pragma unroll
        for (int p = 0; p < 8; ++p)
          if (row0 + 4 * p + (lane >> 4) < n_rows)
            *reinterpret_cast<float4 *>(dxp + 4 * p * DD) = *reinterpret_cast<const float4 *>(src + 4 * p * DD);
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);   // the tile has been read back before its next DMA
      asm volatile("" ::: "memory");
    }
  };

  bool first = true;
  for (; tile < n_tiles; tile += 2 * stride) {
    do_tile(tile, x0[wave], x1[wave], first);
    first = false;
    if (tile + stride < n_tiles) do_tile(tile + stride, x1[wave], x0[wave], false);
  }

  // ---- per-workgroup partials: the four wavefronts' sums are added in wavefront order
  __syncthreads();                 // every wavefront is done with its tiles: x0 is free
  float *red = &x0[0][0];          // 64 x 64 floats
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int idx = (32 * t + crow(r, half)) * DD + 32 * kt + col;
            red[idx] = (w == 0 ? 0.0f : red[idx]) + dw[t][kt][r];
          }
    }
    __syncthreads();
  }
  float *out = part + (int64_t)blockIdx.x * PART;
  for (int i = tid; i < DD * DD; i += 256) out[i] = red[i];
  // db: lanes l and l+32 hold different rows of the same column; d(alpha): every lane
#pragma unroll
  for (int t = 0; t < 2; ++t) db_acc[t] += __shfl_xor(db_acc[t], 32, kWave);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) da_acc += __shfl_xor(da_acc, off, kWave);
  if (half == 0) {
    red_b[wave][col] = db_acc[0];
    red_b[wave][32 + col] = db_acc[1];
  }
  if (lane == 0) red_b[wave][DD] = da_acc;
  __syncthreads();
  if (tid <= DD)
    out[DD * DD + tid] = ((red_b[0][tid] + red_b[1][tid]) + red_b[2][tid]) + red_b[3][tid];
}

// Sum of the per-workgroup partials in a fixed order, routed to dW / db / dalpha: 16 lanes per
// element stride the partials (two accumulators each) and are folded through LDS in lane order.
// One launch: the partials (<= 512 x 17 KB, just written) come out of L2, and a second launch
// would cost more than the whole reduction.
constexpr int NRED = DD * DD + DD + 1;
__global__ __launch_bounds__(256) void dense_reduce_kernel(const float *__restrict__ slab, int32_t n_part,
                                                           float *__restrict__ dW, float *__restrict__ db,
                                                           float *__restrict__ dalpha, int accumulate) {
  __shared__ float red[16][17];
  const int el = threadIdx.x & 15, q = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + el;
  float s0 = 0.0f, s1 = 0.0f;
  if (i < NRED) {
    int32_t p = q;
    for (; p + 16 < n_part; p += 32) {
      s0 += slab[(int64_t)p * PART + i];
      s1 += slab[(int64_t)(p + 16) * PART + i];
    }
    if (p < n_part) s0 += slab[(int64_t)p * PART + i];
  }
  red[q][el] = s0 + s1;
  __syncthreads();
  if (q != 0 || i >= NRED) return;
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += red[k][el];
  // accumulate: the sums are added to what the outputs hold (a weight shared by several layers)
  if (i < DD * DD)
    dW[i] = accumulate ? dW[i] + s : s;
  else if (i < DD * DD + DD) {
    if (db) db[i - DD * DD] = accumulate ? db[i - DD * DD] + s : s;
  } else if (dalpha)
    dalpha[0] = accumulate ? dalpha[0] + s : s;
}

// The same reduction for several layers in one launch (blockIdx.y = job): a backward pass with a
// dozen layer calls otherwise spends a launch of a few microseconds of work after each.  A job
// that lists several slabs (one weight applied several times) adds their sums in list order,
// with the operations the accumulate form of the kernel above would have performed call by call.
constexpr int kJobsPerLaunch = 16;
struct ReduceJobs {
  rk_dense_reduce_job job[kJobsPerLaunch];
};
__global__ __launch_bounds__(1024) void dense_reduce_jobs_kernel(ReduceJobs jobs) {
  // 64 consecutive elements x 16 strides over the partials: a wavefront reads 256 contiguous bytes of a partial
  // (the sums and their order per element are those of the 256-thread kernel above)
  __shared__ float red[16][65];
  const rk_dense_reduce_job &jb = jobs.job[blockIdx.y];
  const int el = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + el;
  float total = 0.0f;
  for (int32_t sl = 0; sl < jb.n_slabs; ++sl) {      // uniform over the workgroup
    const float *__restrict__ slab = jb.slabs[sl];
    float s0 = 0.0f, s1 = 0.0f;
    if (i < NRED) {
      int32_t p = q;
#pragma unroll 4      // eight loads in flight; the additions keep their order
      for (; p + 16 < jb.n_part; p += 32) {
        s0 += slab[(int64_t)p * PART + i];
        s1 += slab[(int64_t)(p + 16) * PART + i];
      }
      if (p < jb.n_part) s0 += slab[(int64_t)p * PART + i];
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
  if (q != 0 || i >= NRED) return;
  if (i < DD * DD)
    jb.dW[i] = total;
  else if (i < DD * DD + DD) {
    if (jb.db) jb.db[i - DD * DD] = total;
  } else if (jb.dalpha)
    jb.dalpha[0] = total;
}

inline int cu_count() {
  static int cached = 0;
  if (cached == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    cached = n;
  }
  return cached;
}

// Persistent grids: as many workgroups per CU as the variant's registers and LDS let be resident.
inline int fwd_grid(int64_t n_rows, int wg_per_cu) {
  return (int)std::min<int64_t>(rk::ceil_div(rk::ceil_div(n_rows, WT), 4), (int64_t)cu_count() * wg_per_cu);
}
constexpr int kBwdMaxGrid = 512;
inline int bwd_grid(int64_t n_rows) {
  return (int)std::min<int64_t>(rk::ceil_div(rk::ceil_div(n_rows, WT), 4),
                                std::min<int64_t>(cu_count(), kBwdMaxGrid));
}

}  // namespace

extern "C" {

int rk_dense_fwd_f32(const float *X1, const float *W1, const float *X2, const float *W2,
                     const float *b1, const float *b2, const float *alpha, float p_drop,
                     uint64_t seed, const uint64_t *seed_step, const float *residual, int64_t n_rows,
                     int32_t K, int32_t D, float *Y, void *stream) {
  RK_CHECK_ARG(n_rows >= 0, "negative row count");
  if (K != D || (D != DD && D != 128))
    return rk::fail(RK_ERR_UNSUPPORTED, "rk_dense_fwd_f32: built for 64 -> 64 and 128 -> 128 layers (got %d -> %d)", K, D);
  if (n_rows == 0) return RK_OK;
  RK_CHECK_ARG(X1 && W1 && Y, "null pointer");
  RK_CHECK_ARG((X2 == nullptr) == (W2 == nullptr), "X2 and W2 go together");
  RK_CHECK_ARG(p_drop >= 0.0f && p_drop < 1.0f, "dropout probability must be in [0, 1)");
  RK_CHECK_ARG(rk::aligned16(X1) && rk::aligned16(X2) && rk::aligned16(W1) && rk::aligned16(W2) &&
                   rk::aligned16(Y),
               "matrices must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (D == 128)
    return rk::dense128_fwd(X1, W1, X2, W2, b1, b2, alpha, p_drop, seed, seed_step, residual, n_rows, Y, st);
  // the plain layer fits two workgroups per CU (256 registers, 64 KB of LDS): a second wavefront per
  // SIMD splits operands and runs the epilogue while the first one multiplies (32.6 -> 29.9 us at
  // 255k rows); the two-input and the residual variants need the whole CU
  const unsigned grid = (unsigned)fwd_grid(n_rows, (X2 || residual) ? 1 : 2);
  if (X2) {
    if (residual)
      dense_fwd_kernel<true, true><<<grid, 256, 0, st>>>(X1, W1, X2, W2, b1, b2, alpha, p_drop, seed, seed_step, residual, n_rows, Y);
    else
      dense_fwd_kernel<true, false><<<grid, 256, 0, st>>>(X1, W1, X2, W2, b1, b2, alpha, p_drop, seed, seed_step, nullptr, n_rows, Y);
  } else {
    if (residual)
      dense_fwd_kernel<false, true><<<grid, 256, 0, st>>>(X1, W1, nullptr, nullptr, b1, nullptr, alpha, p_drop, seed, seed_step,
                                                           residual, n_rows, Y);
    else
      dense_fwd_kernel<false, false><<<grid, 256, 0, st>>>(X1, W1, nullptr, nullptr, b1, nullptr, alpha, p_drop, seed, seed_step,
                                                            nullptr, n_rows, Y);
  }
  RK_CHECK_LAUNCH("dense_fwd_kernel");
  return RK_OK;
}

size_t rk_dense_bwd_ws_bytes(int64_t n_rows, int32_t K, int32_t D) {
  if (n_rows > 0 && K == 128 && D == 128) return rk::dense128_bwd_ws_bytes();
  if (n_rows <= 0 || K != DD || D != DD) return 0;
  // sized for the largest grid, so that the answer does not depend on the device queried
  return ((size_t)kBwdMaxGrid * PART + (size_t)rk::kSlabMaxSeg * NRED) * sizeof(float);
}

int rk_dense_bwd_f32(const float *X, const float *W, const float *b, const float *alpha, float p_drop,
                     uint64_t seed, const uint64_t *seed_step, const float *G, int64_t n_rows, int32_t K,
                     int32_t D, float *dX,
                     float *dW, float *db, float *dalpha, void *ws, size_t ws_bytes, void *stream) {
  return rk_dense_bwd_accum_f32(X, W, b, alpha, p_drop, seed, seed_step, G, n_rows, K, D, dX, dW, db, dalpha, 0, ws,
                                ws_bytes, stream);
}

size_t rk_dense_bwd_slab_bytes(int64_t n_rows, int32_t K, int32_t D) {
  if (n_rows > 0 && K == 128 && D == 128) return rk::dense128_slab_bytes();
  if (n_rows <= 0 || K != DD || D != DD) return 0;
  return (size_t)kBwdMaxGrid * PART * sizeof(float);
}

int rk_dense_bwd_partials_f32(const float *X, const float *W, const float *b, const float *alpha, float p_drop,
                              uint64_t seed, const uint64_t *seed_step, const float *G, int64_t n_rows, int32_t K,
                              int32_t D, float *dX, float *slab, size_t slab_bytes, int32_t *n_part_out, void *stream) {
  RK_CHECK_ARG(n_rows > 0, "empty layer input");
  if (K != D || (D != DD && D != 128))
    return rk::fail(RK_ERR_UNSUPPORTED, "rk_dense_bwd_partials_f32: built for 64 -> 64 and 128 -> 128 layers (got %d -> %d)", K, D);
  RK_CHECK_ARG(X && W && G && slab && n_part_out, "null pointer");
  RK_CHECK_ARG(p_drop >= 0.0f && p_drop < 1.0f, "dropout probability must be in [0, 1)");
  RK_CHECK_ARG(slab_bytes >= rk_dense_bwd_slab_bytes(n_rows, K, D), "slab too small");
  RK_CHECK_ARG(rk::aligned16(X) && rk::aligned16(W) && rk::aligned16(G) && rk::aligned16(dX) && rk::aligned16(slab),
               "matrices must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (D == 128)
    return rk::dense128_bwd_partials(X, W, b, alpha, p_drop, seed, seed_step, G, n_rows, dX, slab, n_part_out, st);
  const int grid = bwd_grid(n_rows);
  if (alpha)
    dense_bwd_kernel<true><<<grid, 256, 0, st>>>(X, W, b, alpha, p_drop, seed, seed_step, G, n_rows, dX, slab);
  else
    dense_bwd_kernel<false><<<grid, 256, 0, st>>>(X, W, b, nullptr, p_drop, seed, seed_step, G, n_rows, dX, slab);
  RK_CHECK_LAUNCH("dense_bwd_kernel");
  *n_part_out = grid;
  return RK_OK;
}

int rk_dense_reduce_jobs_f32(const rk_dense_reduce_job *jobs, int32_t n_jobs, void *stream) {
  return rk_dense_reduce_jobs_d_f32(jobs, n_jobs, DD, stream);
}

int rk_dense_reduce_jobs_d_f32(const rk_dense_reduce_job *jobs, int32_t n_jobs, int32_t D, void *stream) {
  RK_CHECK_ARG(n_jobs >= 0 && (jobs || n_jobs == 0), "bad job list");
  if (D != DD && D != 128)
    return rk::fail(RK_ERR_UNSUPPORTED, "rk_dense_reduce_jobs_d_f32: built for layer widths 64 and 128 (got %d)", D);
  for (int32_t j = 0; j < n_jobs; ++j) {
    RK_CHECK_ARG(jobs[j].n_slabs >= 1 && jobs[j].n_slabs <= RK_DENSE_REDUCE_MAX_SLABS, "a job lists 1..RK_DENSE_REDUCE_MAX_SLABS slabs");
    RK_CHECK_ARG(jobs[j].n_part >= 1 && jobs[j].n_part <= kBwdMaxGrid && jobs[j].dW, "bad job");
    for (int32_t sl = 0; sl < jobs[j].n_slabs; ++sl) RK_CHECK_ARG(jobs[j].slabs[sl], "null slab");
  }
  hipStream_t st = (hipStream_t)stream;
  if (D == 128) return rk::dense128_reduce_jobs(jobs, n_jobs, st);
  for (int32_t j0 = 0; j0 < n_jobs; j0 += kJobsPerLaunch) {
    const int32_t nj = std::min<int32_t>(kJobsPerLaunch, n_jobs - j0);
    ReduceJobs args;
    memset(&args, 0, sizeof args);
    for (int32_t j = 0; j < nj; ++j) args.job[j] = jobs[j0 + j];
    dense_reduce_jobs_kernel<<<dim3((NRED + 63) / 64, nj), 1024, 0, st>>>(args);
    RK_CHECK_LAUNCH("dense_reduce_jobs_kernel");
  }
  return RK_OK;
}

int rk_dense_bwd_accum_f32(const float *X, const float *W, const float *b, const float *alpha, float p_drop,
                           uint64_t seed, const uint64_t *seed_step, const float *G, int64_t n_rows, int32_t K,
                           int32_t D, float *dX, float *dW, float *db, float *dalpha, int32_t accumulate, void *ws,
                           size_t ws_bytes, void *stream) {
  RK_CHECK_ARG(n_rows > 0, "empty layer input");
  if (K != D || (D != DD && D != 128))
    return rk::fail(RK_ERR_UNSUPPORTED, "rk_dense_bwd_f32: built for 64 -> 64 and 128 -> 128 layers (got %d -> %d)", K, D);
  RK_CHECK_ARG(X && W && G && dW && ws, "null pointer");
  RK_CHECK_ARG(p_drop >= 0.0f && p_drop < 1.0f, "dropout probability must be in [0, 1)");
  RK_CHECK_ARG(ws_bytes >= rk_dense_bwd_ws_bytes(n_rows, K, D), "workspace too small");
  RK_CHECK_ARG(rk::aligned16(X) && rk::aligned16(W) && rk::aligned16(G) && rk::aligned16(dX),
               "matrices must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (D == 128)
    return rk::dense128_bwd(X, W, b, alpha, p_drop, seed, seed_step, G, n_rows, dX, dW, db, dalpha, accumulate, ws, st);
  const int grid = bwd_grid(n_rows);
  if (alpha)
    dense_bwd_kernel<true><<<grid, 256, 0, st>>>(X, W, b, alpha, p_drop, seed, seed_step, G, n_rows, dX, (float *)ws);
  else
    dense_bwd_kernel<false><<<grid, 256, 0, st>>>(X, W, b, nullptr, p_drop, seed, seed_step, G, n_rows, dX, (float *)ws);
  RK_CHECK_LAUNCH("dense_bwd_kernel");
  dense_reduce_kernel<<<(NRED + 15) / 16, 256, 0, st>>>((const float *)ws, grid, dW, db, dalpha, accumulate);
  RK_CHECK_LAUNCH("dense_reduce_kernel");
  return RK_OK;
}

}  // extern "C"
