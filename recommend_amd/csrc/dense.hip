// Fused node-wise layer  Y = dropout(prelu(X1 W1^T [+ X2 W2^T] + b1 [+ b2])) [+ R]
// and its backward, for the 64-wide embeddings of the propagation hops
// (gfx950, v_mfma_f32_32x32x2_f32, exact fp32).
//
// These are the D x D layers wrapped around every SpMM hop and inside Combine
// (reference: graph_constructor.py:204, Combin_Layer.py:26-30, out_Layer.py:36,
// Main_Model.py:162-163).  N is the node count, so each layer is a skinny GEMM
// that streams an N x 64 matrix once: HBM-bound.  Forward reads X (and R), writes
// Y.  Backward reads X and G once, RECOMPUTES Z = X W^T + b on the matrix cores
// instead of saving it, regenerates the dropout mask from the same counter hash,
// and produces dX, dW, db and d(alpha) in the same pass; per-workgroup partial
// sums of dW/db/d(alpha) are reduced in workgroup order by a second launch (no
// float atomics, bitwise reproducible).
//
// MFMA operand maps (32x32x2): lane l supplies A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]; the result register r of lane l is C[(r&3) + 8*(r>>2) +
// 4*(l>>5)][l&31].  K is walked as (k, k+32) pairs so that a lane's 32 operand
// values are contiguous in its LDS row.
#include "common.h"
#include "slab_reduce.h"

namespace {

using rk::kWave;
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int DD = 64;        // layer width (in = out)
constexpr int LDW = DD + 4;   // padded LDS row: conflict-free b128 row reads

__device__ __forceinline__ int crow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// counter-based dropout decision, identical in forward and backward
__device__ __forceinline__ bool keep_elem(uint32_t seed_lo, uint32_t seed_hi, uint64_t idx,
                                          uint32_t thresh24) {
  uint32_t x = (uint32_t)idx * 0x9E3779B1u ^ seed_lo;
  x ^= (uint32_t)(idx >> 32) * 0x85EBCA77u;
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x += seed_hi;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return (x >> 8) >= thresh24;
}

// 32 operand values of one lane: row `row` of an LDS tile, columns 32*half .. 32*half+31
__device__ __forceinline__ void load_frag(const float *tile, int row, int half, float (&f)[32]) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float4 v = *reinterpret_cast<const float4 *>(&tile[row * LDW + 32 * half + 4 * q]);
    f[4 * q + 0] = v.x;
    f[4 * q + 1] = v.y;
    f[4 * q + 2] = v.z;
    f[4 * q + 3] = v.w;
  }
}

// copy a rows x 64 fp32 block (global row-major) into a padded LDS tile; rows past n_valid are zero
template <int ROWS>
__device__ __forceinline__ void stage_tile(const float *__restrict__ g, int64_t row0, int64_t n_rows,
                                           float *tile, int tid) {
  constexpr int PER = ROWS * 16 / 256;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int e = tid + 256 * j, r = e >> 4, c = (e & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + r < n_rows) v = *reinterpret_cast<const float4 *>(g + (row0 + r) * DD + c);
    *reinterpret_cast<float4 *>(&tile[r * LDW + c]) = v;
  }
}

// ------------------------------------------------------------------ forward
constexpr int FBM = 128;  // rows per workgroup tile (4 wavefronts x 32)

template <bool DUAL>
__global__ __launch_bounds__(256) void dense_fwd_kernel(
    const float *__restrict__ X1, const float *__restrict__ W1, const float *__restrict__ X2,
    const float *__restrict__ W2, const float *__restrict__ b1, const float *__restrict__ b2,
    const float *__restrict__ alpha, float p_drop, uint32_t seed_lo, uint32_t seed_hi,
    const float *__restrict__ R, int64_t n_rows, float *__restrict__ Y) {
  __shared__ __attribute__((aligned(16))) float Ws[DD * LDW];
  __shared__ __attribute__((aligned(16))) float Xs[FBM * LDW];
  const int tid = threadIdx.x, wave = tid / kWave, lane = tid % kWave;
  const int col = lane & 31, half = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * FBM;

  floatx16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

#pragma unroll
  for (int pass = 0; pass < (DUAL ? 2 : 1); ++pass) {
    if (pass) __syncthreads();
    stage_tile<DD>(pass ? W2 : W1, 0, DD, Ws, tid);
    stage_tile<FBM>(pass ? X2 : X1, row0, n_rows, Xs, tid);
    __syncthreads();
    float a[32], b[32];
    load_frag(Xs, wave * 32 + col, half, a);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      load_frag(Ws, 32 * t + col, half, b);
#pragma unroll
      for (int m = 0; m < 32; ++m)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[m], acc[t], 0, 0, 0);
    }
  }

  const float al = alpha ? alpha[0] : 1.0f;
  const bool act = alpha != nullptr;
  const bool drop = p_drop > 0.0f;
  const uint32_t thresh = (uint32_t)(p_drop * 16777216.0f);
  const float scale = drop ? 1.0f / (1.0f - p_drop) : 1.0f;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int n = 32 * t + col;
    float bias = b1 ? b1[n] : 0.0f;
    if (DUAL && b2) bias += b2[n];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = row0 + wave * 32 + crow(r, half);
      if (row < n_rows) {
        float z = acc[t][r] + bias;
        if (act) z = z > 0.0f ? z : al * z;
        if (drop) z = keep_elem(seed_lo, seed_hi, (uint64_t)row * DD + n, thresh) ? z * scale : 0.0f;
        if (R) z += R[row * DD + n];
        Y[row * DD + n] = z;
      }
    }
  }
}

// ----------------------------------------------------------------- backward
constexpr int BBM = 64;                       // rows per tile
constexpr int PART = DD * DD + DD + 8;        // per-workgroup partial: dW, db, d(alpha) (+pad)

// ACT: Y = dropout(prelu(X W^T + b)); otherwise Y = X W^T + b (dZ = G).
template <bool ACT>
__global__ __launch_bounds__(256) void dense_bwd_kernel(
    const float *__restrict__ X, const float *__restrict__ W, const float *__restrict__ b,
    const float *__restrict__ alpha, float p_drop, uint32_t seed_lo, uint32_t seed_hi,
    const float *__restrict__ G, int64_t n_rows, int32_t n_tiles, float *__restrict__ dX,
    float *__restrict__ part) {
  __shared__ __attribute__((aligned(16))) float Ws[DD * LDW];
  __shared__ __attribute__((aligned(16))) float Xs[BBM * LDW];
  __shared__ __attribute__((aligned(16))) float Zs[BBM * LDW];   // dZ tile
  __shared__ float red[4][DD + 1];
  const int tid = threadIdx.x, wave = tid / kWave, lane = tid % kWave;
  const int col = lane & 31, half = lane >> 5;
  const int rt = wave & 1, ct = wave >> 1;   // this wavefront's 32x32 sub-tile (row tile, column tile)

  stage_tile<DD>(W, 0, DD, Ws, tid);
  const float al = ACT ? alpha[0] : 1.0f;
  const bool drop = p_drop > 0.0f;
  const uint32_t thresh = (uint32_t)(p_drop * 16777216.0f);
  const float scale = drop ? 1.0f / (1.0f - p_drop) : 1.0f;
  const float bias = (ACT && b) ? b[32 * ct + col] : 0.0f;

  floatx16 dw;   // dW[n = 32*rt + row-of-C][k = 32*ct + col], accumulated over all tiles of this workgroup
#pragma unroll
  for (int r = 0; r < 16; ++r) dw[r] = 0.0f;
  float db_acc = 0.0f, da_acc = 0.0f;

  for (int32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row0 = (int64_t)tile * BBM;
    __syncthreads();   // previous tile's readers are done (also orders the W staging the first time)
    stage_tile<BBM>(X, row0, n_rows, Xs, tid);
    if (!ACT) {
      // dZ = G (times the dropout mask when the layer is linear + dropout)
#pragma unroll
      for (int j = 0; j < BBM * 16 / 256; ++j) {
        const int e = tid + 256 * j, r = e >> 4, c = (e & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + r < n_rows) {
          v = *reinterpret_cast<const float4 *>(G + (row0 + r) * DD + c);
          if (drop) {
            const uint64_t idx = (uint64_t)(row0 + r) * DD + c;
            v.x = keep_elem(seed_lo, seed_hi, idx + 0, thresh) ? v.x * scale : 0.0f;
            v.y = keep_elem(seed_lo, seed_hi, idx + 1, thresh) ? v.y * scale : 0.0f;
            v.z = keep_elem(seed_lo, seed_hi, idx + 2, thresh) ? v.z * scale : 0.0f;
            v.w = keep_elem(seed_lo, seed_hi, idx + 3, thresh) ? v.w * scale : 0.0f;
          }
        }
        *reinterpret_cast<float4 *>(&Zs[r * LDW + c]) = v;
      }
    }
    __syncthreads();

    if (ACT) {
      // upstream gradient in accumulator layout, issued before the MFMAs that hide its latency
      float g[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + 32 * rt + crow(r, half);
        g[r] = row < n_rows ? G[row * DD + 32 * ct + col] : 0.0f;
      }
      floatx16 z;
#pragma unroll
      for (int r = 0; r < 16; ++r) z[r] = 0.0f;
      float a[32], bw[32];
      load_frag(Xs, 32 * rt + col, half, a);
      load_frag(Ws, 32 * ct + col, half, bw);
#pragma unroll
      for (int m = 0; m < 32; ++m) z = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bw[m], z, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lrow = 32 * rt + crow(r, half);
        const int64_t row = row0 + lrow;
        const float zz = z[r] + bias;
        float gs = g[r];
        if (drop)
          gs = keep_elem(seed_lo, seed_hi, (uint64_t)row * DD + 32 * ct + col, thresh) ? gs * scale : 0.0f;
        const bool pos = zz > 0.0f;
        const float dz = pos ? gs : al * gs;     // torch prelu backward: z > 0 ? g : alpha * g
        da_acc += pos ? 0.0f : zz * gs;
        db_acc += dz;
        Zs[lrow * LDW + 32 * ct + col] = dz;
      }
      __syncthreads();
    } else {
      // db from the staged gradient: column (tid & 63), rows strided by 4 over the tile
#pragma unroll
      for (int r = 0; r < BBM / 4; ++r) db_acc += Zs[(4 * r + (tid >> 6)) * LDW + (tid & 63)];
    }

    if (dX) {
      // dX[row][k] = sum_n dZ[row][n] W[n][k]   (A = dZ rows, B = W columns)
      floatx16 dx;
#pragma unroll
      for (int r = 0; r < 16; ++r) dx[r] = 0.0f;
      float a[32];
      load_frag(Zs, 32 * rt + col, half, a);
#pragma unroll
      for (int m = 0; m < 32; ++m) {
        const float bv = Ws[(32 * half + m) * LDW + 32 * ct + col];
        dx = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bv, dx, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + 32 * rt + crow(r, half);
        if (row < n_rows) dX[row * DD + 32 * ct + col] = dx[r];
      }
    }
    // dW[n][k] += sum_row dZ[row][n] X[row][k]   (A[i=n][kk=row], B[kk=row][j=k]); rows past the end are zero
#pragma unroll
    for (int kk = 0; kk < BBM; kk += 2) {
      const float av = Zs[(kk + half) * LDW + 32 * rt + col];
      const float bv = Xs[(kk + half) * LDW + 32 * ct + col];
      dw = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, dw, 0, 0, 0);
    }
  }

  // per-workgroup partials: dW[64][64], db[64], d(alpha)
  float *out = part + (int64_t)blockIdx.x * PART;
#pragma unroll
  for (int r = 0; r < 16; ++r) out[(32 * rt + crow(r, half)) * DD + 32 * ct + col] = dw[r];
  __syncthreads();
  if (ACT) {
    // lanes l and l+32 hold different rows of column 32*ct+col; wavefronts rt = 0,1 share the column
    db_acc += __shfl_xor(db_acc, 32, kWave);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) da_acc += __shfl_xor(da_acc, off, kWave);
    if (half == 0) red[wave][col] = db_acc;
    if (lane == 0) red[wave][DD] = da_acc;
    __syncthreads();
    if (tid < DD) out[DD * DD + tid] = red[2 * (tid >> 5)][tid & 31] + red[2 * (tid >> 5) + 1][tid & 31];
    if (tid == 0) out[DD * DD + DD] = ((red[0][DD] + red[1][DD]) + red[2][DD]) + red[3][DD];
  } else {
    red[tid >> 6][tid & 63] = db_acc;
    __syncthreads();
    if (tid < DD) out[DD * DD + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
    if (tid == 0) out[DD * DD + DD] = 0.0f;
  }
}

// last step of the partial reduction: adds the segment sums and routes them to dW / db / dalpha
constexpr int NRED = DD * DD + DD + 1;
__global__ __launch_bounds__(256) void dense_route_kernel(const float *__restrict__ tmp, int32_t n_seg,
                                                          float *__restrict__ dW, float *__restrict__ db,
                                                          float *__restrict__ dalpha) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= NRED) return;
  float s = 0.0f;
  for (int32_t k = 0; k < n_seg; ++k) s += tmp[(int64_t)k * NRED + i];
  if (i < DD * DD)
    dW[i] = s;
  else if (i < DD * DD + DD) {
    if (db) db[i - DD * DD] = s;
  } else if (dalpha)
    dalpha[0] = s;
}

inline int bwd_grid(int64_t n_rows) {
  const int64_t tiles = rk::ceil_div(n_rows, BBM);
  return (int)std::min<int64_t>(tiles, 768);
}

}  // namespace

extern "C" {

int rk_dense_fwd_f32(const float *X1, const float *W1, const float *X2, const float *W2,
                     const float *b1, const float *b2, const float *alpha, float p_drop,
                     uint64_t seed, const float *residual, int64_t n_rows, int32_t K, int32_t D,
                     float *Y, void *stream) {
  RK_CHECK_ARG(n_rows >= 0, "negative row count");
  if (K != DD || D != DD)
    return rk::fail(RK_ERR_UNSUPPORTED, "rk_dense_fwd_f32: only 64 -> 64 layers are built (got %d -> %d)", K, D);
  if (n_rows == 0) return RK_OK;
  RK_CHECK_ARG(X1 && W1 && Y, "null pointer");
  RK_CHECK_ARG((X2 == nullptr) == (W2 == nullptr), "X2 and W2 go together");
  RK_CHECK_ARG(p_drop >= 0.0f && p_drop < 1.0f, "dropout probability must be in [0, 1)");
  RK_CHECK_ARG(rk::aligned16(X1) && rk::aligned16(X2) && rk::aligned16(W1) && rk::aligned16(W2) &&
                   rk::aligned16(Y),
               "matrices must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)rk::ceil_div(n_rows, FBM);
  const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
  if (X2)
    dense_fwd_kernel<true><<<grid, 256, 0, st>>>(X1, W1, X2, W2, b1, b2, alpha, p_drop, lo, hi, residual, n_rows, Y);
  else
    dense_fwd_kernel<false><<<grid, 256, 0, st>>>(X1, W1, nullptr, nullptr, b1, nullptr, alpha, p_drop, lo, hi,
                                                   residual, n_rows, Y);
  RK_CHECK_LAUNCH("dense_fwd_kernel");
  return RK_OK;
}

size_t rk_dense_bwd_ws_bytes(int64_t n_rows, int32_t K, int32_t D) {
  if (n_rows <= 0 || K != DD || D != DD) return 0;
  return ((size_t)bwd_grid(n_rows) * PART + (size_t)rk::kSlabMaxSeg * NRED) * sizeof(float);
}

int rk_dense_bwd_f32(const float *X, const float *W, const float *b, const float *alpha, float p_drop,
                     uint64_t seed, const float *G, int64_t n_rows, int32_t K, int32_t D, float *dX,
                     float *dW, float *db, float *dalpha, void *ws, size_t ws_bytes, void *stream) {
  RK_CHECK_ARG(n_rows > 0, "empty layer input");
  if (K != DD || D != DD)
    return rk::fail(RK_ERR_UNSUPPORTED, "rk_dense_bwd_f32: only 64 -> 64 layers are built (got %d -> %d)", K, D);
  RK_CHECK_ARG(X && W && G && dW && ws, "null pointer");
  RK_CHECK_ARG(p_drop >= 0.0f && p_drop < 1.0f, "dropout probability must be in [0, 1)");
  RK_CHECK_ARG(ws_bytes >= rk_dense_bwd_ws_bytes(n_rows, K, D), "workspace too small");
  RK_CHECK_ARG(rk::aligned16(X) && rk::aligned16(W) && rk::aligned16(G) && rk::aligned16(dX),
               "matrices must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int grid = bwd_grid(n_rows);
  const int32_t tiles = (int32_t)rk::ceil_div(n_rows, BBM);
  const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
  if (alpha)
    dense_bwd_kernel<true><<<grid, 256, 0, st>>>(X, W, b, alpha, p_drop, lo, hi, G, n_rows, tiles, dX, (float *)ws);
  else
    dense_bwd_kernel<false><<<grid, 256, 0, st>>>(X, W, b, nullptr, p_drop, lo, hi, G, n_rows, tiles, dX, (float *)ws);
  RK_CHECK_LAUNCH("dense_bwd_kernel");
  float *tmp = (float *)ws + (size_t)grid * PART;
  const int n_seg = rk::slab_segments(grid);
  dim3 rgrid((unsigned)rk::ceil_div(NRED, 64), (unsigned)n_seg);
  rk::slab_reduce_stage1<<<rgrid, 256, 0, st>>>((const float *)ws, grid, PART, NRED, n_seg, tmp);
  RK_CHECK_LAUNCH("slab_reduce_stage1");
  dense_route_kernel<<<(NRED + 255) / 256, 256, 0, st>>>(tmp, n_seg, dW, db, dalpha);
  RK_CHECK_LAUNCH("dense_route_kernel");
  return RK_OK;
}

}  // extern "C"
