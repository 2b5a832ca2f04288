// Laboratory for the hop fused with the layer that consumes it:
//   X = A h  (stored: the layer's backward needs it)   and   Y = dropout(prelu(X W^T + b))
// in one launch, against the library's two launches (rk_spmm_csr_f32 + rk_dense_fwd_f32) on a
// dumped CSR (tools/dump_csr.py).  X must match bit for bit, Y to 1e-5 of its scale (the library's
// layer multiplies bf16 three-way splits, this kernel fp32 operands; the dropout masks are the same).
//   hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=off \
//         tools/hop_layer_lab.hip -o tools/_variants/hop_layer_lab -Lrecommend_amd/lib -lrecommend_amd \
//         -Wl,-rpath,'$ORIGIN/../../recommend_amd/lib'
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../include/recommend_amd.h"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int kChunk = 2048;
constexpr int LPR = 16, GROUPS = 16, DD = 64;
constexpr int LDX = DD + 4;   // padded LDS row: 16-byte aligned, rows shifted by one float4 of banks

template <int J>
__device__ __forceinline__ int bcast_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x150 + J, 0xF, 0xF, false);
}
template <int J>
__device__ __forceinline__ float bcast_f(float v) {
  return __builtin_bit_cast(float, bcast_i<J>(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ void fma4(v4f &a, float v, const v4f &x) {
  a.x = fmaf(v, x.x, a.x);
  a.y = fmaf(v, x.y, a.y);
  a.z = fmaf(v, x.z, a.z);
  a.w = fmaf(v, x.w, a.w);
}
template <int Q>
__device__ __forceinline__ void gather4(const v4f *__restrict__ h4, int32_t c_l, float v_l, int sub, v4f &acc) {
  const int32_t c0 = bcast_i<Q>(c_l), c1 = bcast_i<Q + 1>(c_l), c2 = bcast_i<Q + 2>(c_l), c3 = bcast_i<Q + 3>(c_l);
  const v4f x0 = h4[(int64_t)c0 * LPR + sub];
  const v4f x1 = h4[(int64_t)c1 * LPR + sub];
  const v4f x2 = h4[(int64_t)c2 * LPR + sub];
  const v4f x3 = h4[(int64_t)c3 * LPR + sub];
  fma4(acc, bcast_f<Q>(v_l), x0);
  fma4(acc, bcast_f<Q + 1>(v_l), x1);
  fma4(acc, bcast_f<Q + 2>(v_l), x2);
  fma4(acc, bcast_f<Q + 3>(v_l), x3);
}
template <int Q>
__device__ __forceinline__ void batch_from(const v4f *__restrict__ h4, int32_t c_l, float v_l, int sub, int cnt, v4f &acc) {
  if constexpr (Q < LPR) {
    if (Q < cnt) {
      gather4<Q>(h4, c_l, v_l, sub, acc);
      batch_from<Q + 4>(h4, c_l, v_l, sub, cnt, acc);
    }
  }
}
__device__ __forceinline__ void accumulate_edges(const int32_t *__restrict__ col, const float *__restrict__ val,
                                                 const v4f *__restrict__ h4, int32_t b, int32_t e, int sub, v4f &acc) {
  for (int32_t e0 = b; e0 < e; e0 += LPR) {
    const int cnt = min(LPR, e - e0);
    const int32_t mine = e0 + min(sub, cnt - 1);
    const int32_t c_l = col[mine];
    const float v_l = sub < cnt ? val[mine] : 0.0f;
    batch_from<0>(h4, c_l, v_l, sub, cnt, acc);
  }
}

// the layer kernels' dropout hash (one hash per 2 x 2 block of elements)
struct DropQuad {
  uint32_t lo, hi;
};
__device__ __forceinline__ DropQuad drop_quad(uint32_t seed_lo, uint32_t seed_hi, uint32_t row_pair, uint32_t col_term) {
  uint32_t x = row_pair * 0x9E3779B1u + col_term;
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

struct LayerArgs {
  const float *W, *b;   // [64][64] row-major (out, in), [64]
  float alpha, p_drop;
  uint32_t seed_lo, seed_hi;
  v4f *Y4;
};

// one row through the layer on the vector ALU (rows finished by the long tier): x[64] in LDS,
// lane n produces column n; called by one whole wavefront
__device__ __forceinline__ void row_layer(const float *x, int64_t row, const LayerArgs &la, int lane) {
  float z = la.b ? la.b[lane] : 0.0f;
  const v4f *w4 = reinterpret_cast<const v4f *>(la.W + lane * DD);
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const v4f w = w4[q];
    const v4f xv = *reinterpret_cast<const v4f *>(x + 4 * q);
    z = fmaf(w.x, xv.x, z);
    z = fmaf(w.y, xv.y, z);
    z = fmaf(w.z, xv.z, z);
    z = fmaf(w.w, xv.w, z);
  }
  z = z > 0.0f ? z : la.alpha * z;
  if (la.p_drop > 0.0f) {
    const uint32_t thresh = (uint32_t)(la.p_drop * 65536.0f + 0.5f);
    const DropQuad dq = drop_quad(la.seed_lo, la.seed_hi, (uint32_t)(row >> 1),
                                  (uint32_t)(lane >> 1) * 0x85EBCA77u ^ la.seed_lo);
    z = kept((row & 1) ? dq.hi : dq.lo, lane & 1, thresh) ? z * (1.0f / (1.0f - la.p_drop)) : 0.0f;
  }
  reinterpret_cast<float *>(la.Y4)[row * DD + lane] = z;
}

struct PlanArgs {
  int32_t long_thresh, n_chunks, n_multi;
  const int32_t *chunk_row, *chunk_begin, *chunk_end, *chunk_slot, *multi_slot_ptr;
  v4f *partial4;
  int32_t *arrive;
};

__device__ __forceinline__ void store_through(v4f *p, const v4f &v) {
  unsigned long long *q = reinterpret_cast<unsigned long long *>(p);
  const unsigned long long lo = (unsigned long long)__float_as_uint(v.x) | ((unsigned long long)__float_as_uint(v.y) << 32);
  const unsigned long long hi = (unsigned long long)__float_as_uint(v.z) | ((unsigned long long)__float_as_uint(v.w) << 32);
  __hip_atomic_store(q, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ v4f load_through(const v4f *p) {
  unsigned long long *q = reinterpret_cast<unsigned long long *>(const_cast<v4f *>(p));
  const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return v4f{__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)), __uint_as_float((unsigned)hi),
             __uint_as_float((unsigned)(hi >> 32))};
}

// PASSES x 16 consecutive rows per short-tier workgroup: the weight fragments are fetched once
// per workgroup and serve PASSES groups of rows.
template <int PASSES, bool LAYER>
__global__ __launch_bounds__(256) void hop_layer_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                        const float *__restrict__ val, int32_t n_rows,
                                                        const v4f *__restrict__ h4, v4f *__restrict__ out4, PlanArgs pl,
                                                        LayerArgs la) {
  __shared__ v4f red[GROUPS][LPR];
  __shared__ __attribute__((aligned(16))) float xs[GROUPS * LDX];
  __shared__ __attribute__((aligned(16))) float ys[GROUPS * LDX];
  const int tid = threadIdx.x;
  const int sub = tid % LPR, grp = tid / LPR;
  const int wave = tid / 64, lane = tid % 64;
  if ((int32_t)blockIdx.x >= pl.n_chunks) {
    const int64_t row0 = (int64_t)((int32_t)blockIdx.x - pl.n_chunks) * (GROUPS * PASSES);
    // B operand of v_mfma_f32_16x16x4_f32: lane l supplies B[k][j = l & 15] with k = l >> 4 of the
    // step's four; step s stands for k in {s, 16 + s, 32 + s, 48 + s}, so that a lane's sixteen
    // values are 16 consecutive floats of its weight row (and of its X row)
    const int j = lane & 15, kq = lane >> 4;
    const int n = 16 * wave + j;
    float wf[16];
    float bias = 0.0f;
    uint32_t col_term = 0;
    if constexpr (LAYER) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const v4f w = *reinterpret_cast<const v4f *>(la.W + n * DD + 16 * kq + 4 * q);
        wf[4 * q + 0] = w.x, wf[4 * q + 1] = w.y, wf[4 * q + 2] = w.z, wf[4 * q + 3] = w.w;
      }
      bias = la.b ? la.b[n] : 0.0f;
      col_term = (uint32_t)(n >> 1) * 0x85EBCA77u ^ la.seed_lo;
    }
    const bool drop = la.p_drop > 0.0f;
    const uint32_t thresh = (uint32_t)(la.p_drop * 65536.0f + 0.5f);
    const float scale = drop ? 1.0f / (1.0f - la.p_drop) : 1.0f;
#pragma unroll 1
    for (int pass = 0; pass < PASSES; ++pass) {
      const int64_t rbase = row0 + pass * GROUPS;
      if (rbase >= n_rows) break;                 // uniform over the workgroup
      const int64_t row = rbase + grp;
      bool mine = row < n_rows;
      v4f acc = {0.f, 0.f, 0.f, 0.f};
      if (mine) {
        const int32_t b = rowptr[row], e = rowptr[row + 1];
        mine = e - b <= pl.long_thresh;           // longer rows belong to the long tier
        if (mine) {
          accumulate_edges(col, val, h4, b, e, sub, acc);
          out4[row * LPR + sub] = acc;
        }
      }
      if constexpr (LAYER) {
        *reinterpret_cast<v4f *>(xs + grp * LDX + 4 * sub) = acc;
        __syncthreads();
        v4f c = {0.f, 0.f, 0.f, 0.f};
        float xa[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const v4f x = *reinterpret_cast<const v4f *>(xs + j * LDX + 16 * kq + 4 * q);
          xa[4 * q + 0] = x.x, xa[4 * q + 1] = x.y, xa[4 * q + 2] = x.z, xa[4 * q + 3] = x.w;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], wf[s], c, 0, 0, 0);
        // C: lane holds rows 4 * kq + r (r < 4) of column n
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const int64_t grow = rbase + 4 * kq + r;   // even
          DropQuad dq{0u, 0u};
          if (drop) dq = drop_quad(la.seed_lo, la.seed_hi, (uint32_t)(grow >> 1), col_term);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            float z = c[r + e] + bias;
            z = z > 0.0f ? z : la.alpha * z;
            if (drop) z = kept(e ? dq.hi : dq.lo, n & 1, thresh) ? z * scale : 0.0f;
            ys[(4 * kq + r + e) * LDX + n] = z;
          }
        }
        __syncthreads();
        if (mine) la.Y4[row * LPR + sub] = *reinterpret_cast<const v4f *>(ys + grp * LDX + 4 * sub);
      }
    }
    return;
  }
  // long tier: as in the library; the wavefront that finishes a row sends it through the layer
  const int32_t chunk = blockIdx.x;
  const int32_t cb = pl.chunk_begin[chunk], ce = pl.chunk_end[chunk];
  const int32_t span = ((ce - cb + GROUPS - 1) / GROUPS + LPR - 1) / LPR * LPR;
  const int32_t b = min(cb + grp * span, ce), e = min(b + span, ce);
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  accumulate_edges(col, val, h4, b, e, sub, acc);
  red[grp][sub] = acc;
  __syncthreads();
  const int32_t slot = pl.chunk_slot[chunk];
  if (grp == 0) {
    v4f s = red[0][sub];
#pragma unroll
    for (int g = 1; g < GROUPS; ++g) s += red[g][sub];
    if (slot < 0) {
      out4[(int64_t)pl.chunk_row[chunk] * LPR + sub] = s;
      if constexpr (LAYER) *reinterpret_cast<v4f *>(xs + 4 * sub) = s;
    } else {
      store_through(&pl.partial4[(int64_t)slot * LPR + sub], s);
    }
  }
  if (slot < 0) {
    if constexpr (LAYER) {
      if (tid < 64) {                       // lanes 0..15 wrote xs: same wavefront, LDS in order
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        row_layer(xs, pl.chunk_row[chunk], la, lane);
      }
    }
    return;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int32_t *flag = reinterpret_cast<int32_t *>(&red[0][0]);
  if (tid == 0) {
    int32_t lo = 0, hi = pl.n_multi;
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
  if (mi < 0 || tid >= 64) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  {
    constexpr int GPW = 64 / LPR;
    const int g = tid / LPR;
    v4f a = {0.f, 0.f, 0.f, 0.f};
    for (int32_t sl = pl.multi_slot_ptr[mi] + g; sl < pl.multi_slot_ptr[mi + 1]; sl += GPW)
      a += load_through(&pl.partial4[(int64_t)sl * LPR + sub]);
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
      a.x += __shfl_xor(a.x, off, 64);
      a.y += __shfl_xor(a.y, off, 64);
      a.z += __shfl_xor(a.z, off, 64);
      a.w += __shfl_xor(a.w, off, 64);
    }
    if (g == 0) {
      out4[(int64_t)pl.chunk_row[chunk] * LPR + sub] = a;
      if constexpr (LAYER) *reinterpret_cast<v4f *>(xs + 4 * sub) = a;
    }
    if constexpr (LAYER) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      row_layer(xs, pl.chunk_row[chunk], la, lane);
    }
  }
}

template <class T>
T *to_dev(const std::vector<T> &v) {
  T *p = nullptr;
  CK(hipMalloc(&p, std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (!v.empty()) CK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return p;
}

int main(int argc, char **argv) {
  const char *path = argc > 1 ? argv[1] : "tools/_variants/elec_csr.bin";
  const int reps = argc > 2 ? atoi(argv[2]) : 100;
  FILE *f = fopen(path, "rb");
  if (!f) {
    fprintf(stderr, "cannot open %s\n", path);
    return 1;
  }
  int64_t hdr[3];
  if (fread(hdr, 8, 3, f) != 3) return 1;
  const int64_t N = hdr[0], nnz = hdr[1];
  std::vector<int32_t> rowptr(N + 1), colv(nnz);
  std::vector<float> valv(nnz);
  if (fread(rowptr.data(), 4, N + 1, f) != (size_t)N + 1) return 1;
  if (fread(colv.data(), 4, nnz, f) != (size_t)nnz) return 1;
  if (fread(valv.data(), 4, nnz, f) != (size_t)nnz) return 1;
  fclose(f);
  printf("graph: N %lld nnz %lld\n", (long long)N, (long long)nnz);

  int32_t nc = 0, nm = 0, ns = 0;
  rk_spmm_plan_host(rowptr.data(), (int32_t)N, 64, &nc, &nm, &ns, 0, 0, 0, 0, 0, 0);
  std::vector<int32_t> crow(std::max(nc, 1)), cbeg(std::max(nc, 1)), cend(std::max(nc, 1)), cslot(std::max(nc, 1)),
      mrows(std::max(nm, 1)), mptr(nm + 1);
  rk_spmm_plan_host(rowptr.data(), (int32_t)N, 64, &nc, &nm, &ns, crow.data(), cbeg.data(), cend.data(), cslot.data(),
                    mrows.data(), mptr.data());
  printf("plan: %d chunks, %d multi rows, %d slots\n", nc, nm, ns);

  int32_t *d_rowptr = to_dev(rowptr), *d_col = to_dev(colv);
  float *d_val = to_dev(valv);
  int32_t *d_crow = to_dev(crow), *d_cbeg = to_dev(cbeg), *d_cend = to_dev(cend), *d_cslot = to_dev(cslot);
  int32_t *d_mrows = to_dev(mrows), *d_mptr = to_dev(mptr);
  float *d_partial;
  CK(hipMalloc(&d_partial, std::max(ns, 1) * DD * 4));
  int32_t *d_arrive;
  CK(hipMalloc(&d_arrive, std::max(nm, 1) * 4));
  CK(hipMemset(d_arrive, 0, std::max(nm, 1) * 4));
  std::vector<float> h((size_t)N * DD), W(DD * DD), bvec(DD);
  srand(1);
  for (auto &x : h) x = (float)rand() / RAND_MAX - 0.5f;
  for (auto &x : W) x = ((float)rand() / RAND_MAX - 0.5f) * 0.3f;
  for (auto &x : bvec) x = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
  const float alpha = 0.25f, p_drop = 0.3f;
  const uint64_t seed = 0x1234567ABCDEFull;
  float *d_h = to_dev(h), *d_W = to_dev(W), *d_b = to_dev(bvec), *d_alpha;
  CK(hipMalloc(&d_alpha, 4));
  CK(hipMemcpy(d_alpha, &alpha, 4, hipMemcpyHostToDevice));
  float *d_x, *d_y, *d_xr, *d_yr;
  for (float **p : {&d_x, &d_y, &d_xr, &d_yr}) CK(hipMalloc(p, (size_t)N * DD * 4));

  rk_spmm_plan plan{};
  plan.long_thresh = 64;
  plan.n_chunks = nc;
  plan.chunk_row = d_crow, plan.chunk_begin = d_cbeg, plan.chunk_end = d_cend, plan.chunk_slot = d_cslot;
  plan.n_multi = nm, plan.multi_rows = d_mrows, plan.multi_slot_ptr = d_mptr, plan.n_slots = ns;
  plan.partial = d_partial;
  plan.arrive = d_arrive;
  PlanArgs pl{64, nc, nm, d_crow, d_cbeg, d_cend, d_cslot, d_mptr, (v4f *)d_partial, d_arrive};
  LayerArgs la{d_W, d_b, alpha, p_drop, (uint32_t)seed, (uint32_t)(seed >> 32), (v4f *)d_y};

  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto timeit = [&](const char *name, auto fn) {
    for (int i = 0; i < 3; ++i) fn();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) fn();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-58s %8.1f us\n", name, ms * 1e3 / reps);
    fflush(stdout);
  };
  auto check = [&](const char *what) {
    std::vector<float> a((size_t)N * DD), b((size_t)N * DD);
    CK(hipMemcpy(a.data(), d_x, a.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), d_xr, b.size() * 4, hipMemcpyDeviceToHost));
    const bool xbits = memcmp(a.data(), b.data(), a.size() * 4) == 0;
    CK(hipMemcpy(a.data(), d_y, a.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), d_yr, b.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    int64_t zero_mismatch = 0;
    for (size_t i = 0; i < a.size(); ++i) {
      worst = std::max(worst, (double)fabsf(a[i] - b[i]));
      scale = std::max(scale, (double)fabsf(b[i]));
      zero_mismatch += (a[i] == 0.0f) != (b[i] == 0.0f);
    }
    printf("  %s: X %s, Y max |diff| %.3g of scale %.3g (%.2e), dropout pattern mismatches %lld\n", what,
           xbits ? "bits=ref" : "BITS DIFFER", worst, scale, worst / scale, (long long)zero_mismatch);
  };

  auto lib_hop = [&]() {
    if (rk_spmm_csr_f32(d_rowptr, d_col, d_val, (int32_t)N, d_h, DD, d_xr, nullptr, 0.f, 1.f, &plan, nullptr)) {
      fprintf(stderr, "spmm failed: %s\n", rk_last_error());
      exit(1);
    }
  };
  auto lib_layer = [&]() {
    if (rk_dense_fwd_f32(d_xr, d_W, nullptr, nullptr, d_b, nullptr, d_alpha, p_drop, seed, nullptr, nullptr, N, DD, DD,
                         d_yr, nullptr)) {
      fprintf(stderr, "dense failed: %s\n", rk_last_error());
      exit(1);
    }
  };
  timeit("library hop", lib_hop);
  timeit("library layer", lib_layer);
  timeit("library hop + layer (two launches)", [&]() { lib_hop(); lib_layer(); });

#define RUN(PASSES, LAYER)                                                                                   \
  {                                                                                                          \
    CK(hipMemset(d_x, 0xFF, (size_t)N * DD * 4));                                                             \
    CK(hipMemset(d_y, 0xFF, (size_t)N * DD * 4));                                                             \
    const int32_t nb = nc + (int32_t)((N + 16 * PASSES - 1) / (16 * PASSES));                                \
    timeit(LAYER ? "fused hop+layer, " #PASSES " x 16 rows per workgroup" : "hop alone, " #PASSES " x 16 rows per workgroup", \
           [&]() { hop_layer_kernel<PASSES, LAYER><<<nb, 256>>>(d_rowptr, d_col, d_val, (int32_t)N, (const v4f *)d_h, \
                                                               (v4f *)d_x, pl, la); });                      \
    CK(hipDeviceSynchronize());                                                                              \
    if (LAYER) check("fused");                                                                               \
  }
  RUN(1, false)
  RUN(4, false)
  RUN(1, true)
  RUN(2, true)
  RUN(4, true)
  RUN(8, true)
  return 0;
}
