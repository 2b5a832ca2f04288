// SpMM kernel laboratory: times variants of the hop kernel on a dumped CSR (tools/dump_csr.py)
// against the library's rk_spmm_csr_f32 and checks every variant's output bit for bit.
//   hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=off \
//         tools/spmm_lab.hip -o tools/_variants/spmm_lab -Lrecommend_amd/lib -lrecommend_amd \
//         -Wl,-rpath,'$ORIGIN/../../recommend_amd/lib'
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <numeric>
#include <string>
#include <utility>
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

struct Seg {
  int32_t r0, r1;       // rows of the segment
  int32_t c0, n_long;   // its long-tier chunks
  int32_t n_work;       // n_long + short row blocks
  int32_t n_hot;
  const int32_t *hot_cols;
};
struct SegArgs {
  int32_t n_seg;
  Seg seg[4];
  const int32_t *chunk_row, *chunk_begin, *chunk_end, *chunk_slot;
  v4f *partial4;
  int32_t long_thresh;
};

template <int LPR, int J>
__device__ __forceinline__ int bcast_i(int v) {
  if constexpr (LPR == 16)
    return __builtin_amdgcn_update_dpp(0, v, 0x150 + J, 0xF, 0xF, false);
  else
    return __shfl(v, J, LPR);
}
template <int LPR, int J>
__device__ __forceinline__ float bcast_f(float v) {
  return __builtin_bit_cast(float, bcast_i<LPR, J>(__builtin_bit_cast(int, v)));
}

__device__ __forceinline__ void fma4(v4f &a, float v, const v4f &x) {
  a.x = fmaf(v, x.x, a.x);
  a.y = fmaf(v, x.y, a.y);
  a.z = fmaf(v, x.z, a.z);
  a.w = fmaf(v, x.w, a.w);
}

template <int LPR, int HOT, int MODE>
__device__ __forceinline__ v4f load_row(const v4f *__restrict__ h4, const v4f *hot, int32_t c, int sub) {
  if constexpr (HOT == 0) {
    return h4[(int64_t)c * LPR + sub];
  } else if constexpr (MODE == 1) {   // one flat load: the address decides between LDS and memory
    const v4f *p = c < 0 ? hot + (~c) * LPR + sub : h4 + (int64_t)c * LPR + sub;
    return *p;
  } else {
    if (c < 0) return hot[(~c) * LPR + sub];
    return h4[(int64_t)c * LPR + sub];
  }
}

template <int LPR, int HOT, int MODE, int Q>
__device__ __forceinline__ void gather4(const v4f *__restrict__ h4, const v4f *hot, int32_t c_l, float v_l,
                                        int sub, v4f &acc) {
  const int32_t c0 = bcast_i<LPR, Q>(c_l), c1 = bcast_i<LPR, Q + 1>(c_l);
  const int32_t c2 = bcast_i<LPR, Q + 2>(c_l), c3 = bcast_i<LPR, Q + 3>(c_l);
  const v4f x0 = load_row<LPR, HOT, MODE>(h4, hot, c0, sub);
  const v4f x1 = load_row<LPR, HOT, MODE>(h4, hot, c1, sub);
  const v4f x2 = load_row<LPR, HOT, MODE>(h4, hot, c2, sub);
  const v4f x3 = load_row<LPR, HOT, MODE>(h4, hot, c3, sub);
  fma4(acc, bcast_f<LPR, Q>(v_l), x0);
  fma4(acc, bcast_f<LPR, Q + 1>(v_l), x1);
  fma4(acc, bcast_f<LPR, Q + 2>(v_l), x2);
  fma4(acc, bcast_f<LPR, Q + 3>(v_l), x3);
}

template <int LPR, int HOT, int MODE, int Q>
__device__ __forceinline__ void batch_from(const v4f *__restrict__ h4, const v4f *hot, int32_t c_l, float v_l,
                                           int sub, int cnt, v4f &acc) {
  if constexpr (Q < LPR) {
    if (Q < cnt) {
      gather4<LPR, HOT, MODE, Q>(h4, hot, c_l, v_l, sub, acc);
      batch_from<LPR, HOT, MODE, Q + 4>(h4, hot, c_l, v_l, sub, cnt, acc);
    }
  }
}

// G gathers in flight: all loads of a step are issued before its first fma
template <int LPR, int G, int Q>
__device__ __forceinline__ void gatherN(const v4f *__restrict__ h4, int32_t c_l, float v_l, int sub, v4f &acc) {
  v4f x[G];
  int32_t c[G];
#pragma unroll
  for (int j = 0; j < G; ++j) c[j] = 0;
  [&]<int... J>(std::integer_sequence<int, J...>) {
    ((c[J] = bcast_i<LPR, Q + J>(c_l)), ...);
  }(std::make_integer_sequence<int, G>{});
#pragma unroll
  for (int j = 0; j < G; ++j) x[j] = h4[(int64_t)c[j] * LPR + sub];
  [&]<int... J>(std::integer_sequence<int, J...>) {
    ((fma4(acc, bcast_f<LPR, Q + J>(v_l), x[J])), ...);
  }(std::make_integer_sequence<int, G>{});
}
template <int LPR, int G, int Q>
__device__ __forceinline__ void batch_fromN(const v4f *__restrict__ h4, int32_t c_l, float v_l, int sub, int cnt,
                                            v4f &acc) {
  if constexpr (Q < LPR) {
    if (Q < cnt) {
      gatherN<LPR, G, Q>(h4, c_l, v_l, sub, acc);
      batch_fromN<LPR, G, Q + G>(h4, c_l, v_l, sub, cnt, acc);
    }
  }
}
template <int LPR, int G>
__device__ __forceinline__ void accumulate_edgesN(const int32_t *__restrict__ col, const float *__restrict__ val,
                                                  const v4f *__restrict__ h4, int32_t b, int32_t e, int sub,
                                                  v4f &acc) {
  for (int32_t e0 = b; e0 < e; e0 += LPR) {
    const int cnt = min(LPR, e - e0);
    const int32_t mine = e0 + min(sub, cnt - 1);
    const int32_t c_l = col[mine];
    float v_l = val[mine];
    if (sub >= cnt) v_l = 0.0f;
    batch_fromN<LPR, G, 0>(h4, c_l, v_l, sub, cnt, acc);
  }
}

template <int LPR, int HOT, int MODE, bool NT>
__device__ __forceinline__ void accumulate_edges(const int32_t *__restrict__ col, const float *__restrict__ val,
                                                 const v4f *__restrict__ h4, const v4f *hot, int32_t b,
                                                 int32_t e, int sub, v4f &acc) {
  for (int32_t e0 = b; e0 < e; e0 += LPR) {
    const int cnt = min(LPR, e - e0);
    const int32_t mine = e0 + min(sub, cnt - 1);
    const int32_t c_l = NT ? __builtin_nontemporal_load(col + mine) : col[mine];
    float v_l = NT ? __builtin_nontemporal_load(val + mine) : val[mine];
    if (sub >= cnt) v_l = 0.0f;
    batch_from<LPR, HOT, MODE, 0>(h4, hot, c_l, v_l, sub, cnt, acc);
  }
}

template <bool NT>
__device__ __forceinline__ void store4(v4f *p, v4f v) {
  if (NT)
    __builtin_nontemporal_store(v, p);
  else
    *p = v;
}

// MODE: 0 = branch between LDS and memory, 1 = flat load.  HOT = cached rows per segment (0: none)
template <int LPR, int HOT, int MODE, bool NT>
__global__ __launch_bounds__(256) void spmm_seg_kernel(const int32_t *__restrict__ rowptr,
                                                       const int32_t *__restrict__ col,
                                                       const float *__restrict__ val,
                                                       const v4f *__restrict__ h4, v4f *__restrict__ out4,
                                                       SegArgs sa) {
  constexpr int GROUPS = 256 / LPR;
  __shared__ v4f red[GROUPS][LPR];
  __shared__ v4f hot[HOT > 0 ? HOT * LPR : 1];
  const int tid = threadIdx.x;
  const int sub = tid % LPR, grp = tid / LPR;
  for (int s = 0; s < sa.n_seg; ++s) {
    const Seg sg = sa.seg[s];
    if constexpr (HOT > 0) {
      if (s > 0) __syncthreads();
      for (int i = tid; i < sg.n_hot * LPR; i += 256) hot[i] = h4[(int64_t)sg.hot_cols[i / LPR] * LPR + i % LPR];
      __syncthreads();
    }
    for (int32_t w = blockIdx.x; w < sg.n_work; w += gridDim.x) {
      if (w >= sg.n_long) {
        const int32_t row = sg.r0 + (w - sg.n_long) * GROUPS + grp;
        if (row < sg.r1) {
          const int32_t b = rowptr[row], e = rowptr[row + 1];
          if (e - b <= sa.long_thresh) {
            v4f acc = {0.f, 0.f, 0.f, 0.f};
            accumulate_edges<LPR, HOT, MODE, NT>(col, val, h4, hot, b, e, sub, acc);
            store4<NT>(out4 + (int64_t)row * LPR + sub, acc);
          }
        }
      } else {
        const int32_t chunk = sg.c0 + w;
        const int32_t cb = sa.chunk_begin[chunk], ce = sa.chunk_end[chunk];
        const int32_t span = ((ce - cb + GROUPS - 1) / GROUPS + LPR - 1) / LPR * LPR;
        const int32_t b = min(cb + grp * span, ce), e = min(b + span, ce);
        v4f acc = {0.f, 0.f, 0.f, 0.f};
        accumulate_edges<LPR, HOT, MODE, NT>(col, val, h4, hot, b, e, sub, acc);
        red[grp][sub] = acc;
        __syncthreads();
        if (grp == 0) {
          v4f t = red[0][sub];
#pragma unroll
          for (int g = 1; g < GROUPS; ++g) t += red[g][sub];
          const int32_t slot = sa.chunk_slot[chunk];
          if (slot < 0)
            store4<NT>(out4 + (int64_t)sa.chunk_row[chunk] * LPR + sub, t);
          else
            sa.partial4[(int64_t)slot * LPR + sub] = t;
        }
        __syncthreads();
      }
    }
  }
}

// non-persistent form of the library kernel with a row processing order and a store policy
// ST: 0 plain, 1 sc1 (write-through, line not kept in L2), 2 nt
template <int ST>
__device__ __forceinline__ void store_pol(v4f *p, v4f v) {
  if constexpr (ST == 1)
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else if constexpr (ST == 2)
    __builtin_nontemporal_store(v, p);
  else
    *p = v;
}
template <int LPR, int ST, bool PERM, int G = 4>
__global__ __launch_bounds__(256) void spmm_np_kernel(const int32_t *__restrict__ rowptr,
                                                      const int32_t *__restrict__ col,
                                                      const float *__restrict__ val, int32_t n_rows,
                                                      const int32_t *__restrict__ order,
                                                      const v4f *__restrict__ h4, v4f *__restrict__ out4,
                                                      SegArgs sa, int32_t n_chunks) {
  constexpr int GROUPS = 256 / LPR;
  __shared__ v4f red[GROUPS][LPR];
  const int tid = threadIdx.x;
  const int sub = tid % LPR, grp = tid / LPR;
  if ((int32_t)blockIdx.x >= n_chunks) {
    const int64_t idx = (int64_t)((int32_t)blockIdx.x - n_chunks) * GROUPS + grp;
    if (idx >= n_rows) return;
    const int32_t row = PERM ? order[idx] : (int32_t)idx;
    const int32_t b = rowptr[row], e = rowptr[row + 1];
    if (e - b > sa.long_thresh) return;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    accumulate_edgesN<LPR, G>(col, val, h4, b, e, sub, acc);
    store_pol<ST>(out4 + (int64_t)row * LPR + sub, acc);
    return;
  }
  const int32_t chunk = blockIdx.x;
  const int32_t cb = sa.chunk_begin[chunk], ce = sa.chunk_end[chunk];
  const int32_t span = ((ce - cb + GROUPS - 1) / GROUPS + LPR - 1) / LPR * LPR;
  const int32_t b = min(cb + grp * span, ce), e = min(b + span, ce);
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  accumulate_edgesN<LPR, G>(col, val, h4, b, e, sub, acc);
  red[grp][sub] = acc;
  __syncthreads();
  if (grp == 0) {
    v4f t = red[0][sub];
#pragma unroll
    for (int g = 1; g < GROUPS; ++g) t += red[g][sub];
    const int32_t slot = sa.chunk_slot[chunk];
    if (slot < 0)
      store_pol<ST>(out4 + (int64_t)sa.chunk_row[chunk] * LPR + sub, t);
    else
      sa.partial4[(int64_t)slot * LPR + sub] = t;
  }
}

// one wavefront per row that was cut into several chunks (as in the library)
template <int LPR>
__global__ __launch_bounds__(256) void multi_reduce_kernel(const int32_t *__restrict__ multi_rows,
                                                           const int32_t *__restrict__ multi_slot_ptr,
                                                           int32_t n_multi, const v4f *__restrict__ partial4,
                                                           v4f *__restrict__ out4) {
  constexpr int GPW = 64 / LPR;
  const int lane = threadIdx.x % 64;
  const int sub = lane % LPR, g = lane / LPR;
  const int32_t i = blockIdx.x * 4 + threadIdx.x / 64;
  if (i >= n_multi) return;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  for (int32_t s = multi_slot_ptr[i] + g; s < multi_slot_ptr[i + 1]; s += GPW) acc += partial4[(int64_t)s * LPR + sub];
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off, 64);
    acc.y += __shfl_xor(acc.y, off, 64);
    acc.z += __shfl_xor(acc.z, off, 64);
    acc.w += __shfl_xor(acc.w, off, 64);
  }
  if (g == 0) out4[(int64_t)multi_rows[i] * LPR + sub] = acc;
}

template <class T>
T *to_dev(const std::vector<T> &v) {
  T *p = nullptr;
  CK(hipMalloc(&p, std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (!v.empty()) CK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return p;
}

struct Graph {
  int64_t N, nnz, U;
  std::vector<int32_t> rowptr, col;
  std::vector<float> val;
};

struct HotEnc {
  int32_t *col_dev;
  int32_t *hot_dev[2];
  int32_t n_hot[2];
  double covered[2];
};

// per segment: the `hot` most frequent columns; edges to them are re-coded as ~slot
HotEnc encode_hot(const Graph &g, int hot, bool by_segment) {
  HotEnc he{};
  std::vector<int32_t> enc = g.col;
  const int64_t bounds[3] = {0, by_segment ? g.U : g.N, g.N};
  for (int s = 0; s < 2; ++s) {
    const int64_t r0 = bounds[s], r1 = bounds[s + 1];
    std::vector<int32_t> hot_cols;
    he.covered[s] = 0;
    if (hot > 0 && r1 > r0) {
      std::vector<int32_t> cnt(g.N, 0);
      for (int64_t p = g.rowptr[r0]; p < g.rowptr[r1]; ++p) cnt[g.col[p]]++;
      std::vector<int32_t> order(g.N);
      std::iota(order.begin(), order.end(), 0);
      const int k = (int)std::min<int64_t>(hot, g.N);
      std::partial_sort(order.begin(), order.begin() + k, order.end(),
                        [&](int a, int b) { return cnt[a] != cnt[b] ? cnt[a] > cnt[b] : a < b; });
      std::vector<int32_t> slot(g.N, -1);
      for (int i = 0; i < k; ++i)
        if (cnt[order[i]] > 1) {
          slot[order[i]] = (int32_t)hot_cols.size();
          hot_cols.push_back(order[i]);
        }
      int64_t hit = 0;
      for (int64_t p = g.rowptr[r0]; p < g.rowptr[r1]; ++p)
        if (slot[g.col[p]] >= 0) {
          enc[p] = ~slot[g.col[p]];
          ++hit;
        }
      he.covered[s] = (double)hit / std::max<int64_t>(1, g.rowptr[r1] - g.rowptr[r0]);
    }
    he.n_hot[s] = (int32_t)hot_cols.size();
    he.hot_dev[s] = to_dev(hot_cols);
  }
  he.col_dev = to_dev(enc);
  return he;
}

int main(int argc, char **argv) {
  const char *path = argc > 1 ? argv[1] : "tools/_variants/elec_csr.bin";
  const int reps = argc > 2 ? atoi(argv[2]) : 100;
  FILE *f = fopen(path, "rb");
  if (!f) {
    fprintf(stderr, "cannot open %s\n", path);
    return 1;
  }
  Graph g;
  int64_t hdr[3];
  if (fread(hdr, 8, 3, f) != 3) return 1;
  g.N = hdr[0], g.nnz = hdr[1], g.U = hdr[2];
  g.rowptr.resize(g.N + 1);
  g.col.resize(g.nnz);
  g.val.resize(g.nnz);
  if (fread(g.rowptr.data(), 4, g.N + 1, f) != (size_t)g.N + 1) return 1;
  if (fread(g.col.data(), 4, g.nnz, f) != (size_t)g.nnz) return 1;
  if (fread(g.val.data(), 4, g.nnz, f) != (size_t)g.nnz) return 1;
  fclose(f);
  if (const char *cm = getenv("LAB_COLMOD")) {   // calibration: fold every column into a table of n rows
    const int n = atoi(cm);
    for (auto &c : g.col) c %= n;
    printf("columns folded modulo %d (%.1f MB table)\n", n, n * 256e-6);
  }
  const bool quick = getenv("LAB_QUICK") != nullptr;
  constexpr int D = 64, LPR = 16;
  printf("graph: N %lld nnz %lld U %lld\n", (long long)g.N, (long long)g.nnz, (long long)g.U);

  // plan (library)
  int32_t nc = 0, nm = 0, ns = 0;
  rk_spmm_plan_host(g.rowptr.data(), (int32_t)g.N, 64, &nc, &nm, &ns, 0, 0, 0, 0, 0, 0);
  std::vector<int32_t> crow(std::max(nc, 1)), cbeg(std::max(nc, 1)), cend(std::max(nc, 1)), cslot(std::max(nc, 1)),
      mrows(std::max(nm, 1)), mptr(nm + 1);
  rk_spmm_plan_host(g.rowptr.data(), (int32_t)g.N, 64, &nc, &nm, &ns, crow.data(), cbeg.data(), cend.data(),
                    cslot.data(), mrows.data(), mptr.data());
  printf("plan: %d chunks, %d multi rows, %d slots\n", nc, nm, ns);
  int32_t c_split = 0;
  while (c_split < nc && crow[c_split] < g.U) ++c_split;

  int32_t *d_rowptr = to_dev(g.rowptr), *d_col = to_dev(g.col);
  float *d_val = to_dev(g.val);
  int32_t *d_crow = to_dev(crow), *d_cbeg = to_dev(cbeg), *d_cend = to_dev(cend), *d_cslot = to_dev(cslot);
  int32_t *d_mrows = to_dev(mrows), *d_mptr = to_dev(mptr);
  float *d_partial;
  CK(hipMalloc(&d_partial, std::max(ns, 1) * D * 4));
  std::vector<float> h((size_t)g.N * D);
  srand(1);
  for (auto &x : h) x = (float)rand() / RAND_MAX - 0.5f;
  float *d_h = to_dev(h), *d_out, *d_ref;
  CK(hipMalloc(&d_out, (size_t)g.N * D * 4));
  CK(hipMalloc(&d_ref, (size_t)g.N * D * 4));

  rk_spmm_plan plan{};
  plan.long_thresh = 64;
  plan.n_chunks = nc;
  plan.chunk_row = d_crow, plan.chunk_begin = d_cbeg, plan.chunk_end = d_cend, plan.chunk_slot = d_cslot;
  plan.n_multi = nm, plan.multi_rows = d_mrows, plan.multi_slot_ptr = d_mptr, plan.n_slots = ns;
  plan.partial = d_partial;
  int32_t *d_arrive;
  CK(hipMalloc(&d_arrive, std::max(nm, 1) * 4));
  CK(hipMemset(d_arrive, 0, std::max(nm, 1) * 4));
  plan.arrive = d_arrive;

  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double bytes = g.nnz * 8.0 + (g.N + 1) * 4.0 + 2.0 * g.N * D * 4;
  std::vector<float> ref((size_t)g.N * D), got((size_t)g.N * D);

  auto timeit = [&](const char *name, auto fn, bool check) {
    CK(hipMemset(d_out, 0xFF, (size_t)g.N * D * 4));
    for (int i = 0; i < 3; ++i) fn();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) fn();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    const char *verdict = "";
    if (check) {
      CK(hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost));
      verdict = memcmp(got.data(), ref.data(), got.size() * 4) == 0 ? "bits=ref" : "BITS DIFFER";
    }
    printf("%-44s %8.1f us  %7.0f GB/s alg  frac %.3f  %s\n", name, us, bytes / us / 1e3, bytes / us / 1e3 / 8000.0,
           verdict);
    fflush(stdout);
  };

  auto base = [&]() {
    int rc = rk_spmm_csr_f32(d_rowptr, d_col, d_val, (int32_t)g.N, d_h, D, d_out, nullptr, 0.f, 1.f, &plan, nullptr);
    if (rc) {
      fprintf(stderr, "baseline failed: %s\n", rk_last_error());
      exit(1);
    }
  };
  timeit("library rk_spmm_csr_f32", base, false);
  CK(hipMemcpy(ref.data(), d_out, ref.size() * 4, hipMemcpyDeviceToHost));

  auto make_args = [&](const HotEnc &he, int which /*0 both, 1 users, 2 items*/) {
    SegArgs sa{};
    sa.chunk_row = d_crow, sa.chunk_begin = d_cbeg, sa.chunk_end = d_cend, sa.chunk_slot = d_cslot;
    sa.partial4 = (v4f *)d_partial;
    sa.long_thresh = 64;
    const int64_t bounds[3] = {0, g.U, g.N};
    const int32_t cb[3] = {0, c_split, nc};
    for (int s = 0; s < 2; ++s) {
      if (which == 1 && s == 1) continue;
      if (which == 2 && s == 0) continue;
      Seg &sg = sa.seg[sa.n_seg++];
      sg.r0 = (int32_t)bounds[s], sg.r1 = (int32_t)bounds[s + 1];
      sg.c0 = cb[s], sg.n_long = cb[s + 1] - cb[s];
      sg.n_work = sg.n_long + (int32_t)((sg.r1 - sg.r0 + 15) / 16);
      sg.n_hot = he.n_hot[s], sg.hot_cols = he.hot_dev[s];
    }
    return sa;
  };
  auto multi = [&]() {
    if (nm > 0)
      multi_reduce_kernel<LPR><<<(nm + 3) / 4, 256>>>(d_mrows, d_mptr, nm, (const v4f *)d_partial, (v4f *)d_out);
  };

#define RUN(HOT, MODE, NT, WGPC, WHICH, HE)                                                              \
  {                                                                                                      \
    SegArgs sa = make_args(HE, WHICH);                                                                   \
    char nm_[128];                                                                                       \
    snprintf(nm_, sizeof nm_, "seg hot=%d mode=%d nt=%d wg/cu=%d rows=%s", HOT, MODE, (int)NT, WGPC,     \
             WHICH == 0 ? "all" : WHICH == 1 ? "users" : "items");                                       \
    timeit(nm_, [&]() {                                                                                  \
      spmm_seg_kernel<LPR, HOT, MODE, NT><<<256 * WGPC, 256>>>(d_rowptr, HE.col_dev, d_val, (const v4f *)d_h, \
                                                                 (v4f *)d_out, sa);                      \
      if (WHICH == 0) multi();                                                                           \
    }, WHICH == 0);                                                                                      \
  }

  if (quick) return 0;
  HotEnc h0 = encode_hot(g, 0, true);

  {
    SegArgs sa = make_args(h0, 0);
    const int32_t nb = nc + (int32_t)((g.N + 15) / 16);
    // processing orders
    std::vector<int32_t> len(g.N), ident(g.N), by_len(g.N), by_rare(g.N);
    for (int64_t r = 0; r < g.N; ++r) len[r] = g.rowptr[r + 1] - g.rowptr[r];
    std::iota(ident.begin(), ident.end(), 0);
    by_len = ident;
    std::stable_sort(by_len.begin(), by_len.begin() + g.U, [&](int a, int b) { return len[a] > len[b]; });
    std::stable_sort(by_len.begin() + g.U, by_len.end(), [&](int a, int b) { return len[a] > len[b]; });
    // key of a row: its least frequent column (rows sharing a rare source row become neighbours)
    std::vector<int32_t> freq(g.N, 0), key(g.N);
    for (auto c : g.col) freq[c]++;
    for (int64_t r = 0; r < g.N; ++r) {
      int32_t best = -1;
      for (int32_t p = g.rowptr[r]; p < g.rowptr[r + 1]; ++p) {
        const int32_t c = g.col[p];
        if (c == r) continue;
        if (best < 0 || freq[c] < freq[best] || (freq[c] == freq[best] && c < best)) best = c;
      }
      key[r] = best < 0 ? (int32_t)r : best;
    }
    by_rare = ident;
    std::stable_sort(by_rare.begin(), by_rare.begin() + g.U, [&](int a, int b) { return key[a] < key[b]; });
    std::stable_sort(by_rare.begin() + g.U, by_rare.end(), [&](int a, int b) { return key[a] < key[b]; });
    int32_t *d_ident = to_dev(ident), *d_len = to_dev(by_len), *d_rare = to_dev(by_rare);
#define RUN_NP(ST, PERM, ORD, LABEL)                                                                     \
    timeit("np st=" #ST " order=" LABEL, [&]() {                                                         \
      spmm_np_kernel<LPR, ST, PERM><<<nb, 256>>>(d_rowptr, d_col, d_val, (int32_t)g.N, ORD, (const v4f *)d_h, \
                                                  (v4f *)d_out, sa, nc);                                 \
      multi();                                                                                           \
    }, true);
    RUN_NP(0, false, d_ident, "natural")
    RUN_NP(1, false, d_ident, "natural")
    RUN_NP(2, false, d_ident, "natural")
#define RUN_G(G)                                                                                          \
    timeit("np gathers in flight " #G, [&]() {                                                           \
      spmm_np_kernel<LPR, 0, false, G><<<nb, 256>>>(d_rowptr, d_col, d_val, (int32_t)g.N, d_ident, (const v4f *)d_h, \
                                                  (v4f *)d_out, sa, nc);                                 \
      multi();                                                                                           \
    }, true);
    RUN_G(2) RUN_G(4) RUN_G(8) RUN_G(16)
    if (getenv("LAB_G_ONLY")) return 0;
    RUN_NP(0, true, d_ident, "identity-array")
    RUN_NP(0, true, d_len, "by-length")
    RUN_NP(1, true, d_len, "by-length")
    RUN_NP(0, true, d_rare, "by-rarest-column")
    RUN_NP(1, true, d_rare, "by-rarest-column")
  }
  if (getenv("LAB_NP_ONLY")) return 0;
  RUN(0, 0, false, 8, 0, h0)
  RUN(0, 0, true, 8, 0, h0)
  RUN(0, 0, false, 8, 1, h0)
  RUN(0, 0, false, 8, 2, h0)
  RUN(0, 0, true, 8, 1, h0)
  RUN(0, 0, true, 8, 2, h0)
  {
    HotEnc he = encode_hot(g, 64, true);
    printf("hot 64: coverage users-rows %.3f item-rows %.3f\n", he.covered[0], he.covered[1]);
    RUN(64, 0, false, 8, 0, he)
    RUN(64, 0, true, 8, 0, he)
    RUN(64, 1, true, 8, 0, he)
    RUN(64, 0, true, 8, 1, he)
    RUN(64, 0, true, 8, 2, he)
  }
  {
    HotEnc he = encode_hot(g, 128, true);
    printf("hot 128: coverage users-rows %.3f item-rows %.3f\n", he.covered[0], he.covered[1]);
    RUN(128, 0, true, 4, 0, he)
    RUN(128, 1, true, 4, 0, he)
    RUN(128, 0, true, 4, 1, he)
    RUN(128, 0, true, 4, 2, he)
  }
  {
    HotEnc he = encode_hot(g, 256, true);
    printf("hot 256: coverage users-rows %.3f item-rows %.3f\n", he.covered[0], he.covered[1]);
    RUN(256, 0, true, 2, 0, he)
    RUN(256, 1, true, 2, 0, he)
  }
  {
    HotEnc he = encode_hot(g, 512, true);
    printf("hot 512: coverage users-rows %.3f item-rows %.3f\n", he.covered[0], he.covered[1]);
    RUN(512, 0, true, 1, 0, he)
    RUN(512, 1, true, 1, 0, he)
  }
  return 0;
}
