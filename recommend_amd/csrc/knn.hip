// Fused similarity + top-K over WIDE feature rows (gfx950): out[u] = the K largest of
// <A[u, :], B[i, :]> over i, for row widths of hundreds to thousands of floats, without ever
// holding a block of the score matrix.  Replaces the dense pipeline of the item-item kNN graph,
// model/sgrec/graph.py:100-105 (score = normalized_feat @ normalized_feat.T; torch.topk(score, k)).
//
// The evaluation kernel (topk_kernel.h) keeps a wavefront's 64 user rows in registers, which ends at
// 128 floats per row.  Here the feature width is a loop: a workgroup owns 256 A rows (64 per
// wavefront) x 128 B rows at a time and walks the width in chunks of 64 floats, accumulating the
// 64 x 128 scores of every wavefront in registers (v_mfma_f32_32x32x2_f32: exact fp32 products, the
// same k order per chunk as the evaluation kernel's score chain).  The B chunk (128 rows x 256 B)
// goes through double-buffered LDS shared by the four wavefronts, the A chunk of each wavefront
// comes straight from memory into registers one chunk ahead.  Only when a group of 128 B rows has
// seen the whole width are its scores offered to the per-lane sorted lists (one user per lane after
// a half swap, as in the evaluation kernel; ties go to the lower id).  Per 64-float chunk a
// workgroup loads 96 KB for 4.2 Mflop, i.e. the launch is bound by the fp32-input MFMA rate.
//
// The same kernel ranks users against items for embedding widths the evaluation kernel is not built
// for (rk_score_topk_f32 with D = 192, 256, ...): it then takes the train mask (CSR per A row, sorted;
// masked pairs score -1e10, evalue/evaluator.py:55) and the id offset of an item shard.
#include <algorithm>

#include "topk_kernel.h"
#include "topk_wide.h"

namespace {

constexpr int GT = 4;          // 32-row B tiles per group
constexpr int GI = GT * TI;    // B rows per group
constexpr int CH = 64;         // floats per chunk of the width
typedef float floatx4 __attribute__((ext_vector_type(4)));   // (arrays of HIP's float4 struct end up in scratch)
constexpr int LDB = CH + 4;    // LDS row pitch in floats (shifts consecutive rows by four banks)

template <int KPAD>
__global__ __launch_bounds__(256, 1) void wide_score_topk_kernel(
    const float *__restrict__ A, int32_t n_a, const float *__restrict__ B, int32_t n_b, int32_t F,
    const int32_t *__restrict__ mask_ptr, const int32_t *__restrict__ mask_items, int32_t id_offset,
    int32_t nsplit, float *__restrict__ part_s, int32_t *__restrict__ part_id) {
  __shared__ __attribute__((aligned(16))) float Bs[2][GI][LDB];

  const int tid = threadIdx.x;
  const int wave = tid / kWave, lane = tid % kWave;
  const int col = lane & 31, half = lane >> 5;
  const int32_t split = blockIdx.x % nsplit;
  const int32_t ubase = (blockIdx.x / nsplit) * UPB + wave * UPW;
  const int32_t user = ubase + lane;
  // every wavefront takes part in the workgroup barriers, also those past the end of A: their rows
  // are clamped to the last one and their lists never written
  const int64_t row_a = min(ubase + col, n_a - 1), row_b = min(ubase + 32 + col, n_a - 1);
  const float *pa = A + row_a * F + 32 * half, *pb = A + row_b * F + 32 * half;

  const int32_t groups_total = (n_b + GI - 1) / GI;
  const int32_t g_lo = (int32_t)((int64_t)groups_total * split / nsplit);
  const int32_t g_hi = (int32_t)((int64_t)groups_total * (split + 1) / nsplit);
  const int32_t n_chunks = F / CH;
  const int64_t steps = (int64_t)(g_hi - g_lo) * n_chunks;

  LaneList<KPAD> list;
  list.init();
  float thr = -INFINITY;

  // train-mask cursor of this lane's row of A (item ids ascending), at the first entry of this slice
  int32_t mcur = 0, mend = 0;
  if (mask_ptr != nullptr && user < n_a) {
    mcur = mask_ptr[user];
    mend = mask_ptr[user + 1];
    const int32_t first = g_lo * GI;
    int32_t lo = mcur, hi = mend;
    while (lo < hi) {
      const int32_t mid = (lo + hi) >> 1;
      if (mask_items[mid] < first)
        lo = mid + 1;
      else
        hi = mid;
    }
    mcur = lo;
  }
  int32_t next_m = mcur < mend ? mask_items[mcur] : INT32_MAX;

  // registers of the step in flight: this thread's 8 float4 of the next B chunk (row e / 16,
  // float4 e % 16 for e = tid + 256 i) and the next A chunk of the two user columns of this lane
  floatx4 bpre[8];
  float ua[32], ub[32], ua_n[32], ub_n[32];
  // where this thread's 8 float4 of a B chunk come from: fixed per group, advanced by the chunk
  const float *bp[8];
  auto set_group = [&](int32_t g) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + 256 * i;
      const int64_t row = min(g * GI + e / 16, n_b - 1);   // rows past the end repeat the last; discarded later
      bp[i] = B + row * F + (e % 16) * 4;
    }
  };
  auto fetch = [&](int32_t c) {
#pragma unroll
    for (int i = 0; i < 8; ++i) bpre[i] = *reinterpret_cast<const floatx4 *>(bp[i] + c * CH);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const floatx4 va = *reinterpret_cast<const floatx4 *>(pa + c * CH + 4 * q);
      const floatx4 vb = *reinterpret_cast<const floatx4 *>(pb + c * CH + 4 * q);
      ua_n[4 * q + 0] = va.x; ua_n[4 * q + 1] = va.y; ua_n[4 * q + 2] = va.z; ua_n[4 * q + 3] = va.w;
      ub_n[4 * q + 0] = vb.x; ub_n[4 * q + 1] = vb.y; ub_n[4 * q + 2] = vb.z; ub_n[4 * q + 3] = vb.w;
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + 256 * i;
      *reinterpret_cast<floatx4 *>(&Bs[buf][e / 16][(e % 16) * 4]) = bpre[i];
    }
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      ua[m] = ua_n[m];
      ub[m] = ub_n[m];
    }
  };

  if (steps > 0) {
    set_group(g_lo);
    fetch(0);
    stage(0);
  }
  __syncthreads();
  int64_t s = 0;
  for (int32_t g = g_lo; g < g_hi; ++g) {
    // (the accumulators are set up per group and carried only through the MFMAs of the chunk
    // loop: a zero that travels around the loop as well makes the compiler move all of them
    // between the two register files every step)
    floatx16 acc_a[GT], acc_b[GT];
#pragma unroll
    for (int j = 0; j < GT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc_a[j][r] = 0.0f;
        acc_b[j][r] = 0.0f;
      }
    for (int32_t c = 0; c < n_chunks; ++c, ++s) {
      const int buf = (int)(s & 1);
      const bool more = s + 1 < steps;     // uniform over the workgroup
      if (more) {
        if (c + 1 == n_chunks) set_group(g + 1);
        fetch(c + 1 < n_chunks ? c + 1 : 0);
      }
      // the fragment of tile j+1 is read from LDS while the MFMAs of tile j run
      floatx4 bf[2][8];
      auto read_frag = [&](int j) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
          bf[j & 1][q] = *reinterpret_cast<const floatx4 *>(&Bs[buf][TI * j + col][32 * half + 4 * q]);
      };
      read_frag(0);
#pragma unroll
      for (int j = 0; j < GT; ++j) {
        if (j + 1 < GT) read_frag(j + 1);
        // B rows on the MFMA rows, users on the columns: chain k = 0, 32, 1, 33, ... of this chunk
#pragma unroll
        for (int m = 0; m < 32; ++m) {
          acc_a[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j & 1][m / 4][m % 4], ua[m], acc_a[j], 0, 0, 0);
          acc_b[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j & 1][m / 4][m % 4], ub[m], acc_b[j], 0, 0, 0);
        }
      }
      if (more) stage(buf ^ 1);
      __syncthreads();
    }
    // the group has seen the whole width: offer its 128 scores per user, ascending ids
    uint32_t mbits[GT] = {0u, 0u, 0u, 0u};   // masked items of this lane's user among the group's 128
    while (next_m < (g + 1) * GI) {
      const int32_t rel = next_m - g * GI;
#pragma unroll
      for (int j = 0; j < GT; ++j) mbits[j] |= (rel >> 5) == j ? 1u << (rel & 31) : 0u;
      ++mcur;
      next_m = mcur < mend ? mask_items[mcur] : INT32_MAX;
    }
#pragma unroll
    for (int j = 0; j < GT; ++j) {
      // lane l holds rows (r&3)+8*(r>>2)+4*half of user col in acc_a and of user 32+col in
      // acc_b; after the swap lane l holds, for user ubase+l, rows (r&3)+8*(r>>2) in lo[r] and
      // rows 4+(r&3)+8*(r>>2) in hi[r]  (the compiler's builtin: it places the wait states behind
      // the last MFMA pass)
      float lo[16], hi[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        lo[r] = acc_a[j][r];
        hi[r] = acc_b[j][r];
        rk::swap_halves32(lo[r], hi[r]);
      }
      const int32_t item0 = (g * GT + j) * TI;
      if (item0 < n_b) {                 // uniform
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int32_t id = item0 + 8 * q + i;
            float v = i < 4 ? lo[4 * q + i] : hi[4 * q + i - 4];
            if ((mbits[j] >> (8 * q + i)) & 1u) v = kMasked;
            if (id >= n_b) v = -INFINITY;
            if (__any(v > thr)) {
              list.insert(v, id);
              thr = list.s[KPAD - 1];
            }
          }
        }
      }
    }
  }

  if (user < n_a) {
    const int64_t base = ((int64_t)user * nsplit + split) * KPAD;
#pragma unroll
    for (int j = 0; j < KPAD; ++j) {
      part_s[base + j] = list.s[j];
      part_id[base + j] = list.id[j] < 0 ? -1 : list.id[j] + id_offset;
    }
  }
}

// slices of the B rows per block of 256 A rows: enough workgroups to cover the CUs, whole groups each
inline int wide_nsplit(int32_t n_a, int32_t n_b) {
  const int64_t utiles = rk::ceil_div(n_a, UPB);
  const int64_t groups = rk::ceil_div(n_b, GI);
  const int64_t want = rk::ceil_div(cu_count(), utiles);
  return (int)std::max<int64_t>(1, std::min<int64_t>({want, groups, 16}));
}

}  // namespace

namespace rk {

size_t wide_part_entries(int32_t n_a, int32_t n_b, int32_t K) {
  return (size_t)n_a * wide_nsplit(n_a, n_b) * kpad_for(K);
}

int wide_topk(const float *A, int32_t n_a, const float *B, int32_t n_b, int32_t F, const int32_t *mask_ptr,
              const int32_t *mask_items, int32_t K, int32_t id_offset, float *out_scores, int32_t *out_ids,
              float *ps, hipStream_t st) {
  const int kp = kpad_for(K), nsplit = wide_nsplit(n_a, n_b);
  int32_t *pi = (int32_t *)(ps + (size_t)n_a * nsplit * kp);
  const int grid = (int)ceil_div(n_a, UPB) * nsplit;
#define RK_WIDE_CASE(KP)                                                                                           \
  case KP:                                                                                                         \
    wide_score_topk_kernel<KP><<<grid, 256, 0, st>>>(A, n_a, B, n_b, F, mask_ptr, mask_items, id_offset, nsplit, ps, pi); \
    break;
  switch (kp) {
    RK_WIDE_CASE(4) RK_WIDE_CASE(8) RK_WIDE_CASE(12) RK_WIDE_CASE(16) RK_WIDE_CASE(20) RK_WIDE_CASE(24) RK_WIDE_CASE(28)
    default:
      wide_score_topk_kernel<32><<<grid, 256, 0, st>>>(A, n_a, B, n_b, F, mask_ptr, mask_items, id_offset, nsplit, ps, pi);
      break;
  }
#undef RK_WIDE_CASE
  RK_CHECK_LAUNCH("wide_score_topk_kernel");
  return rk_topk_merge_f32(ps, pi, n_a, nsplit, kp, K, out_scores, out_ids, (void *)st);
}

}  // namespace rk

extern "C" {

size_t rk_score_topk_wide_ws_bytes(int32_t n_a, int32_t n_b, int32_t K) {
  if (n_a <= 0 || n_b <= 0 || K <= 0 || K > 32) return 0;
  return (size_t)n_a * wide_nsplit(n_a, n_b) * kpad_for(K) * (sizeof(float) + sizeof(int32_t));
}

int rk_score_topk_wide_f32(const float *A, int32_t n_a, const float *B, int32_t n_b, int32_t F, int32_t K,
                           float *out_scores, int32_t *out_ids, void *ws, size_t ws_bytes, void *stream) {
  RK_CHECK_ARG(n_a >= 0 && n_b > 0 && F > 0, "bad shape");
  RK_CHECK_ARG(K >= 1 && K <= n_b, "k must be in the range [1, num_items]");
  if (K > 32) return rk::fail(RK_ERR_UNSUPPORTED, "rk_score_topk_wide_f32: K=%d above the built maximum 32", K);
  if (F % CH != 0)
    return rk::fail(RK_ERR_UNSUPPORTED, "rk_score_topk_wide_f32: row width %d is not a multiple of %d (pad with zero columns)", F, CH);
  if (n_a == 0) return RK_OK;
  RK_CHECK_ARG(A && B && out_scores && out_ids && ws, "null pointer");
  RK_CHECK_ARG(((uintptr_t)A | (uintptr_t)B) % 16 == 0, "rows must be 16-byte aligned");
  RK_CHECK_ARG(ws_bytes >= rk_score_topk_wide_ws_bytes(n_a, n_b, K), "workspace too small");
  return rk::wide_topk(A, n_a, B, n_b, F, nullptr, nullptr, K, 0, out_scores, out_ids, (float *)ws, (hipStream_t)stream);
}

}  // extern "C"
