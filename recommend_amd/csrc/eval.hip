// Full-ranking evaluation: fused score (fp32 MFMA) + train-mask + per-lane
// top-K that never writes the user x item score matrix, a list merge, and the
// ranking-metric reductions (gfx950).
//
// score_topk_kernel.  Every wavefront works alone: it owns 64 users and walks one
// slice of the item range in tiles of 32 items.  It stages its own tiles by LDS-DMA
// (global_load_lds, 1 KiB per instruction) into a private double buffer whose 16-byte
// chunks are XOR-swizzled with the row number (conflict-free b128 fragment reads), so
// the main loop has no workgroup barrier: the only wait on a tile is the wavefront's
// own vmcnt.  Per tile it runs two chains of v_mfma_f32_32x32x2_f32 (items on the MFMA
// rows, users 0-31 and users 32-63 on the columns).  A 32x32 result leaves each user's
// 32 scores split over lanes l and l+32; one v_permlane32_swap per register pair
// gives lane l all 32 scores of user l and lane l+32 all 32 scores of user 32+l.  So
// the whole top-K state of a user (a sorted list of KPAD scores and ids) lives in ONE
// lane's registers, and there is one partial list per user and item slice.
//
// A score enters a lane's list only if it beats the lane's current KPAD-th best.
// Candidates are parked in an 8-deep per-lane LDS queue and the unrolled sorted
// insertion runs only when some lane's queue is nearly full, all lanes inserting
// together.  Every lane sees its items in ascending id order and insertion uses a
// strict '>', so equal scores keep the lower item id first; the merge of the
// per-slice lists orders by (score desc, id asc) explicitly.
#include <cstdlib>
#include "topk_filter.h"
#include "topk_wide.h"
#include "topk_kernel.h"

namespace {

// Top-K of the rows of a score matrix that already exists (evaluate_all_at_k called with
// precomputed scores).  32 lanes per row, lane j scans columns j, j+32, ... (coalesced 128-B
// reads) into its own sorted list; the 32 lists of a row are merged by topk_merge_kernel.
constexpr int ROWS_LANES = 32;

// CEIL: a later page of a ranking longer than one list (see score_topk_kernel): only columns that
// rank strictly after (ceil_s[row], ceil_id[row]) take part.
template <int KPAD, bool CEIL = false>
__global__ __launch_bounds__(256) void rows_topk_kernel(const float *__restrict__ scores, int32_t n_rows,
                                                        int32_t n_items, int64_t ld,
                                                        float *__restrict__ part_s,
                                                        int32_t *__restrict__ part_id,
                                                        const float *__restrict__ ceil_s = nullptr,
                                                        const int32_t *__restrict__ ceil_id = nullptr) {
  const int sub = threadIdx.x % ROWS_LANES;
  const int32_t row = blockIdx.x * (256 / ROWS_LANES) + threadIdx.x / ROWS_LANES;
  if (row >= n_rows) return;
  LaneList<KPAD> list;
  list.init();
  const float *src = scores + (int64_t)row * ld;
  float page_s = INFINITY;
  int32_t page_id = -1;
  if constexpr (CEIL) {
    page_s = ceil_s[row];
    page_id = ceil_id[row];
  }
  for (int32_t i = sub; i < n_items; i += ROWS_LANES) {
    const float v = src[i];
    if constexpr (CEIL) {
      if (v > page_s || (v == page_s && i <= page_id)) continue;
    }
    if (v > list.s[KPAD - 1]) list.insert(v, i);   // ascending ids + strict '>' keep lower ids first
  }
  const int64_t base = ((int64_t)row * ROWS_LANES + sub) * KPAD;
#pragma unroll
  for (int j = 0; j < KPAD; ++j) {
    part_s[base + j] = list.s[j];
    part_id[base + j] = list.id[j];
  }
}

// One wavefront per user ranks all n = P*Kin candidates by counting how many
// beat each one; candidates with rank < K are written to their final slot.
constexpr int MERGE_MAX = 1024;

__device__ __forceinline__ bool beats(float sa, int32_t ia, int ja, float sb, int32_t ib, int jb) {
  const bool va = ia >= 0, vb = ib >= 0;
  if (va != vb) return va;
  if (!va) return ja < jb;
  return sa > sb || (sa == sb && (ia < ib || (ia == ib && ja < jb)));
}

// Which rows a merge launch works on (user_list / user_count / gate_cap as for
// score_topk_kernel: with a list, row `row_in` is a position of the compacted launch and stands
// for user user_list[row_in]); returns false if this wavefront has nothing to do.
__device__ __forceinline__ bool merge_row(int32_t n_rows, const int32_t *skip, const int32_t *user_list,
                                          const int32_t *user_count, int32_t gate_cap, int32_t row_in,
                                          int32_t &row_out) {
  const bool listed = user_list != nullptr;
  if (user_count != nullptr) {
    const int32_t cnt = user_count[0];
    if (listed ? cnt > gate_cap : cnt <= gate_cap) return false;
    if (listed) n_rows = min(n_rows, cnt);
  }
  if (row_in >= n_rows) return false;  // uniform over the lanes that share a row
  row_out = listed ? user_list[row_in] : row_in;
  return skip == nullptr || skip[row_out] == 0;   // flagged rows of out are already final
}

// Candidates of the four rows of a workgroup: scores and ids of row w at [2 * n * w, ...).  Sized by
// the launch (n entries per row, not MERGE_MAX): a merge lives on the number of rows in flight.
extern __shared__ float merge_lds[];

// One wavefront per row, four rows per workgroup.  Workgroups walk the rows with a grid stride, so
// a launch that is gated off on the device does not pay for one workgroup per four rows of the table.
__device__ __forceinline__ void merge_sync() {
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS accesses are done
}

__global__ __launch_bounds__(256) void topk_merge_kernel(
    const float *__restrict__ scores, const int32_t *__restrict__ ids,
    int32_t n_users, int32_t n, int32_t K, const int32_t *__restrict__ skip,
    const int32_t *__restrict__ user_list, const int32_t *__restrict__ user_count, int32_t gate_cap,
    float *__restrict__ out_s, int32_t *__restrict__ out_id) {
  constexpr int RPW = 4, LANES = 256 / RPW;
  const int sub_row = threadIdx.x / LANES, lane = threadIdx.x % LANES;
  float *cs_w = merge_lds + (size_t)sub_row * 2 * n;
  int32_t *ci_w = reinterpret_cast<int32_t *>(cs_w + n);
  for (int64_t rb = blockIdx.x; rb * RPW < n_users; rb += gridDim.x) {
    const int32_t pos = (int32_t)(rb * RPW) + sub_row;
    int32_t user;
    if (!merge_row(n_users, skip, user_list, user_count, gate_cap, pos, user)) continue;
    merge_sync();                 // the row before is done with the buffer
    for (int c = lane; c < n; c += LANES) {
      cs_w[c] = scores[(int64_t)pos * n + c];
      ci_w[c] = ids[(int64_t)pos * n + c];
    }
    merge_sync();
    for (int c = lane; c < n; c += LANES) {
      const float sc = cs_w[c];
      const int32_t ic = ci_w[c];
      int rank = 0;
      for (int j = 0; j < n; ++j) rank += beats(cs_w[j], ci_w[j], j, sc, ic, c) ? 1 : 0;
      if (rank < K) {
        out_s[(int64_t)user * K + rank] = ic >= 0 ? sc : -INFINITY;
        out_id[(int64_t)user * K + rank] = ic;
      }
    }
  }
}

// The same for P lists of Kin entries that are each in final order already -- (score desc, id
// asc), padding last -- as the lists of score_topk_kernel and of this kernel are.  An entry is
// beaten by the entries in front of it in its own list and by a prefix of every other list, found
// by bisection: P * log2(Kin) steps per entry instead of P * Kin.
__global__ __launch_bounds__(256) void topk_merge_sorted_kernel(
    const float *__restrict__ scores, const int32_t *__restrict__ ids,
    int32_t n_users, int32_t P, int32_t Kin, int32_t K, const int32_t *__restrict__ skip,
    const int32_t *__restrict__ user_list, const int32_t *__restrict__ user_count, int32_t gate_cap,
    float *__restrict__ out_s, int32_t *__restrict__ out_id) {
  constexpr int RPW = 4, LANES = 256 / RPW;
  const int sub_row = threadIdx.x / LANES, lane = threadIdx.x % LANES;
  const int n = P * Kin;
  float *cs_w = merge_lds + (size_t)sub_row * 2 * n;
  int32_t *ci_w = reinterpret_cast<int32_t *>(cs_w + n);
  for (int64_t rb = blockIdx.x; rb * RPW < n_users; rb += gridDim.x) {
    const int32_t pos = (int32_t)(rb * RPW) + sub_row;
    int32_t user;
    if (!merge_row(n_users, skip, user_list, user_count, gate_cap, pos, user)) continue;
    merge_sync();                 // the row before is done with the buffer
    for (int c = lane; c < n; c += LANES) {
      cs_w[c] = scores[(int64_t)pos * n + c];
      ci_w[c] = ids[(int64_t)pos * n + c];
    }
    merge_sync();
    for (int c = lane; c < n; c += LANES) {
      const float sc = cs_w[c];
      const int32_t ic = ci_w[c];
      const int own = c / Kin;
      int rank = c - own * Kin;
      for (int q = 0; q < P; ++q) {
        if (q == own) continue;
        int lo = 0, hi = Kin;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1, j = q * Kin + mid;
          if (beats(cs_w[j], ci_w[j], j, sc, ic, c))
            lo = mid + 1;
          else
            hi = mid;
        }
        rank += lo;
      }
      if (rank < K) {
        out_s[(int64_t)user * K + rank] = ic >= 0 ? sc : -INFINITY;
        out_id[(int64_t)user * K + rank] = ic;
      }
    }
  }
}

// The merge of the compacted launch: up to 256 sorted lists per listed user (its many short item
// slices), one wavefront per user, one level.  The lists go to LDS; every lane keeps the heads of
// up to four lists and, in registers, the key of the best of them: (score, then the smaller id) as
// a pair of unsigned words.  Each of K rounds takes the largest key of the wavefront -- the larger
// high word by a DPP reduction, then the larger low word among the lanes that hold it --, the lane
// that owns it writes the entry out, moves on in that list and looks at its heads again.  Padding
// (id < 0) ends a list.  (Bisection merges as above, in two levels of 16 lists with a workgroup per
// row, took 41 us for 48 lists and 89 us for 256: chains of dependent LDS reads behind ~100-cycle
// compares; this kernel with the key reduced by twelve dependent shuffles per round 15 and 24 us.)
constexpr int kHeadsMaxLists = 256;
// max over the wavefront of an unsigned word, in every lane: shifts inside the rows of 16 lanes,
// the row's last lane broadcast to the next row(s), the wavefront's last lane read back (a maximum
// may count a lane twice, and 0, which a lane without a source keeps, is its identity)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false));   // row_shr:1
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false));   // row_shr:2
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false));   // row_shr:4
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false));   // row_shr:8
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));   // row_bcast:15 into rows 1, 3
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));   // row_bcast:31 into rows 2, 3
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// (score desc, id asc) as a pair of unsigned words: the float's bits made monotone (-0 and +0 tie,
// as they compare; never 0 for a score that can be listed), and the complemented id; (0, 0): no entry
__device__ __forceinline__ void head_key(float s, int32_t id, uint32_t &hi, uint32_t &lo) {
  uint32_t u = __float_as_uint(s == 0.0f ? 0.0f : s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  hi = id < 0 ? 0u : u;
  lo = id < 0 ? 0u : (uint32_t)(0x7FFFFFFF - id);
}
__global__ __launch_bounds__(64) void topk_merge_heads_kernel(
    const float *__restrict__ scores, const int32_t *__restrict__ ids, int32_t n_rows, int32_t P, int32_t Kin,
    int32_t K, const int32_t *__restrict__ skip, const int32_t *__restrict__ user_list,
    const int32_t *__restrict__ user_count, int32_t gate_cap, float *__restrict__ out_s,
    int32_t *__restrict__ out_id) {
  const int lane = threadIdx.x, n = P * Kin;
  float *ls = merge_lds;
  int32_t *li = reinterpret_cast<int32_t *>(ls + n);
  for (int32_t pos = blockIdx.x; pos < n_rows; pos += gridDim.x) {
    int32_t user;
    if (!merge_row(n_rows, skip, user_list, user_count, gate_cap, pos, user)) continue;
    merge_sync();                  // the row before is done with the buffer
    for (int c = lane; c < n; c += 64) {
      ls[c] = scores[(int64_t)pos * n + c];
      li[c] = ids[(int64_t)pos * n + c];
    }
    merge_sync();
    int h0 = 0, h1 = 0, h2 = 0, h3 = 0;
    uint32_t best_hi = 0, best_lo = 0;
    int bt = 0, be = 0;
    auto look = [&]() {            // the best of this lane's heads
      best_hi = best_lo = 0;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int q = lane + 64 * t, h = t == 0 ? h0 : t == 1 ? h1 : t == 2 ? h2 : h3;
        if (q < P && h < Kin) {
          const int e = q * Kin + h;
          uint32_t hi, lo;
          head_key(ls[e], li[e], hi, lo);
          if (hi > best_hi || (hi == best_hi && lo > best_lo)) {
            best_hi = hi;
            best_lo = lo;
            bt = t;
            be = e;
          }
        }
      }
    };
    look();
    for (int r = 0; r < K; ++r) {
      const uint32_t m_hi = wave_max_u32(best_hi);
      if (m_hi == 0) {                // every list is used up: the rest of the row is padding
        for (int j = r + lane; j < K; j += 64) {
          out_s[(int64_t)user * K + j] = -INFINITY;
          out_id[(int64_t)user * K + j] = -1;
        }
        break;
      }
      const uint32_t m_lo = wave_max_u32(best_hi == m_hi ? best_lo : 0u);
      if (best_hi == m_hi && best_lo == m_lo) {   // (ids are distinct: one lane)
        out_s[(int64_t)user * K + r] = ls[be];
        out_id[(int64_t)user * K + r] = li[be];
        h0 += bt == 0;
        h1 += bt == 1;
        h2 += bt == 2;
        h3 += bt == 3;
        look();
      }
    }
  }
}

// Rankings longer than one list (K > 32, up to the number of items) are produced a page of kPage places at a time:
// page p is the kPage best among the items that rank strictly after the last entry of page p - 1.
// This kernel files a finished page (n_rows x kPage, local ids) under columns col0.. of the
// K-wide outputs and records every row's last entry as the ceiling of the next page (a page
// that is not full has exhausted its row: nothing ranks after (-inf, INT32_MAX)).
constexpr int kPage = 32;
constexpr int kMaxK = 128;   // longest ranking whose ids the metrics kernel holds in registers (longer: rank_metrics_stream_kernel)
__global__ __launch_bounds__(256) void place_page_kernel(const float *__restrict__ page_s, const int32_t *__restrict__ page_id,
                                                         int32_t n_rows, int32_t K, int32_t col0, int32_t id_offset,
                                                         float *__restrict__ out_s, int32_t *__restrict__ out_id,
                                                         float *__restrict__ ceil_s, int32_t *__restrict__ ceil_id) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int32_t row = (int32_t)(t / kPage), j = (int32_t)(t % kPage);
  if (row >= n_rows) return;
  const float v = page_s[t];
  const int32_t id = page_id[t];
  if (col0 + j < K) {
    out_s[(int64_t)row * K + col0 + j] = id >= 0 ? v : -INFINITY;
    out_id[(int64_t)row * K + col0 + j] = id >= 0 ? id + id_offset : -1;
  }
  if (j == kPage - 1) {
    ceil_s[row] = id >= 0 ? v : -INFINITY;
    ceil_id[row] = id >= 0 ? id : INT32_MAX;
  }
}

// ---------------------------------------------------------------- M -------
constexpr int MAXK = 8;  // k values per call

struct KList {
  int32_t k[MAXK];
};

// NW: words of relevance bits (1 for k <= 32; 4 for rankings of up to kMaxK places); NR: ranked ids a
// lane holds in registers and compares every target item with (>= the largest k: 20 for the presets'
// k <= 20, where 32 would spend a third of the kernel's compares on places that do not exist)
template <int NW, int NR>
__global__ __launch_bounds__(256) void rank_metrics_kernel(
    const int32_t *__restrict__ topk_ids, int32_t n_users, int32_t id_stride,
    const int32_t *__restrict__ tgt_ptr, const int32_t *__restrict__ tgt_items,
    KList kl, int32_t nk, int32_t kmax, const float *__restrict__ disc,
    double *__restrict__ part) {
  __shared__ double red[4][MAXK * 6];
  const int32_t u = blockIdx.x * 256 + threadIdx.x;
  float vals[MAXK * 6];
#pragma unroll
  for (int i = 0; i < MAXK * 6; ++i) vals[i] = 0.0f;
  if (u < n_users) {
    const int32_t tb = tgt_ptr[u], te = tgt_ptr[u + 1];
    const int32_t n_pos = te - tb;
    static_assert(NR <= 32 * NW, "a relevance bit per ranked id");
    uint32_t rel[NW];  // bit j: j-th ranked item is a target item
#pragma unroll
    for (int w = 0; w < NW; ++w) rel[w] = 0;
    // The ranked ids go to registers with independent loads.  A user with a handful of target
    // items compares each of them with all ranked ids (a bisection per ranked item is a chain of
    // dependent loads four deep, kmax times over); the few users with long target lists bisect (a
    // walk over thousands of targets is thousands of dependent loads in one lane: 2.5 ms at Elec).
    int32_t ranked[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) ranked[j] = j < kmax ? topk_ids[(int64_t)u * id_stride + j] : -1;
    if (n_pos <= 16) {
      // (four target items per round, loaded together: one at a time is a chain of up to 16 round trips)
      for (int32_t t = tb; t < te; t += 4) {
        int32_t v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = t + e < te ? tgt_items[t + e] : -1;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int j = 0; j < NR; ++j) rel[j >> 5] |= (ranked[j] == v[e] && v[e] >= 0 ? 1u : 0u) << (j & 31);
      }
    } else if constexpr (NW == 1) {
      // long target lists (sorted): every ranked item bisects at the same time, so that a round is
      // kmax independent loads (one item after the other is a chain of kmax * log2(n_pos) dependent
      // loads in one lane -- 59 us for the whole launch at 192k users, set by its slowest wavefront)
      int32_t lo[NR], hi[NR];
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        lo[j] = tb;
        hi[j] = j < kmax && ranked[j] >= 0 ? te : tb;
      }
      const int rounds = 32 - __clz(n_pos);        // a bisection of n entries ends within floor(log2 n) + 1 steps
      for (int s = 0; s < rounds; ++s) {
        int32_t v[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) v[j] = lo[j] < hi[j] ? tgt_items[(lo[j] + hi[j]) >> 1] : 0;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          const int32_t mid = (lo[j] + hi[j]) >> 1;
          if (lo[j] < hi[j]) {
            if (v[j] < ranked[j])
              lo[j] = mid + 1;
            else
              hi[j] = mid;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const bool look = j < kmax && ranked[j] >= 0 && lo[j] < te;
        const int32_t v = look ? tgt_items[lo[j]] : -1;
        rel[0] |= (look && v == ranked[j] ? 1u : 0u) << j;
      }
    } else {
      // (rankings of more than 32 places: a bisection per ranked item, one after the other)
#pragma unroll 1
      for (int j = 0; j < kmax; ++j) {
        int32_t it = ranked[0];
#pragma unroll
        for (int q = 1; q < NR; ++q) it = q == j ? ranked[q] : it;   // (no indexed access: the ids stay in registers)
        int32_t lo = tb, hi = te;
        while (lo < hi) {
          const int32_t mid = (lo + hi) >> 1;
          if (tgt_items[mid] < it)
            lo = mid + 1;
          else
            hi = mid;
        }
        if (it >= 0 && lo < te && tgt_items[lo] == it) {
#pragma unroll
          for (int w = 0; w < NW; ++w) rel[w] |= (w == (j >> 5) ? 1u : 0u) << (j & 31);
        }
      }
    }
    // evalue/metric.py:119-155, per user, fp32
    const float positives = (float)max(n_pos, 1);
#pragma unroll
    for (int ki = 0; ki < MAXK; ++ki) {
      if (ki < nk) {
        const int k = kl.k[ki];
        float hits = 0.f, dcg = 0.f, idcg = 0.f, ap = 0.f, cum = 0.f;
        int first = 0;
        const int ideal_k = min(n_pos, k);
        for (int j = 0; j < k; ++j) {
          uint32_t word = rel[0];
#pragma unroll
          for (int w = 1; w < NW; ++w) word = w == (j >> 5) ? rel[w] : word;
          const float rj = (float)((word >> (j & 31)) & 1u);
          hits += rj;
          dcg += rj * disc[j];
          if (j < ideal_k) idcg += disc[j];
          cum += rj;
          ap += (cum / (float)(j + 1)) * rj;
          if (rj != 0.f && first == 0) first = j + 1;
        }
        vals[ki * 6 + 0] = hits / (float)k;
        vals[ki * 6 + 1] = hits / positives;
        vals[ki * 6 + 2] = hits > 0.f ? 1.f : 0.f;
        vals[ki * 6 + 3] = idcg > 0.f ? dcg / idcg : 0.f;
        vals[ki * 6 + 4] = ap / positives;
        vals[ki * 6 + 5] = first > 0 ? 1.0f / (float)first : 0.f;
      }
    }
  }
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
#pragma unroll
  for (int i = 0; i < MAXK * 6; ++i) {
    if (i >= nk * 6) {                 // (uniform; a 64-bit shuffle reduction is 12 LDS-pipe instructions)
      if (lane == 0) red[wave][i] = 0.0;
      continue;
    }
    double v = (double)vals[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < MAXK * 6)
    part[(int64_t)blockIdx.x * (MAXK * 6) + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// Rankings longer than kMaxK places: the ranked ids stay in memory and a lane walks its user's
// ranking once, bisecting the (sorted) target list per place.  The sums run over the places in
// ascending order, the same operations in the same order as in the kernel above, and a k's metrics
// are read off when its last place has been added: the same values, at the speed of one dependent
// chain per user (rankings this long are rare: evalue/metric.py:9-10 allows any k <= num_items).
__global__ __launch_bounds__(256) void rank_metrics_stream_kernel(
    const int32_t *__restrict__ topk_ids, int32_t n_users, int32_t id_stride,
    const int32_t *__restrict__ tgt_ptr, const int32_t *__restrict__ tgt_items,
    KList kl, int32_t nk, int32_t kmax, const float *__restrict__ disc,
    double *__restrict__ part) {
  __shared__ double red[4][MAXK * 6];
  const int32_t u = blockIdx.x * 256 + threadIdx.x;
  float vals[MAXK * 6];
#pragma unroll
  for (int i = 0; i < MAXK * 6; ++i) vals[i] = 0.0f;
  if (u < n_users) {
    const int32_t tb = tgt_ptr[u], te = tgt_ptr[u + 1];
    const int32_t n_pos = te - tb;
    const float positives = (float)max(n_pos, 1);
    float hits = 0.f, dcg = 0.f, idcg = 0.f, ap = 0.f, cum = 0.f;
    int first = 0;
    for (int32_t j = 0; j < kmax; ++j) {
      const int32_t it = topk_ids[(int64_t)u * id_stride + j];
      int32_t lo = tb, hi = te;
      while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (tgt_items[mid] < it)
          lo = mid + 1;
        else
          hi = mid;
      }
      const float rj = (it >= 0 && lo < te && tgt_items[lo] == it) ? 1.f : 0.f;
      hits += rj;
      dcg += rj * disc[j];
      if (j < n_pos) idcg += disc[j];      // (for a k: the places below min(n_pos, k))
      cum += rj;
      ap += (cum / (float)(j + 1)) * rj;
      if (rj != 0.f && first == 0) first = j + 1;
#pragma unroll
      for (int ki = 0; ki < MAXK; ++ki) {
        if (ki < nk && kl.k[ki] == j + 1) {
          vals[ki * 6 + 0] = hits / (float)(j + 1);
          vals[ki * 6 + 1] = hits / positives;
          vals[ki * 6 + 2] = hits > 0.f ? 1.f : 0.f;
          vals[ki * 6 + 3] = idcg > 0.f ? dcg / idcg : 0.f;
          vals[ki * 6 + 4] = ap / positives;
          vals[ki * 6 + 5] = first > 0 ? 1.0f / (float)first : 0.f;
        }
      }
    }
  }
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
#pragma unroll
  for (int i = 0; i < MAXK * 6; ++i) {
    if (i >= nk * 6) {
      if (lane == 0) red[wave][i] = 0.0;
      continue;
    }
    double v = (double)vals[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < MAXK * 6)
    part[(int64_t)blockIdx.x * (MAXK * 6) + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// Sum of the per-workgroup partials in a fixed order: 16 threads per (k, metric) take the blocks
// b = q, q + 16, ... in ascending order and thread q = 0 adds the 16 sums in order (one thread
// walking all blocks took 180 us for the 752 blocks of 192k users).
__global__ __launch_bounds__(1024) void rank_metrics_final_kernel(const double *__restrict__ part, int32_t n_part,
                                                                  int32_t nk, double denom, float *__restrict__ out,
                                                                  double *__restrict__ sums_out) {
  __shared__ double red[16][64];
  const int i = threadIdx.x & 63, q = threadIdx.x >> 6;
  static_assert(MAXK * 6 <= 64, "one lane per (k, metric)");
  double s0 = 0.0, s1 = 0.0;
  if (i < nk * 6) {
    int32_t b = q;
    for (; b + 16 < n_part; b += 32) {
      s0 += part[(int64_t)b * (MAXK * 6) + i];
      s1 += part[(int64_t)(b + 16) * (MAXK * 6) + i];
    }
    if (b < n_part) s0 += part[(int64_t)b * (MAXK * 6) + i];
  }
  red[q][i] = s0 + s1;
  __syncthreads();
  if (q != 0 || i >= nk * 6) return;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += red[k][i];
  if (sums_out) sums_out[i] = s;
  out[i] = (float)(s / denom);
}

template <int KPAD>
int launch_score_topk(int DC, int grid, hipStream_t st, const float *Uemb, int32_t n_users,
                      const float *Iemb, int32_t n_items, const int32_t *mask_ptr,
                      const int32_t *mask_items, int32_t nsplit, int32_t off, const int32_t *skip,
                      const int32_t *user_list, const int32_t *user_count, int32_t gate_cap, float *ps,
                      int32_t *pi, const float *ceil_s = nullptr, const int32_t *ceil_id = nullptr) {
  if (ceil_s != nullptr) {
    // a later page of a long ranking: full-length lists, all users
    if constexpr (KPAD == kPage) {
      switch (DC) {
        case 1:
          score_topk_kernel<KPAD, 1, false, true><<<grid, 256, 0, st>>>(Uemb, n_users, Iemb, n_items, mask_ptr, mask_items,
                                                                        nsplit, off, skip, user_list, user_count, gate_cap,
                                                                        ps, pi, nullptr, ceil_s, ceil_id);
          break;
        case 2:
          score_topk_kernel<KPAD, 2, false, true><<<grid, 256, 0, st>>>(Uemb, n_users, Iemb, n_items, mask_ptr, mask_items,
                                                                        nsplit, off, skip, user_list, user_count, gate_cap,
                                                                        ps, pi, nullptr, ceil_s, ceil_id);
          break;
        default:
          return rk::fail(RK_ERR_UNSUPPORTED, "rk_score_topk_f32: embedding width %d not built (64/128)", 64 * DC);
      }
      RK_CHECK_LAUNCH("score_topk_kernel");
      return RK_OK;
    } else {
      return rk::fail(RK_ERR_ARG, "pages are ranked with full-length lists");
    }
  }
  switch (DC) {
    case 1:
      score_topk_kernel<KPAD, 1, false><<<grid, 256, 0, st>>>(Uemb, n_users, Iemb, n_items, mask_ptr,
                                                               mask_items, nsplit, off, skip, user_list, user_count,
                                                               gate_cap, ps, pi, nullptr);
      break;
    case 2:
      score_topk_kernel<KPAD, 2, false><<<grid, 256, 0, st>>>(Uemb, n_users, Iemb, n_items, mask_ptr,
                                                               mask_items, nsplit, off, skip, user_list, user_count,
                                                               gate_cap, ps, pi, nullptr);
      break;
    default:
      return rk::fail(RK_ERR_UNSUPPORTED, "rk_score_topk_f32: embedding width %d not built (64/128)",
                      64 * DC);
  }
  RK_CHECK_LAUNCH("score_topk_kernel");
  return RK_OK;
}

struct UserSel {   // which users a launch works on; see score_topk_kernel and merge_row
  const int32_t *skip = nullptr, *list = nullptr, *count = nullptr;
  int32_t cap = 0;
};

// sorted_lists: every list is in (score desc, id asc) order already, padding last.
int merge_lists(const float *scores, const int32_t *ids, int32_t n_users, int32_t P, int32_t Kin,
                int32_t K, const UserSel &sel, bool sorted_lists, float *out_scores, int32_t *out_ids, hipStream_t st) {
  RK_CHECK_ARG((int64_t)P * Kin <= MERGE_MAX, "too many candidates per user (%lld > %d)",
               (long long)P * Kin, MERGE_MAX);
  RK_CHECK_ARG(K <= P * Kin, "K larger than the candidate count");
  const int grid = (int)std::min<int64_t>(rk::ceil_div(n_users, 4), 8 * (int64_t)cu_count());
  const size_t lds = (size_t)4 * 2 * P * Kin * sizeof(float);
  if (sorted_lists)
    topk_merge_sorted_kernel<<<grid, 256, lds, st>>>(scores, ids, n_users, P, Kin, K, sel.skip, sel.list, sel.count, sel.cap,
                                                   out_scores, out_ids);
  else
    topk_merge_kernel<<<grid, 256, lds, st>>>(scores, ids, n_users, P * Kin, K, sel.skip, sel.list, sel.count, sel.cap,
                                            out_scores, out_ids);
  RK_CHECK_LAUNCH("topk_merge_kernel");
  return RK_OK;
}

// exact kernel and merge for n_users users (positions, if sel names a list) over nsplit item slices
int exact_topk(const float *Uemb, int32_t n_users, const float *Iemb, int32_t n_items, int32_t D,
               const int32_t *mask_ptr, const int32_t *mask_items, int32_t K, int32_t item_id_offset,
               int nsplit, const UserSel &sel, float *out_scores, int32_t *out_ids, float *ps, hipStream_t st,
               const float *ceil_s = nullptr, const int32_t *ceil_id = nullptr) {
  const int kp = kpad_for(K);
  int32_t *pi = (int32_t *)(ps + (size_t)n_users * nsplit * kp);
  const int grid = (int)rk::ceil_div(n_users, UPB) * nsplit;
  int rc;
#define RK_EXACT_CASE(KP)                                                                                   \
  case KP:                                                                                                  \
    rc = launch_score_topk<KP>(D / 64, grid, st, Uemb, n_users, Iemb, n_items, mask_ptr, mask_items, nsplit, \
                               item_id_offset, sel.skip, sel.list, sel.count, sel.cap, ps, pi);             \
    break;
  switch (kp) {
    RK_EXACT_CASE(4) RK_EXACT_CASE(8) RK_EXACT_CASE(12) RK_EXACT_CASE(16) RK_EXACT_CASE(20) RK_EXACT_CASE(24)
    RK_EXACT_CASE(28)
    default:
      rc = launch_score_topk<32>(D / 64, grid, st, Uemb, n_users, Iemb, n_items, mask_ptr, mask_items, nsplit,
                                 item_id_offset, sel.skip, sel.list, sel.count, sel.cap, ps, pi, ceil_s, ceil_id);
      break;
  }
#undef RK_EXACT_CASE
  if (rc != RK_OK) return rc;
  if (sel.list != nullptr) {   // the compacted launch: one wavefront per listed user, any number of slices up to 256
    RK_CHECK_ARG(nsplit <= kHeadsMaxLists, "compacted launch: %d item slices (at most %d)", nsplit, kHeadsMaxLists);
    const int mgrid = (int)std::min<int64_t>(n_users, 8 * (int64_t)cu_count());
    topk_merge_heads_kernel<<<mgrid, 64, (size_t)2 * nsplit * kp * sizeof(float), st>>>(
        ps, pi, n_users, nsplit, kp, K, sel.skip, sel.list, sel.count, sel.cap, out_scores, out_ids);
    RK_CHECK_LAUNCH("topk_merge_heads_kernel");
    return RK_OK;
  }
  return merge_lists(ps, pi, n_users, nsplit, kp, K, sel, true, out_scores, out_ids, st);
}

// Users the filter could not certify are few (ties at the K-th place, degenerate embeddings).
// Up to kOpenCap of them are listed and ranked by a launch of their own: packed 64 to a wavefront
// and cut into many item slices, it takes a fraction of a millisecond, where the same users
// scattered over the full grid would each hold a wavefront for a whole pass over the items.
constexpr int kOpenCap = 1024;
inline int open_positions(int32_t n_users) { return (int)std::min<int64_t>(kOpenCap, rk::ceil_div(n_users, UPB) * UPB); }
inline int open_nsplit(int32_t n_items, int32_t K) {
  // as many slices as the one-level merge takes, down to one tile per slice: the launch costs about
  // 15 us + 7 us per tile of a slice (a lone wavefront per SIMD whose lists never leave their start-up),
  // the merge 11 -> 20 us from 55 to 220 lists (19k x 7k: 63 -> 42 us for the pair)
  const int64_t tiles = rk::ceil_div(n_items, TI);
  return (int)std::max<int64_t>(1, std::min<int64_t>(kHeadsMaxLists, tiles));
}
// floats + ints of the compacted launch: its partial lists
inline size_t open_part_entries(int32_t n_users, int32_t n_items, int32_t K) {
  return (size_t)open_positions(n_users) * ((size_t)open_nsplit(n_items, K) + 1) * kpad_for(K);
}

// candidate list length of the filter pass for a final K (0: no filter pass, K too large)
// Item slices of the filter launch: as for the exact one, at the filter's occupancy, but at least
// two on long tables (measured at 1969 tiles: a single slice, with every workgroup of the grid
// walking the table at the same position, runs half as fast per tile), and no more than the exact
// launch uses (the workspace is sized for that).  Short tables (an item shard of a multi-GPU
// evaluation) take ONE slice: every slice starts an empty list per user, and the ~K ln(n/K)
// insertions of a list's start-up are what such a launch spends its time on; one slice also needs
// no merge.  Measured at 192k users: 246 tiles 2.36 -> 2.17 ms per call, 492 tiles even.
constexpr int kShortTableTiles = 512;
inline int filter_nsplit(int n_users, int n_items, int kc, int D) {
  if (rk::ceil_div(n_items, TI) <= kShortTableTiles) return 1;
  // (long tables used to take two slices at least: a single slice looked half as fast per tile,
  // which was one user with a very long training list holding up its workgroup: kHeavyUsers)
  const int ns = choose_nsplit(n_users, n_items, wg_per_cu(kc, D / 64, true));
  return std::min(ns, std::max(1, choose_nsplit(n_users, n_items)));
}

// (8 places of margin between the K-th exact and the last filter score, 4 at least)
inline int filter_len(int K) { return K + 8 <= 32 ? std::max(8, kpad_for(K + 8)) : (K + 4 <= 32 ? 32 : 0); }

// Item slices of the threshold filter's two passes: whatever fills the chip.  A slice costs no list
// start-up there (topk_threshold.h), only its 64 strata maxima and candidate places per user; slices
// are whole groups of four tiles.
inline int threshold_nsplit(int n_users, int n_items, int D) {
  const int64_t groups = rk::ceil_div(rk::ceil_div(n_items, TI), 4);
  const int ns = choose_nsplit(n_users, n_items, D == 64 ? 3 : 2);
  return (int)std::max<int64_t>(1, std::min<int64_t>(ns, groups));
}

// Which form of the filter a shape takes.  The list form (score_topk_kernel on fp16 rows) pays the
// start-up of a sorted list per user and item slice; on a long table walked in ONE slice that is
// amortised and one pass over the scores beats two (Elec, 192k x 63k: 3.0 ms against 1.7 + 2.3).
// Whenever the list form would cut the table into three or more slices to fill the chip (few users:
// a small data set, a rank's block of a user-sharded evaluation) or the table is short (an item
// shard), the threshold form (topk_threshold.h) is faster: 19k x 7k 0.66 -> 0.37 ms, 36k x 18k
// 0.90 -> 0.60, 39k x 23k 0.91 -> 0.72, 48k x 63k 1.58 -> 1.27, 24k x 63k 1.16 -> 0.79 (whole calls; at
// two slices, 96k x 63k, the lists still lead 2.23 to 2.31).  Both give the exact kernel's result.
// RK_FILTER_FORM=lists|threshold in the environment (read at every call) overrides the choice:
// a knob for the parity tests, which run every case through both forms.
inline bool threshold_form(int n_users, int n_items, int kc, int D) {
  if (const char *f = getenv("RK_FILTER_FORM")) {
    if (f[0] == 'l') return false;
    if (f[0] == 't') return true;
  }
  return filter_nsplit(n_users, n_items, kc, D) > 2 || rk::ceil_div(n_items, TI) <= kShortTableTiles;
}

inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

struct TopkWorkspace {
  size_t part, uhalf, ihalf, unorm, umax, uexp, istats, open_count, heavy, stats_part, open_list, cand_s, cand_id, done, thr, cnt, cand, total;
};
inline TopkWorkspace topk_workspace(int32_t n_users, int32_t n_items, int32_t D, int32_t K) {
  TopkWorkspace w{};
  if (K > kPage) {
    // a paged ranking: the exact kernel's partial lists, one finished page (in the cand_* slots)
    // and the rows' ceilings (unorm: scores, uexp: ids)
    const int nsplit = choose_nsplit(n_users, n_items);
    size_t off = 0;
    w.part = off;    off += align256((size_t)n_users * nsplit * kPage * 8);
    w.cand_s = off;  off += align256((size_t)n_users * kPage * 4);
    w.cand_id = off; off += align256((size_t)n_users * kPage * 4);
    w.unorm = off;   off += align256((size_t)n_users * 4);
    w.uexp = off;    off += align256((size_t)n_users * 4);
    w.total = off;
    return w;
  }
  const int kc = filter_len(K);
  // (the filter launch may use fewer slices than the exact one, never more)
  const int nsplit = choose_nsplit(n_users, n_items);
  const size_t klen = (size_t)std::max(kpad_for(K), kc);
  size_t off = 0;
  w.part = off;
  size_t part_entries = std::max((size_t)n_users * std::max(nsplit, 2) * klen, open_part_entries(n_users, n_items, K));
  if (D != 64 && D != 128) part_entries = std::max(part_entries, rk::wide_part_entries(n_users, n_items, K));
  // (the strata maxima of the threshold filter live here too: dead before the exact kernel's lists are written)
  const int tsplit = threshold_nsplit(n_users, n_items, D);
  part_entries = std::max(part_entries, rk::threshold_strata_bytes(n_users, tsplit) / 8 + 1);
  off += align256(part_entries * 8);
  w.uhalf = off;   off += align256((size_t)n_users * D * 2);
  w.ihalf = off;   off += align256((size_t)n_items * D * 2);
  w.unorm = off;   off += align256((size_t)n_users * 4);
  w.umax = off;    off += align256((size_t)n_users * 4);
  w.uexp = off;    off += align256((size_t)n_users * 4);
  w.istats = off;  off += 128;   // item statistics and open_count share one 256-byte block, zeroed together
  w.open_count = off; off += 128;
  w.heavy = off;      off += 256;   // histogram of training-list lengths + the heavy-user threshold (zeroed with the rest)
  w.stats_part = off; off += align256((size_t)2 * rk::kStatSlots * 4);   // follows directly: one memset
  w.open_list = off;  off += align256((size_t)kOpenCap * 4);
  w.cand_s = off;  off += align256((size_t)n_users * std::max(kc, 1) * 4);
  w.cand_id = off; off += align256((size_t)n_users * std::max(kc, 1) * 4);
  w.done = off;    off += align256((size_t)n_users * 4);
  w.thr = off;     off += align256((size_t)n_users * 4);
  w.cnt = off;     off += align256((size_t)n_users * tsplit * 4);
  w.cand = off;    off += align256(rk::threshold_cand_bytes(n_users, tsplit));
  w.total = off;
  return w;
}

// K > kPage: pages of kPage places by the exact kernel, each bounded above by the page before
int paged_topk(const float *Uemb, int32_t n_users, const float *Iemb, int32_t n_items, int32_t D,
               const int32_t *mask_ptr, const int32_t *mask_items, int32_t K, int32_t item_id_offset,
               float *out_scores, int32_t *out_ids, void *ws, hipStream_t st) {
  const TopkWorkspace w = topk_workspace(n_users, n_items, D, K);
  char *base = (char *)ws;
  float *ps = (float *)(base + w.part), *page_s = (float *)(base + w.cand_s), *ceil_s = (float *)(base + w.unorm);
  int32_t *page_id = (int32_t *)(base + w.cand_id), *ceil_id = (int32_t *)(base + w.uexp);
  const int nsplit = choose_nsplit(n_users, n_items);
  const int grid = (int)rk::ceil_div((int64_t)n_users * kPage, 256);
  for (int32_t col0 = 0; col0 < K; col0 += kPage) {
    const int rc = exact_topk(Uemb, n_users, Iemb, n_items, D, mask_ptr, mask_items, kPage, 0, nsplit, UserSel{}, page_s,
                              page_id, ps, st, col0 > 0 ? ceil_s : nullptr, col0 > 0 ? ceil_id : nullptr);
    if (rc != RK_OK) return rc;
    place_page_kernel<<<grid, 256, 0, st>>>(page_s, page_id, n_users, K, col0, item_id_offset, out_scores, out_ids, ceil_s,
                                            ceil_id);
    RK_CHECK_LAUNCH("place_page_kernel");
  }
  return RK_OK;
}

int check_topk_args(const float *Uemb, int32_t n_users, const float *Iemb, int32_t n_items, int32_t D,
                    const int32_t *mask_ptr, const int32_t *mask_items, int32_t K, float *out_scores,
                    int32_t *out_ids, void *ws, size_t ws_bytes) {
  RK_CHECK_ARG(n_users > 0 && n_items > 0, "empty score matrix");
  // evalue/metric.py:9-10
  RK_CHECK_ARG(K >= 1 && K <= n_items, "k must be in the range [1, num_items]");
  if (K > kPage && D != 64 && D != 128)
    return rk::fail(RK_ERR_UNSUPPORTED, "rk_score_topk_f32: K=%d above %d is built for embedding widths 64 and 128 (got %d)", K,
                    kPage, D);
  RK_CHECK_ARG(D > 0 && D % 64 == 0, "embedding width must be a multiple of 64");
  RK_CHECK_ARG(Uemb && Iemb && out_scores && out_ids && ws, "null pointer");
  RK_CHECK_ARG((mask_ptr == nullptr) == (mask_items == nullptr), "mask_ptr and mask_items go together");
  RK_CHECK_ARG(rk::aligned16(Uemb) && rk::aligned16(Iemb) && rk::aligned16(ws),
               "embeddings and workspace must be 16-byte aligned");
  RK_CHECK_ARG(ws_bytes >= topk_workspace(n_users, n_items, D, K).total, "workspace too small");
  return RK_OK;
}

}  // namespace

extern "C" {

size_t rk_score_topk_ws_bytes(int32_t n_users, int32_t n_items, int32_t D, int32_t K) {
  if (n_users <= 0 || n_items <= 0 || D <= 0 || K <= 0 || K > n_items) return 0;
  return topk_workspace(n_users, n_items, D, K).total;
}

int rk_score_topk_open_users(int32_t n_users, int32_t n_items, int32_t D, int32_t K, const void *ws,
                             size_t ws_bytes, void *stream, int32_t *open_users) {
  RK_CHECK_ARG(ws && open_users, "null pointer");
  RK_CHECK_ARG(n_users > 0 && n_items > 0 && D > 0 && K >= 1 && K <= n_items, "bad sizes");
  *open_users = 0;
  if (filter_len(K) == 0 || (D != 64 && D != 128)) return RK_OK;
  const TopkWorkspace w = topk_workspace(n_users, n_items, D, K);
  RK_CHECK_ARG(ws_bytes >= w.total, "workspace too small");
  RK_CHECK_HIP(hipMemcpyAsync(open_users, (const char *)ws + w.open_count, sizeof(int32_t), hipMemcpyDeviceToHost,
                              (hipStream_t)stream));
  RK_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  return RK_OK;
}

int rk_topk_merge_f32(const float *scores, const int32_t *ids, int32_t n_users, int32_t P,
                      int32_t Kin, int32_t K, float *out_scores, int32_t *out_ids, void *stream) {
  RK_CHECK_ARG(n_users >= 0 && P > 0 && Kin > 0 && K > 0, "bad sizes");
  if (n_users == 0) return RK_OK;
  RK_CHECK_ARG(scores && ids && out_scores && out_ids, "null pointer");
  return merge_lists(scores, ids, n_users, P, Kin, K, UserSel{}, false, out_scores, out_ids, (hipStream_t)stream);
}

int rk_score_topk_exact_f32(const float *Uemb, int32_t n_users, const float *Iemb, int32_t n_items,
                            int32_t D, const int32_t *mask_ptr, const int32_t *mask_items, int32_t K,
                            int32_t item_id_offset, float *out_scores, int32_t *out_ids, void *ws,
                            size_t ws_bytes, void *stream) {
  const int rc = check_topk_args(Uemb, n_users, Iemb, n_items, D, mask_ptr, mask_items, K, out_scores,
                                 out_ids, ws, ws_bytes);
  if (rc != RK_OK) return rc;
  if (K > kPage)
    return paged_topk(Uemb, n_users, Iemb, n_items, D, mask_ptr, mask_items, K, item_id_offset, out_scores, out_ids, ws,
                      (hipStream_t)stream);
  const TopkWorkspace w = topk_workspace(n_users, n_items, D, K);
  // widths the register-resident kernel is not built for: the kernel that walks the width in chunks
  if (D != 64 && D != 128)
    return rk::wide_topk(Uemb, n_users, Iemb, n_items, D, mask_ptr, mask_items, K, item_id_offset, out_scores, out_ids,
                         (float *)((char *)ws + w.part), (hipStream_t)stream);
  return exact_topk(Uemb, n_users, Iemb, n_items, D, mask_ptr, mask_items, K, item_id_offset,
                    choose_nsplit(n_users, n_items), UserSel{}, out_scores, out_ids,
                    (float *)((char *)ws + w.part), (hipStream_t)stream);
}

int rk_score_topk_f32(const float *Uemb, int32_t n_users, const float *Iemb, int32_t n_items,
                      int32_t D, const int32_t *mask_ptr, const int32_t *mask_items, int32_t K,
                      int32_t item_id_offset, float *out_scores, int32_t *out_ids, void *ws,
                      size_t ws_bytes, void *stream) {
  // The filter's dozen launches cost a few tenths of a millisecond whatever the size; measured:
  // 19k x 7k (1.4e8 scores) exact 0.46 ms / filter 0.37 ms, 36k x 18k (6.5e8) 1.45 / 0.61,
  // 39k x 23k (9.1e8) 1.64 / 0.74.  From 2^27 scores up the filter wins.
  if ((int64_t)n_users * n_items < ((int64_t)1 << 27))
    return rk_score_topk_exact_f32(Uemb, n_users, Iemb, n_items, D, mask_ptr, mask_items, K, item_id_offset,
                                   out_scores, out_ids, ws, ws_bytes, stream);
  return rk_score_topk_filtered_f32(Uemb, n_users, Iemb, n_items, D, mask_ptr, mask_items, K, item_id_offset,
                                    out_scores, out_ids, ws, ws_bytes, stream);
}

int rk_score_topk_filtered_f32(const float *Uemb, int32_t n_users, const float *Iemb, int32_t n_items,
                               int32_t D, const int32_t *mask_ptr, const int32_t *mask_items, int32_t K,
                               int32_t item_id_offset, float *out_scores, int32_t *out_ids, void *ws,
                               size_t ws_bytes, void *stream) {
  const int kc = filter_len(K);
  if (kc == 0 || (D != 64 && D != 128))
    return rk_score_topk_exact_f32(Uemb, n_users, Iemb, n_items, D, mask_ptr, mask_items, K, item_id_offset,
                                   out_scores, out_ids, ws, ws_bytes, stream);
  int rc = check_topk_args(Uemb, n_users, Iemb, n_items, D, mask_ptr, mask_items, K, out_scores, out_ids, ws,
                           ws_bytes);
  if (rc != RK_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  const TopkWorkspace w = topk_workspace(n_users, n_items, D, K);
  char *base = (char *)ws;
  const int nsplit = choose_nsplit(n_users, n_items);
  const int fsplit = filter_nsplit(n_users, n_items, kc, D);
  float *uhalf = (float *)(base + w.uhalf), *ihalf = (float *)(base + w.ihalf);
  float *unorm = (float *)(base + w.unorm), *umax = (float *)(base + w.umax), *istats = (float *)(base + w.istats);
  int32_t *uexp = (int32_t *)(base + w.uexp);
  float *cand_s = (float *)(base + w.cand_s);
  int32_t *cand_id = (int32_t *)(base + w.cand_id), *done = (int32_t *)(base + w.done);
  int32_t *open_count = (int32_t *)(base + w.open_count), *open_list = (int32_t *)(base + w.open_list);

  // 1. fp16 copies of both tables; user norms and the largest item norm for the error bound
  RK_CHECK_HIP(hipMemsetAsync(istats, 0, 512 + 2 * rk::kStatSlots * 4, st));   // statistics, open_count, heavy, partial maxima
  int32_t *heavy = mask_ptr ? (int32_t *)(base + w.heavy) : nullptr;
  if (heavy && (rc = rk::heavy_threshold(mask_ptr, n_users, n_items, heavy, st)) != RK_OK) return rc;
  const int32_t *heavy_thresh = heavy ? heavy + 32 : nullptr;
  unsigned int *stats_part = (unsigned int *)(base + w.stats_part);
  if ((rc = rk::half_rows(Uemb, n_users, D, true, uhalf, unorm, umax, uexp, nullptr, nullptr, st)) != RK_OK) return rc;
  if ((rc = rk::half_rows(Iemb, n_items, D, false, ihalf, nullptr, nullptr, nullptr, stats_part, istats, st)) != RK_OK)
    return rc;
  float *ps = (float *)(base + w.part);
  if (threshold_form(n_users, n_items, kc, D)) {
    // 2. strata maxima -> per-user threshold -> every item at or above it as a candidate
    const int tsplit = threshold_nsplit(n_users, n_items, D);
    float *thr = (float *)(base + w.thr);
    int32_t *cnt = (int32_t *)(base + w.cnt);
    void *cand = base + w.cand;
    if ((rc = rk::threshold_candidates(kc, D, tsplit, st, uhalf, n_users, ihalf, n_items, mask_ptr, mask_items, heavy_thresh,
                                       ps, thr, cand, cnt)) != RK_OK)
      return rc;
    // 3. exact scores of the candidates, final order, certification
    if ((rc = rk::refine_threshold_candidates(Uemb, Iemb, n_users, n_items, D, cand, cnt, tsplit, thr, K, mask_ptr, heavy_thresh,
                                              unorm, uexp, istats, item_id_offset, out_scores, out_ids, done, open_count,
                                              open_list, kOpenCap, st)) != RK_OK)
      return rc;
  } else {
  // 2. the kc best filter scores per user (local item ids)
  int32_t *pi = (int32_t *)(ps + (size_t)n_users * fsplit * kc);
  const int grid = (int)rk::ceil_div(n_users, UPB) * fsplit;
  // a single slice's lists ARE the candidates (sorted, unfilled places -inf / -1): no merge
  const bool direct = fsplit == 1 && n_items >= kc;
  if ((rc = rk::launch_filter_topk(kc, D / 64, grid, st, uhalf, n_users, ihalf, n_items, mask_ptr, mask_items,
                                   fsplit, direct ? cand_s : ps, direct ? cand_id : pi, heavy_thresh)) != RK_OK)
    return rc;
  // two slices: the refine kernel merges the two lists itself; more: a merge launch first
  const bool merge_in_refine = !direct && fsplit == 2 && n_items >= kc;
  if (!direct && !merge_in_refine &&
      (rc = merge_lists(ps, pi, n_users, fsplit, kc, std::min(kc, n_items), UserSel{}, true, cand_s, cand_id, st)) != RK_OK)
    return rc;
  // 3. exact scores of the candidates, final order, certification
  if ((rc = rk::refine_candidates(Uemb, Iemb, n_users, n_items, D, merge_in_refine ? ps : cand_s,
                                  merge_in_refine ? pi : cand_id, merge_in_refine ? 2 : 1, std::min(kc, n_items), K, mask_ptr,
                                  heavy_thresh, unorm,
                                  uexp, istats, item_id_offset, out_scores, out_ids, done, open_count, open_list,
                                  kOpenCap, st)) != RK_OK)
    return rc;
  }
  // 4. the exact kernel for the users that could not be certified: the listed ones as a compact
  //    launch; if the list overflowed (then that launch does nothing), everybody whose flag is 0
  UserSel listed, flagged;
  listed.list = open_list; listed.count = open_count; listed.cap = kOpenCap;
  flagged.skip = done; flagged.count = open_count; flagged.cap = kOpenCap;
  if ((rc = exact_topk(Uemb, open_positions(n_users), Iemb, n_items, D, mask_ptr, mask_items, K, item_id_offset,
                       open_nsplit(n_items, K), listed, out_scores, out_ids, ps, st)) != RK_OK)
    return rc;
  return exact_topk(Uemb, n_users, Iemb, n_items, D, mask_ptr, mask_items, K, item_id_offset, nsplit, flagged,
                    out_scores, out_ids, ps, st);
}

size_t rk_rows_topk_ws_bytes(int32_t n_rows, int32_t K) {
  if (n_rows <= 0 || K <= 0) return 0;
  const size_t lists = (size_t)n_rows * ROWS_LANES * kpad_for(std::min(K, kPage)) * (sizeof(float) + sizeof(int32_t));
  if (K <= kPage) return lists;
  // paged: one finished page and the rows' ceilings behind the partial lists
  return align256(lists) + 2 * align256((size_t)n_rows * kPage * 4) + 2 * align256((size_t)n_rows * 4);
}

int rk_rows_topk_f32(const float *scores, int32_t n_rows, int32_t n_items, int64_t ld, int32_t K,
                     float *out_scores, int32_t *out_ids, void *ws, size_t ws_bytes, void *stream) {
  RK_CHECK_ARG(n_rows >= 0 && n_items > 0 && ld >= n_items, "bad score matrix shape");
  // evalue/metric.py:9-10
  RK_CHECK_ARG(K >= 1 && K <= n_items, "k must be in the range [1, num_items]");
  if (n_rows == 0) return RK_OK;
  RK_CHECK_ARG(scores && out_scores && out_ids && ws, "null pointer");
  RK_CHECK_ARG(ws_bytes >= rk_rows_topk_ws_bytes(n_rows, K), "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int grid = (int)rk::ceil_div(n_rows, 256 / ROWS_LANES);
  if (K > kPage) {
    // pages of kPage places, each bounded above by the last entry of the page before
    char *base = (char *)ws;
    float *ps = (float *)base;
    int32_t *pi = (int32_t *)(ps + (size_t)n_rows * ROWS_LANES * kPage);
    size_t off = align256((size_t)n_rows * ROWS_LANES * kPage * 8);
    float *page_s = (float *)(base + off);    off += align256((size_t)n_rows * kPage * 4);
    int32_t *page_id = (int32_t *)(base + off); off += align256((size_t)n_rows * kPage * 4);
    float *ceil_s = (float *)(base + off);    off += align256((size_t)n_rows * 4);
    int32_t *ceil_id = (int32_t *)(base + off);
    const int pgrid = (int)rk::ceil_div((int64_t)n_rows * kPage, 256);
    for (int32_t col0 = 0; col0 < K; col0 += kPage) {
      if (col0 == 0)
        rows_topk_kernel<kPage, false><<<grid, 256, 0, st>>>(scores, n_rows, n_items, ld, ps, pi);
      else
        rows_topk_kernel<kPage, true><<<grid, 256, 0, st>>>(scores, n_rows, n_items, ld, ps, pi, ceil_s, ceil_id);
      RK_CHECK_LAUNCH("rows_topk_kernel");
      const int rc = rk_topk_merge_f32(ps, pi, n_rows, ROWS_LANES, kPage, kPage, page_s, page_id, stream);
      if (rc != RK_OK) return rc;
      place_page_kernel<<<pgrid, 256, 0, st>>>(page_s, page_id, n_rows, K, col0, 0, out_scores, out_ids, ceil_s, ceil_id);
      RK_CHECK_LAUNCH("place_page_kernel");
    }
    return RK_OK;
  }
  const int kp = kpad_for(K);
  float *ps = (float *)ws;
  int32_t *pi = (int32_t *)(ps + (size_t)n_rows * ROWS_LANES * kp);
  switch (kp) {
    case 4: rows_topk_kernel<4><<<grid, 256, 0, st>>>(scores, n_rows, n_items, ld, ps, pi); break;
    case 8: rows_topk_kernel<8><<<grid, 256, 0, st>>>(scores, n_rows, n_items, ld, ps, pi); break;
    case 12: rows_topk_kernel<12><<<grid, 256, 0, st>>>(scores, n_rows, n_items, ld, ps, pi); break;
    case 16: rows_topk_kernel<16><<<grid, 256, 0, st>>>(scores, n_rows, n_items, ld, ps, pi); break;
    case 20: rows_topk_kernel<20><<<grid, 256, 0, st>>>(scores, n_rows, n_items, ld, ps, pi); break;
    case 24: rows_topk_kernel<24><<<grid, 256, 0, st>>>(scores, n_rows, n_items, ld, ps, pi); break;
    case 28: rows_topk_kernel<28><<<grid, 256, 0, st>>>(scores, n_rows, n_items, ld, ps, pi); break;
    default: rows_topk_kernel<32><<<grid, 256, 0, st>>>(scores, n_rows, n_items, ld, ps, pi); break;
  }
  RK_CHECK_LAUNCH("rows_topk_kernel");
  return rk_topk_merge_f32(ps, pi, n_rows, ROWS_LANES, kp, K, out_scores, out_ids, stream);
}

size_t rk_rank_metrics_ws_bytes(int32_t n_users, int32_t nk) {
  (void)nk;
  if (n_users < 0) return 0;
  return (size_t)(rk::ceil_div(n_users, 256) + 1) * MAXK * 6 * sizeof(double);
}

int rk_rank_metrics_f32(const int32_t *topk_ids, int32_t n_users, int32_t id_stride,
                        const int32_t *tgt_ptr, const int32_t *tgt_items,
                        const int32_t *k_list_host, int32_t nk, const float *discounts,
                        float *out, double *sums_out, int64_t mean_denominator, void *ws,
                        size_t ws_bytes, void *stream) {
  RK_CHECK_ARG(n_users > 0 && nk > 0 && nk <= MAXK, "need 1..%d k values and at least one user", MAXK);
  RK_CHECK_ARG(topk_ids && tgt_ptr && tgt_items && k_list_host && discounts && out && ws, "null pointer");
  RK_CHECK_ARG(ws_bytes >= rk_rank_metrics_ws_bytes(n_users, nk), "workspace too small");
  RK_CHECK_ARG(mean_denominator > 0, "mean denominator must be positive");
  KList kl;
  int kmax = 0;
  for (int i = 0; i < MAXK; ++i) {
    kl.k[i] = i < nk ? k_list_host[i] : 0;
    if (i < nk) {
      RK_CHECK_ARG(kl.k[i] >= 1 && kl.k[i] <= id_stride, "k=%d outside [1, id_stride = %d]", kl.k[i], id_stride);
      if (kl.k[i] > kmax) kmax = kl.k[i];
    }
  }
  hipStream_t st = (hipStream_t)stream;
  const int nb = (int)rk::ceil_div(n_users, 256);
  if (kmax <= 20)
    rank_metrics_kernel<1, 20><<<nb, 256, 0, st>>>(topk_ids, n_users, id_stride, tgt_ptr, tgt_items, kl, nk, kmax, discounts,
                                                  (double *)ws);
  else if (kmax <= 32)
    rank_metrics_kernel<1, 32><<<nb, 256, 0, st>>>(topk_ids, n_users, id_stride, tgt_ptr, tgt_items, kl, nk, kmax, discounts,
                                                (double *)ws);
  else if (kmax <= kMaxK)
    rank_metrics_kernel<kMaxK / 32, kMaxK><<<nb, 256, 0, st>>>(topk_ids, n_users, id_stride, tgt_ptr, tgt_items, kl, nk, kmax,
                                                        discounts, (double *)ws);
  else
    rank_metrics_stream_kernel<<<nb, 256, 0, st>>>(topk_ids, n_users, id_stride, tgt_ptr, tgt_items, kl, nk, kmax, discounts,
                                                   (double *)ws);
  RK_CHECK_LAUNCH("rank_metrics_kernel");
  rank_metrics_final_kernel<<<1, 1024, 0, st>>>((const double *)ws, nb, nk, (double)mean_denominator,
                                               out, sums_out);
  RK_CHECK_LAUNCH("rank_metrics_final_kernel");
  return RK_OK;
}

}  // extern "C"
