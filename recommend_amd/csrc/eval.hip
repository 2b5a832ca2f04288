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
#include "common.h"

namespace {

using rk::kWave;
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));

constexpr int TI = 32;     // items per LDS tile (MFMA rows)
constexpr int UPW = 64;    // users per wavefront
constexpr int UPB = 256;   // users per workgroup (4 independent wavefronts)
constexpr int QS = 8;      // queue slots per lane
constexpr float kMasked = -1e10f;  // evalue/evaluator.py:55

// list length the kernel is compiled for: K rounded up to a multiple of 4 (an insertion costs one
// step per list entry, and a longer list also lowers the threshold that filters candidates)
__host__ __device__ inline int kpad_for(int K) { return K <= 4 ? 4 : (K + 3) / 4 * 4; }

// Item slices per user tile.  Every slice restarts its lists from nothing and pays the K*ln(n/K)
// insertions of a streaming top-K again, so fewer slices are cheaper; what slices buy is a grid
// that fills the chip evenly.  The estimate below is launch rounds (two workgroups fit a CU)
// times the work per round, with 4 % of a slice's time charged per extra slice.
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

inline int choose_nsplit(int n_users, int n_items) {
  const int64_t utiles = rk::ceil_div(n_users, UPB);
  const int64_t tiles = rk::ceil_div(n_items, TI);
  const int64_t slots = 2 * (int64_t)cu_count();
  int best = 1;
  double best_cost = 0.0;
  for (int ns = 1; ns <= 16; ++ns) {  // ns * 32 <= MERGE_MAX
    if (ns > 1 && tiles / ns < 4) break;
    const int64_t rounds = rk::ceil_div(utiles * ns, slots);
    const double cost = (double)rounds / ns * (1.0 + 0.04 * ns);
    if (ns == 1 || cost < best_cost * 0.999) {
      best = ns;
      best_cost = cost;
    }
  }
  return best;
}

template <int KPAD>
struct LaneList {
  float s[KPAD];
  int32_t id[KPAD];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int j = 0; j < KPAD; ++j) {
      s[j] = -INFINITY;
      id[j] = -1;
    }
  }
  // Sorted insertion; a no-op for lanes whose v does not beat the last entry.  With
  // s[j-1] >= s[j], the new s[j] is the median of (s[j-1], s[j], v): one v_med3_f32.  The ids
  // follow the same three cases through the strict compares, so equal scores stay in arrival order.
  __device__ __forceinline__ void insert(float v, int32_t vid) {
    bool c_hi = v > s[KPAD - 1];
#pragma unroll
    for (int j = KPAD - 1; j >= 1; --j) {
      const bool c_lo = v > s[j - 1];
      id[j] = c_lo ? id[j - 1] : (c_hi ? vid : id[j]);
      s[j] = __builtin_amdgcn_fmed3f(s[j - 1], s[j], v);
      c_hi = c_lo;
    }
    id[0] = c_hi ? vid : id[0];
    s[0] = fmaxf(s[0], v);
  }
};

template <int KPAD, int DC>
__global__ __launch_bounds__(256, (DC == 1 && KPAD <= 24) ? 2 : 1) void score_topk_kernel(
    const float *__restrict__ Uemb, int32_t n_users,
    const float *__restrict__ Iemb, int32_t n_items,
    const int32_t *__restrict__ mask_ptr, const int32_t *__restrict__ mask_items,
    int32_t nsplit, int32_t item_id_offset, float *__restrict__ part_s,
    int32_t *__restrict__ part_id) {
  constexpr int D = 64 * DC;
  // two separate arrays, not one indexed by a buffer number: the compiler then knows that a DMA
  // into one cannot alias a read of the other and does not wait for the prefetch before reading
  __shared__ __attribute__((aligned(1024))) float tile0[4][TI * D];
  __shared__ __attribute__((aligned(1024))) float tile1[4][TI * D];
  // candidate queues: [wave][0] scores, [wave][1] item ids (one array, so that a score and its
  // id, a fixed distance apart, go out in one ds_write2)
  __shared__ uint32_t queue[4][2][QS][kWave];

  const int tid = threadIdx.x;
  const int wave = tid / kWave, lane = tid % kWave;
  const int col = lane & 31, half = lane >> 5;
  const int32_t split = blockIdx.x % nsplit;
  const int32_t utile = blockIdx.x / nsplit;
  const int32_t ubase = utile * UPB + wave * UPW;
  const int32_t user = ubase + lane;          // the user whose list this lane keeps
  const bool user_ok = user < n_users;

  const int32_t tiles_total = (n_items + TI - 1) / TI;
  const int32_t t_lo = (int32_t)((int64_t)tiles_total * split / nsplit);
  const int32_t t_hi = (int32_t)((int64_t)tiles_total * (split + 1) / nsplit);

  // user fragments B[k][j = col]: chain A multiplies users ubase+col, chain B users ubase+32+col;
  // this lane holds k = 64c + 32*half + m
  float bfa[DC][32], bfb[DC][32];
#pragma unroll
  for (int c = 0; c < DC; ++c)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
      const int64_t off = 64 * c + 32 * half + 4 * q;
      if (ubase + col < n_users)
        va = *reinterpret_cast<const float4 *>(Uemb + (int64_t)(ubase + col) * D + off);
      if (ubase + 32 + col < n_users)
        vb = *reinterpret_cast<const float4 *>(Uemb + (int64_t)(ubase + 32 + col) * D + off);
      bfa[c][4 * q + 0] = va.x; bfa[c][4 * q + 1] = va.y; bfa[c][4 * q + 2] = va.z; bfa[c][4 * q + 3] = va.w;
      bfb[c][4 * q + 0] = vb.x; bfb[c][4 * q + 1] = vb.y; bfb[c][4 * q + 2] = vb.z; bfb[c][4 * q + 3] = vb.w;
    }

  // train-mask cursor of this lane's user (item ids ascending)
  int32_t mcur = 0, mend = 0;
  if (mask_ptr && user_ok) {
    mcur = mask_ptr[user];
    mend = mask_ptr[user + 1];
    int32_t lo = mcur, hi = mend;      // skip entries in front of this slice
    const int32_t first = t_lo * TI;
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

  LaneList<KPAD> list;
  list.init();
  float thr = -INFINITY;
  int qc = 0;

  // Tile staging by LDS-DMA: one wave-instruction writes 64 x 16 B = 1 KiB of LDS linearly from
  // per-lane global addresses; slot s of row r holds chunk s ^ (r & 15) of that item's embedding.
  constexpr int CPR = D / 4;     // 16-byte chunks per row
  constexpr int RPP = 64 / CPR;  // rows per 1-KiB piece
  constexpr int PIECES = TI / RPP;
  // A full tile is one contiguous block of the table: the per-lane byte offset of piece p is a
  // loop-invariant pattern XOR a per-piece constant (rows p*RPP + l with l < RPP never carry into
  // the bits that p*RPP sets), so a piece costs one XOR besides the DMA instruction itself.
  const int l_row = lane / CPR, l_slot = lane % CPR;
  uint32_t lane_off = (uint32_t)(l_row * D * 4 + (((l_slot & ~15) | ((l_slot & 15) ^ l_row)) * 16));
  auto issue_tile = [&](int32_t t, float *dst) {
    asm volatile("" : "+v"(lane_off));  // keep the per-piece offsets from being hoisted into registers
    if ((t + 1) * TI <= n_items) {
      const char *tile_base = reinterpret_cast<const char *>(Iemb + (int64_t)t * TI * D);
#pragma unroll
      for (int p = 0; p < PIECES; ++p) {
        const uint32_t off = (lane_off ^ (uint32_t)(((p * RPP) & 15) * 16)) + (uint32_t)(p * 1024);
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)(tile_base + off),
            (__attribute__((address_space(3))) void *)(dst + p * 256), 16, 0, 0);
      }
      return;
    }
    // the last, partial tile: rows past the end of the table re-read the last item (their scores
    // are discarded after the multiplication); same offset pattern, row clamped per lane
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      const uint32_t off = lane_off ^ (uint32_t)(((p * RPP) & 15) * 16);
      const int32_t item = min(t * TI + p * RPP + (int32_t)(off / (D * 4)), n_items - 1);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void *)(
              reinterpret_cast<const char *>(Iemb + (int64_t)item * D) + (off & (D * 4 - 1))),
          (__attribute__((address_space(3))) void *)(dst + p * 256), 16, 0, 0);
    }
  };
  auto flush = [&]() {
#pragma unroll 1
    for (int q = 0; q < QS; ++q) {
      const bool has = q < qc;
      if (!__any(has)) break;
      const float v = has ? __uint_as_float(queue[wave][0][q][lane]) : -INFINITY;
      const int32_t vid = (int32_t)queue[wave][1][q][lane];
      list.insert(v, vid);
    }
    qc = 0;
    thr = list.s[KPAD - 1];
  };
  // Four scores of this lane's user (ascending item ids).  Late in the stream hardly any score
  // beats its lane's threshold, so the wavefront first asks whether any lane has a candidate at
  // all; only then are the four scores parked, branch free: the store always happens and the
  // slot advances only for a candidate.
  auto consider4 = [&](float v0, float v1, float v2, float v3, int32_t id0) {
    const float m = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
    if (!__any(m > thr)) return;
    const float v[4] = {v0, v1, v2, v3};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      queue[wave][0][qc][lane] = __float_as_uint(v[i]);
      queue[wave][1][qc][lane] = (uint32_t)(id0 + i);
      qc += v[i] > thr ? 1 : 0;
    }
    // a queue holds QS entries: insert when some lane could overflow within the next 4 scores
    if (__any(qc > QS - 4)) flush();
  };

  uint32_t frag_off = (uint32_t)(col * D * 4 + ((8 * half) ^ (col & 15)) * 16);
  // one tile: prefetch the next one into `nxt`, score and select from `cur`
  auto process_tile = [&](int32_t t, const float *cur, float *nxt) {
    if (t + 1 < t_hi) issue_tile(t + 1, nxt);   // lands during this tile's MFMA + selection
    asm volatile("" : "+v"(frag_off));  // as for lane_off: eight hoisted addresses cost eight registers

    floatx16 acc_a, acc_b;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc_a[r] = 0.0f;
      acc_b[r] = 0.0f;
    }
#pragma unroll
    for (int c = 0; c < DC; ++c) {
      float afrag[32];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        // slot 16c + ((8*half + q) ^ (col & 15)) of row col; q < 8 only touches bits the other
        // terms leave alone, so the eight addresses are one per-lane value XOR q*16
        const float4 v = *reinterpret_cast<const float4 *>(
            reinterpret_cast<const char *>(cur) + ((frag_off ^ (uint32_t)(q * 16)) + (uint32_t)(c * 256)));
        afrag[4 * q + 0] = v.x;
        afrag[4 * q + 1] = v.y;
        afrag[4 * q + 2] = v.z;
        afrag[4 * q + 3] = v.w;
      }
#pragma unroll
      for (int m = 0; m < 32; ++m) {
        acc_a = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[m], bfa[c][m], acc_a, 0, 0, 0);
        acc_b = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[m], bfb[c][m], acc_b, 0, 0, 0);
      }
    }

    // Lane l holds rows (r&3)+8*(r>>2)+4*half of user col in acc_a and of user 32+col in acc_b.
    // After the swap every lane holds, for ITS user (ubase + lane), rows (r&3)+8*(r>>2) in lo[r]
    // and rows 4+(r&3)+8*(r>>2) in hi[r].
    // The swap is written as inline assembly: the compiler's builtin returns its first result for
    // both halves.  The s_nop pair covers the 18 wait states between the last MFMA pass and a
    // VALU read of its result, which the compiler does not insert in front of inline assembly.
    // (the accumulators are operands of the s_nop statement so that it stays behind the MFMAs)
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc_a), "+v"(acc_b));
    float lo[16], hi[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      lo[r] = acc_a[r];
      hi[r] = acc_b[r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(lo[r]), "+v"(hi[r]));

    const int32_t item0 = t * TI;
    // rows past the end of the item table never rank
    if (item0 + TI > n_items) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (item0 + (r & 3) + 8 * (r >> 2) >= n_items) lo[r] = -INFINITY;
        if (item0 + 4 + (r & 3) + 8 * (r >> 2) >= n_items) hi[r] = -INFINITY;
      }
    }
    // training interactions of this user inside the tile get the mask score
    if (__any(next_m < item0 + TI)) {
      uint32_t mbits = 0;
      while (next_m < item0 + TI) {
        mbits |= 1u << (next_m - item0);
        ++mcur;
        next_m = mcur < mend ? mask_items[mcur] : INT32_MAX;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // sign-extended bit (0 or all ones) selects between the score and the mask score
        const uint32_t ml = (uint32_t)__builtin_amdgcn_sbfe((int32_t)mbits, (r & 3) + 8 * (r >> 2), 1);
        const uint32_t mh = (uint32_t)__builtin_amdgcn_sbfe((int32_t)mbits, 4 + (r & 3) + 8 * (r >> 2), 1);
        lo[r] = __uint_as_float((ml & __float_as_uint(kMasked)) | (~ml & __float_as_uint(lo[r])));
        hi[r] = __uint_as_float((mh & __float_as_uint(kMasked)) | (~mh & __float_as_uint(hi[r])));
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      consider4(lo[4 * g], lo[4 * g + 1], lo[4 * g + 2], lo[4 * g + 3], item0 + 8 * g);
      consider4(hi[4 * g], hi[4 * g + 1], hi[4 * g + 2], hi[4 * g + 3], item0 + 8 * g + 4);
    }

    // The DMA of the next tile must have landed before this wavefront reads it.  The wait is the
    // builtin (vmcnt(0), other counters untouched), not inline assembly: the compiler's own
    // wait-count bookkeeping sees it and does not add a second, earlier wait in the next tile.
    __builtin_amdgcn_s_waitcnt(0x0F70);
  };

  if (t_lo < t_hi) issue_tile(t_lo, tile0[wave]);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  for (int32_t t = t_lo; t < t_hi; t += 2) {
    process_tile(t, tile0[wave], tile1[wave]);
    if (t + 1 < t_hi) process_tile(t + 1, tile1[wave], tile0[wave]);
  }
  flush();

  if (user_ok) {
    const int64_t base = ((int64_t)user * nsplit + split) * KPAD;
#pragma unroll
    for (int j = 0; j < KPAD; ++j) {
      part_s[base + j] = list.s[j];
      part_id[base + j] = list.id[j] < 0 ? -1 : list.id[j] + item_id_offset;
    }
  }
}

// Top-K of the rows of a score matrix that already exists (evaluate_all_at_k called with
// precomputed scores).  32 lanes per row, lane j scans columns j, j+32, ... (coalesced 128-B
// reads) into its own sorted list; the 32 lists of a row are merged by topk_merge_kernel.
constexpr int ROWS_LANES = 32;

template <int KPAD>
__global__ __launch_bounds__(256) void rows_topk_kernel(const float *__restrict__ scores, int32_t n_rows,
                                                        int32_t n_items, int64_t ld,
                                                        float *__restrict__ part_s,
                                                        int32_t *__restrict__ part_id) {
  const int sub = threadIdx.x % ROWS_LANES;
  const int32_t row = blockIdx.x * (256 / ROWS_LANES) + threadIdx.x / ROWS_LANES;
  if (row >= n_rows) return;
  LaneList<KPAD> list;
  list.init();
  const float *src = scores + (int64_t)row * ld;
  for (int32_t i = sub; i < n_items; i += ROWS_LANES) {
    const float v = src[i];
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

__global__ __launch_bounds__(256) void topk_merge_kernel(
    const float *__restrict__ scores, const int32_t *__restrict__ ids,
    int32_t n_users, int32_t n, int32_t K, float *__restrict__ out_s,
    int32_t *__restrict__ out_id) {
  __shared__ float cs[4][MERGE_MAX];
  __shared__ int32_t ci[4][MERGE_MAX];
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  const int32_t user = blockIdx.x * 4 + wave;
  if (user >= n_users) return;  // wave-uniform; no block barrier below
  for (int c = lane; c < n; c += kWave) {
    cs[wave][c] = scores[(int64_t)user * n + c];
    ci[wave][c] = ids[(int64_t)user * n + c];
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes are done
  for (int c = lane; c < n; c += kWave) {
    const float sc = cs[wave][c];
    const int32_t ic = ci[wave][c];
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += beats(cs[wave][j], ci[wave][j], j, sc, ic, c) ? 1 : 0;
    if (rank < K) {
      out_s[(int64_t)user * K + rank] = ic >= 0 ? sc : -INFINITY;
      out_id[(int64_t)user * K + rank] = ic;
    }
  }
}

// ---------------------------------------------------------------- M -------
constexpr int MAXK = 8;  // k values per call

struct KList {
  int32_t k[MAXK];
};

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
    uint32_t rel = 0;  // bit j: j-th ranked item is a target item
    for (int j = 0; j < kmax; ++j) {
      const int32_t it = topk_ids[(int64_t)u * id_stride + j];
      int32_t lo = tb, hi = te;
      while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (tgt_items[mid] < it)
          lo = mid + 1;
        else
          hi = mid;
      }
      if (it >= 0 && lo < te && tgt_items[lo] == it) rel |= 1u << j;
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
          const float rj = (float)((rel >> j) & 1u);
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

__global__ void rank_metrics_final_kernel(const double *__restrict__ part, int32_t n_part,
                                          int32_t nk, double denom, float *__restrict__ out,
                                          double *__restrict__ sums_out) {
  const int i = threadIdx.x;  // one thread per (k, metric), sequential over blocks
  if (i >= nk * 6) return;
  double s = 0.0;
  for (int32_t b = 0; b < n_part; ++b) s += part[(int64_t)b * (MAXK * 6) + i];
  if (sums_out) sums_out[i] = s;
  out[i] = (float)(s / denom);
}

template <int KPAD>
int launch_score_topk(int DC, int grid, hipStream_t st, const float *Uemb, int32_t n_users,
                      const float *Iemb, int32_t n_items, const int32_t *mask_ptr,
                      const int32_t *mask_items, int32_t nsplit, int32_t off, float *ps,
                      int32_t *pi) {
  switch (DC) {
    case 1:
      score_topk_kernel<KPAD, 1><<<grid, 256, 0, st>>>(Uemb, n_users, Iemb, n_items, mask_ptr,
                                                        mask_items, nsplit, off, ps, pi);
      break;
    case 2:
      score_topk_kernel<KPAD, 2><<<grid, 256, 0, st>>>(Uemb, n_users, Iemb, n_items, mask_ptr,
                                                        mask_items, nsplit, off, ps, pi);
      break;
    default:
      return rk::fail(RK_ERR_UNSUPPORTED, "rk_score_topk_f32: embedding width %d not built (64/128)",
                      64 * DC);
  }
  RK_CHECK_LAUNCH("score_topk_kernel");
  return RK_OK;
}

}  // namespace

extern "C" {

size_t rk_score_topk_ws_bytes(int32_t n_users, int32_t n_items, int32_t D, int32_t K) {
  (void)D;
  if (n_users <= 0 || n_items <= 0 || K <= 0 || K > 32) return 0;
  const int nsplit = choose_nsplit(n_users, n_items);
  return (size_t)n_users * nsplit * kpad_for(K) * (sizeof(float) + sizeof(int32_t));
}

int rk_topk_merge_f32(const float *scores, const int32_t *ids, int32_t n_users, int32_t P,
                      int32_t Kin, int32_t K, float *out_scores, int32_t *out_ids, void *stream) {
  RK_CHECK_ARG(n_users >= 0 && P > 0 && Kin > 0 && K > 0, "bad sizes");
  if (n_users == 0) return RK_OK;
  RK_CHECK_ARG(scores && ids && out_scores && out_ids, "null pointer");
  RK_CHECK_ARG((int64_t)P * Kin <= MERGE_MAX, "too many candidates per user (%lld > %d)",
               (long long)P * Kin, MERGE_MAX);
  RK_CHECK_ARG(K <= P * Kin, "K larger than the candidate count");
  topk_merge_kernel<<<(int)rk::ceil_div(n_users, 4), 256, 0, (hipStream_t)stream>>>(
      scores, ids, n_users, P * Kin, K, out_scores, out_ids);
  RK_CHECK_LAUNCH("topk_merge_kernel");
  return RK_OK;
}

int rk_score_topk_f32(const float *Uemb, int32_t n_users, const float *Iemb, int32_t n_items,
                      int32_t D, const int32_t *mask_ptr, const int32_t *mask_items, int32_t K,
                      int32_t item_id_offset, float *out_scores, int32_t *out_ids, void *ws,
                      size_t ws_bytes, void *stream) {
  RK_CHECK_ARG(n_users > 0 && n_items > 0, "empty score matrix");
  // evalue/metric.py:9-10
  RK_CHECK_ARG(K >= 1 && K <= n_items, "k must be in the range [1, num_items]");
  if (K > 32) return rk::fail(RK_ERR_UNSUPPORTED, "rk_score_topk_f32: K=%d above the built maximum 32", K);
  RK_CHECK_ARG(D > 0 && D % 64 == 0, "embedding width must be a multiple of 64");
  RK_CHECK_ARG(Uemb && Iemb && out_scores && out_ids && ws, "null pointer");
  RK_CHECK_ARG((mask_ptr == nullptr) == (mask_items == nullptr), "mask_ptr and mask_items go together");
  RK_CHECK_ARG(rk::aligned16(Uemb) && rk::aligned16(Iemb), "embeddings must be 16-byte aligned");
  RK_CHECK_ARG(ws_bytes >= rk_score_topk_ws_bytes(n_users, n_items, D, K), "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int nsplit = choose_nsplit(n_users, n_items);
  const int kp = kpad_for(K);
  const int grid = (int)rk::ceil_div(n_users, UPB) * nsplit;
  float *ps = (float *)ws;
  int32_t *pi = (int32_t *)(ps + (size_t)n_users * nsplit * kp);
  int rc;
  switch (kp) {
    case 4: rc = launch_score_topk<4>(D / 64, grid, st, Uemb, n_users, Iemb, n_items, mask_ptr, mask_items, nsplit, item_id_offset, ps, pi); break;
    case 8: rc = launch_score_topk<8>(D / 64, grid, st, Uemb, n_users, Iemb, n_items, mask_ptr, mask_items, nsplit, item_id_offset, ps, pi); break;
    case 12: rc = launch_score_topk<12>(D / 64, grid, st, Uemb, n_users, Iemb, n_items, mask_ptr, mask_items, nsplit, item_id_offset, ps, pi); break;
    case 16: rc = launch_score_topk<16>(D / 64, grid, st, Uemb, n_users, Iemb, n_items, mask_ptr, mask_items, nsplit, item_id_offset, ps, pi); break;
    case 20: rc = launch_score_topk<20>(D / 64, grid, st, Uemb, n_users, Iemb, n_items, mask_ptr, mask_items, nsplit, item_id_offset, ps, pi); break;
    case 24: rc = launch_score_topk<24>(D / 64, grid, st, Uemb, n_users, Iemb, n_items, mask_ptr, mask_items, nsplit, item_id_offset, ps, pi); break;
    case 28: rc = launch_score_topk<28>(D / 64, grid, st, Uemb, n_users, Iemb, n_items, mask_ptr, mask_items, nsplit, item_id_offset, ps, pi); break;
    default: rc = launch_score_topk<32>(D / 64, grid, st, Uemb, n_users, Iemb, n_items, mask_ptr, mask_items, nsplit, item_id_offset, ps, pi); break;
  }
  if (rc != RK_OK) return rc;
  return rk_topk_merge_f32(ps, pi, n_users, nsplit, kp, K, out_scores, out_ids, stream);
}

size_t rk_rows_topk_ws_bytes(int32_t n_rows, int32_t K) {
  if (n_rows <= 0 || K <= 0 || K > 32) return 0;
  return (size_t)n_rows * ROWS_LANES * kpad_for(K) * (sizeof(float) + sizeof(int32_t));
}

int rk_rows_topk_f32(const float *scores, int32_t n_rows, int32_t n_items, int64_t ld, int32_t K,
                     float *out_scores, int32_t *out_ids, void *ws, size_t ws_bytes, void *stream) {
  RK_CHECK_ARG(n_rows >= 0 && n_items > 0 && ld >= n_items, "bad score matrix shape");
  // evalue/metric.py:9-10
  RK_CHECK_ARG(K >= 1 && K <= n_items, "k must be in the range [1, num_items]");
  if (K > 32) return rk::fail(RK_ERR_UNSUPPORTED, "rk_rows_topk_f32: K=%d above the built maximum 32", K);
  if (n_rows == 0) return RK_OK;
  RK_CHECK_ARG(scores && out_scores && out_ids && ws, "null pointer");
  RK_CHECK_ARG(ws_bytes >= rk_rows_topk_ws_bytes(n_rows, K), "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int kp = kpad_for(K);
  float *ps = (float *)ws;
  int32_t *pi = (int32_t *)(ps + (size_t)n_rows * ROWS_LANES * kp);
  const int grid = (int)rk::ceil_div(n_rows, 256 / ROWS_LANES);
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
      RK_CHECK_ARG(kl.k[i] >= 1 && kl.k[i] <= 32 && kl.k[i] <= id_stride,
                   "k=%d outside [1, min(32, id_stride)]", kl.k[i]);
      if (kl.k[i] > kmax) kmax = kl.k[i];
    }
  }
  hipStream_t st = (hipStream_t)stream;
  const int nb = (int)rk::ceil_div(n_users, 256);
  rank_metrics_kernel<<<nb, 256, 0, st>>>(topk_ids, n_users, id_stride, tgt_ptr, tgt_items, kl, nk,
                                           kmax, discounts, (double *)ws);
  RK_CHECK_LAUNCH("rank_metrics_kernel");
  rank_metrics_final_kernel<<<1, 64, 0, st>>>((const double *)ws, nb, nk, (double)mean_denominator,
                                               out, sums_out);
  RK_CHECK_LAUNCH("rank_metrics_final_kernel");
  return RK_OK;
}

}  // extern "C"
