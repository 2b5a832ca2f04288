// Shared by eval.hip (exact fp32 scores) and eval_filter.hip (fp16 filter scores): the fused
// score + mask + top-K kernel.  See eval.hip for the description of the kernel.
#pragma once
#include "common.h"

namespace {

using rk::kWave;
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int TI = 32;     // items per LDS tile (MFMA rows)
constexpr int UPW = 64;    // users per wavefront
constexpr int UPB = 256;   // users per workgroup (4 independent wavefronts)
constexpr int QS = 8;      // queue slots per lane
constexpr float kMasked = -1e10f;  // evalue/evaluator.py:55

// Users with very long training lists are not masked by the filter pass at all and always handed
// to the exact kernel: a lane that meets three or more masked items in a tile waits for a memory
// round trip per item, behind the tile prefetch, and one such lane holds up its whole workgroup --
// with one slice per user tile, the whole launch (Elec: one user with 45k training items made a
// 3 ms launch take 6 ms, and is why long tables used to be cut into two slices).  The exact kernel
// meets the same users in the compacted launch, cut into hundreds of item slices.  "Very long" is
// more than heavy_thresh[0] items, a device value computed per call (heavy_threshold() in
// eval_filter.hip): at least a quarter of an item per tile of the table, and high enough that no
// more than kHeavyUsers users are above it, whatever the data set looks like.
constexpr int kHeavyUsers = 512;

// list length the kernel is compiled for: K rounded up to a multiple of 4 (an insertion costs one
// step per list entry, and a longer list also lowers the threshold that filters candidates)
__host__ __device__ inline int kpad_for(int K) { return K <= 4 ? 4 : (K + 3) / 4 * 4; }

// Item slices per user tile.  Every slice restarts its lists from nothing and pays the K*ln(n/K)
// insertions of a streaming top-K again, so fewer slices are cheaper; what slices buy is a grid
// that fills the chip evenly.  The estimate below is launch rounds (wgs_per_cu workgroups fit a CU)
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

// workgroups of score_topk_kernel a CU holds (registers: 512 / waves per SIMD each)
__host__ __device__ constexpr int wg_per_cu(int kpad, int dc, bool filter) {
  return filter ? (dc == 1 ? 3 : 2) : (dc == 1 && kpad <= 24 ? 2 : 1);
}

inline int choose_nsplit(int n_users, int n_items, int wgs_per_cu = 2) {
  const int64_t utiles = rk::ceil_div(n_users, UPB);
  const int64_t tiles = rk::ceil_div(n_items, TI);
  const int64_t slots = wgs_per_cu * (int64_t)cu_count();
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

// First position in [lo, hi) of the ascending list `items` whose entry is >= first.  A kernel's
// prologue waits for this chain of dependent loads before its first tile, so it is a 4-ary search
// (three independent probes per round: half the rounds of a bisection), and no search at all for a
// slice that starts at the table's first item.
__device__ __forceinline__ int32_t mask_lower_bound(const int32_t *__restrict__ items, int32_t lo, int32_t hi,
                                                    int32_t first) {
  if (first <= 0) return lo;
  while (lo < hi) {
    const int32_t n = hi - lo;
    const int32_t p1 = lo + (n >> 2), p2 = lo + (n >> 1), p3 = lo + n - 1 - (n >> 2);
    const int32_t v1 = items[p1], v2 = items[p2], v3 = items[p3];
    if (v3 < first) {
      lo = p3 + 1;
    } else if (v2 < first) {
      lo = p2 + 1;
      hi = p3;
    } else if (v1 < first) {
      lo = p1 + 1;
      hi = p2;
    } else {
      hi = p1;
    }
  }
  return lo;
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
      // (operands read first: a lazily evaluated ?: becomes a load through a pointer phi, which
      // keeps part of the list in memory)
      const int32_t above = id[j - 1], here = id[j];
      id[j] = c_lo ? above : (c_hi ? vid : here);
      s[j] = __builtin_amdgcn_fmed3f(s[j - 1], s[j], v);
      c_hi = c_lo;
    }
    const int32_t first = id[0];
    id[0] = c_hi ? vid : first;
    s[0] = fmaxf(s[0], v);
  }
};

// FILTER = false: Uemb / Iemb are the fp32 embeddings and a score is the exact fp32 fma chain.
// FILTER = true: they point at fp16 copies of the rows (D halfs per row, scaled by powers of two
// into the fp16 range) and a score is their product on v_mfma_f32_32x32x16_f16 -- a filter score
// within 2^-10 * |u| * |i| of the exact one (times the scale), used to pick candidates that are
// then re-scored exactly.  The fp16 MFMA runs on the matrix pipe, which (unlike the fp32-input
// MFMA) the selection's vector instructions overlap with, and one of them covers 16 k where the
// fp32 form needs eight instructions of twice the length.
// skip (may be NULL): users with a non-zero flag need no result; a wavefront whose 64 users are
// all flagged returns at once.
// user_list / user_count / gate_cap select which users the launch works on:
//   user_list == NULL, user_count == NULL: users 0..n_users-1 (rows of Uemb, of the mask, of part_*).
//   user_list == NULL, user_count != NULL: the same, but only if *user_count > gate_cap; otherwise
//     every wavefront returns at once (the launch is the overflow path of the compacted one).
//   user_list != NULL: position p < *user_count stands for user user_list[p] (row of Uemb and of
//     the mask); part_* rows are positions; nothing runs if *user_count > gate_cap.  n_users is
//     then the number of positions the grid covers.
#ifdef RK_STAMP
__device__ inline int n_tiles_stamp(int a, int b) { return b - a; }
// Diagnostic build: cycles of wavefront lifetime per phase of the filter loop, summed over all
// wavefronts (s_memtime around the phases; the stamps themselves cost about a tenth).
__device__ unsigned long long g_stamp[10];
#define RK_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define RK_ACC(i, a, b) st_acc[i] += (b) - (a)
#else
#define RK_T(var)
#define RK_ACC(i, a, b)
#endif

// CEIL (exact scores only): a later page of a ranking longer than one list.  Only items that rank
// strictly AFTER (ceil_s[u], ceil_id[u]) in the order (score descending, id ascending) -- the last
// entry of the page before -- take part; ids are local to this table, like the lists'.
template <int KPAD, int DC, bool FILTER, bool CEIL = false>
__global__ __launch_bounds__(256, wg_per_cu(KPAD, DC, FILTER)) void score_topk_kernel(
    const float *__restrict__ Uemb, int32_t n_users,
    const float *__restrict__ Iemb, int32_t n_items,
    const int32_t *__restrict__ mask_ptr, const int32_t *__restrict__ mask_items,
    int32_t nsplit, int32_t item_id_offset, const int32_t *__restrict__ skip,
    const int32_t *__restrict__ user_list, const int32_t *__restrict__ user_count, int32_t gate_cap,
    float *__restrict__ part_s, int32_t *__restrict__ part_id, const int32_t *__restrict__ heavy_thresh,
    const float *__restrict__ ceil_s = nullptr, const int32_t *__restrict__ ceil_id = nullptr) {
  static_assert(!(CEIL && FILTER), "pages are ranked by the exact kernel");
  constexpr int D = 64 * DC;
  // Exact scores: every wavefront streams its own copy of the item tiles through a private pair
  // of buffers (the MFMA chain is the bound, nothing is shared, no barriers).
  // Filter scores: fast enough to be bound by the tile stream otherwise (four private copies of
  // the table per workgroup out of L2 / Infinity Cache, one tile of lookahead against ~1 us of
  // latency), so the four wavefronts share a ring of two stages of RN tiles, each wavefront
  // loading a quarter of every tile, with one workgroup barrier per stage.
  constexpr int RN = 4;   // tiles per stage
  __shared__ __attribute__((aligned(1024))) float tile0[FILTER ? 1 : 4][FILTER ? 1 : TI * D];
  __shared__ __attribute__((aligned(1024))) float tile1[FILTER ? 1 : 4][FILTER ? 1 : TI * D];
  __shared__ __attribute__((aligned(1024))) float ring[FILTER ? 2 * RN : 1][FILTER ? TI * D / 2 : 1];
  // candidate queues: [wave][0] scores, [wave][1] item ids (one array, so that a score and its
  // id, a fixed distance apart, go out in one ds_write2)
  __shared__ uint32_t queue[4][2][QS][kWave];

  const int tid = threadIdx.x;
  const int wave = tid / kWave, lane = tid % kWave;
  const int col = lane & 31, half = lane >> 5;
  const int32_t split = blockIdx.x % nsplit;
  const int32_t utile = blockIdx.x / nsplit;
  const int32_t ubase = utile * UPB + wave * UPW;
  const int32_t user = ubase + lane;          // the position whose list this lane keeps
  // the filter's wavefronts share tiles and barriers, all of them stay; it takes neither a skip
  // list nor a user list
  if constexpr (!FILTER) {
    if (user_count != nullptr) {
      const int32_t cnt = user_count[0];
      if (user_list == nullptr ? cnt <= gate_cap : cnt > gate_cap) return;
      if (user_list != nullptr) n_users = min(n_users, cnt);
    }
    if (ubase >= n_users) return;             // wave-uniform
  }
  const bool user_ok = user < n_users;
  // rows of Uemb / of the mask behind positions ubase+col, ubase+32+col and ubase+lane
  int64_t row_a = ubase + col, row_b = ubase + 32 + col, row_l = user;
  if constexpr (!FILTER) {
    if (user_list != nullptr) {
      row_a = ubase + col < n_users ? user_list[ubase + col] : 0;
      row_b = ubase + 32 + col < n_users ? user_list[ubase + 32 + col] : 0;
      row_l = user_ok ? user_list[user] : 0;
    }
    if (skip != nullptr && __all(!user_ok || skip[user_ok ? row_l : 0] != 0)) return;   // wave-uniform
  }

  const int32_t tiles_total = (n_items + TI - 1) / TI;
  const int32_t t_lo = (int32_t)((int64_t)tiles_total * split / nsplit);
  const int32_t t_hi = (int32_t)((int64_t)tiles_total * (split + 1) / nsplit);

  // user fragments B[k][j = col]: chain A multiplies users ubase+col, chain B users ubase+32+col.
  // fp32: this lane holds k = 64c + 32*half + m.  fp16 rows: 16 k per MFMA, this lane holds
  // k = 16*kb + 8*half + 0..7.
  constexpr int KB = D / 16;
  float bfa[FILTER ? 1 : DC][FILTER ? 1 : 32], bfb[FILTER ? 1 : DC][FILTER ? 1 : 32];
  half8 uha[FILTER ? KB : 1], uhb[FILTER ? KB : 1];
  if constexpr (FILTER) {
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      uint4 ha = zero, hb = zero;
      const int64_t off = (2 * kb + half) * 16;   // bytes
      if (ubase + col < n_users)
        ha = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(Uemb) + row_a * (D * 2) + off);
      if (ubase + 32 + col < n_users)
        hb = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(Uemb) + row_b * (D * 2) + off);
      uha[kb] = __builtin_bit_cast(half8, ha);
      uhb[kb] = __builtin_bit_cast(half8, hb);
    }
  } else {
#pragma unroll
    for (int c = 0; c < DC; ++c)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
        const int64_t off = 64 * c + 32 * half + 4 * q;
        if (ubase + col < n_users)
          va = *reinterpret_cast<const float4 *>(Uemb + row_a * D + off);
        if (ubase + 32 + col < n_users)
          vb = *reinterpret_cast<const float4 *>(Uemb + row_b * D + off);
        bfa[FILTER ? 0 : c][FILTER ? 0 : 4 * q + 0] = va.x; bfa[FILTER ? 0 : c][FILTER ? 0 : 4 * q + 1] = va.y;
        bfa[FILTER ? 0 : c][FILTER ? 0 : 4 * q + 2] = va.z; bfa[FILTER ? 0 : c][FILTER ? 0 : 4 * q + 3] = va.w;
        bfb[FILTER ? 0 : c][FILTER ? 0 : 4 * q + 0] = vb.x; bfb[FILTER ? 0 : c][FILTER ? 0 : 4 * q + 1] = vb.y;
        bfb[FILTER ? 0 : c][FILTER ? 0 : 4 * q + 2] = vb.z; bfb[FILTER ? 0 : c][FILTER ? 0 : 4 * q + 3] = vb.w;
      }
  }

  // train-mask cursor of this lane's user (item ids ascending)
  int32_t mcur = 0, mend = 0;
  if (mask_ptr && user_ok) {
    mcur = mask_ptr[row_l];
    mend = mask_ptr[row_l + 1];
    if constexpr (FILTER) {
      if (heavy_thresh != nullptr && mend - mcur > heavy_thresh[0]) mcur = mend;   // left to the exact kernel (kHeavyUsers)
    }
    mcur = mask_lower_bound(mask_items, mcur, mend, t_lo * TI);   // skip entries in front of this slice
  }
  // The cursor holds the next TWO masked items: the entry consumed in a tile was loaded at
  // least one consumption earlier, so the loop below does not sit out a memory round trip
  // (behind every DMA in flight -- the counter is in order) for each masked item of the tile.
  int32_t next_m = mcur < mend ? mask_items[mcur] : INT32_MAX;
  int32_t next2_m = mcur + 1 < mend ? mask_items[mcur + 1] : INT32_MAX;
  float page_s = INFINITY;
  int32_t page_id = -1;
  if constexpr (CEIL) {
    if (user_ok) {
      page_s = ceil_s[row_l];
      page_id = ceil_id[row_l];
    }
  }

#ifdef RK_STAMP
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  LaneList<KPAD> list;
  list.init();
  float thr = -INFINITY;
  int qc = 0;

  // Tile staging by LDS-DMA: one wave-instruction writes 64 x 16 B = 1 KiB of LDS linearly from
  // per-lane global addresses.  Rows are RB bytes; slot s of row r holds 16-byte chunk
  // s ^ swz(r) of that item's row, where swz spreads the rows that share LDS banks (16 rows of
  // 256 B: r & 15; 128-byte rows alias in pairs: (r >> 1) & 7), so that the b128 fragment reads
  // of 32 rows at one chunk index are conflict-free.
  constexpr int RB = FILTER ? D * 2 : D * 4;   // bytes per row
  constexpr int CPR = RB / 16;                 // 16-byte chunks per row
  constexpr int SW = CPR < 16 ? CPR : 16;      // chunks a swizzle permutes
  constexpr int RPP = 64 / CPR;                // rows per 1-KiB piece
  constexpr int PIECES = TI / RPP;
  static_assert(RB == 128 || RB % 256 == 0, "swizzle written for 128-byte rows and multiples of 256");
  auto swz = [](int r) { return RB == 128 ? (r >> 1) & 7 : r & 15; };
  // A full tile is one contiguous block of the table: the per-lane byte offset of piece p is a
  // loop-invariant pattern XOR a per-piece constant (swz(p*RPP + l) = swz(p*RPP) ^ swz(l) for
  // l < RPP: the two never set the same bits), so a piece costs one XOR besides the DMA itself.
  // Exact: a wavefront issues all pieces of its private tile.  Filter: piece range NP*wave..
  constexpr int NP = FILTER ? PIECES / 4 : PIECES;
  static_assert(NP >= 1, "a tile has at least one piece per wavefront");
  const int pbase = FILTER ? __builtin_amdgcn_readfirstlane(wave) * NP : 0;
  const int l_row = lane / CPR, l_slot = lane % CPR;
  uint32_t lane_off = (uint32_t)(l_row * RB + (((l_slot & ~(SW - 1)) | ((l_slot & (SW - 1)) ^ swz(l_row))) * 16));
  // The DMA instructions are inline assembly, and on purpose: the compiler's wait-count pass
  // tracks LDS-DMA per LDS array and makes later reads of an array wait for "its" DMAs, but with
  // four buffers and the loads of the mask cursor in between it falls back to vmcnt(0) in front
  // of some tile reads, which drains the whole prefetch ring.  Written this way the pass knows
  // nothing of the DMAs; the kernel's own s_waitcnt (and, for the shared ring, barrier) in front
  // of every first read of a tile is the only synchronisation, and it is sufficient.  More
  // operations in the in-order counter than the compiler knows of only make its own vmcnt(N)
  // waits stronger.
  // Hazard rules these two statements rely on: (1) a scalar write of M0 needs one wait state before
  // an LDS-DMA reads it: the s_nop 0 inside the statement; (2) completion is counted by vmcnt in
  // issue order: the kernel's own s_waitcnt vmcnt(N) (the builtin, visible to the compiler) stands
  // in front of the first read of every tile; (3) the LDS range a DMA overwrites has been read
  // completely before it is issued: an lgkmcnt(0) (and, for the shared ring, the stage barrier)
  // precedes every issue.  No VGPR destination: nothing for the register allocator to move early.
  auto dma_piece = [&](const char *base, uint32_t off, uint32_t lds_addr) {   // uniform base + lane offset
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 : : "v"(off), "s"(base), "s"(lds_addr) : "memory", "m0");
  };
  auto dma_piece_lane = [&](const char *addr, uint32_t lds_addr) {            // per-lane address
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 : : "v"(addr), "s"(lds_addr) : "memory", "m0");
  };
  auto issue_tile = [&](int32_t t, float *dst) {
    asm volatile("" : "+v"(lane_off));  // keep the per-piece offsets from being hoisted into registers
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(
        (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)dst + (uint32_t)pbase * 1024u);
    if ((t + 1) * TI <= n_items) {
      const char *tile_base = reinterpret_cast<const char *>(Iemb) + (int64_t)t * TI * RB + (int64_t)pbase * 1024;
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int p = pbase + j;
        const uint32_t off = (lane_off ^ (uint32_t)(swz(p * RPP) * 16)) + (uint32_t)(j * 1024);
        dma_piece(tile_base, off, lds0 + (uint32_t)(j * 1024));
      }
      return;
    }
    // the last, partial tile: rows past the end of the table re-read the last item (their scores
    // are discarded after the multiplication); same offset pattern, row clamped per lane
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int p = pbase + j;
      const uint32_t off = lane_off ^ (uint32_t)(swz(p * RPP) * 16);
      const int32_t item = min(t * TI + p * RPP + (int32_t)(off / RB), n_items - 1);
      dma_piece_lane(reinterpret_cast<const char *>(Iemb) + (int64_t)item * RB + (off & (RB - 1)),
                     lds0 + (uint32_t)(j * 1024));
    }
  };
  // The queue is read and written through inline assembly.  The compiler's wait-count pass makes
  // every LDS access that does not alias a pending LDS-DMA wait for ALL pending DMAs (vmcnt(0)),
  // which would stall each queue access on the tile prefetch issued just before; it does not look
  // into assembly.  LDS operations of a wavefront execute in order, so the writes need no wait;
  // the reads carry their own.  The statements are volatile, which keeps them in program order
  // among themselves, and deliberately without a "memory" clobber: with one, the optimiser must
  // assume they modify the lists behind the lambdas' references and leaves part of a list in
  // memory.
  const uint32_t qbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)&queue[wave][0][0][lane];
  constexpr int kIdPlane = QS * kWave * 4;   // bytes from a score to its id
  static_assert(kIdPlane == 8 * 256, "ds_write2st64 offset1");
  auto flush = [&]() {
    // Bottom-tested with a single exit (an empty first round inserts -inf, a no-op): any other
    // shape makes the loop carry part of the list in two sets of registers, with a copy per entry
    // and round.
    RK_T(f0);
    int q = 0;
#pragma unroll 1
    do {
      const bool has = q < qc;
      uint32_t qs, qid;
      // (self-contained: its own lgkmcnt(0) inside the statement, early-clobber outputs; LDS
      // operations of a wavefront complete in order, so it sees the ds_write2 statements before it)
      asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:2048\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(qs), "=&v"(qid) : "v"(qbase + (uint32_t)q * 256u));
      list.insert(has ? __uint_as_float(qs) : -INFINITY, (int32_t)qid);
      ++q;
    } while (__any(q < qc));
    qc = 0;
    thr = list.s[KPAD - 1];
    RK_T(f1);
    RK_ACC(6, f0, f1);
  };
  // Four scores of this lane's user (ascending item ids).  Late in the stream hardly any score
  // beats its lane's threshold, so the wavefront first asks whether any lane has a candidate at
  // all; only then are the four scores parked, branch free: the store always happens and the
  // slot advances only for a candidate.
  auto consider4 = [&](float v0, float v1, float v2, float v3, int32_t id0) {
    // (v_max3 written out: fmaxf() on values the compiler does not know to be canonical costs a
    // canonicalising v_max per operand; a NaN operand is ignored either way.  The operands are results
    // of the half swap, a vector instruction: no wait states are owed between the two.)
    float m;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(v0), "v"(v1), "v"(v2));
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(v3));
    if (!__any(m > thr)) return;
    RK_T(c0);
    const float v[4] = {v0, v1, v2, v3};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // (address and data are vector-ALU results: ordinary register dependencies, interlocked by the hardware)
      asm volatile("ds_write2st64_b32 %0, %1, %2 offset1:8"
                   : : "v"(qbase + (uint32_t)qc * 256u), "v"(v[i]), "v"(id0 + i));
      qc += v[i] > thr ? 1 : 0;
    }
    // a queue holds QS entries: insert when some lane could overflow within the next 4 scores
    RK_T(c1);
    RK_ACC(7, c0, c1);
    if (__any(qc > QS - 4)) flush();
  };

  // fp32 rows: slot 16c + ((8*half + q) ^ (col & 15)) of row col; q < 8 only touches bits the other
  // terms leave alone, so the eight addresses are one per-lane value XOR q*16.  fp16 rows: the
  // 16-byte chunk of k-block kb is 2*kb + half; again the lane part (half, the swizzle) and the
  // static part (2*kb) combine by XOR.
  uint32_t frag_off = FILTER ? (uint32_t)(col * RB + (half ^ swz(col)) * 16)
                             : (uint32_t)(col * RB + ((8 * half) ^ (col & 15)) * 16);
  // Lane l holds rows (r&3)+8*(r>>2)+4*half of user col in acc_a and of user 32+col in acc_b.
  // After the swap every lane holds, for ITS user (ubase + lane), rows (r&3)+8*(r>>2) in lo[r]
  // and rows 4+(r&3)+8*(r>>2) in hi[r].
  // (rk::swap_halves32 is the compiler's builtin: the wait states between the last MFMA pass and
  // this first vector read of its result are the compiler's to place)
  auto swap_halves = [&](floatx16 &acc_a, floatx16 &acc_b, float (&lo)[16], float (&hi)[16]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      lo[r] = acc_a[r];
      hi[r] = acc_b[r];
      rk::swap_halves32(lo[r], hi[r]);
    }
  };
  // scores of tile t that do not rank as computed: rows past the end of the item table (never),
  // training interactions of this lane's user (the mask score)
  auto range_and_mask = [&](int32_t t, float (&lo)[16], float (&hi)[16]) {
    const int32_t item0 = t * TI;
    if (item0 + TI > n_items) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (item0 + (r & 3) + 8 * (r >> 2) >= n_items) lo[r] = -INFINITY;
        if (item0 + 4 + (r & 3) + 8 * (r >> 2) >= n_items) hi[r] = -INFINITY;
      }
    }
    if (__any(next_m < item0 + TI)) {
      uint32_t mbits = 0;
      while (next_m < item0 + TI) {
        mbits |= 1u << (next_m - item0);
        ++mcur;
        next_m = next2_m;
        next2_m = mcur + 1 < mend ? mask_items[mcur + 1] : INT32_MAX;
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
    if constexpr (CEIL) {
      // (after the mask: a masked item ranks with the mask score, on this page or an earlier one)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int32_t il = item0 + (r & 3) + 8 * (r >> 2), ih = il + 4;
        if (lo[r] > page_s || (lo[r] == page_s && il <= page_id)) lo[r] = -INFINITY;
        if (hi[r] > page_s || (hi[r] == page_s && ih <= page_id)) hi[r] = -INFINITY;
      }
    }
  };
  // the g-th eight scores of a tile (g < 4): rows 8g..8g+3 are lo[4g..], rows 8g+4..8g+7 hi[4g..]
  auto consider_lo = [&](int32_t t, int g, const float (&lo)[16]) {
    consider4(lo[4 * g], lo[4 * g + 1], lo[4 * g + 2], lo[4 * g + 3], t * TI + 8 * g);
  };
  auto consider_hi = [&](int32_t t, int g, const float (&hi)[16]) {
    consider4(hi[4 * g], hi[4 * g + 1], hi[4 * g + 2], hi[4 * g + 3], t * TI + 8 * g + 4);
  };

  if constexpr (FILTER) {
    // The tiles come through a ring of two stages of RN tiles.  Before the first tile of a stage
    // is read every wavefront waits for its own quarters of that stage (issued a whole stage
    // earlier) and meets the others at the barrier; the barrier also says that everybody is done
    // with the previous stage, whose buffers the quarters of the next stage then overwrite.  One
    // barrier per RN tiles rather than per tile: the wavefronts' tile times differ a lot (a list
    // insertion now and then), and every barrier costs the slowest one's.  Past the end of the
    // slice the last tile is loaded again; those copies are never read.
    //
    // The MFMAs of a tile are not overlapped with the selection of the previous one inside the
    // wavefront: that costs 32 more registers, and at this size the third wavefront per SIMD that
    // the registers buy instead hides more.
    floatx16 acc_a, acc_b;
    // the item fragment of k-block kb lives in fi[kb & 1] and is read one k-block ahead of its
    // MFMAs, so that the LDS latency is not waited for
    half8 fi[2];
    auto load_frags = [&](int kb, const float *src) {
      const uint32_t a = frag_off ^ (uint32_t)((2 * kb) * 16);
      fi[kb & 1] = __builtin_bit_cast(half8, *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(src) + a));
    };
    auto issue_stage = [&](int32_t first, int half_idx) {
#pragma unroll
      for (int j = 0; j < RN; ++j) issue_tile(min(first + j, t_hi - 1), ring[half_idx * RN + j]);
    };
    // The user fragments and the mask cursor are global loads too: all of them have to be back
    // before the in-order counter is used to tell the stages apart.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (t_lo < t_hi) issue_stage(t_lo, 0);
    int half_idx = 0;
    for (int32_t tb = t_lo; tb < t_hi; tb += RN, half_idx ^= 1) {
      RK_T(s0);
      __builtin_amdgcn_s_waitcnt(0x0F70);
      RK_T(s1);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      RK_T(s2);
      issue_stage(tb + RN, half_idx ^ 1);
      RK_T(s3);
      RK_ACC(0, s0, s1); RK_ACC(1, s1, s2); RK_ACC(2, s2, s3);
#pragma unroll
      for (int j = 0; j < RN; ++j) {
        const int32_t t = tb + j;
        if (t >= t_hi) break;                      // uniform over the workgroup
        const float *cur = ring[half_idx * RN + j];
        asm volatile("" : "+v"(frag_off));         // as for lane_off: hoisted addresses cost registers
        RK_T(p0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc_a[r] = 0.0f;
          acc_b[r] = 0.0f;
        }
        load_frags(0, cur);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          if (kb + 1 < KB) load_frags(kb + 1, cur);
          acc_a = __builtin_amdgcn_mfma_f32_32x32x16_f16(fi[kb & 1], uha[FILTER ? kb : 0], acc_a, 0, 0, 0);
          acc_b = __builtin_amdgcn_mfma_f32_32x32x16_f16(fi[kb & 1], uhb[FILTER ? kb : 0], acc_b, 0, 0, 0);
        }
        float lo[16], hi[16];
        swap_halves(acc_a, acc_b, lo, hi);
#ifdef RK_STAMP
        asm volatile("" : "+v"(lo[0]), "+v"(hi[15]));
#endif
        RK_T(p1);
        range_and_mask(t, lo, hi);
#ifdef RK_STAMP
        asm volatile("" : "+v"(lo[0]), "+v"(hi[15]));
#endif
        RK_T(p2);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          consider_lo(t, g, lo);
          consider_hi(t, g, hi);
        }
        RK_T(p3);
        RK_ACC(3, p0, p1); RK_ACC(4, p1, p2); RK_ACC(5, p2, p3);
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
#ifdef RK_STAMP
    if (lane == 0) {
      for (int i = 0; i < 8; ++i) atomicAdd(&g_stamp[i], st_acc[i]);
      atomicAdd(&g_stamp[8], (unsigned long long)n_tiles_stamp(t_lo, t_hi));
      atomicAdd(&g_stamp[9], 1ull);
    }
#endif
  } else {
    // one tile: score and select from `cur`, the next tile on its way into `nxt`
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
          const float4 v = *reinterpret_cast<const float4 *>(
              reinterpret_cast<const char *>(cur) + ((frag_off ^ (uint32_t)(q * 16)) + (uint32_t)(c * 256)));
          afrag[4 * q + 0] = v.x;
          afrag[4 * q + 1] = v.y;
          afrag[4 * q + 2] = v.z;
          afrag[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int m = 0; m < 32; ++m) {
          acc_a = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[m], bfa[FILTER ? 0 : c][FILTER ? 0 : m], acc_a, 0, 0, 0);
          acc_b = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[m], bfb[FILTER ? 0 : c][FILTER ? 0 : m], acc_b, 0, 0, 0);
        }
      }
      float lo[16], hi[16];
      swap_halves(acc_a, acc_b, lo, hi);
      range_and_mask(t, lo, hi);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        consider_lo(t, g, lo);
        consider_hi(t, g, hi);
      }
      // The DMA of the next tile must have landed before this wavefront reads it.  The wait is
      // the builtin (vmcnt(0), other counters untouched), not inline assembly: the compiler's own
      // wait-count bookkeeping sees it and does not add a second, earlier wait in the next tile.
      __builtin_amdgcn_s_waitcnt(0x0F70);
    };
    if (t_lo < t_hi) issue_tile(t_lo, tile0[wave]);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (int32_t t = t_lo; t < t_hi; t += 2) {
      process_tile(t, tile0[wave], tile1[wave]);
      if (t + 1 < t_hi) process_tile(t + 1, tile1[wave], tile0[wave]);
    }
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

}  // namespace
