// The candidate filter of the full-ranking evaluation as two streaming passes over the fp16 scores
// (eval_filter.hip launches them; see there for the arithmetic of the filter scores).
//
// A streaming top-K keeps a sorted list per user and pays, per list, K ln(n/K) insertions plus a
// gate on every score against a threshold that is weak while the list fills; with 64 users per
// wavefront some lane has a candidate in most groups of four scores for the first several thousand
// items.  That start-up is paid again by every item slice, which is what short item tables, item
// shards and small user blocks (all of which need slices to fill the chip) spend their time on.
//
// Here no list exists while the scores stream by.
//   Pass A (strata_max_kernel): every user keeps kStrata running maxima, one per stratum of the
//     item table (item position in its 32-item tile x parity of the tile pair): one v_max per score
//     and nothing else.  The strata are disjoint sets of items, so the kc largest of a user's
//     strata maxima are kc different items, and the kc-th largest of them is a LOWER BOUND of the
//     user's kc-th best score -- a nearly tight one: with iid scores about 37 items lie at or above
//     it at kc = 28, whatever the length of the table.  Slices add strata (kStrata per slice).
//   strata_threshold_kernel: that kc-th largest maximum per user.
//   Pass B (collect_kernel): the same stream again (the matrix pipe is idle most of the time in
//     either pass), every score >= the user's threshold is a candidate and is stored to the user's
//     candidate places in memory.  The threshold never moves, so nothing is sorted or inserted and
//     the gate is as strong at the first tile as at the last.
// Every item with a filter score >= threshold is a candidate; all of them are scored exactly
// afterwards (refine_wide_kernel in eval_filter.hip) and the result is certified against the
// threshold.  A user with more than kCandCap candidates (ties, degenerate embeddings) is ranked by
// the exact kernel, like every user the certificate fails for.
#pragma once
#include "topk_kernel.h"

namespace {

constexpr int kStrata = 64;    // running maxima per user and item slice
constexpr int kCandCap = 64;   // candidate places per user and item slice; at most this many per user take part
// the score of a row past the end of the item table: a quiet NaN is ignored by v_max / v_max3 and
// fails every comparison, so such a row neither feeds a stratum maximum nor passes a threshold --
// not even the threshold -inf of a user with fewer than kc finite maxima, which -inf itself would
#define kNoItem __builtin_nanf("")


// Geometry of the fp16 item tiles in LDS (as in score_topk_kernel's filter form): rows of RB bytes,
// 16-byte chunk s of row r stored at slot s ^ swz(r); a tile is PIECES pieces of 1 KiB, each
// wavefront of the workgroup stages NP of them.
template <int DC>
struct HalfTiles {
  static constexpr int D = 64 * DC, RB = D * 2, CPR = RB / 16, SW = CPR < 16 ? CPR : 16, RPP = 64 / CPR,
                       PIECES = TI / RPP, NP = PIECES / 4, KB = D / 16, RN = 4;
  static_assert(RB == 128 || RB % 256 == 0, "swizzle written for 128-byte rows and multiples of 256");
  static_assert(NP >= 1, "a tile has at least one piece per wavefront");
  __device__ static __forceinline__ int swz(int r) { return RB == 128 ? (r >> 1) & 7 : r & 15; }
  __device__ static __forceinline__ uint32_t lane_offset(int lane) {
    const int l_row = lane / CPR, l_slot = lane % CPR;
    return (uint32_t)(l_row * RB + (((l_slot & ~(SW - 1)) | ((l_slot & (SW - 1)) ^ swz(l_row))) * 16));
  }
  __device__ static __forceinline__ uint32_t frag_offset(int col, int half) {
    return (uint32_t)(col * RB + (half ^ swz(col)) * 16);
  }
};

// This wavefront's quarter of item tile t into dst (LDS) by LDS-DMA; see score_topk_kernel for why
// the DMA is inline assembly and what the offsets are.
// LDS-DMA written as inline assembly, as in topk_kernel.h and for the reason given there.  Rules
// relied on: one wait state between the scalar write of M0 and the DMA that reads it (the s_nop 0);
// completion by vmcnt, waited for by the callers' own s_waitcnt (the builtin) + stage barrier in
// front of a stage's first read; every reader of the stage a DMA overwrites has passed that barrier.
template <int DC>
__device__ __forceinline__ void issue_half_tile(const float *__restrict__ Iemb, int32_t n_items, int32_t t,
                                                float *dst, uint32_t &lane_off, int pbase) {
  using G = HalfTiles<DC>;
  asm volatile("" : "+v"(lane_off));
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)dst + (uint32_t)pbase * 1024u);
  if ((t + 1) * TI <= n_items) {
    const char *tile_base = reinterpret_cast<const char *>(Iemb) + (int64_t)t * TI * G::RB + (int64_t)pbase * 1024;
#pragma unroll
    for (int j = 0; j < G::NP; ++j) {
      const int p = pbase + j;
      const uint32_t off = (lane_off ^ (uint32_t)(G::swz(p * G::RPP) * 16)) + (uint32_t)(j * 1024);
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                   : : "v"(off), "s"(tile_base), "s"(lds0 + (uint32_t)(j * 1024)) : "memory", "m0");
    }
    return;
  }
  // the last, partial tile: rows past the end re-read the last item (their scores are discarded)
#pragma unroll
  for (int j = 0; j < G::NP; ++j) {
    const int p = pbase + j;
    const uint32_t off = lane_off ^ (uint32_t)(G::swz(p * G::RPP) * 16);
    const int32_t item = min(t * TI + p * G::RPP + (int32_t)(off / G::RB), n_items - 1);
    const char *addr = reinterpret_cast<const char *>(Iemb) + (int64_t)item * G::RB + (off & (G::RB - 1));
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 : : "v"(addr), "s"(lds0 + (uint32_t)(j * 1024)) : "memory", "m0");
  }
}

// Slices are whole groups of four tiles (one ring stage), so that a tile's place in its stage --
// a compile-time index in the unrolled loop -- tells its stratum.
__host__ __device__ inline void slice_tiles(int32_t n_items, int32_t split, int32_t nsplit, int32_t &t_lo, int32_t &t_hi) {
  const int32_t tiles = (n_items + TI - 1) / TI, groups = (tiles + 3) / 4;
  t_lo = 4 * (int32_t)((int64_t)groups * split / nsplit);
  t_hi = min(tiles, 4 * (int32_t)((int64_t)groups * (split + 1) / nsplit));
}

// State both passes share: user fragments, the lane's mask cursor, staging offsets.
template <int DC>
struct StreamState {
  using G = HalfTiles<DC>;
  half8 uha[G::KB], uhb[G::KB];
  int32_t mcur, mend, next_m, next2_m;
  uint32_t lane_off, frag_off;
  int pbase;

  __device__ __forceinline__ void init(const float *__restrict__ Uemb, int32_t n_users, int32_t ubase, int lane, int wave,
                                       const int32_t *__restrict__ mask_ptr, const int32_t *__restrict__ mask_items,
                                       const int32_t *__restrict__ heavy_thresh, int32_t t_lo) {
    const int col = lane & 31, half = lane >> 5;
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int kb = 0; kb < G::KB; ++kb) {
      uint4 ha = zero, hb = zero;
      const int64_t off = (2 * kb + half) * 16;   // bytes
      if (ubase + col < n_users)
        ha = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(Uemb) + (int64_t)(ubase + col) * (G::D * 2) + off);
      if (ubase + 32 + col < n_users)
        hb = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(Uemb) + (int64_t)(ubase + 32 + col) * (G::D * 2) + off);
      uha[kb] = __builtin_bit_cast(half8, ha);
      uhb[kb] = __builtin_bit_cast(half8, hb);
    }
    const int32_t user = ubase + lane;
    mcur = mend = 0;
    if (mask_ptr && user < n_users) {
      mcur = mask_ptr[user];
      mend = mask_ptr[user + 1];
      if (heavy_thresh != nullptr && mend - mcur > heavy_thresh[0]) mcur = mend;   // left to the exact kernel (kHeavyUsers)
      mcur = mask_lower_bound(mask_items, mcur, mend, t_lo * TI);   // skip entries in front of this slice
    }
    // the cursor holds the next TWO masked items (see score_topk_kernel)
    next_m = mcur < mend ? mask_items[mcur] : INT32_MAX;
    next2_m = mcur + 1 < mend ? mask_items[mcur + 1] : INT32_MAX;
    lane_off = G::lane_offset(lane);
    frag_off = G::frag_offset(col, half);
    pbase = __builtin_amdgcn_readfirstlane(wave) * G::NP;
  }

  // bit i set: item item0 + i is a training item of this lane's user (the cursor moves past them)
  __device__ __forceinline__ uint32_t mask_bits(int32_t item0, const int32_t *__restrict__ mask_items) {
    uint32_t mbits = 0;
    while (next_m < item0 + TI) {
      mbits |= 1u << (next_m - item0);
      ++mcur;
      next_m = next2_m;
      next2_m = mcur + 1 < mend ? mask_items[mcur + 1] : INT32_MAX;
    }
    return mbits;
  }

  // The two MFMA chains of one tile: scores of users ubase + col (a) and ubase + 32 + col (b); the
  // item fragment of k-block kb is read from LDS one k-block ahead of its MFMAs.
  __device__ __forceinline__ void scores(const float *cur, floatx16 &acc_a, floatx16 &acc_b) {
    asm volatile("" : "+v"(frag_off));   // hoisted addresses cost registers
    const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    half8 fi[2];
    auto load_frags = [&](int kb) {
      const uint32_t a = frag_off ^ (uint32_t)((2 * kb) * 16);
      fi[kb & 1] = __builtin_bit_cast(half8, *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(cur) + a));
    };
    load_frags(0);
#pragma unroll
    for (int kb = 0; kb < G::KB; ++kb) {
      if (kb + 1 < G::KB) load_frags(kb + 1);
      acc_a = __builtin_amdgcn_mfma_f32_32x32x16_f16(fi[kb & 1], uha[kb], kb == 0 ? zero : acc_a, 0, 0, 0);
      acc_b = __builtin_amdgcn_mfma_f32_32x32x16_f16(fi[kb & 1], uhb[kb], kb == 0 ? zero : acc_b, 0, 0, 0);
    }
    // Hazard rule: 18 wait states between the last pass of a 32x32 MFMA and a vector-ALU read of
    // its result.  The first readers of these accumulators are single-instruction inline assembly
    // statements (RK_VMAX, RK_MASK_IF_BIT below), in front of which the compiler places none: the
    // s_nop pair (20 wait states) stands here for them.  The accumulators are operands of the
    // statement so that it stays behind the MFMAs.
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc_a), "+v"(acc_b));
  }
};

// v_max in place (fmaxf() on values the compiler does not know to be canonical costs a second
// v_max per operand; a NaN operand is ignored either way)
#define RK_VMAX(dst, src) asm("v_max_f32 %0, %0, %1" : "+v"(dst) : "v"(src))
// x = bit `pos` of `bits` set ? the mask score : x, in two instructions (sign-extended bit, bit-field
// insert); the portable (m & K) | (~m & x) came out as five per score, 164 for a tile
#define RK_MASK_IF_BIT(x, bits, pos)                                                              \
  do {                                                                                            \
    uint32_t rk_m_;                                                                               \
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(rk_m_) : "v"(bits), "n"(pos));                          \
    asm("v_bfi_b32 %0, %1, %2, %0" : "+v"(x) : "v"(rk_m_), "v"(kMasked));                        \
  } while (0)

// The ring both passes stream the item tiles through holds NST stages of RN tiles; NST - 1 stages
// are in flight while one is consumed.  Before a stage is read every wavefront waits for its own
// quarters of it -- the in-order counter may still hold the DMAs of the NST - 2 stages issued
// after it -- and meets the others at the barrier, which also says that everybody is done with the
// stage before, whose buffers the newest stage then overwrites.  Past the end of the slice the
// last tile is loaded again (never read), so every stage is the same number of DMAs.
//
// What bounds these passes (Elec, 192k users, one slice, counters in profiles/): per SIMD the
// vector ALU is busy 1.45 M cycles and the matrix pipe 1.48 M of a 3.5 M-cycle launch, and hardly
// ever together (0.3 M).  Issue is in order: a wavefront that issues its eight MFMAs in a burst
// and then its vector work gets the SUM of the two even with three wavefronts per SIMD, and only
// vector instructions that stand between the MFMAs in program order run in their shadow
// (tools/mfma_valu_probe.hip: 8 MFMAs + 32 v_max per round, three wavefronts per SIMD: 343 ticks
// per round one after the other, 245 = the matrix pipe's time interleaved; at two wavefronts per
// SIMD, which a second accumulator set costs pass A, the interleaved form is no faster than the
// plain one at three).  Pass B was built that way at three wavefronts per SIMD (second accumulator
// set, half swap / group maxima / compares between the k-blocks of the next tile, visiting only
// groups that hold a candidate): correct, and slower, 2.3 -> 3.1 ms at Elec -- 156 vector
// instructions per wave-tile (copies of the finished set for the visit, ~44 on average for the
// mask path, ~30 per visit) make it bound by the vector ALU whatever overlaps.
// Without a second set -- the two chains of a tile half a phase apart, each chain's maxima folded
// between the MFMAs of the other -- pass A was correct and slower as well (19k x 7k 46 -> 52 us,
// 36k x 18k 124 -> 152, 24k x 63k 219 -> 268): a chain of four MFMAs on ONE accumulator issues at half
// the pipe's rate (each waits for the one before), where the two chains of a tile issued
// alternately keep the pipe full.

// ---------------------------------------------------------------------------------------------
// Pass A.  strata[(user * nsplit + split) * kStrata + ...]: the running maxima of this slice, in no
// particular order.  The maxima are kept in the MFMA's own result layout (lane l: 16 rows of the
// tile for user col in acc_a, for user 32 + col in acc_b), so no half swap is needed; the mask bits
// of the two users a lane holds scores of are shuffled in from the lanes that own them.
// Masked items count with the mask score, rows past the end of the table never (kNoItem).
// HALF: 32 strata per slice (the position in the tile only) instead of 64 (position x parity of the
// tile pair), for tables cut into two or more slices: a user still has 32 * nsplit >= 64 strata, and
// the threshold kernel, which sorts them, has half as many to look at.  (With the 32 registers
// that frees, a second accumulator set and the maxima of tile t folded between the MFMAs of tile
// t + 1 at three wavefronts per SIMD was built as well -- 157 registers, the instruction stream
// interleaved as intended -- and ran no faster: 19k x 7k 46 -> 42 us, 36k x 18k 124 -> 130, 24k x 63k
// 219 -> 221.  What the probe shows for MFMAs and v_max alone does not carry over to this loop.)
template <int DC, int NST, bool HALF = false>
__global__ __launch_bounds__(256, DC == 1 ? 3 : 2) void strata_max_kernel(
    const float *__restrict__ Uemb, int32_t n_users, const float *__restrict__ Iemb, int32_t n_items,
    const int32_t *__restrict__ mask_ptr, const int32_t *__restrict__ mask_items, int32_t nsplit,
    const int32_t *__restrict__ heavy_thresh, float *__restrict__ strata) {
  using G = HalfTiles<DC>;
  constexpr int RN = G::RN;
  __shared__ __attribute__((aligned(1024))) float ring[NST * RN][TI * G::D / 2];
  const int tid = threadIdx.x;
  const int wave = tid / kWave, lane = tid % kWave;
  const int col = lane & 31, half = lane >> 5;
  const int32_t split = blockIdx.x % nsplit, utile = blockIdx.x / nsplit;
  const int32_t ubase = utile * UPB + wave * UPW;
  int32_t t_lo, t_hi;
  slice_tiles(n_items, split, nsplit, t_lo, t_hi);

  StreamState<DC> st;
  st.init(Uemb, n_users, ubase, lane, wave, mask_ptr, mask_items, heavy_thresh, t_lo);

  constexpr int NG = HALF ? 1 : 2;
  float sa[NG][16], sb[NG][16];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) sa[g][r] = sb[g][r] = -INFINITY;

  constexpr int kInFlight = (NST - 2) * RN * G::NP;
  static_assert(NST >= 2 && kInFlight <= 15, "vmcnt immediate");
  auto issue_stage = [&](int32_t first, int slot) {
#pragma unroll
    for (int j = 0; j < RN; ++j)
      issue_half_tile<DC>(Iemb, n_items, min(first + j, t_hi - 1), ring[slot * RN + j], st.lane_off, st.pbase);
  };
  __builtin_amdgcn_s_waitcnt(0x0F70);   // the user fragments and the mask cursor are loads of the same counter
  if (t_lo < t_hi) {
#pragma unroll
    for (int s0 = 0; s0 < NST - 1; ++s0) issue_stage(t_lo + s0 * RN, s0);
  }
  int slot = 0;
  for (int32_t tb = t_lo; tb < t_hi; tb += RN, slot = slot + 1 == NST ? 0 : slot + 1) {
    __builtin_amdgcn_s_waitcnt(0x0F70 | kInFlight);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_stage(tb + (NST - 1) * RN, slot == 0 ? NST - 1 : slot - 1);
#pragma unroll
    for (int j = 0; j < RN; ++j) {
      const int32_t t = tb + j;
      if (t >= t_hi) break;                      // uniform over the workgroup
      floatx16 acc_a, acc_b;
      st.scores(ring[slot * RN + j], acc_a, acc_b);
      // (everything below works on the accumulators in place: copies of them that are changed on
      // the rare paths only are not coalesced away, 30 moves per tile)
      const int32_t item0 = t * TI;
      if (item0 + TI > n_items) {
        // (the empty statement keeps this a branch: turned into selects it costs every tile 2 x 32 of them)
        asm volatile("");
        const int32_t lim = n_items - item0 - 4 * half;   // rows (r&3) + 8*(r>>2) + 4*half of the tile
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if ((r & 3) + 8 * (r >> 2) >= lim) acc_a[r] = acc_b[r] = kNoItem;
      }
      if (__any(st.next_m < item0 + TI)) {
        const uint32_t own = st.mask_bits(item0, mask_items);
        // ma: bits of user col, mb: bits of user 32 + col, in every lane.  (Shuffles, not one
        // v_permlane32_swap of two copies: written as inline assembly that instruction read its
        // operands, just written by vector instructions, too early -- the compiler adds no wait
        // states in front of inline assembly -- and some training items went unmasked: harmless
        // here, where a maximum that is too high only costs candidates, but not intended.)
        uint32_t ma = (uint32_t)__shfl((int)own, col), mb = (uint32_t)__shfl((int)own, 32 + col);
        ma >>= 4 * half;
        mb >>= 4 * half;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float xa = acc_a[r], xb = acc_b[r];
          RK_MASK_IF_BIT(xa, ma, (r & 3) + 8 * (r >> 2));
          RK_MASK_IF_BIT(xb, mb, (r & 3) + 8 * (r >> 2));
          acc_a[r] = xa;
          acc_b[r] = xb;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        RK_VMAX(sa[HALF ? 0 : j >> 1][r], acc_a[r]);
        RK_VMAX(sb[HALF ? 0 : j >> 1][r], acc_b[r]);
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);

  // user col: values [16*NG*half, 16*NG*(half + 1)) of its kStrata places; user 32 + col likewise
  if (ubase + col < n_users) {
    float *dst = strata + ((int64_t)(ubase + col) * nsplit + split) * kStrata + 16 * NG * half;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4 *>(dst + 16 * g + 4 * q) =
            make_float4(sa[g][4 * q], sa[g][4 * q + 1], sa[g][4 * q + 2], sa[g][4 * q + 3]);
  }
  if (ubase + 32 + col < n_users) {
    float *dst = strata + ((int64_t)(ubase + 32 + col) * nsplit + split) * kStrata + 16 * NG * half;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4 *>(dst + 16 * g + 4 * q) =
            make_float4(sb[g][4 * q], sb[g][4 * q + 1], sb[g][4 * q + 2], sb[g][4 * q + 3]);
  }
}

// thr[user] = the KC-th largest of the user's nsplit * kStrata strata maxima (-inf if fewer are finite:
// every item is a candidate then).  One lane per user and item slice sorts the KC largest of its
// slice's kStrata maxima into registers (reads of 256 contiguous bytes per lane).  With several slices
// the nsplit lanes of a user sit side by side in one wavefront (64 / nsplit users per wavefront) and
// merge their lists together: KC rounds of "the largest head of the group" by shuffles, the lane that
// owns it moving on to its next value -- each lane reads only its own list, which sits in LDS only to
// be indexed by the head.  (One thread per user over all slices took 84 us for 19k users x 10 slices,
// longer than pass A: strided reads by a grid of 76 workgroups; one thread walking the heads of all
// lists in LDS took 57 us, a chain of dependent LDS reads.)
template <int KC>
__global__ __launch_bounds__(256) void strata_threshold_kernel(const float *__restrict__ strata, int32_t n_users,
                                                               int32_t nsplit, int32_t n_strata, float *__restrict__ thr) {
  __shared__ float lists[256][KC + 1];     // (odd row length: the lanes' accesses spread over the banks)
  const int tid = threadIdx.x, lane = tid & 63;
  const int upw = 64 / nsplit;             // users per wavefront (nsplit <= 16)
  const int lu = lane / nsplit, sl = lane - lu * nsplit;
  const int32_t user = (blockIdx.x * 4 + (tid >> 6)) * upw + lu;
  const bool ok = lu < upw && user < n_users;
  float s[KC];
#pragma unroll
  for (int j = 0; j < KC; ++j) s[j] = -INFINITY;
  const float *p = strata + ((int64_t)(ok ? user : 0) * nsplit + (ok ? sl : 0)) * kStrata;
  for (int32_t i = 0; i < n_strata; i += 4) {   // (n_strata of the slice's kStrata places are in use)
    const float4 v4 = *reinterpret_cast<const float4 *>(p + i);
    const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (!__any(ok && v[e] > s[KC - 1])) continue;
      const float x = ok ? v[e] : -INFINITY;
#pragma unroll
      for (int j = KC - 1; j >= 1; --j) s[j] = __builtin_amdgcn_fmed3f(s[j - 1], s[j], x);
      s[0] = fmaxf(s[0], x);
    }
  }
  if (nsplit == 1) {                       // uniform: the slice's list is the user's
    if (ok) thr[user] = s[KC - 1];
    return;
  }
#pragma unroll
  for (int j = 0; j < KC; ++j) lists[tid][j] = s[j];
  float cur = s[0], last = -INFINITY;
  int head = 0;
  const int base = lane - sl;              // the group's first lane
  const uint64_t group = ((1ull << nsplit) - 1) << base;
  for (int round = 0; round < KC; ++round) {
    float m = cur;                         // segmented max: lane `base` ends up with the group's
    for (int d = 1; d < nsplit; d <<= 1) {
      const float t = __shfl_down(m, d);
      if (sl + d < nsplit) m = fmaxf(m, t);
    }
    m = __shfl(m, base);
    last = m;
    const uint64_t owners = __ballot(cur == m) & group;
    if (owners != 0 && lane == __builtin_ctzll(owners)) {   // (ties: one lane moves on per round)
      ++head;
      cur = head < KC ? lists[tid][head] : -INFINITY;
    }
  }
  if (ok && sl == 0) thr[user] = last;
}

// ---------------------------------------------------------------------------------------------
// Pass B.  cand[(user * nsplit + split) * kCandCap + j] = (filter score bits, local item id) of the
// j-th item of this slice whose score is >= thr[user]; cnt[user * nsplit + split] = how many there
// were (places exist for the first kCandCap).
template <int DC, int NST>
__global__ __launch_bounds__(256, DC == 1 ? 3 : 2) void collect_kernel(
    const float *__restrict__ Uemb, int32_t n_users, const float *__restrict__ Iemb, int32_t n_items,
    const int32_t *__restrict__ mask_ptr, const int32_t *__restrict__ mask_items, int32_t nsplit,
    const int32_t *__restrict__ heavy_thresh, const float *__restrict__ thr_in, uintx2 *__restrict__ cand,
    int32_t *__restrict__ cnt) {
  using G = HalfTiles<DC>;
  constexpr int RN = G::RN;
  __shared__ __attribute__((aligned(1024))) float ring[NST * RN][TI * G::D / 2];
  const int tid = threadIdx.x;
  const int wave = tid / kWave, lane = tid % kWave;
  const int32_t split = blockIdx.x % nsplit, utile = blockIdx.x / nsplit;
  const int32_t ubase = utile * UPB + wave * UPW;
  const int32_t user = ubase + lane;
  const bool user_ok = user < n_users;
  int32_t t_lo, t_hi;
  slice_tiles(n_items, split, nsplit, t_lo, t_hi);

  StreamState<DC> st;
  st.init(Uemb, n_users, ubase, lane, wave, mask_ptr, mask_items, heavy_thresh, t_lo);
  const float thr = user_ok ? thr_in[user] : INFINITY;   // lanes without a user never have a candidate
  // byte offset of this lane's next candidate place (the host checks that the places fit 32 bits)
  uint32_t place = (uint32_t)(((int64_t)(user_ok ? user : 0) * nsplit + split) * kCandCap * 8);
  const uint32_t place_end = place + kCandCap * 8;
  int32_t cu = 0;   // candidates of this lane's user so far

  // Four scores of this lane's user (ascending item ids).  Hardly any group of four holds a
  // candidate of any lane (a user has about 37 in the whole table), so the wavefront first asks
  // whether one does; the candidates themselves -- one lane or two at a time -- are stored straight
  // to the user's places in memory under the execution mask: no queue, nothing to flush.
  // (Taking the eight maxima and compares of a tile first and visiting only the groups that hold a
  // candidate, instead of a gate and a branch per group, was measured: 2.3 -> 2.9 ms at Elec.)
  auto consider4 = [&](float v0, float v1, float v2, float v3, int32_t id0) {
    float m;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(v0), "v"(v1), "v"(v2));
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(v3));
    if (!__any(m >= thr)) return;
    const float v[4] = {v0, v1, v2, v3};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (v[i] >= thr) {
        if (place < place_end) {
          uintx2 e;
          e.x = __float_as_uint(v[i]);
          e.y = (uint32_t)(id0 + i);
          *reinterpret_cast<uintx2 *>(reinterpret_cast<char *>(cand) + place) = e;
        }
        place += 8;
        ++cu;
      }
    }
  };

  constexpr int kInFlight = (NST - 2) * RN * G::NP;
  static_assert(NST >= 2 && kInFlight <= 15, "vmcnt immediate");
  auto issue_stage = [&](int32_t first, int slot) {
#pragma unroll
    for (int j = 0; j < RN; ++j)
      issue_half_tile<DC>(Iemb, n_items, min(first + j, t_hi - 1), ring[slot * RN + j], st.lane_off, st.pbase);
  };
  __builtin_amdgcn_s_waitcnt(0x0F70);
  if (t_lo < t_hi) {
#pragma unroll
    for (int s0 = 0; s0 < NST - 1; ++s0) issue_stage(t_lo + s0 * RN, s0);
  }
  int slot = 0;
  for (int32_t tb = t_lo; tb < t_hi; tb += RN, slot = slot + 1 == NST ? 0 : slot + 1) {
    // (the candidate stores share the in-order counter with the DMAs: they can only make this wait longer)
    __builtin_amdgcn_s_waitcnt(0x0F70 | kInFlight);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_stage(tb + (NST - 1) * RN, slot == 0 ? NST - 1 : slot - 1);
#pragma unroll
    for (int j = 0; j < RN; ++j) {
      const int32_t t = tb + j;
      if (t >= t_hi) break;                      // uniform over the workgroup
      floatx16 acc_a, acc_b;
      st.scores(ring[slot * RN + j], acc_a, acc_b);
      // after the swap lane l holds, for ITS user, rows (r&3)+8*(r>>2) in lo = acc_a, 4 more in
      // hi = acc_b (in place: see strata_max_kernel)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float x = acc_a[r], y = acc_b[r];
        rk::swap_halves32(x, y);
        acc_a[r] = x;
        acc_b[r] = y;
      }
      const int32_t item0 = t * TI;
      if (item0 + TI > n_items) {
        asm volatile("");   // (stays a branch)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (item0 + (r & 3) + 8 * (r >> 2) >= n_items) acc_a[r] = kNoItem;
          if (item0 + 4 + (r & 3) + 8 * (r >> 2) >= n_items) acc_b[r] = kNoItem;
        }
      }
      if (__any(st.next_m < item0 + TI)) {
        const uint32_t mbits = st.mask_bits(item0, mask_items);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float xl = acc_a[r], xh = acc_b[r];
          RK_MASK_IF_BIT(xl, mbits, (r & 3) + 8 * (r >> 2));
          RK_MASK_IF_BIT(xh, mbits, 4 + (r & 3) + 8 * (r >> 2));
          acc_a[r] = xl;
          acc_b[r] = xh;
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        consider4(acc_a[4 * g], acc_a[4 * g + 1], acc_a[4 * g + 2], acc_a[4 * g + 3], item0 + 8 * g);
        consider4(acc_b[4 * g], acc_b[4 * g + 1], acc_b[4 * g + 2], acc_b[4 * g + 3], item0 + 8 * g + 4);
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  if (user_ok) cnt[(int64_t)user * nsplit + split] = cu;
}

}  // namespace
