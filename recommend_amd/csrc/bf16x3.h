// fp32 products on the bf16 matrix pipe (gfx950): three-way bf16 operand splits and the six-MFMA
// product they feed.  Shared by the 64-wide (dense.hip) and 128-wide (dense_wide.hip) layer kernels.
#pragma once
#include "common.h"

namespace rk {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// fp32 products on the bf16 matrix pipe.  x = h + m + l with h, m, l bf16 (8 significant bits
// each: all 24 of an fp32 mantissa, every subtraction below exact), and
//   x * w = hh + hm + mh + mm + hl + lh   (+ ml + lm + ll, below 2^-24 |x| |w|, dropped):
// six v_mfma_f32_32x32x16_bf16 of 32 cycles per 16 k against eight fp32-input MFMAs of 64, and,
// unlike those, not on the vector ALU's issue port.  Sums are fp32 inside the MFMA as before.
template <int NV>
struct SplitN {
  bf16x8 h[NV / 8], m[NV / 8], l[NV / 8];   // element 8s + j is the lane's reduction index 8s + j
};
using Split3 = SplitN<32>;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// Two values at a time, so that every step is one packed instruction or one per value: convert
// (v_cvt_pk_bf16_f32), widen the two halves back (shift / mask), subtract (v_pk_add_f32):
// 9 instructions per pair.
__device__ __forceinline__ f32x2 widen(uint32_t pk) {
  return f32x2{__uint_as_float(pk << 16), __uint_as_float(pk & 0xFFFF0000u)};
}
template <int NV>
__device__ __forceinline__ void split3(const float (&x)[NV], SplitN<NV> &o) {
#pragma unroll
  for (int s4 = 0; s4 < NV / 8; ++s4) {
    u32x4 hh, mm, ll;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x2 v = {x[8 * s4 + 2 * q], x[8 * s4 + 2 * q + 1]};
      const uint32_t h = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
      const f32x2 r1 = v - widen(h);
      const uint32_t m = __builtin_bit_cast(uint32_t, __builtin_convertvector(r1, bf16x2));
      const f32x2 r2 = r1 - widen(m);
      hh[q] = h;
      mm[q] = m;
      ll[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r2, bf16x2));
    }
    o.h[s4] = __builtin_bit_cast(bf16x8, hh);
    o.m[s4] = __builtin_bit_cast(bf16x8, mm);
    o.l[s4] = __builtin_bit_cast(bf16x8, ll);
  }
}
// acc += A (32 rows x this lane's NV reduction indices) * B^T in six bf16 products per step of 8;
// the small terms first, the leading one last
template <int NV>
__device__ __forceinline__ void mfma_split3(const SplitN<NV> &a, const SplitN<NV> &b, floatx16 &acc) {
#pragma unroll
  for (int s4 = 0; s4 < NV / 8; ++s4) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l[s4], b.h[s4], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h[s4], b.l[s4], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m[s4], b.m[s4], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m[s4], b.h[s4], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h[s4], b.m[s4], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h[s4], b.h[s4], acc, 0, 0, 0);
  }
}

// Two independent products that share one operand, their MFMAs issued alternately.  A chain of these
// MFMAs on ONE accumulator issues nearly at the pipe's rate (36 cycles each against 34 with several
// chains in flight, measured with one wavefront per SIMD); alternating gives those few per cent and
// lets the operand fragments of the second product be read while the first computes.  Each
// accumulator sees the same operations in the same order as with mfma_split3: the same bits.
template <int NV>
__device__ __forceinline__ void mfma_split3_shared_a(const SplitN<NV> &a, const SplitN<NV> &b0, const SplitN<NV> &b1,
                                                     floatx16 &acc0, floatx16 &acc1) {
#pragma unroll
  for (int s4 = 0; s4 < NV / 8; ++s4) {
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l[s4], b0.h[s4], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l[s4], b1.h[s4], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h[s4], b0.l[s4], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h[s4], b1.l[s4], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m[s4], b0.m[s4], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m[s4], b1.m[s4], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m[s4], b0.h[s4], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m[s4], b1.h[s4], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h[s4], b0.m[s4], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h[s4], b1.m[s4], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h[s4], b0.h[s4], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h[s4], b1.h[s4], acc1, 0, 0, 0);
  }
}
// Four independent products (two A operands x two B operands), one MFMA of each in turn (see above:
// a few per cent over one chain after the other).  Per accumulator the operations and their order
// are mfma_split3's.
template <int NV>
__device__ __forceinline__ void mfma_split3_2x2(const SplitN<NV> &a0, const SplitN<NV> &a1, const SplitN<NV> &b0,
                                                const SplitN<NV> &b1, floatx16 &c00, floatx16 &c01, floatx16 &c10,
                                                floatx16 &c11) {
#define RK_MFMA4(PA, PB)                                                                 \
  c00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.PA[s4], b0.PB[s4], c00, 0, 0, 0);     \
  c01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0.PA[s4], b1.PB[s4], c01, 0, 0, 0);     \
  c10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.PA[s4], b0.PB[s4], c10, 0, 0, 0);     \
  c11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1.PA[s4], b1.PB[s4], c11, 0, 0, 0);
#pragma unroll
  for (int s4 = 0; s4 < NV / 8; ++s4) {
    RK_MFMA4(l, h) RK_MFMA4(h, l) RK_MFMA4(m, m) RK_MFMA4(m, h) RK_MFMA4(h, m) RK_MFMA4(h, h)
  }
#undef RK_MFMA4
}

// the same split for four values that leave as three 8-byte packs (h, m, l of x, y, z, w)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_quad(float x, float y, float z, float w, u32x2 &h, u32x2 &m, u32x2 &l) {
  const f32x2 v[2] = {f32x2{x, y}, f32x2{z, w}};
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const uint32_t hh = __builtin_bit_cast(uint32_t, __builtin_convertvector(v[q], bf16x2));
    const f32x2 r1 = v[q] - widen(hh);
    const uint32_t mm = __builtin_bit_cast(uint32_t, __builtin_convertvector(r1, bf16x2));
    const f32x2 r2 = r1 - widen(mm);
    h[q] = hh;
    m[q] = mm;
    l[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r2, bf16x2));
  }
}

}  // namespace rk
