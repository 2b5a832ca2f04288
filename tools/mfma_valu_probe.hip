// Probe: do v_mfma_f32_32x32x16_f16 and vector instructions of ONE wavefront run at the same time,
// and what does it depend on?  Every variant runs REPS rounds of 8 MFMAs (two chains of 4) and 32
// v_max, placed differently, with W wavefronts per SIMD; prints cycles per round (s_memtime, wave 0).
//   serial      : 8 MFMAs, wait for the result, 32 v_max on the result (the kernels' first form)
//   interleave  : the 32 v_max (on the PREVIOUS round's result, other accumulator set) between the MFMAs
//   indep       : 32 v_max on registers no MFMA ever touches, between the MFMAs
//   mfma_only / valu_only : the two halves alone
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_probe tools/mfma_valu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
constexpr int REPS = 2000;

#define VMAX(d, s) asm volatile("v_max_f32 %0, %0, %1" : "+v"(d) : "v"(s))

template <int MODE>
__global__ __launch_bounds__(256) void probe(const float *in, float *out, unsigned long long *cyc) {
  half8 a[4], b[4];
  for (int k = 0; k < 4; ++k)
    for (int e = 0; e < 8; ++e) {
      a[k][e] = (_Float16)in[(threadIdx.x + 8 * k + e) & 1023];
      b[k][e] = (_Float16)in[(threadIdx.x * 3 + 8 * k + e) & 1023];
    }
  floatx16 acc[2][2];
  float m[32], x[32];
  for (int r = 0; r < 32; ++r) { m[r] = -1e30f; x[r] = in[(threadIdx.x + r) & 1023]; }
  for (int s = 0; s < 2; ++s) for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) acc[s][c][r] = 0.f;

  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 2
  for (int it = 0; it < REPS; ++it) {
    floatx16 &na = acc[it & 1][0], &nb = acc[it & 1][1], &oa = acc[(it + 1) & 1][0], &ob = acc[(it + 1) & 1][1];
    if (MODE == 0) {          // serial
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        na = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[k], na, 0, 0, 0);
        nb = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[3 - k], nb, 0, 0, 0);
      }
      asm volatile("s_nop 15\n\ts_nop 3" : "+v"(na), "+v"(nb));
#pragma unroll
      for (int r = 0; r < 16; ++r) { VMAX(m[r], na[r]); VMAX(m[16 + r], nb[r]); }
    } else if (MODE == 1 || MODE == 2) {   // interleave (1: previous result, 2: independent registers)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        na = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[k], na, 0, 0, 0);
        nb = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[3 - k], nb, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 4 * k; r < 4 * k + 4; ++r) {
          if (MODE == 1) { VMAX(m[r], oa[r]); VMAX(m[16 + r], ob[r]); }
          else { VMAX(m[r], x[r]); VMAX(m[16 + r], x[16 + r]); }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (MODE == 3) {   // mfma only
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        na = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[k], na, 0, 0, 0);
        nb = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[3 - k], nb, 0, 0, 0);
      }
      asm volatile("" : "+v"(na), "+v"(nb));
    } else {                  // valu only
#pragma unroll
      for (int r = 0; r < 32; ++r) VMAX(m[r], x[r]);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int r = 0; r < 32; ++r) s += m[r];
  for (int q = 0; q < 2; ++q) for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) s += acc[q][c][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// Are the previous round's results complete when vector instructions placed behind the first two
// MFMAs of the next round read them?  dist = how many MFMAs of the new round precede the first read.
template <int DIST>
__global__ __launch_bounds__(256) void check(const float *in, float *out_fast, float *out_safe) {
  half8 a[4], b[4];
  for (int k = 0; k < 4; ++k)
    for (int e = 0; e < 8; ++e) {
      a[k][e] = (_Float16)in[(threadIdx.x + 8 * k + e) & 1023];
      b[k][e] = (_Float16)in[(threadIdx.x * 3 + 8 * k + e) & 1023];
    }
  floatx16 acc[2][2];
  for (int s = 0; s < 2; ++s) for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) acc[s][c][r] = 0.f;
  const floatx16 zero = acc[0][0];
  float fast = 0.f, safe = 0.f;
#pragma unroll 2
  for (int it = 0; it < 64; ++it) {
    floatx16 &na = acc[it & 1][0], &nb = acc[it & 1][1], &oa = acc[(it + 1) & 1][0], &ob = acc[(it + 1) & 1][1];
    half8 av = a[it & 3];
    av[0] = (_Float16)(0.001f * it);
    float got = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      na = __builtin_amdgcn_mfma_f32_32x32x16_f16(k == 0 ? av : a[k], b[k], k == 0 ? zero : na, 0, 0, 0);
      nb = __builtin_amdgcn_mfma_f32_32x32x16_f16(k == 0 ? av : a[k], b[3 - k], k == 0 ? zero : nb, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (2 * (k + 1) == DIST && it > 0) {
        // read the LAST registers the previous round wrote (its final MFMA pair), unguarded
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          asm volatile("v_add_f32 %0, %0, %1" : "+v"(got) : "v"(oa[r]));
          asm volatile("v_add_f32 %0, %0, %1" : "+v"(got) : "v"(ob[r]));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    fast += got;
    // the same sum read with the full guard, one round later (from the set just finished)
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(na), "+v"(nb));
    float ref = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(ref) : "v"(na[r]));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(ref) : "v"(nb[r]));
    }
    if (it < 63) safe += ref;
  }
  out_fast[blockIdx.x * 256 + threadIdx.x] = fast;
  out_safe[blockIdx.x * 256 + threadIdx.x] = safe;
}

template <int DIST>
void run_check(const float *in, float *o1, float *o2) {
  check<DIST><<<768, 256>>>(in, o1, o2);
  hipDeviceSynchronize();
  static float h1[768 * 256], h2[768 * 256];
  hipMemcpy(h1, o1, sizeof(h1), hipMemcpyDeviceToHost);
  hipMemcpy(h2, o2, sizeof(h2), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 768 * 256; ++i) bad += h1[i] != h2[i];
  printf("reads of the previous round's result behind %d MFMAs of the next: %d of %d lanes differ from the guarded read\n", DIST, bad, 768 * 256);
}

int main() {
  float *in, *out; unsigned long long *cyc;
  hipMalloc(&in, 4096); hipMalloc(&out, 4 * 256 * 4096); hipMalloc(&cyc, 8 * 4096);
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
  run_check<2>(in, out, out + 768 * 256);
  run_check<4>(in, out, out + 768 * 256);
  run_check<6>(in, out, out + 768 * 256);
  const char *names[5] = {"serial", "interleave(prev result)", "interleave(independent)", "mfma only", "valu only"};
  for (int wps = 1; wps <= 3; ++wps) {          // wavefronts per SIMD = workgroups of 256 per CU
    const int grid = 256 * wps;
    for (int mode = 0; mode < 5; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        switch (mode) {
          case 0: probe<0><<<grid, 256>>>(in, out, cyc); break;
          case 1: probe<1><<<grid, 256>>>(in, out, cyc); break;
          case 2: probe<2><<<grid, 256>>>(in, out, cyc); break;
          case 3: probe<3><<<grid, 256>>>(in, out, cyc); break;
          default: probe<4><<<grid, 256>>>(in, out, cyc); break;
        }
      }
      hipDeviceSynchronize();
      unsigned long long c[4096]; hipMemcpy(c, cyc, 8 * grid, hipMemcpyDeviceToHost);
      double sum = 0; for (int i = 0; i < grid; ++i) sum += (double)c[i];
      printf("%d wave(s)/SIMD  %-26s %8.1f cycles per round of 8 MFMA + 32 v_max (per wave)\n", wps, names[mode], sum / grid / REPS);
    }
  }
  return 0;
}
