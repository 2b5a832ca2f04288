// Probe: issue rate of v_mfma_f32_32x32x16_bf16 for one wavefront per SIMD as a function of the number of
// independent accumulator chains it alternates between (a chain = MFMAs that accumulate onto each other).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_chain_probe.hip -o /tmp/mfma_chain_probe && /tmp/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CHAINS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void chains(int iters, float *out) {
  floatx16 acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
  bf16x8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x & 3); b[j] = (__bf16)1.0f; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int rep = 0; rep < 8 / CHAINS + (CHAINS > 8); ++rep)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0];
  if (s == 123.456f) out[0] = s;
}

template <int CHAINS, int WAVES>
void run(float *out) {
  const int iters = 20000, per_iter = CHAINS > 8 ? CHAINS : 8;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  chains<CHAINS, WAVES><<<256, 64 * WAVES>>>(iters, out);     // one workgroup per CU, WAVES / 4 wavefronts per SIMD
  CK(hipEventRecord(a));
  chains<CHAINS, WAVES><<<256, 64 * WAVES>>>(iters, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double ns_per_mfma = ms * 1e6 / ((double)iters * per_iter);
  printf("wavefronts per SIMD %d  chains %2d : %6.2f ns per MFMA and wavefront = %5.1f cycles at 2.4 GHz; %6.1f TFLOP/s\n", WAVES / 4, CHAINS,
         ns_per_mfma, ns_per_mfma * 2.4, 256.0 * WAVES * iters * per_iter * 32768.0 / (ms * 1e-3) / 1e12);
}

int main() {
  float *out; CK(hipMalloc(&out, 4));
  run<1, 4>(out); run<2, 4>(out); run<4, 4>(out); run<8, 4>(out);
  run<1, 8>(out); run<2, 8>(out); run<4, 8>(out);
  return 0;
}
