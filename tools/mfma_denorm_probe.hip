// Does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs?  (The error bound of the candidate
// filter charges 2^-25 per element below 2^-14; a matrix unit that flushed them would need 2^-14.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void k(float *out, float a_val, float b_val) {
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
  floatx16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.0f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
  float *d, h;
  hipMalloc(&d, 4);
  const float cases[3][2] = {{0x1p-20f, 1.0f}, {0x1p-24f, 1.0f}, {0x1p-20f, 0x1p-20f}};
  for (auto &cs : cases) {
    k<<<1, 64>>>(d, cs[0], cs[1]);
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("a=%g b=%g  sum of 16 products = %g (expected %g)\n", cs[0], cs[1], h, 16.0 * cs[0] * cs[1]);
  }
  return 0;
}
