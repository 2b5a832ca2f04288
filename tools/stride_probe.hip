// Probe: read bandwidth of a row-major [rows x 4096] fp32 matrix when every workgroup walks its own
// block of rows in steps of SEG bytes per row (the access pattern of a GEMM A-tile with BK = SEG/4).
//   hipcc --offload-arch=gfx950 -O3 tools/stride_probe.hip -o /tmp/stride_probe && /tmp/stride_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int ROWS, int SEG, int DEPTH>
__global__ __launch_bounds__(256) void walk(const float4 *__restrict__ X, int n_rows, int F4, float *out) {
  // 256 threads cover ROWS x SEG bytes per step with LD loads each
  constexpr int PER_ROW = SEG / 16, LD = ROWS * PER_ROW / 256;
  const int r0 = blockIdx.x * ROWS;
  float4 acc = {0, 0, 0, 0};
  const int steps = F4 / PER_ROW;
  float4 v[DEPTH][LD];
  auto load = [&](int st, float4 (&dst)[LD]) {
#pragma unroll
    for (int j = 0; j < LD; ++j) {
      const int e = threadIdx.x + 256 * j, r = e / PER_ROW, c = e % PER_ROW;
      const int row = min(r0 + r, n_rows - 1);
      dst[j] = X[(size_t)row * F4 + (size_t)st * PER_ROW + c];
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load(d, v[d]);
  for (int st = 0; st < steps; st += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int j = 0; j < LD; ++j) { acc.x += v[d][j].x; acc.y += v[d][j].y; acc.z += v[d][j].z; acc.w += v[d][j].w; }
      if (st + d + DEPTH < steps) load(st + d + DEPTH, v[d]);
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.0f;
}

template <int ROWS, int SEG, int DEPTH>
void run(const float4 *X, int n_rows, int F, float *out) {
  const int grid = (n_rows + ROWS - 1) / ROWS;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) walk<ROWS, SEG, DEPTH><<<grid, 256>>>(X, n_rows, F / 4, out);
  CK(hipEventRecord(a));
  for (int i = 0; i < 10; ++i) walk<ROWS, SEG, DEPTH><<<grid, 256>>>(X, n_rows, F / 4, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms * 100.0;
  printf("rows/wg %4d  seg %5d B  depth %d  grid %5d : %7.1f us  %5.2f TB/s\n", ROWS, SEG, DEPTH, grid, us,
         (double)n_rows * F * 4 / us / 1e6);
}

int main() {
  const int n_rows = 63001, F = 4096;
  float4 *X; float *out;
  CK(hipMalloc(&X, (size_t)n_rows * F * 4)); CK(hipMalloc(&out, 4));
  CK(hipMemset(X, 0, (size_t)n_rows * F * 4));
  run<128, 128, 1>(X, n_rows, F, out);
  run<128, 128, 2>(X, n_rows, F, out);
  run<128, 128, 4>(X, n_rows, F, out);
  run<128, 256, 2>(X, n_rows, F, out);
  run<64, 256, 2>(X, n_rows, F, out);
  run<64, 512, 2>(X, n_rows, F, out);
  run<32, 512, 2>(X, n_rows, F, out);
  run<32, 1024, 2>(X, n_rows, F, out);
  run<16, 1024, 4>(X, n_rows, F, out);
  run<16, 4096, 1>(X, n_rows, F, out);
  return 0;
}
