// Probe: how fast a [63001 x 4096] fp32 matrix streams through LDS-DMA in the tile shapes a projection
// kernel with TALL workgroup tiles would use (ROWS rows x KSTEP columns per stage, STAGES stages in
// flight, one workgroup per CU), with and without the pre-split weight chunk (WB bytes per stage, the
// same for every workgroup: served by L2) that each workgroup re-reads.  No arithmetic: the floor of
// such a kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/proj_stream_probe.hip -o /tmp/proj_stream_probe && /tmp/proj_stream_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#ifndef SPIN
#define SPIN 0
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int ROWS, int KSTEP, int STAGES, int WB>
__global__ __launch_bounds__(256, 1) void stream(const float *__restrict__ X, int n_rows, int F, const char *__restrict__ Wp,
                                                  float *out) {
  const int k_lo = blockIdx.y * (F / gridDim.y), k_hi = k_lo + F / gridDim.y;
  constexpr int XB = ROWS * KSTEP * 4;                 // bytes of X per stage
  constexpr int NI = XB / 1024 / 4;                    // X DMA instructions per wavefront and stage
  constexpr int NW = WB / 1024 / 4;                    // weight DMA instructions per wavefront and stage
  constexpr int PPR = KSTEP / 4;                       // 16-byte pieces per row
  constexpr int RPI = 64 / PPR;                        // rows per DMA instruction
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  uint32_t voff[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    int64_t r = row0 + (wave * NI + i) * RPI + lane / PPR;
    r = r < n_rows ? r : n_rows - 1;
    voff[i] = (uint32_t)((r - row0) * F * 4 + (lane % PPR) * 16);
  }
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)lds);
  const int n_steps = (k_hi - k_lo) / KSTEP;
  auto issue = [&](int st) {
    const int buf = st % STAGES;
    const char *base = reinterpret_cast<const char *>(X + row0 * F + k_lo + st * KSTEP);
    const uint32_t dst = lds0 + buf * (XB + WB) + wave * NI * 1024;
#pragma unroll
    for (int i = 0; i < NI; ++i)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                   : : "v"(voff[i]), "s"(base), "s"(dst + i * 1024) : "memory", "m0");
    if constexpr (WB > 0) {
      const char *wb = Wp + (size_t)(k_lo / KSTEP + st) * WB;
      const uint32_t wdst = lds0 + buf * (XB + WB) + XB + wave * NW * 1024;
#pragma unroll
      for (int i = 0; i < NW; ++i)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     : : "v"((uint32_t)((wave * NW + i) * 1024 + lane * 16)), "s"(wb), "s"(wdst + i * 1024) : "memory", "m0");
    }
  };
  for (int s = 0; s < STAGES - 1 && s < n_steps; ++s) issue(s);
  float4 acc = {0, 0, 0, 0};
  for (int st = 0; st < n_steps; ++st) {
    if (st + STAGES - 1 < n_steps) {
      issue(st + STAGES - 1);
      if constexpr (STAGES == 3) __builtin_amdgcn_s_waitcnt(0x0F70 | (((2 * (NI + NW)) & 15)) | (((2 * (NI + NW)) >> 4) << 14));
      if constexpr (STAGES == 2) __builtin_amdgcn_s_waitcnt(0x0F70 | (((NI + NW) & 15)) | (((NI + NW) >> 4) << 14));
    } else {
      __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char *cur = lds + (st % STAGES) * (XB + WB);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4 *>(cur + ((tid * 16 + q * 4096) % XB));
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if constexpr (WB > 0) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const float4 v = *reinterpret_cast<const float4 *>(cur + XB + ((tid * 16 + q * 4096) % WB));
        acc.x += v.x; acc.y += v.y;
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.0f;
}

template <int ROWS, int KSTEP, int STAGES, int WB, int KSPLIT>
void run(const float *X, int n_rows, int F, const char *Wp, float *out) {
  const int grid = (n_rows + ROWS - 1) / ROWS;
  const size_t lds = (size_t)STAGES * (ROWS * KSTEP * 4 + WB);
  auto k = stream<ROWS, KSTEP, STAGES, WB>;
  CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto launch = [&]() { k<<<dim3(grid, KSPLIT), 256, lds>>>(X, n_rows, F, Wp, out); };
  for (int i = 0; i < 2; ++i) launch();
  CK(hipEventRecord(a));
  for (int i = 0; i < 10; ++i) launch();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  CK(hipGetLastError());
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms * 100.0;
  printf("rows/wg %4d  kstep %3d  stages %d  weight bytes/stage %6d  k-split %d  grid %4d  lds %6zu : %7.1f us  %5.2f TB/s of X\n",
         ROWS, KSTEP, STAGES, WB, KSPLIT, grid, lds, us, (double)n_rows * F * 4 / us / 1e6);
}

// The same stream with ONE extra wavefront issuing every DMA (NP producers) and the other four only reading.
template <int ROWS, int KSTEP, int STAGES, int WB, int NP>
__global__ __launch_bounds__(256 + 64 * NP) void stream_producer(const float *__restrict__ X, int n_rows, int F,
                                                                const char *__restrict__ Wp, float *out) {
  constexpr int XB = ROWS * KSTEP * 4, NI = XB / 1024 / NP, NW = WB / 1024 / NP, PPR = KSTEP / 4, RPI = 64 / PPR;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  const int n_steps = F / KSTEP;
  if (wave >= 4) {
    const int pw = wave - 4;
    uint32_t voff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int64_t r = row0 + (pw * NI + i) * RPI + lane / PPR;
      r = r < n_rows ? r : n_rows - 1;
      voff[i] = (uint32_t)((r - row0) * F * 4 + (lane % PPR) * 16);
    }
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)lds);
    auto issue = [&](int st) {
      const int buf = st % STAGES;
      const char *base = reinterpret_cast<const char *>(X + row0 * F + st * KSTEP);
      const uint32_t dst = lds0 + buf * (XB + WB) + pw * NI * 1024;
#pragma unroll
      for (int i = 0; i < NI; ++i)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     : : "v"(voff[i]), "s"(base), "s"(dst + i * 1024) : "memory", "m0");
      const char *wb = Wp + (size_t)st * WB;
      const uint32_t wdst = lds0 + buf * (XB + WB) + XB + pw * NW * 1024;
#pragma unroll
      for (int i = 0; i < NW; ++i)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     : : "v"((uint32_t)((pw * NW + i) * 1024 + lane * 16)), "s"(wb), "s"(wdst + i * 1024) : "memory", "m0");
    };
    constexpr int PS = NI + NW;
    issue(0); issue(1);
    for (int st = 0; st < n_steps; ++st) {
      if (st + 1 < n_steps) __builtin_amdgcn_s_waitcnt(0x0F70 | (PS & 15) | ((PS >> 4) << 14));
      else __builtin_amdgcn_s_waitcnt(0x0F70);
      __builtin_amdgcn_s_barrier();
      if (st + 2 < n_steps) issue(st + 2);
    }
    return;
  }
  float4 acc = {0, 0, 0, 0};
  for (int st = 0; st < n_steps; ++st) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char *cur = lds + (st % STAGES) * (XB + WB);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 v = *reinterpret_cast<const float4 *>(cur + ((tid * 16 + q * 4096) % XB));
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const float4 v = *reinterpret_cast<const float4 *>(cur + XB + ((tid * 16 + q * 4096) % WB));
      acc.x += v.x; acc.y += v.y;
    }
    // stand-in for the arithmetic of a stage
    for (int w = 0; w < SPIN; ++w) __builtin_amdgcn_s_sleep(8);
    __builtin_amdgcn_s_waitcnt(0xC07F);
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.0f;
}

template <int ROWS, int KSTEP, int STAGES, int WB, int NP>
void run_producer(const float *X, int n_rows, int F, const char *Wp, float *out) {
  const int grid = (n_rows + ROWS - 1) / ROWS;
  const size_t lds = (size_t)STAGES * (ROWS * KSTEP * 4 + WB);
  auto k = stream_producer<ROWS, KSTEP, STAGES, WB, NP>;
  CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) k<<<grid, 256 + 64 * NP, lds>>>(X, n_rows, F, Wp, out);
  CK(hipEventRecord(a));
  for (int i = 0; i < 10; ++i) k<<<grid, 256 + 64 * NP, lds>>>(X, n_rows, F, Wp, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  CK(hipGetLastError());
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  printf("producer form: rows/wg %4d  kstep %3d  stages %d  producers %d  spin %d : %7.1f us\n", ROWS, KSTEP, STAGES, NP, SPIN, ms * 100.0);
}

int main() {
  const int n_rows = 63001, F = 4096;
  float *X, *out; char *Wp;
  CK(hipMalloc(&X, (size_t)n_rows * F * 4)); CK(hipMalloc(&out, 4)); CK(hipMalloc(&Wp, (size_t)F * 64 * 6 + 65536));
  CK(hipMemset(X, 0, (size_t)n_rows * F * 4)); CK(hipMemset(Wp, 0, (size_t)F * 64 * 6 + 65536));
  run_producer<256, 32, 3, 12288, 1>(X, n_rows, F, Wp, out);
  run_producer<256, 32, 3, 12288, 2>(X, n_rows, F, Wp, out);
  run_producer<256, 32, 3, 12288, 4>(X, n_rows, F, Wp, out);
  run<128, 32, 3, 0, 1>(X, n_rows, F, Wp, out);
  run<128, 32, 3, 12288, 1>(X, n_rows, F, Wp, out);
  run<256, 32, 3, 0, 1>(X, n_rows, F, Wp, out);
  run<256, 32, 3, 12288, 1>(X, n_rows, F, Wp, out);
  run<256, 32, 2, 12288, 1>(X, n_rows, F, Wp, out);
  run<512, 32, 2, 12288, 2>(X, n_rows, F, Wp, out);
  return 0;
}
