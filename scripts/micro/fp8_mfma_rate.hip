// Developer probe (GPU box): issue rate of the fp8 MFMAs a weight-gradient kernel can use, 4 independent accumulator chains per
// wave, 2 waves per SIMD, every CU busy: cycles per instruction (s_memtime) and the FLOP rate, for
//   v_mfma_f32_32x32x16_bf16, v_mfma_f32_32x32x16_bf8_fp8 (K = 16), v_mfma_f32_32x32x64_f8f6f4 with e5m2 x e4m3 operands (K = 64).
//   hipcc --offload-arch=gfx950 -O3 -o fp8_mfma_rate fp8_mfma_rate.hip && ./fp8_mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
constexpr int kIters = 4096;

template <int MODE>
__global__ __launch_bounds__(512, 2) void rate(float* out, unsigned long long* cycles) {
  f32x16 acc[4];
  for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
  const int seed = threadIdx.x * 2654435761u;
  bf16x8 a16, b16;
  for (int e = 0; e < 8; ++e) { a16[e] = static_cast<__bf16>(float((seed >> e) & 3)); b16[e] = static_cast<__bf16>(float((seed >> (e + 8)) & 3)); }
  long a8 = 0x3838383838383838L ^ (seed & 0x0101010101010101L), b8 = 0x3838383838383838L;
  i32x8 a64, b64;
  for (int d = 0; d < 8; ++d) { a64[d] = 0x38383838 ^ ((seed >> d) & 0x01010101); b64[d] = 0x38383838; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if constexpr (MODE == 0) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a16, b16, acc[c], 0, 0, 0);
      if constexpr (MODE == 1) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf8_fp8(a8, b8, acc[c], 0, 0, 0);
      if constexpr (MODE == 2) acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a64, b64, acc[c], 1, 0, 0, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.0f;
  for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, double flop_per_inst) {
  float* out; unsigned long long* cyc;
  const int blocks = 256;
  hipMalloc(&out, blocks * 512 * sizeof(float));
  hipMalloc(&cyc, blocks * sizeof(unsigned long long));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(512), 0, 0, out, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(512), 0, 0, out, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0; for (int i = 0; i < blocks; ++i) mean += h[i]; mean /= blocks;
  // s_memtime counts at 100 MHz on gfx9: convert with the event time instead - report both
  const double insts_per_simd = 2.0 * 4 * kIters;      // two waves per SIMD
  const double total_flop = flop_per_inst * 4 * kIters * 8 * blocks;
  printf("%-44s %8.3f ms  %8.1f TFLOP/s  memtime ticks per wave-loop %.0f  (%.2f us per MFMA per SIMD)\n", name, ms, total_flop / ms / 1e9, mean,
         ms * 1e3 / insts_per_simd);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0>("v_mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16);
  run<1>("v_mfma_f32_32x32x16_bf8_fp8", 2.0 * 32 * 32 * 16);
  run<2>("v_mfma_f32_32x32x64_f8f6f4 (e5m2 x e4m3)", 2.0 * 32 * 32 * 64);
  run<0>("v_mfma_f32_32x32x16_bf16 (again)", 2.0 * 32 * 32 * 16);
  return 0;
}
