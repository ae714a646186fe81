// Probe (gfx950): does MODE.FP16_OVFL (hwreg MODE bit 23) make v_cvt_scalef32_pk_{fp8,bf8}_bf16 saturate - e4m3 overflow -> 0x7E
// (448) instead of 0x7F (NaN), e5m2 overflow -> 0x7B (57344) instead of 0x7C (inf)?  And is anything else the training kernels
// use touched by the bit (v_cvt_pk_bf16_f32, fp32 VALU)?
// Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/fp16_ovfl_probe.hip -o exp_libs/fp16_ovfl_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

__global__ void probe(const float* in, int n, float scale, int ovfl, unsigned* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1" ::: "memory");
  const float a = in[2 * i], b = in[2 * i + 1];
  bf16x2 v = {static_cast<__bf16>(a), static_cast<__bf16>(b)};
  s16x2 r = {0, 0}, q = {0, 0};
  asm volatile("" : "+v"(v));
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(r, v, scale, false);
  q = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(q, v, scale, false);
  int d0 = 0;
  d0 = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, d0, false);
  int d1 = 0;
  d1 = __builtin_amdgcn_cvt_pk_bf8_f32(a * (1.0f / scale), b * (1.0f / scale), d1, false);
  if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 0" ::: "memory");
  if (i < n) {
    out[4 * i + 0] = static_cast<unsigned short>(r[0]);
    out[4 * i + 1] = static_cast<unsigned short>(q[0]);
    out[4 * i + 2] = d0 & 0xffff;
    out[4 * i + 3] = d1 & 0xffff;
  }
}

int main() {
  float h[] = {1.0f, 2.0f, 448.f, 464.f, 500.f, 1e5f, -1e5f, -600.f, 57344.f, 61440.f, 70000.f, 1e9f, -0.0f, 3e-5f, NAN, INFINITY, 240.f, -448.f, 3.4e38f, -3.4e38f};
  const int N = sizeof(h) / sizeof(float) / 2;
  float* d_in; unsigned* d_out;
  hipMalloc(&d_in, sizeof(h)); hipMalloc(&d_out, 4 * 64 * 4);
  hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
  unsigned o[2][4 * 64];
  for (int ovfl = 0; ovfl < 2; ++ovfl) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_in, N, 1.0f, ovfl, d_out);
    hipMemcpy(o[ovfl], d_out, sizeof(o[0]), hipMemcpyDeviceToHost);
  }
  for (int i = 0; i < N; ++i)
    printf("(%g, %g): scalef32 e4m3 %04x -> %04x | scalef32 e5m2 %04x -> %04x | f32 cvt e4m3 %04x -> %04x | f32 cvt e5m2 %04x -> %04x   (FP16_OVFL 0 -> 1)\n", h[2 * i], h[2 * i + 1],
           o[0][4 * i], o[1][4 * i], o[0][4 * i + 1], o[1][4 * i + 1], o[0][4 * i + 2], o[1][4 * i + 2], o[0][4 * i + 3], o[1][4 * i + 3]);
  return 0;
}
