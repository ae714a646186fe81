// Probe (gfx950): does MODE.FP16_OVFL (bit 23 of the MODE register) make v_cvt_scalef32_pk_{fp8,bf8}_bf16 SATURATE (e4m3 -> 448,
// e5m2 -> 57344) instead of producing NaN / inf on overflow?  (cvt_scale_probe.hip measured the default mode: no saturation - which is
// why the training kernels clamp with v_med3_f32 before they convert.)  Also: v_cvt_pk_bf16_f32 and v_pk_max_i16 results under the bit.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/cvt_sat_probe.hip -o exp_libs/cvt_sat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

template <int OVFL>
__global__ void probe(const float* in, int n, float scale, unsigned* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (OVFL) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
  if (i >= n) return;
  const float a = in[2 * i], b = in[2 * i + 1];
  unsigned pk;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(a), "v"(b));
  unsigned r8 = 0, r5 = 0;
  asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2" : "+v"(r8) : "v"(pk), "v"(scale));
  asm volatile("v_cvt_scalef32_pk_bf8_bf16 %0, %1, %2" : "+v"(r5) : "v"(pk), "v"(scale));
  unsigned f8 = 0;
  asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "+v"(f8) : "v"(a), "v"(b));
  unsigned mode;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_MODE)" : "=s"(mode));
  out[5 * i + 0] = pk; out[5 * i + 1] = r8 & 0xffff; out[5 * i + 2] = r5 & 0xffff; out[5 * i + 3] = f8 & 0xffff; out[5 * i + 4] = mode;
}

int main() {
  float special[] = {1.0f, 2.0f, 448.f, 449.f, 464.f, 480.f, 500.f, 1e5f, -500.f, -1e5f, 57344.f, 70000.f, INFINITY, -INFINITY, NAN, 3.0e38f,
                     0.001f, 1e-8f, -0.0f, 0.0f, 240.f, 447.f, 61440.f, 65536.f};
  const int ns = sizeof(special) / sizeof(float), N = ns / 2;
  float* d_in; unsigned* d_out;
  hipMalloc(&d_in, sizeof(special)); hipMalloc(&d_out, 5 * N * 4);
  hipMemcpy(d_in, special, sizeof(special), hipMemcpyHostToDevice);
  unsigned o[2][5 * 64];
  for (int ov = 0; ov < 2; ++ov) {
    if (ov) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, d_in, N, 1.0f, d_out);
    else hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, d_in, N, 1.0f, d_out);
    hipMemcpy(o[ov], d_out, 5 * N * 4, hipMemcpyDeviceToHost);
  }
  printf("MODE register: default %08x, with FP16_OVFL %08x\n", o[0][4], o[1][4]);
  for (int i = 0; i < N; ++i)
    printf("(%g, %g): bf16 pair %08x | scalef32 e4m3 %04x -> %04x  e5m2 %04x -> %04x | f32 cvt e4m3 %04x -> %04x   (default -> FP16_OVFL)\n", special[2 * i], special[2 * i + 1],
           o[0][5 * i], o[0][5 * i + 1], o[1][5 * i + 1], o[0][5 * i + 2], o[1][5 * i + 2], o[0][5 * i + 3], o[1][5 * i + 3]);
  return 0;
}
