// Probe (gfx950): v_cvt_scalef32_pk_{fp8,bf8}_bf16 - scale direction, saturation, rounding - against v_cvt_pk_{fp8,bf8}_f32 of
// the same values (clamped with v_med3_f32, the form the 32-point training kernels use).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/cvt_scale_probe.hip -o exp_libs/cvt_scale_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

__global__ void probe(const float* in, int n, float scale, unsigned* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = in[2 * i], b = in[2 * i + 1];
  bf16x2 v = {static_cast<__bf16>(a), static_cast<__bf16>(b)};
  s16x2 r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(r, v, scale, false);
  s16x2 q = {0, 0};
  q = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(q, v, scale, false);
  const float fa = static_cast<float>(v[0]), fb = static_cast<float>(v[1]);
  // reference forms: value / scale and value * scale, clamped, through the f32 converts
  int d0 = 0, d1 = 0, m0 = 0, m1 = 0;
  d0 = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(fa / scale, -448.f, 448.f), __builtin_amdgcn_fmed3f(fb / scale, -448.f, 448.f), d0, false);
  m0 = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(fa * scale, -448.f, 448.f), __builtin_amdgcn_fmed3f(fb * scale, -448.f, 448.f), m0, false);
  d1 = __builtin_amdgcn_cvt_pk_bf8_f32(__builtin_amdgcn_fmed3f(fa / scale, -57344.f, 57344.f), __builtin_amdgcn_fmed3f(fb / scale, -57344.f, 57344.f), d1, false);
  m1 = __builtin_amdgcn_cvt_pk_bf8_f32(__builtin_amdgcn_fmed3f(fa * scale, -57344.f, 57344.f), __builtin_amdgcn_fmed3f(fb * scale, -57344.f, 57344.f), m1, false);
  int u0 = 0;   // unclamped f32 convert: what does an overflow give?
  u0 = __builtin_amdgcn_cvt_pk_fp8_f32(fa, fb, u0, false);
  out[6 * i + 0] = static_cast<unsigned short>(r[0]);
  out[6 * i + 1] = static_cast<unsigned short>(q[0]);
  out[6 * i + 2] = d0 & 0xffff; out[6 * i + 3] = m0 & 0xffff;
  out[6 * i + 4] = (d1 & 0xffff) | ((m1 & 0xffff) << 16);
  out[6 * i + 5] = u0 & 0xffff;
}

int main() {
  const int N = 4096;
  static float h[2 * N];
  float special[] = {1.0f, 2.0f, 448.f, 500.f, 1e5f, -1e5f, 0.001f, 0.002f, 57344.f, 70000.f, -0.0f, 0.0f, 1e-8f, 3e-5f, NAN, INFINITY,
                     0.0019f, 0.0009f, 1.0625f, 1.1875f, 0.4375f, 15.5f, 240.f, 464.f};
  int ns = sizeof(special) / sizeof(float);
  for (int i = 0; i < 2 * N; ++i) {
    if (i < ns) h[i] = special[i];
    else { unsigned s = i * 2654435761u; float e = (float)((s >> 8) % 40) - 24.f; float m = 1.f + (float)(s & 255) / 256.f; h[i] = ((s >> 20) & 1 ? -1.f : 1.f) * ldexpf(m, (int)e); }
  }
  float* d_in; unsigned* d_out;
  hipMalloc(&d_in, sizeof(h)); hipMalloc(&d_out, 6 * N * 4);
  hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
  static unsigned o[6 * N];
  for (float scale : {1.0f, 0.5f, 4.0f, 1.52587890625e-05f}) {
    hipLaunchKernelGGL(probe, dim3(N / 64), dim3(64), 0, 0, d_in, N, scale, d_out);
    hipMemcpy(o, d_out, sizeof(o), hipMemcpyDeviceToHost);
    int eq_div8 = 0, eq_mul8 = 0, eq_div5 = 0, eq_mul5 = 0;
    for (int i = 0; i < N; ++i) {
      eq_div8 += o[6 * i] == o[6 * i + 2]; eq_mul8 += o[6 * i] == o[6 * i + 3];
      eq_div5 += o[6 * i + 1] == (o[6 * i + 4] & 0xffff); eq_mul5 += o[6 * i + 1] == (o[6 * i + 4] >> 16);
    }
    printf("scale %g: e4m3 == cvt(clamp(x/scale)) on %d / %d pairs, == cvt(clamp(x*scale)) on %d; e5m2: %d / %d\n", scale, eq_div8, N, eq_mul8, eq_div5, eq_mul5);
    if (scale == 1.0f)
      for (int i = 0; i < ns / 2; ++i)
        printf("  (%g, %g): scalef32 e4m3 %04x e5m2 %04x | clamped f32 cvt e4m3 %04x e5m2 %04x | UNclamped f32 cvt e4m3 %04x\n", h[2 * i], h[2 * i + 1], o[6 * i], o[6 * i + 1],
               o[6 * i + 2], o[6 * i + 4] & 0xffff, o[6 * i + 5]);
    if (scale != 1.0f) {
      int shown = 0;
      for (int i = 0; i < N && shown < 6; ++i)
        if (o[6 * i] != o[6 * i + 2] && o[6 * i] != o[6 * i + 3]) { printf("  neither: (%g, %g) -> %04x, div %04x mul %04x\n", h[2 * i], h[2 * i + 1], o[6 * i], o[6 * i + 2], o[6 * i + 3]); ++shown; }
    }
  }
  return 0;
}
