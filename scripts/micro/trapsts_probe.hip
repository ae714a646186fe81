// Probe (gfx950): does an f32 -> f16 conversion overflow (and a NaN produced by an MFMA / by VALU) leave a sticky bit in TRAPSTS.EXCP
// without traps enabled?  Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/trapsts_probe.hip -o gpurun_out/trapsts_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ unsigned trapsts() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS)" : "=s"(v));
  return v;
}

__global__ void probe(const float* in, unsigned* out, float* fout) {
  const int lane = threadIdx.x;
  out[0] = trapsts();
  f2 a = {in[0], in[1]};                      // finite, in range
  h2 h = __builtin_convertvector(a, h2);
  asm volatile("" :: "v"(h));
  out[1] = trapsts();
  f2 b = {in[2], in[3]};                      // 1e6: beyond fp16
  h2 g = __builtin_convertvector(b, h2);
  asm volatile("" : "+v"(g));
  out[2] = trapsts();
  out[3] = __builtin_bit_cast(unsigned, g);
  // MFMA with an inf operand against mixed-sign weights -> NaN?  what sign?
  f16x8 A, B;
  for (int e = 0; e < 8; ++e) { A[e] = (_Float16)((e & 1) ? 1.0f : -1.0f); B[e] = g[0]; }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, c, 0, 0, 0);
  if (lane == 0) { fout[0] = c[0]; }
  out[4] = trapsts();
  f2 cc = {c[0], c[1]};
  h2 hc = __builtin_convertvector(cc, h2);
  asm volatile("" : "+v"(hc));
  out[5] = trapsts();
  out[6] = __builtin_bit_cast(unsigned, hc);
  float s = in[4] - in[5];                    // inf - inf on the VALU
  asm volatile("" : "+v"(s));
  out[7] = trapsts();
  if (lane == 0) fout[1] = s;
  unsigned mode;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_MODE)" : "=s"(mode));
  out[8] = mode;
}

int main() {
  float h_in[6] = {1.5f, -2.0f, 1.0e6f, 3.0f, __builtin_inff(), __builtin_inff()};
  float* d_in; unsigned* d_out; float* d_f;
  hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, 64 * 4); hipMalloc(&d_f, 64 * 4);
  hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_in, d_out, d_f);
  unsigned o[16]; float f[4];
  hipMemcpy(o, d_out, sizeof(o), hipMemcpyDeviceToHost); hipMemcpy(f, d_f, sizeof(f), hipMemcpyDeviceToHost);
  const char* names[] = {"start", "after in-range cvt", "after overflowing cvt", "(packed halves)", "after MFMA with inf", "after cvt of the MFMA result",
                         "(packed halves)", "after VALU inf-inf", "MODE"};
  for (int i = 0; i < 9; ++i) printf("%-32s 0x%08x   EXCP[8:0]=0x%03x\n", names[i], o[i], o[i] & 0x1ff);
  unsigned b0, b1; memcpy(&b0, &f[0], 4); memcpy(&b1, &f[1], 4);
  printf("MFMA(inf, +-1) = %f (bits 0x%08x), VALU inf-inf bits 0x%08x\n", f[0], b0, b1);
  return 0;
}
