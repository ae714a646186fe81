// Developer probe (GPU box): lane / byte mapping of gfx950's 8-bit transposing LDS read ds_read_b64_tr_b8, the read an
// 8-bit weight-gradient kernel would stage its operands with (HISTORY.md section 4.6b).  LDS byte i holds (i & 255); every lane
// reads at lane * 8 (the natural lane-linear image) and at a row-major [16 rows][stride] image; the program prints, per lane,
// the 8 bytes it received.   hipcc --offload-arch=gfx950 -O2 -o tr8_probe tr8_probe.hip && ./tr8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(unsigned long long* out, int stride) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = static_cast<unsigned char>(i & 255);
  __syncthreads();
  const int lane = threadIdx.x;
  // image A: lane-linear, 8 bytes per lane
  i32x2 a = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)(lds + lane * 8));
  // image B: 16-lane group g reads rows of `stride` bytes: lane (g, li) -> row li / 2 ... (address pattern under test)
  const int li = lane & 15, g = lane >> 4;
  i32x2 b = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)(lds + 1024 + g * 2048 + (li >> 1) * stride + (li & 1) * 8));
  out[lane] = (static_cast<unsigned long long>(static_cast<unsigned>(a[1])) << 32) | static_cast<unsigned>(a[0]);
  out[64 + lane] = (static_cast<unsigned long long>(static_cast<unsigned>(b[1])) << 32) | static_cast<unsigned>(b[0]);
}
int main() {
  unsigned long long* d;
  hipMalloc(&d, 128 * 8);
  for (int stride : {16, 64}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, stride);
    unsigned long long h[128];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("== lane-linear image (byte value = LDS offset & 255; lane reads at lane*8)\n");
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int e = 0; e < 8; ++e) printf(" %3llu", (h[l] >> (8 * e)) & 255); printf("\n"); }
    printf("== row image, stride %d (group g base 1024 + 2048 g; lane li reads row li/2, half li&1; values = offset & 255)\n", stride);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int e = 0; e < 8; ++e) printf(" %3llu", (h[64 + l] >> (8 * e)) & 255); printf("\n"); }
  }
  return 0;
}
