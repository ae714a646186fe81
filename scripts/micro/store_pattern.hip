// Developer microbenchmark: HBM write rate of the training-forward store pattern.
//   A: [tile32][slot][64 lanes][16 B]  - every wave owns a private 148 KiB record and walks its slots (current layout)
//   B: [wgtile][slot][wave][64][16 B]  - the 8 waves of a workgroup write one 8 KiB block per slot
// Both: 256 persistent workgroups x 8 waves, each wave stores 1 KiB per slot, slots in order, `pace` dummy FMAs between
// slots to mimic the MFMA work between stores.   build: hipcc -O3 --offload-arch=gfx950 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(char* out, long long n_wgtiles, int slots, int pace) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = lane;
  for (long long t = blockIdx.x; t < n_wgtiles; t += gridDim.x) {
    for (int s = 0; s < slots; ++s) {
      for (int i = 0; i < pace; ++i) acc = __builtin_fmaf(acc, 1.0001f, 0.5f);
      char* p;
      if (MODE == 0) p = out + (((t * 8 + wave) * slots + s) * 64 + lane) * 16;
      else p = out + (((t * slots + s) * 8 + wave) * 64 + lane) * 16;
      f32x4 v = {acc, acc, acc, acc};
      *reinterpret_cast<f32x4*>(p) = v;
    }
  }
}

int main(int argc, char** argv) {
  const int slots = 148;
  const long long n_wgtiles = 3072;                      // 786,432 points
  const size_t bytes = static_cast<size_t>(n_wgtiles) * 8 * slots * 1024;
  char* buf;
  if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int pace : {0, 64, 256, 512, 1024}) {
    for (int mode = 0; mode < 2; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, buf, n_wgtiles, slots, pace);
        else hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, buf, n_wgtiles, slots, pace);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("pace %4d  layout %c: %.3f ms  %.2f TB/s\n", pace, mode ? 'B' : 'A', best, bytes / best / 1e9);
    }
  }
  hipFree(buf);
  return 0;
}
