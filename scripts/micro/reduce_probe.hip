// Probe (gfx950): what does the cross-workgroup reduction of the weight-gradient partials cost?  256 workgroups of 512 threads,
// 8 "layers" of `elems` floats; the 32 workgroups of a layer each add their partial to the layer's dW.
//   mode 0: atomicAdd, lanes -> 16-byte runs scattered over the row (the accumulator layout of the kernel today)
//   mode 1: atomicAdd, 64 lanes -> 256 contiguous bytes
//   mode 2: plain 16-byte stores of the partial to scratch [wg][elems] (what a two-phase reduction writes)
//   mode 3: mode 2 + every workgroup then sums ITS 1/32 slice over the 32 partials of its layer and stores it (no ordering
//           between the phases here: traffic only)
// Build: hipcc --offload-arch=gfx950 -O3 scripts/micro/reduce_probe.hip -o exp_libs/reduce_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(512) void probe(float* dW, float* scratch, int elems, int mode, int per_layer) {
  const int layer = blockIdx.x / per_layer, w = blockIdx.x % per_layer;
  float* dst = dW + static_cast<long long>(layer) * elems;
  const int tid = threadIdx.x;
  if (mode == 0) {
    for (int base = 0; base < elems; base += 512 * 16) {      // each thread: 16 values, like 16 accumulator registers
      for (int r = 0; r < 16; ++r) {
        const int lane = tid & 63, wave = tid >> 6;
        // 4-float runs: lane -> run (lane % 16) * 4 within a 64-float span ... rows differ per register / lane half
        const int idx = base + wave * 1024 + r * 64 + ((lane & 15) * 4 + (lane >> 4));
        if (idx < elems) atomicAdd(dst + ((idx * 37) % elems / 4 * 4 + (idx & 3)) , 1.0f);
      }
    }
  } else if (mode == 1) {
    for (int idx = tid; idx < elems; idx += 512) atomicAdd(dst + idx, 1.0f);
  } else {
    float4* part = reinterpret_cast<float4*>(scratch + static_cast<long long>(blockIdx.x) * elems);
    for (int idx = tid; idx < elems / 4; idx += 512) part[idx] = make_float4(1.f, 2.f, 3.f, 4.f);
    if (mode == 3) {
      __threadfence();
      const int slice = elems / 4 / per_layer;
      for (int idx = tid; idx < slice; idx += 512) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q = 0; q < per_layer; ++q) {
          const float4 v = reinterpret_cast<const float4*>(scratch + static_cast<long long>(layer * per_layer + q) * elems)[w * slice + idx];
          s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        reinterpret_cast<float4*>(dst)[w * slice + idx] = s;
      }
    }
  }
}

int main(int argc, char** argv) {
  const int elems = argc > 1 ? atoi(argv[1]) : 128 * 128;
  const int layers = 8, per_layer = argc > 2 ? atoi(argv[2]) : 32;
  float *dW, *scratch;
  hipMalloc(&dW, sizeof(float) * elems * layers);
  hipMalloc(&scratch, sizeof(float) * elems * layers * per_layer);
  hipMemset(dW, 0, sizeof(float) * elems * layers);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe, dim3(layers * per_layer), dim3(512), 0, 0, dW, scratch, elems, mode, per_layer);
    hipEventRecord(a);
    for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(probe, dim3(layers * per_layer), dim3(512), 0, 0, dW, scratch, elems, mode, per_layer);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("elems %d, %d workgroups per layer, mode %d: %.2f us per launch (%.1f M values)\n", elems, per_layer, mode, ms * 1000 / 20,
           1e-6 * elems * layers * per_layer);
  }
  return 0;
}
