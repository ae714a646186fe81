// Shared host/device helpers for libdexnerf_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/dexnerf_hip.h"

namespace dn {

constexpr int kWave = 64;

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(dn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Launch check: hipGetLastError after enqueue (no sync, graph-capture safe).
inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return -static_cast<int>(e);
  }
  return 0;
}

// Per-device launch state (api.cpp).  hipFuncAttributeMaxDynamicSharedMemorySize is set per DEVICE: a flag per thread or per
// template instance is not enough in a process that launches on a second GPU (tests, eval with LOCAL_RANK != first device).
// ensure_big_lds() sets it once per (kernel, current device) per thread - setting it again from another thread is harmless -
// and device_cus() caches the CU count of the current device, so the hot path makes one hipGetDevice call per launch.
int ensure_big_lds(const void* kernel);   // 0 or -(hipError_t), error string set
int device_cus();

// composite.hip: dn_volume_render that also counts the non-finite raw radiance-field values it reads into *nonfinite
// (device word, NULL = do not count) - the fp16 render guard of dn_render_rays
int volume_render_counting(const float* rf, const float* z, const float* rd, int rd_stride, const float* noise, float noise_std,
                           int white_background, const float* h_m_thres, int n_thres, int64_t n_rays, int n_samples, float* rgb,
                           float* disp, float* acc, float* weights, float* depth, float* dex, unsigned* nonfinite,
                           dn_stream_t stream, const uint32_t* rng_state = nullptr, uint32_t rng_stream = 0);

// mlp_fused.hip: dn_run_network with the fp16 range flag (a device word the 48-point fp16 kernel bumps when a hidden activation
// left fp16's range; NULL = not wanted)
int run_network_flagged(const dn_mlp_desc* desc, int precision, const void* packed, const float* pts, const float* viewdirs,
                        const float* rays, int ray_stride, const float* z_vals, int64_t n_rays, int samples_per_ray, float* out,
                        unsigned* range_flag, dn_stream_t stream);

// rays_sampling.hip / composite.hip: the stage entry points with an RNG state (dn_rng.h) - a NULL draw pointer together with a
// state means "draw it in the kernel"
int coarse_depths_rng(const float* rays, int ray_stride, int64_t n_rays, int num_coarse, int lindisp, const float* t_rand,
                      float* z_vals, const uint32_t* rng_state, dn_stream_t stream);
int fine_depths_rng(const float* z_coarse, const float* weights, const float* u, int64_t n_rays, int num_coarse, int num_fine,
                    float* z_fine, float* z_samples, const uint32_t* rng_state, dn_stream_t stream);
int volume_render_backward_rng(const float* rf, const float* z, const float* rd, int rd_stride, const float* noise, float noise_std,
                               int white_background, int64_t n_rays, int n_samples, const float* g_rgb, const float* g_depth,
                               const float* g_acc, const float* g_disp, const float* g_weights, float* g_rf,
                               const uint32_t* rng_state, uint32_t rng_stream, dn_stream_t stream);

#define DN_REQUIRE(cond, ...)     \
  do {                            \
    if (!(cond)) {                \
      dn::set_error(__VA_ARGS__); \
      return DN_E_INVAL;          \
    }                             \
  } while (0)

// ---- wave-level primitives (DPP-lowered shuffles; 64 lanes) --------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// inclusive scans across the 64 lanes of a wave
__device__ __forceinline__ double wave_scan_mul(double v) {
  const int l = lane_id();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    double t = __shfl_up(v, o, 64);
    if (l >= o) v *= t;
  }
  return v;
}

__device__ __forceinline__ double wave_scan_add(double v) {
  const int l = lane_id();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    double t = __shfl_up(v, o, 64);
    if (l >= o) v += t;
  }
  return v;
}

__device__ __forceinline__ float wave_scan_add(float v) {
  const int l = lane_id();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float t = __shfl_up(v, o, 64);
    if (l >= o) v += t;
  }
  return v;
}

// torch.linspace(start, end, steps) element i on CPU, fp32 (FMA form)
__host__ __device__ __forceinline__ float linspace_elem(float start, float end, int steps, int i) {
  if (steps == 1) return start;
  const float step = (end - start) / static_cast<float>(steps - 1);
  return (i < steps / 2) ? fmaf(step, static_cast<float>(i), start)
                         : fmaf(-step, static_cast<float>(steps - 1 - i), end);
}

}  // namespace dn
