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
bool weight_grad_pair_fits(const dn_mlp_desc& d);   // mlp_train.hip: the layers of two such networks fit one weight-gradient batch
struct CompParams;   // composite_body.h
int run_network_flagged(const dn_mlp_desc* desc, int precision, const void* packed, const float* pts, const float* viewdirs,
                        const float* rays, int ray_stride, const float* z_vals, int64_t n_rays, int samples_per_ray, float* out,
                        unsigned* range_flag, dn_stream_t stream, const CompParams* comp = nullptr, int* composited = nullptr);

// rays_sampling.hip / composite.hip: the stage entry points with an RNG state (dn_rng.h) - a NULL draw pointer together with a
// state means "draw it in the kernel"
int coarse_depths_rng(const float* rays, int ray_stride, int64_t n_rays, int num_coarse, int lindisp, const float* t_rand,
                      float* z_vals, const uint32_t* rng_state, dn_stream_t stream);
int fine_depths_rng(const float* z_coarse, const float* weights, const float* u, int64_t n_rays, int num_coarse, int num_fine,
                    float* z_fine, float* z_samples, const uint32_t* rng_state, dn_stream_t stream);
int volume_render_backward_rng(const float* rf, const float* z, const float* rd, int rd_stride, const float* noise, float noise_std,
                               int white_background, int64_t n_rays, int n_samples, const float* g_rgb, const float* g_depth,
                               const float* g_acc, const float* g_disp, const float* g_weights, float* g_rf,
                               const uint32_t* rng_state, uint32_t rng_stream, dn_stream_t stream, unsigned* absmax_part = nullptr);
// mlp_train.hip: dn_mlp_backward_data with the largest |g_out| already formed by the caller's previous kernel (DN_PREC_BF16_S8 with the
// per-launch gradient scale): n_partials words whose maximum it is (composite.hip composite_bwd_kernel) - NULL: a launch of its own finds it
int mlp_backward_data_partials(const dn_mlp_desc* desc, int precision, const void* packed_bwd, const float* g_out, const void* masks,
                               int64_t n_points, void* grads, const unsigned* partials, int n_partials, dn_stream_t stream);
bool s8_scale_is_per_launch();   // mlp_train.hip: dn_set_s8_grad_scale(0), the default

#define DN_REQUIRE(cond, ...)     \
  do {                            \
    if (!(cond)) {                \
      dn::set_error(__VA_ARGS__); \
      return DN_E_INVAL;          \
    }                             \
  } while (0)

// ---- wave-level primitives (DPP-lowered shuffles; 64 lanes) --------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// All-reduce over the 64 lanes without the LDS crossbar (__shfl_xor lowers to ds_bpermute_b32: an address computation, an LDS
// round trip and a wait per step - 36 of them per ray in the compositing kernel).  Steps 1, 2: DPP quad permutes (lane ^ 1, lane ^ 2);
// steps 4, 8: DPP row_half_mirror / row_mirror - not lane ^ 4 / ^ 8, but by then every lane of a quad (of a half row) holds the same
// value, so the partner's value is the same one; steps 16, 32: v_permlane16_swap / v_permlane32_swap of the value with itself
// hand every lane its own and its partner's value in the two results.  The tree is the butterfly's (lane ^ 1 first), every
// lane ends with bitwise the same total.
template <int CTRL>
__device__ __forceinline__ unsigned dpp_bits(unsigned v) {
  return static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, 0xF, 0xF, false));
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;

template <class Op>
__device__ __forceinline__ float wave_allreduce(float v, Op op) {
  auto f = [](unsigned b) { return __builtin_bit_cast(float, b); };
  auto u = [](float x) { return __builtin_bit_cast(unsigned, x); };
  v = op(v, f(dpp_bits<kDppXor1>(u(v))));
  v = op(v, f(dpp_bits<kDppXor2>(u(v))));
  v = op(v, f(dpp_bits<kDppHalfMirror>(u(v))));
  v = op(v, f(dpp_bits<kDppMirror>(u(v))));
  { const auto r = __builtin_amdgcn_permlane16_swap(u(v), u(v), false, false); v = op(f(r[0]), f(r[1])); }
  { const auto r = __builtin_amdgcn_permlane32_swap(u(v), u(v), false, false); v = op(f(r[0]), f(r[1])); }
  return v;
}

__device__ __forceinline__ float wave_sum(float v) { return wave_allreduce(v, [](float a, float b) { return a + b; }); }
__device__ __forceinline__ float wave_max(float v) { return wave_allreduce(v, [](float a, float b) { return fmaxf(a, b); }); }

__device__ __forceinline__ double wave_sum(double v) {
  auto mv = [](double x, auto step) {   // both dwords through the same lane movement
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = step(static_cast<unsigned>(b)), hi = step(static_cast<unsigned>(b >> 32));
    return __builtin_bit_cast(double, (static_cast<unsigned long long>(hi) << 32) | lo);
  };
  v += mv(v, [](unsigned w) { return dpp_bits<kDppXor1>(w); });
  v += mv(v, [](unsigned w) { return dpp_bits<kDppXor2>(w); });
  v += mv(v, [](unsigned w) { return dpp_bits<kDppHalfMirror>(w); });
  v += mv(v, [](unsigned w) { return dpp_bits<kDppMirror>(w); });
  {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const auto lo = __builtin_amdgcn_permlane16_swap(static_cast<unsigned>(b), static_cast<unsigned>(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(static_cast<unsigned>(b >> 32), static_cast<unsigned>(b >> 32), false, false);
    v = __builtin_bit_cast(double, (static_cast<unsigned long long>(hi[0]) << 32) | lo[0]) +
        __builtin_bit_cast(double, (static_cast<unsigned long long>(hi[1]) << 32) | lo[1]);
  }
  {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const auto lo = __builtin_amdgcn_permlane32_swap(static_cast<unsigned>(b), static_cast<unsigned>(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(static_cast<unsigned>(b >> 32), static_cast<unsigned>(b >> 32), false, false);
    v = __builtin_bit_cast(double, (static_cast<unsigned long long>(hi[0]) << 32) | lo[0]) +
        __builtin_bit_cast(double, (static_cast<unsigned long long>(hi[1]) << 32) | lo[1]);
  }
  return v;
}

// inclusive scans across the 64 lanes of a wave
// The transmittance scan of the compositing kernels, on DPP: inclusive scan inside each 16-lane row (row_shr 1, 2, 4, 8; a lane
// whose source falls outside its row receives the identity), then the row totals travel down (row_bcast15 into rows 1 and 3,
// row_bcast31 into rows 2 and 3).  Six lane movements per dword and no LDS round trip (the shuffle form: twelve ds_bpermute).
// The products are doubles rounded to fp32 once, by the caller: the association order does not reach the result.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_or_one(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(static_cast<unsigned>(b)), CTRL, ROW_MASK, 0xF, false));
  const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0x3FF00000, static_cast<int>(static_cast<unsigned>(b >> 32)), CTRL, ROW_MASK, 0xF, false));
  return __builtin_bit_cast(double, (static_cast<unsigned long long>(hi) << 32) | lo);
}
__device__ __forceinline__ double wave_scan_mul(double v) {
  v *= dpp_or_one<0x111, 0xF>(v);   // row_shr:1
  v *= dpp_or_one<0x112, 0xF>(v);   // row_shr:2
  v *= dpp_or_one<0x114, 0xF>(v);   // row_shr:4
  v *= dpp_or_one<0x118, 0xF>(v);   // row_shr:8
  v *= dpp_or_one<0x142, 0xA>(v);   // row_bcast15 -> rows 1, 3
  v *= dpp_or_one<0x143, 0xC>(v);   // row_bcast31 -> rows 2, 3
  return v;
}
// lane l - 1's value (lane 0: `first`); the last lane's value, as a wave-uniform number
__device__ __forceinline__ double wave_shift_up1(double v, double first) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v), f = __builtin_bit_cast(unsigned long long, first);
  const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_update_dpp(static_cast<int>(static_cast<unsigned>(f)), static_cast<int>(static_cast<unsigned>(b)), 0x138, 0xF, 0xF, false));   // wave_shr:1
  const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_update_dpp(static_cast<int>(static_cast<unsigned>(f >> 32)), static_cast<int>(static_cast<unsigned>(b >> 32)), 0x138, 0xF, 0xF, false));
  return __builtin_bit_cast(double, (static_cast<unsigned long long>(hi) << 32) | lo);
}
__device__ __forceinline__ double wave_last(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<unsigned>(b)), 63));
  const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<unsigned>(b >> 32)), 63));
  return __builtin_bit_cast(double, (static_cast<unsigned long long>(hi) << 32) | lo);
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
