// Library-level entry points: error string, ABI version, and the fused predict_and_render_radiance
// forward (reference nerf/train_utils.py:92-202) sequenced on one stream from the per-stage kernels.
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "dn_common.h"
#include "composite_body.h"
#include "dn_rng.h"

namespace dn {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int ensure_big_lds(const void* kernel) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  static thread_local std::vector<std::pair<const void*, int>> done;
  for (const auto& kd : done)
    if (kd.first == kernel && kd.second == dev) return 0;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute(device %d): %s", dev, hipGetErrorString(e)); return -static_cast<int>(e); }
  done.emplace_back(kernel, dev);
  return 0;
}

int device_cus() {
  constexpr int kMaxDev = 64;
  static thread_local int cache[kMaxDev] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return 256;
  if (cache[dev] == 0) {
    int cus = 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    cache[dev] = cus;
  }
  return cache[dev];
}

static size_t align256(size_t v) { return (v + 255) & ~static_cast<size_t>(255); }

struct Workspace {
  float *z_c, *rf_c, *w_c, *z_f, *rf_f, *g_rf;
  unsigned* absmax_part;   // training: one word per workgroup of the compositing backward (largest |gradient| it stored)
  unsigned* status;   // the last 256 bytes: word 0 = non-finite raw radiance-field values met by this call's compositing
  size_t bytes;
};

static Workspace carve(void* base, int64_t n, int nc, int nf, bool train = false) {
  Workspace w{};
  size_t off = 0;
  auto take = [&](size_t floats) {
    float* p = base ? reinterpret_cast<float*>(static_cast<char*>(base) + off) : nullptr;
    off += align256(floats * sizeof(float));
    return p;
  };
  w.z_c = take(static_cast<size_t>(n) * nc);
  w.rf_c = take(static_cast<size_t>(n) * nc * 4);
  w.w_c = take(static_cast<size_t>(n) * nc);
  if (nf > 0) {
    w.z_f = take(static_cast<size_t>(n) * (nc + nf));
    w.rf_f = take(static_cast<size_t>(n) * (nc + nf) * 4);
  }
  if (train) w.g_rf = take(static_cast<size_t>(n) * (nc + nf) * 4);   // d loss / d raw radiance field, one network at a time
  if (train) w.absmax_part = reinterpret_cast<unsigned*>(take(static_cast<size_t>((n + 3) / 4)));
  w.status = reinterpret_cast<unsigned*>(take(64));
  w.bytes = off;
  return w;
}

}  // namespace dn

using namespace dn;

extern "C" int dn_abi_version(void) { return DN_ABI_VERSION; }

extern "C" const char* dn_last_error(void) { return g_err; }

extern "C" size_t dn_render_workspace_bytes(int64_t n_rays, int num_coarse, int num_fine) {
  if (n_rays < 0 || num_coarse < 1 || num_fine < 0) return 0;
  return carve(nullptr, n_rays, num_coarse, num_fine).bytes;
}

extern "C" int dn_render_rays(const dn_mlp_desc* desc_coarse, const void* packed_coarse, const dn_mlp_desc* desc_fine,
                              const void* packed_fine, int precision, const float* rays, int ray_stride,
                              int64_t n_rays, int num_coarse, int num_fine, int lindisp, float noise_std,
                              int white_background, const float* h_m_thres, int n_thres, const float* t_rand,
                              const float* noise_c, const float* u, const float* noise_f, float* rgb_c,
                              float* depth_c, float* acc_c, float* rgb_f, float* depth_f, float* acc_f, float* dex_f,
                              void* workspace, dn_stream_t stream) {
  if (n_rays == 0) return 0;
  DN_REQUIRE(desc_coarse && packed_coarse && rays && workspace && n_rays >= 0, "dn_render_rays: bad arguments");
  DN_REQUIRE(num_fine == 0 || (desc_fine && packed_fine), "dn_render_rays: fine pass requested without a fine net");
  DN_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "dn_render_rays: workspace must be 256-byte aligned");
  if (n_rays == 0) return 0;
  Workspace w = carve(workspace, n_rays, num_coarse, num_fine);
  int rc;
  {
    hipError_t e = hipMemsetAsync(w.status, 0, 256, as_stream(stream));
    if (e != hipSuccess) { set_error("dn_render_rays: hipMemsetAsync: %s", hipGetErrorString(e)); return -static_cast<int>(e); }
  }
  if ((rc = dn_coarse_depths(rays, ray_stride, n_rays, num_coarse, lindisp, t_rand, w.z_c, stream))) return rc;
  const bool fine = num_fine > 0;
  DN_REQUIRE(n_thres >= 0 && n_thres <= kMaxThres, "dn_render_rays: at most %d Dex thresholds", kMaxThres);
  // Without density noise the network launch may composite its own rays (the 48-point fixed-shape instances, samples per ray
  // dividing their 384-point tile: the raw radiance field then never goes to HBM); it says whether it did.
  auto comp_for = [&](float* rgb, float* acc, float* weights, float* depth, float* dex, int k) {
    CompParams c{};
    c.rgb = rgb; c.acc = acc; c.weights = weights; c.depth = depth; c.dex = dex; c.disp = nullptr;
    c.nonfinite = w.status; c.n_rays = n_rays; c.n_thres = k; c.white = white_background;
    for (int i = 0; i < kMaxThres; ++i) c.th.m[i] = (i < k) ? h_m_thres[i] : 0.0f;
    return c;
  };
  const bool may_fuse = !(noise_std > 0.0f);
  int done = 0;
  // Dex depths come from the fine pass (train_utils.py:199-201); coarse-only renders report the coarse ones.
  {
    const CompParams c = comp_for(rgb_c, acc_c, w.w_c, depth_c, fine ? nullptr : dex_f, fine ? 0 : n_thres);
    if ((rc = run_network_flagged(desc_coarse, precision, packed_coarse, nullptr, nullptr, rays, ray_stride, w.z_c, n_rays,
                                  num_coarse, w.rf_c, w.status + 1, stream, (may_fuse && rgb_c) ? &c : nullptr, &done)))
      return rc;
  }
  if (!done && (rc = volume_render_counting(w.rf_c, w.z_c, rays + 3, ray_stride, noise_c, noise_std, white_background, h_m_thres,
                                            fine ? 0 : n_thres, n_rays, num_coarse, rgb_c, nullptr, acc_c, w.w_c, depth_c,
                                            fine ? nullptr : dex_f, w.status, stream)))
    return rc;
  if (!fine) return 0;
  if ((rc = dn_fine_depths(w.z_c, w.w_c, u, n_rays, num_coarse, num_fine, w.z_f, nullptr, stream))) return rc;
  {
    const CompParams c = comp_for(rgb_f, acc_f, nullptr, depth_f, dex_f, n_thres);
    if ((rc = run_network_flagged(desc_fine, precision, packed_fine, nullptr, nullptr, rays, ray_stride, w.z_f, n_rays,
                                  num_coarse + num_fine, w.rf_f, w.status + 1, stream, (may_fuse && rgb_f) ? &c : nullptr, &done)))
      return rc;
  }
  if (done) return 0;
  return volume_render_counting(w.rf_f, w.z_f, rays + 3, ray_stride, noise_f, noise_std, white_background, h_m_thres, n_thres,
                                n_rays, num_coarse + num_fine, rgb_f, nullptr, acc_f, nullptr, depth_f, dex_f, w.status, stream);
}

// ---- predict_and_render_radiance under autograd (reference nerf/train_utils.py:92-202 + loss.backward(),
// train_dexnerf_rgb.py:278): one call forward, one call backward -------------------------------------------------------
extern "C" size_t dn_render_train_workspace_bytes(int64_t n_rays, int num_coarse, int num_fine) {
  if (n_rays < 0 || num_coarse < 1 || num_fine < 0) return 0;
  return carve(nullptr, n_rays, num_coarse, num_fine, true).bytes;
}

extern "C" int dn_render_rays_train(const dn_mlp_desc* desc_coarse, const void* packed_coarse, const dn_mlp_desc* desc_fine,
                                    const void* packed_fine, int precision, const float* rays, int ray_stride,
                                    int64_t n_rays, int num_coarse, int num_fine, int lindisp, float noise_std,
                                    int white_background, const float* h_m_thres, int n_thres, const float* t_rand,
                                    const float* noise_c, const float* u, const float* noise_f, float* rgb_c,
                                    float* depth_c, float* acc_c, float* rgb_f, float* depth_f, float* acc_f, float* dex_f,
                                    void* workspace, void* act_c, void* masks_c, void* act_f, void* masks_f,
                                    const uint32_t* rng_state, int perturb, dn_stream_t stream) {
  if (n_rays == 0) return 0;
  DN_REQUIRE(desc_coarse && packed_coarse && rays && workspace && act_c && masks_c && n_rays >= 0, "dn_render_rays_train: bad arguments");
  DN_REQUIRE(num_fine == 0 || (desc_fine && packed_fine && act_f && masks_f), "dn_render_rays_train: fine pass requested without a fine net / its buffers");
  DN_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "dn_render_rays_train: workspace must be 256-byte aligned");
  Workspace w = carve(workspace, n_rays, num_coarse, num_fine, true);
  int rc;
  // a NULL draw with an RNG state: drawn in the kernels (dn_rng.h) - jitter / resampling u only when `perturb`, density noise
  // whenever noise_std > 0; explicit draws win (parity tests inject the reference's)
  const uint32_t* rng_perturb = perturb ? rng_state : nullptr;
  if ((rc = coarse_depths_rng(rays, ray_stride, n_rays, num_coarse, lindisp, t_rand, w.z_c, rng_perturb, stream))) return rc;
  if ((rc = dn_run_network_train(desc_coarse, precision, packed_coarse, nullptr, nullptr, rays, ray_stride, w.z_c, n_rays,
                                 num_coarse, w.rf_c, act_c, masks_c, stream)))
    return rc;
  const bool fine = num_fine > 0;
  if ((rc = volume_render_counting(w.rf_c, w.z_c, rays + 3, ray_stride, noise_c, noise_std, white_background, h_m_thres,
                                   fine ? 0 : n_thres, n_rays, num_coarse, rgb_c, nullptr, acc_c, w.w_c, depth_c,
                                   fine ? nullptr : dex_f, nullptr, stream, rng_state, kRngStreamNoiseCoarse)))
    return rc;
  if (!fine) return 0;
  if ((rc = fine_depths_rng(w.z_c, w.w_c, u, n_rays, num_coarse, num_fine, w.z_f, nullptr, rng_perturb, stream))) return rc;
  if ((rc = dn_run_network_train(desc_fine, precision, packed_fine, nullptr, nullptr, rays, ray_stride, w.z_f, n_rays,
                                 num_coarse + num_fine, w.rf_f, act_f, masks_f, stream)))
    return rc;
  return volume_render_counting(w.rf_f, w.z_f, rays + 3, ray_stride, noise_f, noise_std, white_background, h_m_thres, n_thres,
                                n_rays, num_coarse + num_fine, rgb_f, nullptr, acc_f, nullptr, depth_f, dex_f, nullptr, stream,
                                rng_state, kRngStreamNoiseFine);
}

extern "C" int dn_render_rays_backward_ws(const dn_mlp_desc* desc_coarse, const void* packed_bwd_coarse,
                                          const dn_mlp_desc* desc_fine, const void* packed_bwd_fine, int precision,
                                          const float* rays, int ray_stride, int64_t n_rays, int num_coarse, int num_fine,
                                          float noise_std, int white_background, const float* noise_c, const float* noise_f,
                                          const float* g_rgb_c, const float* g_depth_c, const float* g_acc_c,
                                          const float* g_rgb_f, const float* g_depth_f, const float* g_acc_f, void* workspace,
                                          const void* act_c, const void* masks_c, void* grads_c, const void* act_f,
                                          const void* masks_f, void* grads_f, float* const* h_dW_c, float* const* h_db_c,
                                          float* const* h_dW_f, float* const* h_db_f, int nets, const uint32_t* rng_state,
                                          void* wg_scratch, size_t wg_scratch_bytes, dn_stream_t stream) {
  if (n_rays == 0) return 0;
  DN_REQUIRE(rays && workspace && n_rays >= 0 && (nets & ~3) == 0 && (wg_scratch != nullptr || wg_scratch_bytes == 0), "dn_render_rays_backward: bad arguments");
  DN_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "dn_render_rays_backward: workspace must be 256-byte aligned");
  Workspace w = carve(workspace, n_rays, num_coarse, num_fine, true);
  int rc;
  // both networks, one architecture: the two backward-data chains first, then ONE weight-gradient launch for the layers of both
  // (dn_mlp_weight_grad_pair) - a caller that wants the fine half finished early (its all-reduce under the coarse half) asks for
  // the networks one at a time
  const bool pair_wgrad = nets == 3 && num_fine > 0 && desc_coarse && desc_fine && std::memcmp(desc_coarse, desc_fine, sizeof(dn_mlp_desc)) == 0 &&
                          weight_grad_pair_fits(*desc_fine);   // (both networks' layers in one batch: two D <= 12 networks)
  auto half = [&](const dn_mlp_desc* desc, const void* packed_bwd, const float* rf, const float* z, int samples,
                  const float* noise, const float* g_rgb, const float* g_depth, const float* g_acc, const void* act,
                  const void* masks, void* grads, float* const* h_dW, float* const* h_db, uint32_t noise_stream) -> int {
    DN_REQUIRE(desc && packed_bwd && act && masks && grads && h_dW && h_db, "dn_render_rays_backward: a network's buffers are missing");
    // 8-bit saved tensors with the per-launch gradient scale: the compositing backward leaves the largest |gradient| of each of its
    // workgroups behind, and the network backward reduces those words itself - no pass over g_rf in between (a few thousand words at
    // most: beyond that the separate reduction kernel is the cheaper one)
    const int64_t n_parts = (n_rays + 3) / 4;
    // (DEXNERF_S8_ABSMAX_KERNEL=1, read per call: the separate reduction kernel regardless - the A/B of tests/test_hip_parity.py)
    unsigned* parts = (precision == DN_PREC_BF16_S8 && s8_scale_is_per_launch() && n_parts <= 4096 && std::getenv("DEXNERF_S8_ABSMAX_KERNEL") == nullptr)
                          ? w.absmax_part : nullptr;
    if ((rc = volume_render_backward_rng(rf, z, rays + 3, ray_stride, noise, noise_std, white_background, n_rays, samples,
                                         g_rgb, g_depth, g_acc, nullptr, nullptr, w.g_rf, rng_state, noise_stream, stream, parts)))
      return rc;
    const int64_t n_points = n_rays * samples;
    if ((rc = mlp_backward_data_partials(desc, precision, packed_bwd, w.g_rf, masks, n_points, grads, parts, static_cast<int>(parts ? n_parts : 0), stream))) return rc;
    if (pair_wgrad) return 0;   // both networks' weight gradients follow in one launch
    // (one network at a time: the launches are stream-ordered, so the scratch is free again when the second one starts)
    return dn_mlp_weight_grad_all_ws(desc, precision, act, grads, n_points, h_dW, h_db, wg_scratch, wg_scratch_bytes, stream);
  };
  // the fine network first: autograd's order too (its graph node is the younger one), and the half a data-parallel caller
  // wants finished first so that its all-reduce overlaps the coarse half
  if ((nets & 2) && num_fine > 0) {
    if ((rc = half(desc_fine, packed_bwd_fine, w.rf_f, w.z_f, num_coarse + num_fine, noise_f, g_rgb_f, g_depth_f, g_acc_f,
                   act_f, masks_f, grads_f, h_dW_f, h_db_f, kRngStreamNoiseFine)))
      return rc;
  }
  if (nets & 1) {
    if ((rc = half(desc_coarse, packed_bwd_coarse, w.rf_c, w.z_c, num_coarse, noise_c, g_rgb_c, g_depth_c, g_acc_c, act_c,
                   masks_c, grads_c, h_dW_c, h_db_c, kRngStreamNoiseCoarse)))
      return rc;
  }
  if (pair_wgrad)
    return dn_mlp_weight_grad_pair_ws(desc_fine, precision, act_f, grads_f, n_rays * (num_coarse + num_fine), h_dW_f, h_db_f, act_c, grads_c,
                                      n_rays * num_coarse, h_dW_c, h_db_c, wg_scratch, wg_scratch_bytes, stream);
  return 0;
}

extern "C" int dn_render_rays_backward(const dn_mlp_desc* desc_coarse, const void* packed_bwd_coarse,
                                       const dn_mlp_desc* desc_fine, const void* packed_bwd_fine, int precision,
                                       const float* rays, int ray_stride, int64_t n_rays, int num_coarse, int num_fine,
                                       float noise_std, int white_background, const float* noise_c, const float* noise_f,
                                       const float* g_rgb_c, const float* g_depth_c, const float* g_acc_c,
                                       const float* g_rgb_f, const float* g_depth_f, const float* g_acc_f, void* workspace,
                                       const void* act_c, const void* masks_c, void* grads_c, const void* act_f,
                                       const void* masks_f, void* grads_f, float* const* h_dW_c, float* const* h_db_c,
                                       float* const* h_dW_f, float* const* h_db_f, int nets, const uint32_t* rng_state,
                                       dn_stream_t stream) {
  return dn_render_rays_backward_ws(desc_coarse, packed_bwd_coarse, desc_fine, packed_bwd_fine, precision, rays, ray_stride, n_rays, num_coarse,
                                    num_fine, noise_std, white_background, noise_c, noise_f, g_rgb_c, g_depth_c, g_acc_c, g_rgb_f, g_depth_f,
                                    g_acc_f, workspace, act_c, masks_c, grads_c, act_f, masks_f, grads_f, h_dW_c, h_db_c, h_dW_f, h_db_f, nets,
                                    rng_state, nullptr, 0, stream);
}
