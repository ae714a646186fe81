// volume_render_radiance_field forward / backward (reference nerf/volume_rendering_utils.py:6-70,
// cumprod_exclusive nerf/nerf_helpers.py:43-64) with the Dex-NeRF fixed-sigma depth readout.
//
// One wave64 per ray, samples laid out lane-contiguous (sample s = 64*chunk + lane) so every wave
// instruction reads one coalesced run of the (ray, sample) row; the transmittance product is a
// wave-level fp64 scan with a carry between 64-sample chunks (ATen's CPU cumprod accumulates in double
// and rounds each prefix to fp32), the Dex readout is a ballot + first-set-bit per threshold.
// HBM-bound: 20 B/sample in (rf float4 + z), (10+K)*4 B/ray out.
#include "composite_body.h"

namespace dn {

constexpr int kRaysPerBlock = 4;  // 256 threads

__global__ __launch_bounds__(256) void composite_fwd_kernel(
    const float4* __restrict__ rf, const float* __restrict__ z, const float* __restrict__ rd, int rd_stride,
    const float* __restrict__ noise, float noise_std, int white, ThresArgs th, int n_thres, int64_t n_rays, int S,
    float* __restrict__ rgb, float* __restrict__ disp, float* __restrict__ acc, float* __restrict__ weights,
    float* __restrict__ depth, float* __restrict__ dex, unsigned* __restrict__ nonfinite, RngRef rng) {
  const int lane = lane_id();
  const int64_t ray = static_cast<int64_t>(blockIdx.x) * kRaysPerBlock + (threadIdx.x >> 6);
  if (ray >= n_rays) return;  // wave-uniform exit; no block-level sync in this kernel
  const float4* rfr = rf + ray * S;
  composite_ray([&](int sc) { return rfr[sc]; }, z + ray * S, rd + ray * rd_stride, ray, lane, noise, noise_std, white, th, n_thres, n_rays, S,
                rgb, disp, acc, weights, depth, dex, nonfinite, rng);
}

// Backward w.r.t. rf (checked against autograd of the reference in fp64):
//   s_i = gC.c_i + gD z_i + gA + gW_i ;  R_i = sum_{k>i} s_k w_k ;  dL/dalpha_i = s_i T_i - R_i / o_i ;
//   dL/dsigma_i = dL/dalpha_i * dist_i * exp(-sigma_i dist_i) ; dL/draw_sigma = [raw+noise > 0] * that ;
//   dL/draw_rgb = w_i * gC * c (1 - c).
template <int MAXC>
__global__ __launch_bounds__(256) void composite_bwd_kernel(
    const float4* __restrict__ rf, const float* __restrict__ z, const float* __restrict__ rd, int rd_stride,
    const float* __restrict__ noise, float noise_std, int white, int64_t n_rays, int S, const float* __restrict__ g_rgb,
    const float* __restrict__ g_depth, const float* __restrict__ g_acc, const float* __restrict__ g_disp,
    const float* __restrict__ g_weights, float4* __restrict__ g_rf, RngRef rng, unsigned* __restrict__ absmax_part) {
  const int lane = lane_id();
  const int64_t ray_of_wave = static_cast<int64_t>(blockIdx.x) * kRaysPerBlock + (threadIdx.x >> 6);
  // absmax_part (the 8-bit-saved-tensor training path, api.cpp): this workgroup's largest finite |gradient it stores| goes to
  // absmax_part[blockIdx.x] - what absmax_kernel (mlp_train48.hip) would otherwise read g_rf again for.  Every workgroup writes its
  // word (nothing needs zeroing), so a wave without a ray stays for the workgroup's exchange: it works on the last ray and stores nothing.
  const bool live = ray_of_wave < n_rays;
  if (!live && absmax_part == nullptr) return;
  const int64_t ray = live ? ray_of_wave : n_rays - 1;
  float gmax = 0.0f;
  const float dx = rd[ray * rd_stride + 0], dy = rd[ray * rd_stride + 1], dz = rd[ray * rd_stride + 2];
  const float rd_norm = sqrtf((dx * dx + dy * dy) + dz * dz);
  const float* zr = z + ray * S;
  const float4* rfr = rf + ray * S;
  float gc0 = 0.f, gc1 = 0.f, gc2 = 0.f, gd = 0.f, ga = 0.f;
  if (g_rgb != nullptr) { gc0 = g_rgb[ray * 3]; gc1 = g_rgb[ray * 3 + 1]; gc2 = g_rgb[ray * 3 + 2]; }
  if (g_depth != nullptr) gd = g_depth[ray];
  if (g_acc != nullptr) ga = g_acc[ray];
  if (white) ga -= (gc0 + gc1 + gc2);

  float w_[MAXC], t_[MAXC], om_[MAXC], de_[MAXC], sw_[MAXC], s_[MAXC], c0_[MAXC], c1_[MAXC], c2_[MAXC];
  bool pos_[MAXC];
  double carry = 1.0;
  float sum_d = 0.f, sum_a = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int base = c * 64;
    const int s = base + lane;
    const bool valid = s < S;
    const int sc = valid ? s : S - 1;
    const float4 raw = rfr[sc];
    const float z0 = zr[sc];
    const float z1 = (sc + 1 < S) ? zr[sc + 1] : z0;
    // (the forward's noise again: the same pure function of (seed, iteration, stream, element) when it was drawn in the kernel)
    const float nz = noise_std > 0.0f ? (noise != nullptr ? noise[ray * S + sc]
                                                          : (rng.state != nullptr ? rng_normal(rng, static_cast<uint64_t>(ray) * S + sc) : 0.0f)) : 0.0f;
    const SampleTerms t = sample_terms(raw.w, nz, noise_std, z0, z1, sc == S - 1, rd_norm);
    const double f = valid ? static_cast<double>(t.one_m_alpha) : 1.0;
    const double incl = wave_scan_mul(f) * carry;
    const double excl = wave_shift_up1(incl, carry);
    carry = wave_last(incl);
    const float trans = (s == 0) ? 1.0f : static_cast<float>(excl);
    const float w = valid ? t.alpha * trans : 0.0f;
    c0_[c] = sigmoidf_(raw.x); c1_[c] = sigmoidf_(raw.y); c2_[c] = sigmoidf_(raw.z);
    w_[c] = w; t_[c] = trans; om_[c] = t.one_m_alpha;
    de_[c] = t.dist * expf(-t.sigma * t.dist);  // dalpha/dsigma
    pos_[c] = valid && (t.sigma > 0.0f);
    s_[c] = z0;  // z for now; turned into s_i below once gD is final
    sum_d += w * z0;
    sum_a += w;
  }
  if (g_disp != nullptr) {
    // disp = 1 / max(1e-10, depth/acc): only the q > 1e-10 branch has a gradient
    sum_d = wave_sum(sum_d);
    sum_a = wave_sum(sum_a);
    const float q = sum_d / sum_a;
    if (q > 1e-10f) {
      const float gq = -g_disp[ray] / (q * q);
      gd += gq / sum_a;
      ga += -gq * sum_d / (sum_a * sum_a);
    }
  }
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int s = c * 64 + lane;
    float si = (gc0 * c0_[c] + gc1 * c1_[c] + gc2 * c2_[c]) + gd * s_[c] + ga;
    if (g_weights != nullptr && s < S) si += g_weights[ray * S + s];
    s_[c] = si;
    sw_[c] = (s < S) ? si * w_[c] : 0.0f;
  }
  // reverse exclusive scan of s_k w_k over the whole ray
  float tail = 0.0f;  // sum over all later chunks
#pragma unroll
  for (int c = MAXC - 1; c >= 0; --c) {
    const float incl = wave_scan_add(sw_[c]);            // prefix within chunk
    const float total = __shfl(incl, 63, 64);
    const float suffix_excl = (total - incl) + tail;     // sum_{k > i} within chunk + later chunks
    tail += total;
    const int s = c * 64 + lane;
    if (s < S) {
      const float dalpha = s_[c] * t_[c] - suffix_excl / om_[c];
      const float dsig = pos_[c] ? dalpha * de_[c] : 0.0f;
      float4 g;
      g.x = w_[c] * gc0 * c0_[c] * (1.0f - c0_[c]);
      g.y = w_[c] * gc1 * c1_[c] * (1.0f - c1_[c]);
      g.z = w_[c] * gc2 * c2_[c] * (1.0f - c2_[c]);
      g.w = dsig;
      if (live) {
        g_rf[ray * S + s] = g;
        const float a0 = fabsf(g.x), a1 = fabsf(g.y), a2 = fabsf(g.z), a3 = fabsf(g.w);
        gmax = (a0 < 3.0e38f) ? fmaxf(gmax, a0) : gmax;   // (a non-finite gradient does not set the scale: absmax_kernel's rule)
        gmax = (a1 < 3.0e38f) ? fmaxf(gmax, a1) : gmax;
        gmax = (a2 < 3.0e38f) ? fmaxf(gmax, a2) : gmax;
        gmax = (a3 < 3.0e38f) ? fmaxf(gmax, a3) : gmax;
      }
    }
  }
  if (absmax_part != nullptr) {   // (workgroup-uniform)
    __shared__ float wave_max[kRaysPerBlock];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o, 64));
    if (lane == 0) wave_max[threadIdx.x >> 6] = gmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      float r = wave_max[0];
#pragma unroll
      for (int w = 1; w < kRaysPerBlock; ++w) r = fmaxf(r, wave_max[w]);
      absmax_part[blockIdx.x] = __float_as_uint(r);
    }
  }
}

}  // namespace dn

using namespace dn;

extern "C" int dn_volume_render(const float* rf, const float* z, const float* rd, int rd_stride, const float* noise,
                                float noise_std, int white_background, const float* h_m_thres, int n_thres,
                                int64_t n_rays, int n_samples, float* rgb, float* disp, float* acc, float* weights,
                                float* depth, float* dex, dn_stream_t stream) {
  return dn::volume_render_counting(rf, z, rd, rd_stride, noise, noise_std, white_background, h_m_thres, n_thres, n_rays, n_samples,
                                    rgb, disp, acc, weights, depth, dex, nullptr, stream);
}

// dn_volume_render + a count of the non-finite raw radiance-field values it met, added to *nonfinite (device word, may be NULL)
int dn::volume_render_counting(const float* rf, const float* z, const float* rd, int rd_stride, const float* noise,
                               float noise_std, int white_background, const float* h_m_thres, int n_thres,
                               int64_t n_rays, int n_samples, float* rgb, float* disp, float* acc, float* weights,
                               float* depth, float* dex, unsigned* nonfinite, dn_stream_t stream, const uint32_t* rng_state,
                               uint32_t rng_stream) {
  if (n_rays == 0) return 0;
  DN_REQUIRE(rf && z && rd && n_rays >= 0 && n_samples >= 1 && rd_stride >= 3, "dn_volume_render: bad arguments");
  DN_REQUIRE(n_thres >= 0 && n_thres <= kMaxThres, "dn_volume_render: at most %d Dex thresholds", kMaxThres);
  DN_REQUIRE(n_thres == 0 || (h_m_thres && dex), "dn_volume_render: thresholds given without dex output");
  DN_REQUIRE((reinterpret_cast<uintptr_t>(rf) & 15) == 0, "dn_volume_render: rf must be 16-byte aligned");
  if (n_rays == 0) return 0;
  ThresArgs th;
  for (int k = 0; k < kMaxThres; ++k) th.m[k] = (k < n_thres) ? h_m_thres[k] : 0.0f;
  const unsigned grid = static_cast<unsigned>((n_rays + kRaysPerBlock - 1) / kRaysPerBlock);
  hipLaunchKernelGGL(composite_fwd_kernel, dim3(grid), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float4*>(rf), z, rd, rd_stride, noise, noise_std, white_background, th,
                     n_thres, n_rays, n_samples, rgb, disp, acc, weights, depth, dex, nonfinite, RngRef{rng_state, rng_stream});
  return check_launch("dn_volume_render");
}

extern "C" int dn_volume_render_backward(const float* rf, const float* z, const float* rd, int rd_stride,
                                         const float* noise, float noise_std, int white_background, int64_t n_rays,
                                         int n_samples, const float* g_rgb, const float* g_depth, const float* g_acc,
                                         const float* g_disp, const float* g_weights, float* g_rf,
                                         dn_stream_t stream) {
  return dn::volume_render_backward_rng(rf, z, rd, rd_stride, noise, noise_std, white_background, n_rays, n_samples, g_rgb, g_depth, g_acc,
                                        g_disp, g_weights, g_rf, nullptr, 0u, stream, nullptr);
}

int dn::volume_render_backward_rng(const float* rf, const float* z, const float* rd, int rd_stride, const float* noise, float noise_std,
                                   int white_background, int64_t n_rays, int n_samples, const float* g_rgb, const float* g_depth,
                                   const float* g_acc, const float* g_disp, const float* g_weights, float* g_rf,
                                   const uint32_t* rng_state, uint32_t rng_stream, dn_stream_t stream, unsigned* absmax_part) {
  if (n_rays == 0) return 0;
  DN_REQUIRE(rf && z && rd && g_rf && n_rays >= 0 && n_samples >= 1 && rd_stride >= 3,
             "dn_volume_render_backward: bad arguments");
  DN_REQUIRE(n_samples <= 1024, "dn_volume_render_backward: at most 1024 samples per ray");
  DN_REQUIRE(((reinterpret_cast<uintptr_t>(rf) | reinterpret_cast<uintptr_t>(g_rf)) & 15) == 0,
             "dn_volume_render_backward: rf / g_rf must be 16-byte aligned");
  if (n_rays == 0) return 0;
  const unsigned grid = static_cast<unsigned>((n_rays + kRaysPerBlock - 1) / kRaysPerBlock);
  const int chunks = (n_samples + 63) / 64;
#define DN_LAUNCH_BWD(MC)                                                                                          \
  hipLaunchKernelGGL(composite_bwd_kernel<MC>, dim3(grid), dim3(256), 0, as_stream(stream),                        \
                     reinterpret_cast<const float4*>(rf), z, rd, rd_stride, noise, noise_std, white_background,    \
                     n_rays, n_samples, g_rgb, g_depth, g_acc, g_disp, g_weights, reinterpret_cast<float4*>(g_rf), RngRef{rng_state, rng_stream}, absmax_part)
  if (chunks <= 1) DN_LAUNCH_BWD(1);
  else if (chunks <= 2) DN_LAUNCH_BWD(2);
  else if (chunks <= 4) DN_LAUNCH_BWD(4);
  else if (chunks <= 8) DN_LAUNCH_BWD(8);
  else DN_LAUNCH_BWD(16);
#undef DN_LAUNCH_BWD
  return check_launch("dn_volume_render_backward");
}
