// The forward compositing of ONE ray by one wave (volume_render_radiance_field, reference nerf/volume_rendering_utils.py:6-70):
// shared by composite_fwd_kernel (composite.hip: raw radiance field from HBM) and the fused network kernel's epilogue
// (mlp_fused48.hip: from the rows the wave tile has just staged in LDS) - one body, so the two produce the same bits.
#pragma once
#include "dn_common.h"
#include "dn_rng.h"

namespace dn {

constexpr int kMaxThres = 64;

struct ThresArgs {
  float m[kMaxThres];
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

struct SampleTerms {
  float sigma, alpha, one_m_alpha, dist;
};

__device__ __forceinline__ SampleTerms sample_terms(float raw_sigma, float noise, float noise_std, float z0, float z1,
                                                    bool last, float rd_norm) {
  SampleTerms t;
  float raw = raw_sigma;
  if (noise_std > 0.0f) raw = raw + noise * noise_std;
  t.sigma = fmaxf(raw, 0.0f);
  t.dist = (last ? 1e10f : (z1 - z0)) * rd_norm;
  t.alpha = 1.0f - expf(-t.sigma * t.dist);
  t.one_m_alpha = (1.0f - t.alpha) + 1e-10f;
  return t;
}

// what the fused network kernel needs to composite the rays it has finished (FwdParams::comp; rgb == NULL: not requested)
struct CompParams {
  float* rgb; float* disp; float* acc; float* weights; float* depth; float* dex;
  unsigned* nonfinite;
  int64_t n_rays;
  int n_thres, white;
  ThresArgs th;
};

// raw_at(sc): the ray's raw radiance-field row of sample sc; zr: its depths; rd3: its direction
template <class RawAt>
__device__ __forceinline__ void composite_ray(RawAt raw_at, const float* __restrict__ zr, const float* __restrict__ rd3, int64_t ray, int lane,
                                              const float* __restrict__ noise, float noise_std, int white, const ThresArgs& th, int n_thres,
                                              int64_t n_rays, int S, float* __restrict__ rgb, float* __restrict__ disp, float* __restrict__ acc,
                                              float* __restrict__ weights, float* __restrict__ depth, float* __restrict__ dex,
                                              unsigned* __restrict__ nonfinite, RngRef rng) {
  unsigned n_bad = 0;   // non-finite raw values of this lane's samples (counted only when the caller asks: fp16 overflow guard)
  const float dx = rd3[0], dy = rd3[1], dz = rd3[2];
  const float rd_norm = sqrtf((dx * dx + dy * dy) + dz * dz);
  double carry = 1.0;  // prod_{j < chunk start} (1 - alpha_j + 1e-10), kept in fp64
  float s_r = 0.f, s_g = 0.f, s_b = 0.f, s_d = 0.f, s_a = 0.f;
  int first_idx = -1;  // lane k tracks threshold k
  unsigned long long found = 0ull;   // wave-uniform: thresholds whose first crossing is already known
  for (int base = 0; base < S; base += 64) {
    const int s = base + lane;
    const bool valid = s < S;
    const int sc = valid ? s : S - 1;
    const float4 raw = raw_at(sc);
    if (nonfinite != nullptr && valid) {
      // x - x is 0 for every finite x and NaN for +-inf / NaN
      const float probe = ((raw.x - raw.x) + (raw.y - raw.y)) + ((raw.z - raw.z) + (raw.w - raw.w));
      n_bad += (probe != 0.0f) ? 1u : 0u;
    }
    const float z0 = zr[sc];
    const float z1 = (sc + 1 < S) ? zr[sc + 1] : z0;
    const float nz = noise_std > 0.0f ? (noise != nullptr ? noise[ray * S + sc]
                                                          : (rng.state != nullptr ? rng_normal(rng, static_cast<uint64_t>(ray) * S + sc) : 0.0f)) : 0.0f;
    const SampleTerms t = sample_terms(raw.w, nz, noise_std, z0, z1, sc == S - 1, rd_norm);
    const double f = valid ? static_cast<double>(t.one_m_alpha) : 1.0;
    const double incl = wave_scan_mul(f) * carry;
    const double excl = wave_shift_up1(incl, carry);
    carry = wave_last(incl);
    // reference: cumprod (fp64 accumulate, fp32 per element), rolled by one, [0] = 1
    const float trans = (s == 0) ? 1.0f : static_cast<float>(excl);
    const float w = valid ? t.alpha * trans : 0.0f;
    if (valid && weights != nullptr) weights[ray * S + s] = w;
    s_r += w * sigmoidf_(raw.x);
    s_g += w * sigmoidf_(raw.y);
    s_b += w * sigmoidf_(raw.z);
    s_d += w * z0;
    s_a += w;
    // Dex readout (volume_rendering_utils.py:51-58): first sample whose sigma exceeds m_k, per threshold.  A ballot per
    // threshold per chunk doubled this kernel's time (346 vs 190 us for 160,000 x 192 samples, K = 20); most (chunk,
    // threshold) pairs cannot hit: the threshold was crossed in an earlier chunk, or no sigma of this chunk reaches it.
    // Both are wave-uniform facts - a found mask and the chunk's maximum - so those pairs cost two scalar branches.
    if (n_thres > 0) {
      float cmax = valid ? t.sigma : 0.0f;   // sigma >= 0 (relu) and every threshold of interest is >= 0 ... but keep it general:
      if (!valid) cmax = -__builtin_inff();
      cmax = wave_max(cmax);
      const float cmax_u = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, cmax)));
      for (int k = 0; k < n_thres; ++k) {
        if ((found >> k) & 1ull) continue;
        const float m = th.m[k];
        if (!(cmax_u > m)) continue;
        const unsigned long long hit = __ballot(valid && (t.sigma > m));   // non-empty: cmax > m
        if (lane == k) first_idx = base + __builtin_ctzll(hit);
        found |= 1ull << k;
      }
    }
  }
  s_r = wave_sum(s_r); s_g = wave_sum(s_g); s_b = wave_sum(s_b); s_d = wave_sum(s_d); s_a = wave_sum(s_a);
  if (nonfinite != nullptr && __ballot(n_bad != 0u) != 0ull) {   // (wave-uniform branch; never taken on healthy weights)
    const unsigned total = static_cast<unsigned>(wave_sum(static_cast<float>(n_bad)));
    if (lane == 0) atomicAdd(nonfinite, total);
  }
  if (lane == 0) {
    if (white) {
      const float bg = 1.0f - s_a;
      s_r += bg; s_g += bg; s_b += bg;
    }
    if (rgb != nullptr) {
      rgb[ray * 3 + 0] = s_r; rgb[ray * 3 + 1] = s_g; rgb[ray * 3 + 2] = s_b;
    }
    if (depth != nullptr) depth[ray] = s_d;
    if (acc != nullptr) acc[ray] = s_a;
    if (disp != nullptr) {
      const float q = s_d / s_a;  // NaN when acc == 0; torch.max propagates it
      disp[ray] = 1.0f / ((q != q) ? q : fmaxf(1e-10f, q));
    }
  }
  if (lane < n_thres && dex != nullptr) {
    // argmax of an all-zero row is index 0 -> z[0] (volume_rendering_utils.py:54-58)
    dex[static_cast<int64_t>(lane) * n_rays + ray] = zr[first_idx < 0 ? 0 : first_idx];
  }
}

}  // namespace dn
