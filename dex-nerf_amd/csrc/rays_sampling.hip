// Ray generation, coarse depths, standalone positional encoding, inverse-CDF sampler and the
// coarse+fine depth merge.  HBM-bound elementwise / wave-scan kernels (one wave64 per ray for the
// sampler); compiled with -ffp-contract=off so plain mul/add sequences round like ATen's.
#include "dn_common.h"
#include "dn_rng.h"

namespace dn {

// Cross-lane hand-off through the wave's OWN LDS rows (one lane writes, another reads): a wave's DS instructions execute in
// order, so no s_barrier is needed - but the compiler must not move the reads above the neighbouring lanes' writes.  The bare
// wave_barrier intrinsic does not order memory for alias analysis; the wavefront-scope fence does (it emits no instruction
// beyond, at most, an s_waitcnt lgkmcnt).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// ------------------------------------------------------------------------------------------------
// S1 get_ray_bundle (reference nerf/nerf_helpers.py:67-112)
// ------------------------------------------------------------------------------------------------
struct RayBundleArgs {
  float rinv[9];
  float origin[3];
  float fx, cx, cy;
  int height, width;
};

__global__ void ray_bundle_kernel(RayBundleArgs a, float* __restrict__ ro, float* __restrict__ rd) {
  const int64_t pix = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t total = static_cast<int64_t>(a.height) * a.width;
  if (pix >= total) return;
  const int row = static_cast<int>(pix / a.width);
  const int col = static_cast<int>(pix - static_cast<int64_t>(row) * a.width);
  // dir = [(ii-cx)/fx, (jj-cy)/fx, 1]: fx divides the y term too (nerf_helpers.py:100-101)
  const float d0 = (static_cast<float>(col) - a.cx) / a.fx;
  const float d1 = (static_cast<float>(row) - a.cy) / a.fx;
  const float d2 = 1.0f;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    // sum over the last dim of dir[None,:] * Rinv  ->  ((p0 + p1) + p2), products rounded first
    const float p0 = d0 * a.rinv[3 * j + 0];
    const float p1 = d1 * a.rinv[3 * j + 1];
    const float p2 = d2 * a.rinv[3 * j + 2];
    rd[pix * 3 + j] = (p0 + p1) + p2;
    ro[pix * 3 + j] = a.origin[j];
  }
}

// Training-ray selection (reference train_dexnerf_rgb.py:229-242 + the packing of train_utils.py:225-250): for each
// chosen pixel build the packed ray row [ro3, rd3, near, far, viewdir3] directly (same arithmetic as
// ray_bundle_kernel for rd; viewdir = rd / ||rd||, train_utils.py:225) and gather the target pixel's RGB.
__global__ void select_rays_kernel(RayBundleArgs a, float near, float far, const int64_t* __restrict__ pix, int64_t n,
                                   const float* __restrict__ image, int channels, float* __restrict__ rays,
                                   float* __restrict__ target) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t px = pix[i];
  const int row = static_cast<int>(px / a.width);
  const int col = static_cast<int>(px - static_cast<int64_t>(row) * a.width);
  const float d0 = (static_cast<float>(col) - a.cx) / a.fx;
  const float d1 = (static_cast<float>(row) - a.cy) / a.fx;
  const float d2 = 1.0f;
  float rd[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float p0 = d0 * a.rinv[3 * j + 0];
    const float p1 = d1 * a.rinv[3 * j + 1];
    const float p2 = d2 * a.rinv[3 * j + 2];
    rd[j] = (p0 + p1) + p2;
  }
  const float nrm = sqrtf((rd[0] * rd[0] + rd[1] * rd[1]) + rd[2] * rd[2]);
  float* r = rays + i * 11;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    r[j] = a.origin[j];
    r[3 + j] = rd[j];
    r[8 + j] = rd[j] / nrm;
  }
  r[6] = near;
  r[7] = far;
  if (target != nullptr) {
#pragma unroll
    for (int c = 0; c < 3; ++c) target[i * 3 + c] = image[px * channels + c];
  }
}

// The same with the camera chosen on the device: `cams` holds one 16-float record per training view
// [rinv9, origin3, fx, cx, cy, -] and `view` is a device scalar, so a captured HIP graph of the whole training
// iteration can be replayed for any view (host-side camera constants would be frozen into the graph).
// pix == NULL: the pixels are DRAWN here - element i of this iteration's draw without replacement, a keyed permutation of the
// H W pixels (dn_rng.h feistel_permute; reference train_dexnerf_rgb.py:229-236: np.random.choice(H W, n, replace=False)) - from
// the RNG state's NEXT iteration counter, which thread 0 then publishes as the CURRENT one for the rest of the iteration.
__global__ void select_rays_indirect_kernel(const float* __restrict__ cams, const int* __restrict__ view, int n_views, int height, int width,
                                            float near, float far, const int64_t* __restrict__ pix, int64_t n,
                                            const float* __restrict__ images, int channels, float* __restrict__ rays,
                                            float* __restrict__ target, uint32_t* __restrict__ rng_state, int64_t* __restrict__ pix_out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  uint32_t iteration = 0;
  if (pix == nullptr) {
    iteration = rng_state[3];
    if (i == 0) rng_state[2] = iteration;   // (nobody in this launch reads word 2)
  }
  if (i >= n) return;
  int v;
  if (view != nullptr) v = *view;
  else {   // the iteration's training view drawn here too (reference: img_idx = np.random.choice(i_train), train_dexnerf_rgb.py:223)
    uint32_t w[4];
    rng_words(rng_state[0], rng_state[1], iteration, kRngStreamView, 0, w);
    v = static_cast<int>(w[0] % static_cast<uint32_t>(n_views));
  }
  const float* cam = cams + static_cast<int64_t>(v) * 16;
  const float fx = cam[12], cx = cam[13], cy = cam[14];
  int64_t px;
  if (pix != nullptr) px = pix[i];
  else {
    px = feistel_permute(static_cast<uint32_t>(i), static_cast<uint32_t>(height) * static_cast<uint32_t>(width), rng_state[0], rng_state[1], iteration);
    if (pix_out != nullptr) pix_out[i] = px;
  }
  const int row = static_cast<int>(px / width);
  const int col = static_cast<int>(px - static_cast<int64_t>(row) * width);
  const float d0 = (static_cast<float>(col) - cx) / fx;
  const float d1 = (static_cast<float>(row) - cy) / fx;
  const float d2 = 1.0f;
  float rd[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float p0 = d0 * cam[3 * j + 0];
    const float p1 = d1 * cam[3 * j + 1];
    const float p2 = d2 * cam[3 * j + 2];
    rd[j] = (p0 + p1) + p2;
  }
  const float nrm = sqrtf((rd[0] * rd[0] + rd[1] * rd[1]) + rd[2] * rd[2]);
  float* r = rays + i * 11;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    r[j] = cam[9 + j];
    r[3 + j] = rd[j];
    r[8 + j] = rd[j] / nrm;
  }
  r[6] = near;
  r[7] = far;
  if (target != nullptr) {
    const float* img = images + static_cast<int64_t>(v) * height * width * channels;
#pragma unroll
    for (int c = 0; c < 3; ++c) target[i * 3 + c] = img[px * channels + c];
  }
}

// Forward-facing NDC warp (reference nerf/nerf_helpers.py:172-199), op for op (compiled -ffp-contract=off).
__global__ void ndc_rays_kernel(double h, double w, double focal, double near_d, const float* __restrict__ ro,
                                const float* __restrict__ rd, int64_t n, float* __restrict__ ro_out,
                                float* __restrict__ rd_out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float dx = rd[i * 3 + 0], dy = rd[i * 3 + 1], dz = rd[i * 3 + 2];
  const float near = static_cast<float>(near_d);
  const float t = -(near + ro[i * 3 + 2]) / dz;
  const float ox = ro[i * 3 + 0] + t * dx, oy = ro[i * 3 + 1] + t * dy, oz = ro[i * 3 + 2] + t * dz;
  // python-float constants are computed in double and then meet fp32 tensors as fp32 scalars
  const float cw = static_cast<float>(-1.0 / (w / (2.0 * focal)));
  const float ch = static_cast<float>(-1.0 / (h / (2.0 * focal)));
  const float two_near = static_cast<float>(2.0 * near_d);
  ro_out[i * 3 + 0] = cw * ox / oz;
  ro_out[i * 3 + 1] = ch * oy / oz;
  ro_out[i * 3 + 2] = 1.0f + two_near / oz;
  rd_out[i * 3 + 0] = cw * (dx / dz - ox / oz);
  rd_out[i * 3 + 1] = ch * (dy / dz - oy / oz);
  rd_out[i * 3 + 2] = -two_near / oz;
}

// ------------------------------------------------------------------------------------------------
// S3 coarse depths (reference nerf/train_utils.py:111-133)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float coarse_z_at(float near, float far, int nc, int i, int lindisp) {
  const float t = linspace_elem(0.0f, 1.0f, nc, i);
  if (!lindisp) return near * (1.0f - t) + far * t;
  return 1.0f / (1.0f / near * (1.0f - t) + 1.0f / far * t);
}

__global__ void coarse_depths_kernel(const float* __restrict__ rays, int ray_stride, int64_t n_rays, int nc,
                                     int lindisp, const float* __restrict__ t_rand, float* __restrict__ z, RngRef rng) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= n_rays * nc) return;
  const int64_t r = idx / nc;
  const int i = static_cast<int>(idx - r * nc);
  const float near = rays[r * ray_stride + 6];
  const float far = rays[r * ray_stride + 7];
  const float zi = coarse_z_at(near, far, nc, i, lindisp);
  if (t_rand == nullptr && rng.state == nullptr) {
    z[idx] = zi;
    return;
  }
  const float z_last = coarse_z_at(near, far, nc, nc - 1, lindisp);
  const float z_first = coarse_z_at(near, far, nc, 0, lindisp);
  const float upper = (i < nc - 1) ? 0.5f * (coarse_z_at(near, far, nc, i + 1, lindisp) + zi) : z_last;
  const float lower = (i > 0) ? 0.5f * (zi + coarse_z_at(near, far, nc, i - 1, lindisp)) : z_first;
  const float t = (t_rand != nullptr) ? t_rand[idx] : rng_uniform(rng, static_cast<uint64_t>(idx));   // (drawn here: dn_rng.h)
  z[idx] = lower + (upper - lower) * t;
}

// ------------------------------------------------------------------------------------------------
// S5 positional_encoding (reference nerf/nerf_helpers.py:115-159)
// ------------------------------------------------------------------------------------------------
struct FreqArgs {
  float f[32];
};

__global__ void posenc_kernel(const float* __restrict__ x, int64_t n_elems, int dim, int num_fns, int include_input,
                              FreqArgs fr, float* __restrict__ out) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= n_elems) return;
  const int64_t p = idx / dim;
  const int c = static_cast<int>(idx - p * dim);
  const int width = dim * ((include_input ? 1 : 0) + 2 * num_fns);
  float* o = out + p * width;
  const float v = x[idx];
  int base = 0;
  if (include_input) {
    o[c] = v;
    base = dim;
  }
  for (int k = 0; k < num_fns; ++k) {
    const float arg = v * fr.f[k];
    o[base + (2 * k) * dim + c] = sinf(arg);
    o[base + (2 * k + 1) * dim + c] = cosf(arg);
  }
}

// ------------------------------------------------------------------------------------------------
// S7 sample_pdf_2 (reference nerf/nerf_helpers.py:262-304) + searchsorted(side="right")
//   One wave64 per ray.  Bit-exact index recipe:
//     sum  : ATen-CPU association order (8-lane vectors, 4 ILP accumulators, cascade levels);
//     cdf  : fp64 prefix sums rounded to fp32 per element - here a wave-level fp64 scan, which is
//            bit-identical to the sequential one because every partial sum of <=2^6 fp32 pdf values
//            in [2^-24, 1] is exactly representable in fp64 (no rounding happens at all);
//     inds : count of cdf entries <= u (upper bound by binary search over the LDS-resident cdf).
// ------------------------------------------------------------------------------------------------
constexpr int kSamplerWaves = 4;  // rays per 256-thread block

// ATen multi_row_sum / row_sum / vectorized_inner_sum order for a contiguous fp32 row of length L held
// in LDS.  Lanes 0..7 each own one vector lane; the result is broadcast to the whole wave.
__device__ float aten_order_sum(const float* w, int L) {
  const int lane = lane_id();
  const int nvec = L >> 3;
  const int groups = nvec >> 2;
  float part = 0.0f;
  if (lane < 8) {
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = 0.0f;
    int ceil_log2 = 0;
    while ((1 << ceil_log2) < groups) ++ceil_log2;
    const int level_power = max(4, ceil_log2 / 4);
    const int level_step = 1 << level_power;
    const int level_mask = level_step - 1;
    int i = 0;
    for (; i + level_step <= groups;) {
      for (int j = 0; j < level_step; ++j, ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[0][k] += w[(i * 4 + k) * 8 + lane];
      }
      for (int j = 1; j < 4; ++j) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          acc[j][k] += acc[j - 1][k];
          acc[j - 1][k] = 0.0f;
        }
        const int mask = level_mask << (j * level_power);
        if ((i & mask) != 0) break;
      }
    }
    for (; i < groups; ++i) {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[0][k] += w[(i * 4 + k) * 8 + lane];
    }
    for (int j = 1; j < 4; ++j) {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[0][k] += acc[j][k];
    }
    for (int v = groups * 4; v < nvec; ++v) acc[0][0] += w[v * 8 + lane];
    part = ((acc[0][0] + acc[0][1]) + acc[0][2]) + acc[0][3];
  }
  float total = 0.0f;
  for (int k = nvec * 8; k < L; ++k) total += w[k];
#pragma unroll
  for (int k = 0; k < 8; ++k) total += __shfl(part, k, 64);
  return total;  // identical in every lane
}

// Builds cdf[0..B) in LDS from w[0..B-1) (already +1e-5) ; w is overwritten with the pdf.
__device__ void build_cdf(float* w, float* cdf, int B) {
  const int lane = lane_id();
  const int L = B - 1;
  const float s = aten_order_sum(w, L);
  double carry = 0.0;
  if (lane == 0) cdf[0] = 0.0f;
  for (int base = 0; base < L; base += 64) {
    const int i = base + lane;
    const float pdf = (i < L) ? w[i] / s : 0.0f;
    const double inc = wave_scan_add(static_cast<double>(pdf)) + carry;
    if (i < L) cdf[i + 1] = static_cast<float>(inc);
    carry = __shfl(inc, 63, 64);
  }
}

__device__ __forceinline__ float invert_cdf(const float* cdf, const float* bins, int B, float u, int* ind_out) {
  int lo = 0, hi = B;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
  }
  *ind_out = lo;
  const int below = max(lo - 1, 0);
  const int above = min(lo, B - 1);
  const float cb = cdf[below], ca = cdf[above];
  const float bb = bins[below], ba = bins[above];
  float denom = ca - cb;
  if (denom < 1e-5f) denom = 1.0f;
  const float t = (u - cb) / denom;
  return bb + t * (ba - bb);
}

// mode 0: bins/weights given (dn_sample_pdf).  mode 1: z_coarse/weights_coarse given (dn_fine_depths):
// bins = z_mid, w = weights[1:-1], then z_fine = sort(cat(z_coarse, samples)).
template <int MODE>
__global__ __launch_bounds__(256) void sampler_kernel(const float* __restrict__ bins_or_z, const float* __restrict__ weights,
                                                      const float* __restrict__ u, int64_t n_rays, int B, int nf,
                                                      float* __restrict__ samples, int64_t* __restrict__ inds,
                                                      float* __restrict__ z_fine, int sort_len, RngRef rng) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = threadIdx.x >> 6;
  const int lane = lane_id();
  const int64_t ray_raw = static_cast<int64_t>(blockIdx.x) * kSamplerWaves + wave;
  const bool live = ray_raw < n_rays;
  const int64_t ray = live ? ray_raw : (n_rays - 1);
  const int per_wave = 3 * (B + 1) + sort_len;
  float* w = lds + wave * per_wave;   // B-1 used
  float* cdf = w + (B + 1);           // B
  float* bins = cdf + (B + 1);        // B
  float* sbuf = bins + (B + 1);       // sort_len (MODE 1)
  const int L = B - 1;
  if (MODE == 0) {
    for (int i = lane; i < B; i += 64) bins[i] = bins_or_z[ray * B + i];
    for (int i = lane; i < L; i += 64) w[i] = weights[ray * L + i] + 1e-5f;
  } else {
    const int nc = B + 1;
    const float* zc = bins_or_z + ray * nc;
    for (int i = lane; i < B; i += 64) bins[i] = 0.5f * (zc[i + 1] + zc[i]);
    for (int i = lane; i < L; i += 64) w[i] = weights[ray * nc + 1 + i] + 1e-5f;
    for (int i = lane; i < nc; i += 64) sbuf[i] = zc[i];
  }
  wave_lds_sync();   // (this wave's own rows: DS instructions of a wave execute in order)
  build_cdf(w, cdf, B);
  wave_lds_sync();
  for (int q = lane; q < nf; q += 64) {
    const float uq = (u != nullptr) ? u[ray * nf + q]
                     : (rng.state != nullptr ? rng_uniform(rng, static_cast<uint64_t>(ray) * nf + q) : linspace_elem(0.0f, 1.0f, nf, q));
    int ind;
    const float s = invert_cdf(cdf, bins, B, uq, &ind);
    if (live) {
      if (samples != nullptr) samples[ray * nf + q] = s;
      if (inds != nullptr) inds[ray * nf + q] = ind;
    }
    if (MODE == 1) sbuf[B + 1 + q] = s;
  }
  if (MODE == 1) {
    const int nc = B + 1;
    const int total = nc + nf;
    // Everything below touches this wave's own LDS rows only: a wave's DS instructions execute in order, so no workgroup
    // barrier is needed between the steps (the round-1 kernel had one per bitonic stage: 36 for 192 depths).
    wave_lds_sync();
    // sort(cat(z_coarse, z_samples)) (train_utils.py:173).  Both halves are usually already ascending - the coarse depths
    // always, the samples whenever u is ascending (deterministic resampling: every validation render) - and then the sort is
    // a MERGE: an element's output slot = its own index + the number of elements of the other half in front of it (coarse
    // depths first on ties), found by binary search.  192 depths: ~7 LDS reads per element instead of 36 compare-exchange
    // stages.  Whether the samples really are ascending is checked on the values (an interpolated sample can land an ulp past
    // its bin edge); anything else takes the bitonic sort, which produces the same multiset in the same order.
    const float* zs = sbuf + nc;
    bool ordered = true;
    for (int q = lane; q < nf; q += 64)
      if (q + 1 < nf && zs[q] > zs[q + 1]) ordered = false;
    for (int i = lane; i < nc; i += 64)
      if (i + 1 < nc && sbuf[i] > sbuf[i + 1]) ordered = false;
    constexpr int kMergeCoarse = 4, kMergeFine = 8;   // merge path: up to 256 coarse + 512 fine depths (values held in registers)
    if (nc <= 64 * kMergeCoarse && nf <= 64 * kMergeFine && __all(ordered)) {
      float vc[kMergeCoarse], vf[kMergeFine];
      int sc[kMergeCoarse], sf[kMergeFine];
#pragma unroll
      for (int e = 0; e < kMergeCoarse; ++e) {        // coarse depth i: samples strictly in front of it
        if (64 * e >= nc) { vc[e] = 0.0f; sc[e] = 0; continue; }   // (wave-uniform: no search for rows that do not exist)
        const int i = lane + 64 * e;
        const float v = sbuf[min(i, nc - 1)];
        int lo = 0, hi = nf;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (zs[mid] < v) lo = mid + 1; else hi = mid; }
        vc[e] = v; sc[e] = i + lo;
      }
#pragma unroll
      for (int e = 0; e < kMergeFine; ++e) {          // sample q: coarse depths in front of it or equal to it
        if (64 * e >= nf) { vf[e] = 0.0f; sf[e] = 0; continue; }
        const int q = lane + 64 * e;
        const float v = zs[min(q, nf - 1)];
        int lo = 0, hi = nc;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (sbuf[mid] <= v) lo = mid + 1; else hi = mid; }
        vf[e] = v; sf[e] = q + lo;
      }
      wave_lds_sync();   // every search has read its operands (one wave, in order): now overwrite in place
#pragma unroll
      for (int e = 0; e < kMergeCoarse; ++e) if (lane + 64 * e < nc) sbuf[sc[e]] = vc[e];
#pragma unroll
      for (int e = 0; e < kMergeFine; ++e) if (lane + 64 * e < nf) sbuf[sf[e]] = vf[e];
      wave_lds_sync();
      if (live)
        for (int i = lane; i < total; i += 64) z_fine[ray * total + i] = sbuf[i];
      return;
    }
    for (int i = total + lane; i < sort_len; i += 64) sbuf[i] = __builtin_inff();
    // bitonic sort of sort_len (power of two) floats by one wave
    for (int k = 2; k <= sort_len; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        wave_lds_sync();
        for (int t = lane; t < (sort_len >> 1); t += 64) {
          const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          const int hi = lo | j;
          const bool up = ((lo & k) == 0);
          const float a = sbuf[lo], b = sbuf[hi];
          if ((a > b) == up) {
            sbuf[lo] = b;
            sbuf[hi] = a;
          }
        }
      }
    }
    wave_lds_sync();
    if (live)
      for (int i = lane; i < total; i += 64) z_fine[ray * total + i] = sbuf[i];
  }
}

static int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

}  // namespace dn

using namespace dn;

extern "C" int dn_ray_bundle(int height, int width, const float* h_rinv9, const float* h_origin3, float fx, float cx,
                             float cy, float* ro, float* rd, dn_stream_t stream) {
  DN_REQUIRE(height > 0 && width > 0 && h_rinv9 && h_origin3 && ro && rd, "dn_ray_bundle: bad arguments");
  RayBundleArgs a;
  for (int i = 0; i < 9; ++i) a.rinv[i] = h_rinv9[i];
  for (int i = 0; i < 3; ++i) a.origin[i] = h_origin3[i];
  a.fx = fx; a.cx = cx; a.cy = cy; a.height = height; a.width = width;
  const int64_t total = static_cast<int64_t>(height) * width;
  const int block = 256;
  const unsigned grid = static_cast<unsigned>((total + block - 1) / block);
  hipLaunchKernelGGL(ray_bundle_kernel, dim3(grid), dim3(block), 0, as_stream(stream), a, ro, rd);
  return check_launch("dn_ray_bundle");
}

extern "C" int dn_select_rays(int height, int width, const float* h_rinv9, const float* h_origin3, float fx, float cx,
                              float cy, float near, float far, const int64_t* pixel_index, int64_t n_rays,
                              const float* image, int channels, float* rays, float* target, dn_stream_t stream) {
  if (n_rays == 0) return 0;
  DN_REQUIRE(height > 0 && width > 0 && h_rinv9 && h_origin3 && pixel_index && rays && n_rays >= 0,
             "dn_select_rays: bad arguments");
  DN_REQUIRE(target == nullptr || (image != nullptr && channels >= 3), "dn_select_rays: target requested without an image of >= 3 channels");
  RayBundleArgs a;
  for (int i = 0; i < 9; ++i) a.rinv[i] = h_rinv9[i];
  for (int i = 0; i < 3; ++i) a.origin[i] = h_origin3[i];
  a.fx = fx; a.cx = cx; a.cy = cy; a.height = height; a.width = width;
  const int block = 256;
  const unsigned grid = static_cast<unsigned>((n_rays + block - 1) / block);
  hipLaunchKernelGGL(select_rays_kernel, dim3(grid), dim3(block), 0, as_stream(stream), a, near, far, pixel_index, n_rays,
                     image, channels, rays, target);
  return check_launch("dn_select_rays");
}

extern "C" int dn_select_rays_indirect(int height, int width, const float* cams, const int32_t* view, float near,
                                       float far, const int64_t* pixel_index, int64_t n_rays, const float* images,
                                       int channels, float* rays, float* target, dn_stream_t stream) {
  if (n_rays == 0) return 0;
  DN_REQUIRE(height > 0 && width > 0 && cams && view && pixel_index && rays && n_rays >= 0, "dn_select_rays_indirect: bad arguments");
  DN_REQUIRE(target == nullptr || (images != nullptr && channels >= 3), "dn_select_rays_indirect: target requested without images of >= 3 channels");
  const int block = 256;
  const unsigned grid = static_cast<unsigned>((n_rays + block - 1) / block);
  hipLaunchKernelGGL(select_rays_indirect_kernel, dim3(grid), dim3(block), 0, as_stream(stream), cams, view, 0, height, width, near,
                     far, pixel_index, n_rays, images, channels, rays, target, static_cast<uint32_t*>(nullptr), static_cast<int64_t*>(nullptr));
  return check_launch("dn_select_rays_indirect");
}

extern "C" int dn_select_rays_draw(int height, int width, const float* cams, const int32_t* view, int n_views, float near, float far,
                                   uint32_t* rng_state, int64_t n_rays, const float* images, int channels, float* rays, float* target,
                                   int64_t* pixel_index_out, dn_stream_t stream) {
  DN_REQUIRE(height > 0 && width > 0 && cams && (view || n_views >= 1) && rng_state && rays && n_rays >= 1, "dn_select_rays_draw: bad arguments");
  DN_REQUIRE(n_rays <= static_cast<int64_t>(height) * width, "dn_select_rays_draw: more rays than pixels (the draw is without replacement)");
  DN_REQUIRE(static_cast<int64_t>(height) * width < (1LL << 31), "dn_select_rays_draw: image too large");
  DN_REQUIRE(target == nullptr || (images != nullptr && channels >= 3), "dn_select_rays_draw: target requested without images of >= 3 channels");
  const int block = 256;
  const unsigned grid = static_cast<unsigned>((n_rays + block - 1) / block);
  hipLaunchKernelGGL(select_rays_indirect_kernel, dim3(grid), dim3(block), 0, as_stream(stream), cams, view, n_views, height, width, near,
                     far, static_cast<const int64_t*>(nullptr), n_rays, images, channels, rays, target, rng_state, pixel_index_out);
  return check_launch("dn_select_rays_draw");
}

// ---- S9 loss head on the device: mse(rgb_coarse, target) + mse(rgb_fine, target) (train_dexnerf_rgb.py:264-277; with
// `luminance` the IR head of train_nerf_ir.py:260-263: both sides through 0.299 r + 0.587 g + 0.114 b first), the upstream
// gradients of the two rgb maps written where dn_render_rays_backward reads them, and the RNG state's iteration counter advanced.
// One workgroup: the sums are formed in a fixed order (deterministic); n is a training batch (<= a few thousand rays).
namespace dn {
__global__ __launch_bounds__(1024) void mse2_loss_kernel(const float* __restrict__ rgb_c, const float* __restrict__ rgb_f,
                                                         const float* __restrict__ target, int64_t n, int luminance,
                                                         float* __restrict__ out3, float* __restrict__ g_c, float* __restrict__ g_f,
                                                         uint32_t* __restrict__ rng_state) {
  __shared__ float part[2][16];
  float sc = 0.0f, sf = 0.0f;
  if (luminance) {
    const float inv = 2.0f / static_cast<float>(n);
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
      const float lt = (0.299f * target[i * 3] + 0.587f * target[i * 3 + 1]) + 0.114f * target[i * 3 + 2];
      const float* src[2] = {rgb_c, rgb_f};
      float* dst[2] = {g_c, g_f};
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (src[k] == nullptr) continue;
        const float d = ((0.299f * src[k][i * 3] + 0.587f * src[k][i * 3 + 1]) + 0.114f * src[k][i * 3 + 2]) - lt;
        (k ? sf : sc) += d * d;
        if (dst[k] != nullptr) { dst[k][i * 3] = inv * d * 0.299f; dst[k][i * 3 + 1] = inv * d * 0.587f; dst[k][i * 3 + 2] = inv * d * 0.114f; }
      }
    }
  } else {
    const float inv = 2.0f / static_cast<float>(3 * n);
    for (int64_t e = threadIdx.x; e < 3 * n; e += blockDim.x) {
      const float t = target[e];
      const float dc = rgb_c[e] - t;
      sc += dc * dc;
      if (g_c != nullptr) g_c[e] = inv * dc;
      if (rgb_f != nullptr) {
        const float df = rgb_f[e] - t;
        sf += df * df;
        if (g_f != nullptr) g_f[e] = inv * df;
      }
    }
  }
  sc = wave_sum(sc); sf = wave_sum(sf);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { part[0][wave] = sc; part[1][wave] = sf; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.0f, b = 0.0f;
    for (int w = 0; w < static_cast<int>(blockDim.x >> 6); ++w) { a += part[0][w]; b += part[1][w]; }
    const float denom = static_cast<float>(luminance ? n : 3 * n);
    out3[1] = a / denom; out3[2] = b / denom; out3[0] = a / denom + b / denom;
    if (rng_state != nullptr) rng_state[3] = rng_state[2] + 1u;   // the next iteration's counter (dn_rng.h)
  }
}
}  // namespace dn

namespace dn {
__global__ void rng_fill_kernel(const uint32_t* __restrict__ state, uint32_t stream, int64_t n, int normal, float* __restrict__ out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const RngRef r{state, stream};
  out[i] = normal ? rng_normal(r, static_cast<uint64_t>(i)) : rng_uniform(r, static_cast<uint64_t>(i));
}
}  // namespace dn

extern "C" int dn_rng_fill(const uint32_t* rng_state, uint32_t stream_id, int64_t n, int normal, float* out, dn_stream_t stream) {
  if (n == 0) return 0;
  DN_REQUIRE(rng_state && out && n >= 0, "dn_rng_fill: bad arguments");
  hipLaunchKernelGGL(rng_fill_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, as_stream(stream), rng_state, stream_id, n, normal, out);
  return check_launch("dn_rng_fill");
}

extern "C" int dn_mse2_loss(const float* rgb_coarse, const float* rgb_fine, const float* target, int64_t n_rays, int luminance,
                            float* loss3, float* g_rgb_coarse, float* g_rgb_fine, uint32_t* rng_state, dn_stream_t stream) {
  DN_REQUIRE(rgb_coarse && target && loss3 && n_rays >= 1, "dn_mse2_loss: bad arguments");
  hipLaunchKernelGGL(mse2_loss_kernel, dim3(1), dim3(1024), 0, as_stream(stream), rgb_coarse, rgb_fine, target, n_rays, luminance, loss3,
                     g_rgb_coarse, g_rgb_fine, rng_state);
  return check_launch("dn_mse2_loss");
}

// ---- S2 ray packing of run_one_iter_of_nerf (nerf/train_utils.py:220-250): (N,3) origins / directions -> the (N, 8 | 11) rows
// [o, d, near, far, d_view / |d_view|] predict_and_render_radiance reads - the reference forms them with a norm, a division, two
// ones_like, two multiplies and a cat (eight launches per image); op for op as those run on the device.
namespace dn {
__global__ void pack_ray_rows_kernel(const float* __restrict__ ro, const float* __restrict__ rd, const float* __restrict__ rd_view,
                                     float near, float far, int64_t n, float* __restrict__ rows) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float* r = rows + i * (rd_view != nullptr ? 11 : 8);
#pragma unroll
  for (int j = 0; j < 3; ++j) { r[j] = ro[i * 3 + j]; r[3 + j] = rd[i * 3 + j]; }
  r[6] = near;
  r[7] = far;
  if (rd_view != nullptr) {
    const float x = rd_view[i * 3], y = rd_view[i * 3 + 1], z = rd_view[i * 3 + 2];
    // the order of torch's device reduction over three elements ((x x + z z) + y y: measured, scripts/rows_diag.py) - this kernel
    // replaces torch ops that ran on the device, and the rows stay bit-identical to them
    const float nrm = sqrtf((x * x + z * z) + y * y);
    r[8] = x / nrm; r[9] = y / nrm; r[10] = z / nrm;
  }
}
}  // namespace dn

extern "C" int dn_pack_ray_rows(const float* rays_o, const float* rays_d, const float* view_d, float near, float far, int64_t n_rays,
                                float* rows, dn_stream_t stream) {
  if (n_rays == 0) return 0;
  DN_REQUIRE(rays_o && rays_d && rows && n_rays >= 0, "dn_pack_ray_rows: bad arguments");
  const int block = 256;
  const unsigned grid = static_cast<unsigned>((n_rays + block - 1) / block);
  hipLaunchKernelGGL(pack_ray_rows_kernel, dim3(grid), dim3(block), 0, as_stream(stream), rays_o, rays_d, view_d, near, far, n_rays, rows);
  return check_launch("dn_pack_ray_rows");
}

extern "C" int dn_ndc_rays(int height, int width, double focal, double near, const float* rays_o, const float* rays_d,
                           int64_t n_rays, float* rays_o_out, float* rays_d_out, dn_stream_t stream) {
  if (n_rays == 0) return 0;
  DN_REQUIRE(rays_o && rays_d && rays_o_out && rays_d_out && n_rays >= 0 && height > 0 && width > 0, "dn_ndc_rays: bad arguments");
  const int block = 256;
  const unsigned grid = static_cast<unsigned>((n_rays + block - 1) / block);
  hipLaunchKernelGGL(ndc_rays_kernel, dim3(grid), dim3(block), 0, as_stream(stream), static_cast<double>(height),
                     static_cast<double>(width), focal, near, rays_o, rays_d, n_rays, rays_o_out, rays_d_out);
  return check_launch("dn_ndc_rays");
}

extern "C" int dn_coarse_depths(const float* rays, int ray_stride, int64_t n_rays, int num_coarse, int lindisp,
                                const float* t_rand, float* z_vals, dn_stream_t stream) {
  return dn::coarse_depths_rng(rays, ray_stride, n_rays, num_coarse, lindisp, t_rand, z_vals, nullptr, stream);
}

// dn_coarse_depths; t_rand == NULL with an RNG state: the stratified jitter is drawn in the kernel (dn_rng.h)
int dn::coarse_depths_rng(const float* rays, int ray_stride, int64_t n_rays, int num_coarse, int lindisp,
                                const float* t_rand, float* z_vals, const uint32_t* rng_state, dn_stream_t stream) {
  if (n_rays == 0) return 0;  // empty tensors carry NULL data pointers
  DN_REQUIRE(rays && z_vals && n_rays >= 0 && num_coarse >= 1 && ray_stride >= 8, "dn_coarse_depths: bad arguments");
  const int64_t total = n_rays * num_coarse;
  const int block = 256;
  const unsigned grid = static_cast<unsigned>((total + block - 1) / block);
  hipLaunchKernelGGL(coarse_depths_kernel, dim3(grid), dim3(block), 0, as_stream(stream), rays, ray_stride, n_rays,
                     num_coarse, lindisp, t_rand, z_vals, RngRef{rng_state, kRngStreamJitter});
  return check_launch("dn_coarse_depths");
}

namespace dn {
// frequency bands as the reference builds them (nerf_helpers.py:134-149): 2**linspace(0, L-1, L) or
// linspace(1, 2**(L-1), L), fp32.
void fill_freqs(float* f, int num_fns, int log_sampling) {
  for (int k = 0; k < num_fns; ++k) {
    if (log_sampling) {
      f[k] = exp2f(linspace_elem(0.0f, static_cast<float>(num_fns - 1), num_fns, k));
    } else {
      f[k] = linspace_elem(1.0f, exp2f(static_cast<float>(num_fns - 1)), num_fns, k);
    }
  }
}
}  // namespace dn

extern "C" int dn_positional_encoding(const float* x, int64_t n_points, int dim, int num_fns, int include_input,
                                      int log_sampling, float* out, dn_stream_t stream) {
  if (n_points == 0) return 0;
  DN_REQUIRE(x && out && n_points >= 0 && dim >= 1 && num_fns >= 0 && num_fns <= 32,
             "dn_positional_encoding: bad arguments (num_fns must be in [0,32])");
  DN_REQUIRE(include_input || num_fns > 0, "dn_positional_encoding: empty encoding");
  if (n_points == 0) return 0;
  FreqArgs fr;
  fill_freqs(fr.f, num_fns, log_sampling);
  const int64_t total = n_points * dim;
  const int block = 256;
  const unsigned grid = static_cast<unsigned>((total + block - 1) / block);
  hipLaunchKernelGGL(posenc_kernel, dim3(grid), dim3(block), 0, as_stream(stream), x, total, dim, num_fns,
                     include_input, fr, out);
  return check_launch("dn_positional_encoding");
}

extern "C" int dn_sample_pdf(const float* bins, const float* weights, const float* u, int64_t n_rays, int n_bins,
                             int n_samples, float* samples, int64_t* inds, dn_stream_t stream) {
  if (n_rays == 0) return 0;
  DN_REQUIRE(bins && weights && samples && n_rays >= 0 && n_samples >= 1, "dn_sample_pdf: bad arguments");
  DN_REQUIRE(n_bins >= 9 && n_bins <= 512, "dn_sample_pdf: n_bins must be in [9, 512] (weights row >= 8 wide)");
  if (n_rays == 0) return 0;
  const size_t lds = static_cast<size_t>(kSamplerWaves) * 3 * (n_bins + 1) * sizeof(float);
  const unsigned grid = static_cast<unsigned>((n_rays + kSamplerWaves - 1) / kSamplerWaves);
  hipLaunchKernelGGL(sampler_kernel<0>, dim3(grid), dim3(256), lds, as_stream(stream), bins, weights, u, n_rays,
                     n_bins, n_samples, samples, inds, static_cast<float*>(nullptr), 0, RngRef{nullptr, 0u});
  return check_launch("dn_sample_pdf");
}

extern "C" int dn_fine_depths(const float* z_coarse, const float* weights, const float* u, int64_t n_rays,
                              int num_coarse, int num_fine, float* z_fine, float* z_samples, dn_stream_t stream) {
  return dn::fine_depths_rng(z_coarse, weights, u, n_rays, num_coarse, num_fine, z_fine, z_samples, nullptr, stream);
}

// dn_fine_depths; u == NULL with an RNG state: the resampling draws are made in the kernel (dn_rng.h), NULL without: deterministic
int dn::fine_depths_rng(const float* z_coarse, const float* weights, const float* u, int64_t n_rays,
                              int num_coarse, int num_fine, float* z_fine, float* z_samples, const uint32_t* rng_state, dn_stream_t stream) {
  if (n_rays == 0) return 0;
  DN_REQUIRE(z_coarse && weights && z_fine && n_rays >= 0 && num_fine >= 1, "dn_fine_depths: bad arguments");
  DN_REQUIRE(num_coarse >= 10 && num_coarse <= 512 && num_coarse + num_fine <= 2048,
             "dn_fine_depths: need 10 <= num_coarse <= 512 and num_coarse + num_fine <= 2048");
  if (n_rays == 0) return 0;
  const int B = num_coarse - 1;
  const int sort_len = next_pow2(num_coarse + num_fine);
  const size_t lds = static_cast<size_t>(kSamplerWaves) * (3 * (B + 1) + sort_len) * sizeof(float);
  const unsigned grid = static_cast<unsigned>((n_rays + kSamplerWaves - 1) / kSamplerWaves);
  hipLaunchKernelGGL(sampler_kernel<1>, dim3(grid), dim3(256), lds, as_stream(stream), z_coarse, weights, u, n_rays, B,
                     num_fine, z_samples, static_cast<int64_t*>(nullptr), z_fine, sort_len, RngRef{rng_state, kRngStreamU});
  return check_launch("dn_fine_depths");
}
