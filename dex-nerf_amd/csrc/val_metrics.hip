// Device-side validation extras: the per-threshold Dex depth-error sweep
// (reference train_dexnerf_rgb.py:391-408 calls compute_err_metric, nerf/train_utils.py:9-30, once per candidate
// on the CPU - K device->host copies per validation) and the colour-coded depth-error image
// (nerf/train_utils.py:31-70).  HBM-bound: (K + 1) x N x 4 B read by the sweep, 3 x N x 4 B written by the image.
#include "dn_common.h"

namespace dn {

// out[k] = { sum |pred*1000 - gt*1000|, #(|gt - pred| > 2e-3), #(> 4e-3), #(> 8e-3), #masked }  (doubles)
__global__ __launch_bounds__(256) void dex_error_sweep_kernel(const float* __restrict__ gt, const float* __restrict__ pred,
                                                              int64_t n, const uint8_t* __restrict__ mask, float lo,
                                                              float hi, double* __restrict__ out) {
  const int k = blockIdx.y;
  const float* p = pred + static_cast<int64_t>(k) * n;
  double s = 0.0;
  unsigned c2 = 0, c4 = 0, c8 = 0, cn = 0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const float g = gt[i];
    const bool on = mask ? (mask[i] != 0) : (g > lo && g < hi);
    if (!on) continue;
    const float e = p[i];
    s += static_cast<double>(fabsf(e * 1000.0f - g * 1000.0f));  // F.l1_loss(pred*1000, gt*1000), fp32 per element
    const float d = fabsf(g - e);
    c2 += d > 2e-3f; c4 += d > 4e-3f; c8 += d > 8e-3f; ++cn;
  }
  s = wave_sum(s);
  double v2 = wave_sum(static_cast<double>(c2)), v4 = wave_sum(static_cast<double>(c4));
  double v8 = wave_sum(static_cast<double>(c8)), vn = wave_sum(static_cast<double>(cn));
  if (lane_id() == 0) {
    double* o = out + k * 5;
    atomicAdd(o + 0, s); atomicAdd(o + 1, v2); atomicAdd(o + 2, v4); atomicAdd(o + 3, v8); atomicAdd(o + 4, vn);
  }
}

struct ColorMap { float lo[11], hi[11], rgb[11][3]; };

__global__ void depth_error_image_kernel(const float* __restrict__ est, const float* __restrict__ gt,
                                         const uint8_t* __restrict__ mask, int height, int width, float abs_thres,
                                         ColorMap cm, float* __restrict__ out) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t total = static_cast<int64_t>(height) * width;
  if (i >= total) return;
  const int row = static_cast<int>(i / width), col = static_cast<int>(i - static_cast<int64_t>(row) * width);
  const bool on = mask[i] != 0;
  float r = 0.f, g = 0.f, b = 0.f;
  if (on) {
    const float err = fabsf(gt[i] - est[i]) / abs_thres;
#pragma unroll
    for (int c = 0; c < 11; ++c) {  // later rows overwrite earlier ones, like the reference's loop
      if (err >= cm.lo[c] && err < cm.hi[c]) { r = cm.rgb[c][0]; g = cm.rgb[c][1]; b = cm.rgb[c][2]; }
    }
  }
  // colour legend: 20-pixel-wide swatches along the top 10 rows (drawn over masked-out pixels too)
  if (row < 10 && col < 11 * 20) { const int c = col / 20; r = cm.rgb[c][0]; g = cm.rgb[c][1]; b = cm.rgb[c][2]; }
  out[i * 3 + 0] = r; out[i * 3 + 1] = g; out[i * 3 + 2] = b;
}

}  // namespace dn

using namespace dn;

extern "C" int dn_dex_error_sweep(const float* depth_gt, const float* depth_pred, int n_candidates, int64_t n_pixels,
                                  const uint8_t* mask, float gt_lo, float gt_hi, double* out, dn_stream_t stream) {
  DN_REQUIRE(n_candidates >= 0 && n_pixels >= 0 && (n_candidates == 0 || out), "dn_dex_error_sweep: bad arguments");
  if (n_candidates == 0) return 0;
  hipError_t e = hipMemsetAsync(out, 0, sizeof(double) * 5 * n_candidates, as_stream(stream));
  if (e != hipSuccess) { set_error("dn_dex_error_sweep: %s", hipGetErrorString(e)); return -static_cast<int>(e); }
  if (n_pixels == 0) return 0;
  DN_REQUIRE(depth_gt && depth_pred, "dn_dex_error_sweep: NULL depth maps");
  int64_t blocks = (n_pixels + 256 * 8 - 1) / (256 * 8);
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(dex_error_sweep_kernel, dim3(static_cast<unsigned>(blocks), n_candidates), dim3(256), 0, as_stream(stream),
                     depth_gt, depth_pred, n_pixels, mask, gt_lo, gt_hi, out);
  return check_launch("dn_dex_error_sweep");
}

extern "C" int dn_depth_error_image(const float* depth_est, const float* depth_gt, const uint8_t* mask, int height,
                                    int width, float abs_thres, float* out_rgb, dn_stream_t stream) {
  DN_REQUIRE(height >= 0 && width >= 0, "dn_depth_error_image: bad size");
  if (height == 0 || width == 0) return 0;
  DN_REQUIRE(depth_est && depth_gt && mask && out_rgb, "dn_depth_error_image: NULL pointer");
  // gen_error_colormap_depth (reference nerf/train_utils.py:31-45): [lo, hi) -> RGB/255, fp32
  static const float edges[12] = {0.f, 0.00001f, 2000.f / 1024, 2000.f / 512, 2000.f / 256, 2000.f / 128, 2000.f / 64,
                                  2000.f / 32, 2000.f / 16, 2000.f / 8, 2000.f / 4, INFINITY};
  static const float rgb[11][3] = {{0, 0, 0}, {49, 54, 149}, {69, 117, 180}, {116, 173, 209}, {171, 217, 233}, {224, 243, 248},
                                   {254, 224, 144}, {253, 174, 97}, {244, 109, 67}, {215, 48, 39}, {165, 0, 38}};
  ColorMap cm;
  for (int c = 0; c < 11; ++c) {
    cm.lo[c] = edges[c]; cm.hi[c] = edges[c + 1];
    for (int j = 0; j < 3; ++j) cm.rgb[c][j] = rgb[c][j] / 255.f;
  }
  const int64_t total = static_cast<int64_t>(height) * width;
  hipLaunchKernelGGL(depth_error_image_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                     depth_est, depth_gt, mask, height, width, abs_thres, cm, out_rgb);
  return check_launch("dn_depth_error_image");
}
