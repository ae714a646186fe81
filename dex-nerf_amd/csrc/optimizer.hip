// The optimizer step of the training loop on flat buffers (reference: torch.optim.Adam with its default betas / eps, built at
// train_dexnerf_rgb.py:146-148 and stepped at :280; the learning-rate schedule of :284-289).
//
// The reference steps 48 parameter tensors (two D8/W256 nets) through torch's multi-tensor Adam; here the parameters, their
// gradients (nerf.parallel.FlatGradBucket) and both moment estimates are each ONE contiguous fp32 buffer, and a step is one
// elementwise pass: 16 B/parameter read, 12 B written (+ 4 when the gradients are cleared for the next iteration in the same pass).
#include "dn_common.h"

namespace dn {

// state (16-byte aligned, 40 bytes): float { step count (as torch keeps it), ticket (uint32 bits), learning rate of the last step,
// - } then double { beta1^t, beta2^t, lr_decay_per_step^t } for the t steps taken so far - running products, so that no thread
// evaluates pow() (one per workgroup on the critical path was most of this kernel on the small nets).  Every workgroup reads the
// state before it takes its ticket; the workgroup that takes the last one publishes the next state and resets the ticket - no
// thread reads a word another workgroup may already have rewritten, and a replayed HIP graph advances by itself.
__global__ __launch_bounds__(256) void adam_kernel(float4* __restrict__ param, float4* __restrict__ grad, float4* __restrict__ m,
                                                   float4* __restrict__ v, int64_t n4, float* __restrict__ state,
                                                   const float* __restrict__ lr_ptr, double lr0, double decay_per_step,
                                                   double beta1, double beta2, double eps, int zero_grads) {
  __shared__ double sh[2];
  double* prod = reinterpret_cast<double*>(state + 4);
  const float count = state[0];
  const double b1p = prod[0] * beta1, b2p = prod[1] * beta2, decay = prod[2];   // beta^(t + 1); the schedule's factor for step t
  const double lr = lr_ptr ? static_cast<double>(*lr_ptr) : lr0 * decay;
  if (threadIdx.x == 0) {
    // torch (_fused_adam, capturable): bias corrections and step size in double
    sh[0] = lr / (1.0 - b1p);
    sh[1] = 1.0 / sqrt(1.0 - b2p);
  }
  __syncthreads();
  // the hyper-parameters stay doubles, as in torch's kernel: every product with one of them is formed in double and rounded to
  // fp32 once, when the element is stored (an fp32-only update is ~4x further from float64 Adam after 25 steps - measured, tests)
  const double step_size = sh[0], inv_bc2_sqrt = sh[1];
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const float4 g = grad[i];
    float4 p = param[i], mm = m[i], vv = v[i];
    auto upd = [&](float& pe, float& me, float& ve, float ge) {
      const double gd = static_cast<double>(ge);
      me = static_cast<float>(static_cast<double>(me) + (gd - static_cast<double>(me)) * (1.0 - beta1));   // lerp(exp_avg, grad, 1 - beta1)
      ve = static_cast<float>(static_cast<double>(ve) * beta2 + (1.0 - beta2) * gd * gd);
      const double denom = static_cast<double>(sqrtf(ve)) * inv_bc2_sqrt + eps;
      pe = static_cast<float>(static_cast<double>(pe) - step_size * static_cast<double>(me) / denom);
    };
    upd(p.x, mm.x, vv.x, g.x); upd(p.y, mm.y, vv.y, g.y); upd(p.z, mm.z, vv.z, g.z); upd(p.w, mm.w, vv.w, g.w);
    param[i] = p; m[i] = mm; v[i] = vv;
    if (zero_grads) grad[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  // acquire-release on the ticket: this workgroup's READS of the state (consumed above) are ordered before its ticket, and the
  // last workgroup's rewrite of the state after every other workgroup's ticket - by the memory model, not only by data dependence
  // (measured: the fence is noise next to the kernel's ~10 us)
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* ticket = reinterpret_cast<unsigned*>(state + 1);
    if (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
      state[0] = count + 1.0f;
      state[2] = static_cast<float>(lr);          // the learning rate this step used (for logs)
      prod[0] = b1p; prod[1] = b2p; prod[2] = decay * decay_per_step;
      *ticket = 0u;
    }
  }
}

}  // namespace dn

extern "C" int dn_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float* state, const float* lr,
                            double lr0, double lr_decay_per_step, double beta1, double beta2, double eps, int zero_grads,
                            dn_stream_t stream) {
  using namespace dn;
  if (n == 0) return 0;
  DN_REQUIRE(params && grads && exp_avg && exp_avg_sq && state && n > 0, "dn_adam_step: bad arguments");
  DN_REQUIRE(n % 4 == 0 && ((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(exp_avg) |
                             reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15u) == 0,
             "dn_adam_step: the flat buffers are 16-byte aligned and padded to a multiple of four elements");
  DN_REQUIRE(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0 && (lr || lr0 >= 0.0) && lr_decay_per_step > 0.0,
             "dn_adam_step: bad hyper-parameters");
  DN_REQUIRE((reinterpret_cast<uintptr_t>(state) & 15u) == 0, "dn_adam_step: the state record is 16-byte aligned");
  const int64_t n4 = n / 4;
  int64_t grid = (n4 + 255) / 256;
  // (one workgroup per CU: every workgroup ends with an atomic on the SAME ticket word, and same-address atomics serialise at ~10 ns
  // each - with 8 workgroups per CU the 1,164 tickets of two D8/W256 networks were 12 of the kernel's 18 us)
  const int64_t cap = static_cast<int64_t>(device_cus());
  if (grid > cap) grid = cap;
  hipLaunchKernelGGL(adam_kernel, dim3(static_cast<unsigned>(grid)), dim3(256), 0, as_stream(stream), reinterpret_cast<float4*>(params),
                     reinterpret_cast<float4*>(grads), reinterpret_cast<float4*>(exp_avg), reinterpret_cast<float4*>(exp_avg_sq), n4, state,
                     lr, lr0, lr_decay_per_step, beta1, beta2, eps, zero_grads);
  return check_launch("dn_adam_step");
}
