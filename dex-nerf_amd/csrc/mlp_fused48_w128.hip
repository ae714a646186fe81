// The W = 128 instances of the 48-point forward kernel (mlp_fused48_kernel.h); launched from mlp_fused48.hip launch_forward48.
#include "mlp_fused48_kernel.h"

namespace dn {

template __global__ void mlp_forward48_kernel<128, 1>(FwdParams, G48Params);
template __global__ void mlp_forward48_kernel<128, 2>(FwdParams, G48Params);
template __global__ void mlp_forward48_kernel<128, 1, 4, 0u, 1>(FwdParams, G48Params);
template __global__ void mlp_forward48_kernel<128, 2, 4, 0u, 1>(FwdParams, G48Params);
template __global__ void mlp_forward48_kernel<128, 1, 4, 0u, 1, 0, 1>(FwdParams, G48Params);
template __global__ void mlp_forward48_kernel<128, 2, 4, 0u, 1, 0, 1>(FwdParams, G48Params);
template __global__ void mlp_forward48_kernel<128, 1, 4, 0u, 1, 0, 0, 1>(FwdParams, G48Params);
template __global__ void mlp_forward48_kernel<128, 2, 4, 0u, 1, 0, 0, 1>(FwdParams, G48Params);
template __global__ void mlp_forward48_kernel<128, 1, 4, 0u, 1, 2>(FwdParams, G48Params);
template __global__ void mlp_forward48_kernel<128, 1, 4, 0u, 1, 3>(FwdParams, G48Params);
template __global__ void mlp_forward48_kernel<128, 1, 0, 0u, 0, 2>(FwdParams, G48Params);

}  // namespace dn
