// The explicit-schedule fp16 instances of the paper network (D8 / W256 / skip 4, view directions) of the 48-point forward kernel
// (mlp_fused48_kernel.h); launched from mlp_fused48.hip launch_forward48.
#include "mlp_fused48_kernel.h"

namespace dn {

template __global__ void mlp_forward48_kernel<256, 2, 8, 0x10u, 1, 0, 2>(FwdParams, G48Params);
template __global__ void mlp_forward48_kernel<256, 2, 8, 0x10u, 1>(FwdParams, G48Params);

}  // namespace dn
