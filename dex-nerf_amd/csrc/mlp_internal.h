// Host-side helpers shared between mlp_fused.hip (inference + packing) and mlp_train.hip (training kernels).
#pragma once
#include "mlp_device.h"

namespace dn {

struct PackPtrs {
  const float* w[kMaxStages];
  const float* b[kMaxStages];
};

int setup_params(const dn_mlp_desc* desc, int precision, const void* packed, FwdParams* p);
struct CompParams;
int dispatch_forward(const dn_mlp_desc& d, int precision, FwdParams& p, hipStream_t stream, const CompParams* comp = nullptr, int* composited = nullptr);
int launch_pack(const NetLayout& L, const PackPtrs& ptrs, void* packed, int precision, hipStream_t stream);

}  // namespace dn
