#!/bin/bash
# Developer tool (GPU box): timing-only ablations of the training forward's epilogue on ONE box, alternated twice
# (profiles/r04_headline_schedule.md section 8).  Build the variants first:
#   scripts/build_exp.sh tf_nomask "-DDN_EXP_TF_NOMASK=1" mlp_fused48.hip ; scripts/build_exp.sh tf_nounit "-DDN_EXP_TF_NOUNIT=1" mlp_fused48.hip
#   scripts/build_exp.sh tf_noepi "-DDN_EXP_NOEPI=1" mlp_fused48.hip ; scripts/build_exp.sh tf_none "-DDN_EXP_NOEPI=1 -DDN_EXP_TF_NOUNIT=1 -DDN_EXP_TF_NOMASK=1" mlp_fused48.hip
for r in 1 2; do
for tag in base tf_nomask tf_nounit tf_noepi tf_none; do
  unset DEXNERF_HIP_LIB
  if [ "$tag" != base ]; then export DEXNERF_HIP_LIB=exp_libs/lib$tag.so; fi
  echo "$tag: $(timeout -k 10 120 python3 scripts/train_kernels_time.py 2>/dev/null | head -1 | cut -c1-90)"
done
done
