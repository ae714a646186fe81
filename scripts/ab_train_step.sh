#!/bin/bash
# Developer tool (GPU box): alternate two builds of the library on ONE box, several rounds - the training step in the 8-bit mode.
#   scripts/ab_train_step.sh ROUNDS tag...   tag = base | <name of exp_libs/lib<name>.so>
R=$1; shift
for r in $(seq $R); do
  for tag in "$@"; do
    unset DEXNERF_HIP_LIB
    if [ "$tag" != base ]; then export DEXNERF_HIP_LIB=exp_libs/lib$tag.so; fi
    echo "$tag: $(python3 scripts/train_step_profile.py bf16-s8 30 2>/dev/null | tail -1)"
  done
done
