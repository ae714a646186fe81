#!/bin/bash
# Developer tool (GPU box): alternate builds on ONE box - the as-shipped 4 x 128 nets' fine launch of config 3 (scripts/quick_time128.py)
#   scripts/ab_time128.sh ROUNDS tag...   tag = base | <name of exp_libs/lib<name>.so>      AB_PREC=fp16|bf16
R=$1; shift
for r in $(seq $R); do
  for tag in "$@"; do
    unset DEXNERF_HIP_LIB
    if [ "$tag" != base ]; then export DEXNERF_HIP_LIB=exp_libs/lib$tag.so; fi
    echo "$tag: $(python3 scripts/quick_time128.py ${AB_PREC:-fp16} 129600 128 2>/dev/null | tail -1)"
  done
done
