#!/bin/bash
# Developer tool (GPU box): alternate builds / geometries of the bf16 fine-net launch on ONE box, several rounds.
#   scripts/ab_time.sh ROUNDS tag...   tag = base | geom32 | <name of exp_libs/lib<name>.so>
R=$1; shift
for r in $(seq $R); do
  for tag in "$@"; do
    unset DEXNERF_HIP_LIB DEXNERF_BF16_GEOM
    if [ "$tag" = geom32 ]; then export DEXNERF_BF16_GEOM=32; elif [ "$tag" != base ]; then export DEXNERF_HIP_LIB=exp_libs/lib$tag.so; fi
    echo "$tag: $(python3 scripts/quick_time.py ${AB_PREC:-bf16} 160000 2>/dev/null | tail -1)"
  done
done
