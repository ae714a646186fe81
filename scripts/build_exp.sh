#!/bin/bash
# Developer helper: build an experiment copy of the library with extra -D flags.
#   scripts/build_exp.sh NAME "-DDN_EXP_FOO=1" [file.hip ...]   (default: both MLP translation units)
# Output: exp_libs/libNAME.so (git-ignored; select it with DEXNERF_HIP_LIB=exp_libs/libNAME.so)
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
C=$REPO/dex-nerf_amd/csrc
NAME=$1; FLAGS=$2; shift 2 || true
FILES=${@:-mlp_fused.hip mlp_train.hip}
# the 48-point forward kernel is one header instantiated by four translation units: naming the first rebuilds all four
if echo " $FILES " | grep -q " mlp_fused48.hip "; then FILES="$FILES mlp_fused48_paper_bf16.hip mlp_fused48_paper_fp16.hip mlp_fused48_w128.hip"; fi
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -DDN_ABLATION_BUILD"
make -C $C -j8 >/dev/null
mkdir -p $REPO/exp_libs $C/build/exp_$NAME
OBJS=""
for s in $(sed -n "s/^SRCS := //p" $C/Makefile); do
  if echo " $FILES " | grep -q " $s "; then
    /opt/rocm/bin/hipcc $F $FLAGS -I$C -x hip -c $C/$s -o $C/build/exp_$NAME/$s.o &
    OBJS="$OBJS $C/build/exp_$NAME/$s.o"
  else
    OBJS="$OBJS $C/build/$s.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $REPO/exp_libs/lib$NAME.so $OBJS
echo "built exp_libs/lib$NAME.so"
