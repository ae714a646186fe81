for r in 1 2; do
for tag in base tf_nomask tf_nounit tf_noepi tf_none; do
  unset DEXNERF_HIP_LIB
  if [ "$tag" != base ]; then export DEXNERF_HIP_LIB=exp_libs/lib$tag.so; fi
  echo "$tag: $(timeout -k 10 120 python3 scripts/train_kernels_time.py 2>/dev/null | head -1 | cut -c1-90)"
done
done
