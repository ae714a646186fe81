#!/bin/bash
# Developer tool (run on the GPU box): cycles vs wall time of the fine-net launch for ablation builds of the library.
#   scripts/pmc_ablate.sh base geom32 nodma noread   (exp_libs/lib<NAME>.so; "base" = the shipped library, geom32 / pt2 = its other geometries)
# Answers whether an ablation saves GPU cycles (pipeline effect) or only wall time (clock / power effect).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for tag in "$@"; do
  unset DEXNERF_HIP_LIB DEXNERF_BF16_PT DEXNERF_BF16_GEOM
  if [ "$tag" = pt2 ]; then export DEXNERF_BF16_PT=2   # shipped library, 4 waves x 64 points
  elif [ "$tag" = geom32 ]; then export DEXNERF_BF16_GEOM=32   # shipped library, the 32-point kernel
  elif [ "$tag" != base ]; then export DEXNERF_HIP_LIB=exp_libs/lib$tag.so; fi
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv \
      -d gpurun_out/abl_$tag -- python3 scripts/quick_time.py bf16 160000 > gpurun_out/abl_$tag.log 2>&1 || { echo "$tag: profiler run failed"; tail -3 gpurun_out/abl_$tag.log; exit 1; }
  echo "done $tag"
done
python3 - "$@" <<'PY'
import csv, glob, sys, collections
for tag in sys.argv[1:]:
    f = glob.glob("gpurun_out/abl_%s/*/*counter_collection.csv" % tag)[0]
    agg = collections.OrderedDict(); n = 0
    for r in csv.DictReader(open(f)):
        if "mlp_forward" in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    k = glob.glob("gpurun_out/abl_%s/*/*kernel_trace.csv" % tag)[0]
    dur = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(k)) if "mlp_forward" in r["Kernel_Name"]]
    last = {c: v[-1] for c, v in agg.items()}
    cyc = last.get("GRBM_GUI_ACTIVE", 0) / 8
    print("%-8s ms=%.2f  cycles/XCD=%.4g  clock=%.3f GHz  mfma_util=%.3f  lds_idx_active/SIMDcyc=%.3f  wait_inst_lds=%.4g" % (
        tag, dur[-1], cyc, cyc / dur[-1] / 1e6, last.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024) if cyc else 0,
        last.get("SQ_LDS_IDX_ACTIVE", 0) / (cyc * 256) if cyc else 0, last.get("SQ_WAIT_INST_LDS", 0)))
PY
