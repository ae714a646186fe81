#!/bin/bash
# Developer tool (GPU box): cycles / clock / matrix-pipe utilisation of the two geometries of mfma_geometry.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/geo_pmc -- exp_libs/mfma_geometry 400 > gpurun_out/geo_pmc.log 2>&1 || { tail -5 gpurun_out/geo_pmc.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/geo_pmc/*/*counter_collection.csv")[0]
k = glob.glob("gpurun_out/geo_pmc/*/*kernel_trace.csv")[0]
dur = {}
for r in csv.DictReader(open(k)):
    dur[r["Dispatch_Id"]] = ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r["Kernel_Name"])
ctr = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    ctr[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
for d in sorted(ctr, key=int):
    ms, name = dur[d]
    c = ctr[d]; cyc = c["GRBM_GUI_ACTIVE"] / 8
    if "geo_kernel" not in name: continue
    mode = "mode1 8x48 16x16x32" if "geo_kernel<1>" in name else "mode0 8x32 32x32x16"
    print("%s  ms=%.2f cycles/XCD=%.4g clock=%.3f GHz mfma_busy/cycle=%.3f lds_active/CUcycle=%.3f" % (
        mode, ms, cyc, cyc / ms / 1e6, c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), c["SQ_LDS_IDX_ACTIVE"] / (cyc * 256)))
PY
