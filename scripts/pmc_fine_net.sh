#!/bin/bash
# Developer tool (GPU box): HBM traffic and L2 hit rate of the bf16 fine-net launch (separate --pmc passes, as the microarch
# guide prescribes).  Output: gpurun_out/pmc_fine_net.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d gpurun_out/pmc_fn_hbm -- python3 scripts/quick_time.py bf16 160000 > gpurun_out/pmc_fn_hbm.log 2>&1 || echo "FETCH/WRITE pass failed"
timeout -k 10 150 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_fn_l2 -- python3 scripts/quick_time.py bf16 160000 > gpurun_out/pmc_fn_l2.log 2>&1 || echo "TCC pass failed"
python3 - <<'PY'
import csv, glob
for st in ("hbm", "l2"):
    fs = glob.glob("gpurun_out/pmc_fn_%s/*/*counter_collection.csv" % st)
    if not fs: print(st, "no output"); continue
    last = {}
    for r in csv.DictReader(open(fs[0])):
        if "mlp_forward" in r["Kernel_Name"]:
            last[r["Counter_Name"]] = float(r["Counter_Value"]); name = r["Kernel_Name"]
    print(st, name[:60], last)
PY
