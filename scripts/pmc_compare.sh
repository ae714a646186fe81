#!/bin/bash
# usage: scripts/pmc_compare.sh <tag> [ENV=VAL ...]   (run on the GPU box; developer tool)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC"
B="SQ_WAIT_INST_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE"
for set in A B; do
  eval ctrs=\$$set
  env "$@" rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/pmc_${tag}_$set -- python3 scripts/quick_time.py bf16 160000 > gpurun_out/pmc_${tag}_$set.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for st in "AB":
    f = glob.glob("gpurun_out/pmc_${tag}_%s/*/*counter_collection.csv" % st)[0]
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "mlp_forward" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] = float(r["Counter_Value"])
    for k, v in agg.items():
        print("${tag} %-32s %.4g" % (k, v))
PY
