#!/bin/bash
# Developer tool (GPU box): PMC passes on the bf16 fine-net launch (scripts/quick_time.py bf16 160000 = 160,000 rays x 192
# points through the kernel bench.py times).  ONE counter group per rocprofv3 run: FETCH_SIZE costs 3 of the 4 TCC slots and
# WRITE_SIZE 2 (MI355X_MICROARCH.md, rocprofv3 PMC slots) - asking for both in one pass aborts the profiler with
# "error code 38: Request exceeds the capabilities of the hardware to collect" (that was round 1's "pass does not complete").
#   usage: scripts/pmc_fine_net.sh [tag] [ENV=VAL ...]      output: gpurun_out/pmc_<tag>.json (+ per-pass logs / csv)
# PMC_SCRIPT / PMC_ARGS pick another launch, e.g. the as-shipped 4 x 128 nets' fine launch of config 3 (PMC_POINTS = its points,
# for the algorithmic bytes):  scripts/pmc_fine_net.sh w128 PMC_SCRIPT=scripts/quick_time128.py "PMC_ARGS=fp16 129600 128" PMC_POINTS=16588800
tag=${1:-fine_net}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for kv in "$@"; do export "$kv"; done
pass() {  # name, counters...
  name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}_$name -- python3 ${PMC_SCRIPT:-scripts/quick_time.py} ${PMC_ARGS:-bf16 160000} > gpurun_out/pmc_${tag}_$name.log 2>&1 || echo "pass $name failed (see gpurun_out/pmc_${tag}_$name.log)"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass l2 TCC_HIT_sum TCC_MISS_sum
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass sqA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC
pass sqB SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16
python3 scripts/pmc_collect.py $tag
