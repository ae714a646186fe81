#!/bin/bash
# Developer tool (GPU box): PMC passes on the kernels of one training step (scripts/train_bench.py <mode> 4096), one counter
# group per rocprofv3 run.   usage: scripts/pmc_train_kernels.sh <mode: bf16 | bf16-s8>   output: gpurun_out/pmc_train_<mode>.json
mode=${1:-bf16-s8}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
pass() {  # name, counters...
  name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_train_${mode}_$name -- python3 ${PMC_TRAIN_CMD:-scripts/train_bench.py $mode 4096} > gpurun_out/pmc_train_${mode}_$name.log 2>&1 || echo "pass $name failed (see gpurun_out/pmc_train_${mode}_$name.log)"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass sqA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC
pass sqB SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT
python3 scripts/pmc_train_collect.py $mode
rm -rf gpurun_out/pmc_train_${mode}_mfma gpurun_out/pmc_train_${mode}_sqA gpurun_out/pmc_train_${mode}_sqB gpurun_out/pmc_train_${mode}_fetch gpurun_out/pmc_train_${mode}_write
