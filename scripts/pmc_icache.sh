#!/bin/bash
# Developer tool (GPU box): instruction-fetch counters on the fine-net launch (is the straight-line 80+ KB tile pass paying for
# instruction-cache misses?).  usage: scripts/pmc_icache.sh [tag] [prec]
tag=${1:-icache}; prec=${2:-bf16}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
pass() { name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}_$name -- python3 scripts/quick_time.py $prec 160000 > gpurun_out/pmc_${tag}_$name.log 2>&1 || echo "pass $name failed"
}
pass ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
pass ic2 SQ_INST_LEVEL_VMEM SQ_IFETCH_LEVEL SQ_WAIT_ANY SQ_BUSY_CYCLES
for n in ic ic2; do f=$(ls gpurun_out/pmc_${tag}_$n/*/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && grep mlp_forward $f | awk -F, '{print $(NF-1), $NF}' | sort | uniq -c | tail -20; done
