#!/bin/bash
# Developer tool (GPU box): the round's measurement batch -> gpurun_out/ (bench line, bench under rocprofv3 --kernel-trace --stats,
# PMC passes on the headline kernel in fp16 and bf16, PMC passes on the training kernels of the 8-bit mode)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python3 bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err && echo "bench done" &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03_bench -- python3 bench.py --steps 5 --no-train > gpurun_out/r03_bench_under_rocprof.json 2> gpurun_out/r03_bench_under_rocprof.err && echo "bench under rocprof done" &&
bash scripts/pmc_fine_net.sh fine_net_fp16 "PMC_ARGS=fp16 160000" > gpurun_out/pmc_fine_net_fp16.log 2>&1 && echo "pmc fp16 done" &&
bash scripts/pmc_fine_net.sh fine_net_bf16 "PMC_ARGS=bf16 160000" > gpurun_out/pmc_fine_net_bf16.log 2>&1 && echo "pmc bf16 done" &&
bash scripts/pmc_train_kernels.sh bf16-s8 > gpurun_out/pmc_train_s8.log 2>&1 && echo "pmc train done"
