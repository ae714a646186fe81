#!/bin/bash
# Developer tool (GPU box): round 4's measurement batch -> gpurun_out/ (bench line, bench under rocprofv3 --kernel-trace --stats,
# PMC passes on the headline kernel in bf16 and fp16 - the records bench.py ties to the kernel sources -, one training step under
# rocprofv3 --kernel-trace --stats in the default 'bf16' mode)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/pmc_fine_net.sh fine_net_bf16 "PMC_ARGS=bf16 160000" > gpurun_out/pmc_fine_net_bf16.log 2>&1 && echo "pmc bf16 done" &&
bash scripts/pmc_fine_net.sh fine_net_fp16 "PMC_ARGS=fp16 160000" > gpurun_out/pmc_fine_net_fp16.log 2>&1 && echo "pmc fp16 done" &&
mkdir -p profiles_new && cp gpurun_out/pmc_fine_net_bf16.json profiles/r04_pmc_fine_net_bf16.json && cp gpurun_out/pmc_fine_net_fp16.json profiles/r04_pmc_fine_net_fp16.json &&
T0=$SECONDS && timeout -k 10 400 python3 bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err && echo "bench done in $((SECONDS - T0)) s (wall, default flags)" &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04_bench -- python3 bench.py --steps 5 --no-train > gpurun_out/r04_bench_under_rocprof.json 2> gpurun_out/r04_bench_under_rocprof.err && echo "bench under rocprof done" &&
DEXNERF_BENCH_NO_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04_train -- python3 scripts/train_step_profile.py bf16 10 > gpurun_out/r04_train_step.log 2>&1 && echo "train step under rocprof done"
