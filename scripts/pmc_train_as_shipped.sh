#!/bin/bash
# Developer tool (GPU box): PMC passes on the network kernels of the as-shipped 4 x 128 nets' training iteration (1024 rays, 64 + 128
# samples, the default 'bf16' mode, eager launches) -> gpurun_out/pmc_train_as_shipped.json (fine-network launches averaged)
export PMC_TRAIN_CMD="dex-nerf_amd/train_dexnerf.py --iters 300 --size 64 --views 8 --num-random-rays 1024 --layers 4 --width 128 --num-fine 64 --validate-every 0 --quiet --precision bf16 --no-hip-graph"
export PMC_KERNELS="mlp_forward48_kernel<128, 1, 4, 0u, 1, 3;mlp_backward48_kernel<128, 4, 1, 2>;weight_grad_batch_kernel_s8_small"
bash scripts/pmc_train_kernels.sh as_shipped
