#!/usr/bin/env python3
"""Developer probe: bench.py's training leg (4096 rays, 64+128, D8/W256 x2, nerf.FusedTrainStep) alone, for rocprofv3
--kernel-trace --stats.  usage: train_step_profile.py [precision] [steps]   (DEXNERF_BENCH_NO_GRAPH=1: eager launches)"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "dex-nerf_amd")]
import nerf
import bench
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16-s8"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
nerf.set_precision(prec)
dev = torch.device("cuda:0")
models, cfg, ro, rd, ex, ed = bench.build_scene(dev, 0)
res = bench.train_rate(models, cfg, ro, rd, ex, ed, steps=steps)
print(f"{prec}: {res['ms_per_step']:.3f} ms per step, graphs per step {res['hip_graphs_per_step']}")
