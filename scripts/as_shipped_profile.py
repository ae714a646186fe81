#!/usr/bin/env python3
"""Developer probe: the as-shipped Dex-NeRF training configuration (4x128 nets, 1024 rays, 64+64 samples; reference loop
train_dexnerf_rgb.py:229-289) for a few hundred iterations - run it under rocprofv3 --kernel-trace to count the kernels of one
captured iteration.  usage: as_shipped_profile.py [precision] [iters]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "dex-nerf_amd")]
import train_dexnerf
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16-s8"
iters = sys.argv[2] if len(sys.argv) > 2 else "600"
res = train_dexnerf.main(["--iters", iters, "--size", "64", "--views", "8", "--num-random-rays", "1024", "--layers", "4", "--width", "128",
                          "--num-fine", "64", "--validate-every", "0", "--quiet", "--precision", prec] + sys.argv[3:])
print(f"{prec}: {res['rays_per_s']:.0f} rays/s overall, steady {res.get('steady_ms_per_iter', float('nan')):.4f} ms per iteration "
      f"({res['hip_graphs']} graph(s) per iteration), final train PSNR {res['history'][-1][2]:.2f} dB")
