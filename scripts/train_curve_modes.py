"""Developer probe (GPU): matched-iteration training curves of the training modes (fp32, bf16-s16 = 16-bit saved tensors, bf16 = the
default: 8-bit saved tensors with the per-launch gradient scale, bf16 with the fixed scale 2^16) on the built-in synthetic teacher scene -
same seeds, same ray draws.  usage: scripts/train_curve_modes.py [iters] [extra train_dexnerf args ...]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd"))
import train_dexnerf
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
extra = sys.argv[2:]
res = {}
for tag, prec, more in (("fp32", "fp32", []), ("bf16-s16", "bf16-s16", []), ("bf16", "bf16", ["--s8-grad-scale", "0"]),
                        ("bf16, scale 2^16", "bf16", ["--s8-grad-scale", "65536"])):
    res[tag] = train_dexnerf.main(["--iters", str(iters), "--size", "64", "--views", "12", "--num-random-rays", "1024", "--layers", "4",
                                   "--width", "128", "--num-fine", "64", "--validate-every", "0", "--quiet", "--precision", prec] + more + extra)
import numpy as np
print("| iteration | " + " | ".join(res) + " |\n|---|" + "---|" * len(res))
marks = sorted(set([0, 500, 1000, 2000, 4000, 8000, iters - 1]) & set(range(iters)))
for it in marks:
    row = []
    for prec in res:
        h = res[prec]["history"]
        # history = [(iteration, loss, psnr)], sampled; average the entries within +-100 iterations of the mark
        vals = [p for (i, _, p) in h if abs(i - it) <= 100]
        row.append(f"{np.mean(vals):.2f}" if vals else "-")
    print(f"| {it} | " + " | ".join(row) + " |")
print("| held-out view | " + " | ".join(f"{res[p]['val_psnr']:.2f}" for p in res) + " |")
print("| rays / s | " + " | ".join(f"{res[p]['rays_per_s'] / 1e6:.2f} M" for p in res) + " |")
