"""Developer probe for scripts/pmc_fine_net.sh: one 400x400, 64+128 render of the bench scene through dn_render_rays, fp16 instance
(PMC_SCRIPT=scripts/fused_composite_pmc.py; DEXNERF_FUSED_COMPOSITE=1 makes the network launches composite their own rays)."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "dex-nerf_amd")]
import nerf, bench
dev = torch.device("cuda:0")
nerf.set_precision("bf16")
models, cfg, ro, rd, ex, ed = bench.build_scene(dev, 0)
for _ in range(3):
    bench.render(models, cfg, ro, rd, ex, ed)
torch.cuda.synchronize()
print("done")
