"""Developer probe (CPU): how far two fp32-level evaluations of the SAME reference arithmetic are apart on the bench scene.

The oracle renders the bench's 16,384-ray sample twice: as is (fp32 ATen), and with only the MLP evaluated in float64 (inputs,
weights and outputs cast; everything else - sampling, compositing - unchanged).  The difference is pure fp32 re-association
noise of the network, amplified by resampling and compositing: the floor under any `max|a-b| <= 1e-4 max|b|` comparison at this
sample size (SURVEY.md section 8c measured 6.6e-5 on 10,000 rays of the lego nets)."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd"))
import bench  # noqa: E402
from nerf import synthetic as syn  # noqa: E402
from oracle import nerf_oracle as oc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
torch.set_num_threads(8)
sd_c = oc.to_torch_sd(syn.synth_state_dict(42, sigma_bias=-150.0, **bench.MODEL_KW))
sd_f = oc.to_torch_sd(syn.synth_state_dict(43, sigma_bias=-20.0, **bench.MODEL_KW))
ro, rd = oc.get_ray_bundle(bench.H, bench.W, syn.scene_pose(7), syn.intrinsic(bench.H, bench.W))
sel = syn.select_rays(bench.H, bench.W, n, seed=0)
ro, rd = ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]
cfg = oc.RenderCfg(num_coarse=bench.NC, num_fine=bench.NF, near=2.0, far=6.0, chunksize=4096, m_thres=bench.M_THRES)
mc = oc.ModelCfg(**bench.MODEL_KW)
with torch.no_grad():
    a = oc.run_one_iter(ro, rd, sd_c, sd_f, mc, mc, cfg)
    orig = oc.flexible_mlp

    def mlp64(sd, x, m):
        return orig({k: v.double() for k, v in sd.items()}, x.double(), m).float()
    oc.flexible_mlp = mlp64
    b = oc.run_one_iter(ro, rd, sd_c, sd_f, mc, mc, cfg)
names = ["rgb_coarse", "depth_coarse", "acc_coarse", "rgb_fine", "depth_fine", "acc_fine"]
for name, x, y in zip(names, a[:6], b[:6]):
    x, y = x.numpy().astype(np.float64), y.numpy().astype(np.float64)
    err = np.abs(x - y).reshape(len(x), -1).max(-1) / np.abs(y).max()
    print(f"{name:13s} max {err.max():.2e}  p99.9 {np.quantile(err, 0.999):.2e}  p99 {np.quantile(err, 0.99):.2e}  rays over 1e-4: {(err > 1e-4).sum()} of {len(err)}")
dex_a = np.stack([t.numpy() for t in a[6:]]); dex_b = np.stack([t.numpy() for t in b[6:]])
miss = np.abs(dex_a - dex_b)
print(f"dex: agree {(miss <= 1e-4 * np.abs(dex_b).max()).mean():.6f}, worst miss {miss.max():.4f} m")
