"""Developer probe: one configuration of the 48- vs 32-point bf16 kernels vs fp32, error statistics."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd"))
import nerf
from nerf import _ops, synthetic as syn
dev = torch.device("cuda:0")
D, view, skip = 5, True, 2
kw = dict(num_layers=D, hidden_size=256, skip_connect_every=skip, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=view)
sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(11 + D, sigma_gain=5.0, sigma_bias=0.0, **kw).items()}
packed = {}
for prec in ("fp32", "bf16"):
    nerf.set_precision(prec)
    m = nerf.models.FlexibleNeRFModel(**kw); m.load_state_dict(sd); m = m.to(dev)
    packed[prec] = (m, m.packed())
gen = torch.Generator(device="cpu").manual_seed(5)
for scale_o in (0.0, 1.0, 3.0):
    n_rays, s = 3, 1000
    vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=gen), dim=-1).to(dev)
    rays = torch.cat([scale_o * torch.randn(n_rays, 3, generator=gen).to(dev), vd * 1.5, torch.zeros(n_rays, 2, device=dev), vd], -1).contiguous()
    z = torch.sort(torch.rand(n_rays, s, generator=gen) * 4 + 2, -1)[0].to(dev).contiguous()
    with torch.no_grad():
        ref = _ops.run_network_rays(packed["fp32"][1], rays, z)
        os.environ["DEXNERF_BF16_GEOM"] = "32"; o32 = _ops.run_network_rays(packed["bf16"][1], rays, z)
        del os.environ["DEXNERF_BF16_GEOM"]; o48 = _ops.run_network_rays(packed["bf16"][1], rays, z)
    sc = float(ref.abs().max())
    for name, a, b in (("48-fp32", o48, ref), ("32-fp32", o32, ref), ("48-32", o48, o32)):
        dd = (a - b).abs()
        print(f"origin scale {scale_o}: {name}: max {float(dd.max())/sc:.3e} mean {float(dd.mean())/sc:.3e}  p99.9 {float(dd.flatten().kthvalue(int(dd.numel()*0.999)).values)/sc:.3e}  argmax {int(dd.argmax())}")
