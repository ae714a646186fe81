"""Developer check: the bf16 inference kernel selected by DEXNERF_BF16_GEOM (48-point default / 32) against the exact-fp32
kernel on the same inputs, several shapes; prints error statistics and the kernel time at the bench size."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd"))
import nerf
from nerf import _ops, synthetic as syn
import bench
dev = torch.device("cuda:0")
print("geometry:", os.environ.get("DEXNERF_BF16_GEOM", "48 (default)"))
worst = 0.0
for (D, view, skip) in [(8, True, 4), (8, False, 4), (5, True, 2), (2, True, 4), (3, False, 100)]:
    kw = dict(num_layers=D, hidden_size=256, skip_connect_every=skip, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=view)
    sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(7 + D, sigma_gain=5.0, sigma_bias=0.0, **kw).items()}
    for n_rays, s in [(1, 1), (7, 5), (61, 64), (500, 192)]:
        pts = torch.randn(n_rays, s, 3, device=dev)
        vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1)
        outs = {}
        for prec in ("fp32", "bf16"):
            nerf.set_precision(prec)
            m = nerf.models.FlexibleNeRFModel(**kw); m.load_state_dict(sd); m = m.to(dev)
            with torch.no_grad():
                outs[prec] = _ops.run_network_pts(m.packed(), pts.reshape(-1, 3), vd if view else None, s)
        ref, got = outs["fp32"], outs["bf16"]
        err = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
        worst = max(worst, err)
        print(f"D{D} view{int(view)} skip{skip} rays {n_rays}x{s}: max|bf16-fp32|/max|fp32| = {err:.3e}  finite={bool(torch.isfinite(got).all())}", flush=True)
print("worst", worst)
# rays + depths path at the bench size, timing
nerf.set_precision("bf16")
models, cfg, ro, rd, ex, ed = bench.build_scene(dev, 0)
n = 160000
rays = torch.cat([ro.reshape(-1, 3)[:n], rd.reshape(-1, 3)[:n], torch.full((n, 1), 2.0, device=dev), torch.full((n, 1), 6.0, device=dev),
                  torch.nn.functional.normalize(rd.reshape(-1, 3)[:n], dim=-1)], -1).contiguous()
z = torch.sort(torch.rand(n, 192, device=dev) * 4 + 2, -1)[0].contiguous()
pk = models[1].packed()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = _ops.run_network_rays(pk, rays, z)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"bench size: {dt*1e3:.2f} ms  {n*192*bench.FLOP_PER_POINT/dt/1e12:.1f} TFLOP/s finite={bool(torch.isfinite(out).all())}")
nerf.set_precision("fp32")
pk32 = models[1].packed()
o32 = _ops.run_network_rays(pk32, rays[:4096].contiguous(), z[:4096].contiguous())
e = (out[:4096] - o32).abs().max().item() / o32.abs().max().item()
print(f"rays path vs fp32 on the first 4096 rays: {e:.3e}")
