"""Developer timing probe: fused network kernel at a few sizes, both precisions (not part of the bench contract)."""
import sys, os, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd"))
import nerf
from nerf import _ops, synthetic as syn
import bench

dev = torch.device("cuda:0")
precs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["fp32", "bf16"]
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [256, 2048, 16384]
for prec in precs:
    nerf.set_precision(prec)
    models, cfg, ro, rd, ex, ed = bench.build_scene(dev, 0)
    for n in sizes:
        rays = torch.cat([ro.reshape(-1, 3)[:n], rd.reshape(-1, 3)[:n], torch.full((n, 1), 2.0, device=dev),
                          torch.full((n, 1), 6.0, device=dev), torch.nn.functional.normalize(rd.reshape(-1, 3)[:n], dim=-1)], -1).contiguous()
        z = torch.sort(torch.rand(n, 192, device=dev) * 4 + 2, -1)[0].contiguous()
        pk = models[1].packed()
        torch.cuda.synchronize()
        for rep in range(3):
            t0 = time.perf_counter()
            out = _ops.run_network_rays(pk, rays, z)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        tf = n * 192 * bench.FLOP_PER_POINT / dt / 1e12
        print(f"{prec} rays={n} pts={n*192} t={dt*1e3:.2f} ms  {tf:.1f} TFLOP/s  finite={bool(torch.isfinite(out).all())}", flush=True)
