"""Developer timing probe: the fused network kernel on the as-shipped 4 x 128 nets (BASELINE config 3 shapes)."""
import sys, os, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd"))
import nerf
from nerf import _ops, synthetic as syn

dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 129600
s = int(sys.argv[3]) if len(sys.argv) > 3 else 128
nerf.set_precision(prec)
kw = dict(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True)
m = nerf.models.FlexibleNeRFModel(**kw)
m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.synth_state_dict(43, sigma_bias=-20.0, **kw).items()})
m = m.to(dev)
rd = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=-1)
rays = torch.cat([torch.randn(n, 3, device=dev), rd, torch.full((n, 1), 0.3, device=dev), torch.full((n, 1), 4.0, device=dev), rd], -1).contiguous()
z = torch.sort(torch.rand(n, s, device=dev) * 3.7 + 0.3, -1)[0].contiguous()
pk = m.packed()
flop_pt = 2 * (63 * 128 + 3 * 128 * 128 + 128 * 128 + 128 + (128 + 27) * 64 + 64 * 3)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = _ops.run_network_rays(pk, rays, z)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{prec} 4x128 rays={n} S={s} t={dt*1e3:.3f} ms  {n * s * flop_pt / dt / 1e12:.1f} TFLOP/s  finite={bool(torch.isfinite(out).all())}", flush=True)
