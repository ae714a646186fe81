"""Developer probe: HIP-event times of the three training kernels (forward-with-save, backward-data, weight-grad) for
the fine and coarse nets of the C2 workload at 4096 rays."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd"))
import nerf
from nerf import _ops, _hip
import bench

dev = torch.device("cuda:0")
nerf.set_precision("bf16")
models, cfg, ro, rd, ex, ed = bench.build_scene(dev, 0)
m = models[1]
pk = m.packed()
S8 = _hip.PREC_BF16_S8   # the default training mode: 8-bit saved tensors, 48-point kernels
_ops.pack_backward(pk, [x.weight for x in m.linear_modules()], S8)
for n_rays, s in ((4096, 192), (4096, 64)):
    n = n_rays * s
    pts = torch.randn(n, 3, device=dev); vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1)
    shapes = [tuple(x.weight.shape) for x in m.linear_modules()]
    def timed(f):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for _ in range(3):
            ev[0].record(); r = f(); ev[1].record(); torch.cuda.synchronize()
        return r, ev[0].elapsed_time(ev[1]) * 1e3
    (out, act, masks), t_f = timed(lambda: _ops.run_network_train(pk, pts, vd, s, prec=S8))
    g = torch.randn(n, 4, device=dev)
    grads, t_b = timed(lambda: _ops.mlp_backward_data(pk, g, masks, n, prec=S8))
    _, t_w = timed(lambda: _ops.mlp_weight_grad_all(pk, act, grads, n, shapes, prec=S8))
    fl = n * bench.FLOP_PER_POINT / 1e6
    print(f"points={n}: fwd+save {t_f:.0f} us ({fl / t_f:.0f} TFLOP/s, {act.numel() / t_f / 1e6:.2f} TB/s written)  "
          f"bwd-data {t_b:.0f} us ({grads.numel() / t_b / 1e6:.2f} TB/s written)  dW {t_w:.0f} us "
          f"({(act.numel() + grads.numel()) / t_w / 1e6:.2f} TB/s read)", flush=True)
