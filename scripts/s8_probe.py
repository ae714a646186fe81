"""Developer probe (GPU): the 8-bit weight-gradient kernel (DN_PREC_BF16_S8) against the bf16 one and the exact-fp32 one on the same
saved tensors - accuracy (per-tensor cosine / relative error of dW, db) and time at 786,432 points."""
import os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import nerf
from nerf import _ops, synthetic as syn

dev = torch.device("cuda:0")
kw = dict(num_layers=8, hidden_size=256, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True)
sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(43, sigma_bias=-20.0, **kw).items()}
n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = 192
n = n_rays * S
gen = torch.Generator(device="cpu").manual_seed(1)
pts = (torch.rand(n, 3, generator=gen) * 2 - 1).to(dev)
vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=gen), dim=-1).to(dev)
g_out = (torch.randn(n, 4, generator=gen) * float(sys.argv[2] if len(sys.argv) > 2 else 1e-4)).to(dev)
res = {}
for prec in ("fp32", "bf16"):
    nerf.set_precision(prec)
    m = nerf.models.FlexibleNeRFModel(**kw); m.load_state_dict(sd); m = m.to(dev)
    pk = m.packed()
    _ops.pack_backward(pk, [x.weight for x in m.linear_modules()])
    out, act, masks = _ops.run_network_train(pk, pts, vd, S)
    grads = _ops.mlp_backward_data(pk, g_out, masks, n)
    shapes = [tuple(x.weight.shape) for x in m.linear_modules()]
    res[prec] = _ops.mlp_weight_grad_all(pk, act, grads, n, shapes)
    if prec == "bf16":
        scale = float(sys.argv[3]) if len(sys.argv) > 3 else 65536.0
        act8 = _ops.convert_saved_s8(pk, 0, act, n)
        grads8 = _ops.convert_saved_s8(pk, 1, grads, n, grad_scale=scale)
        res["s8"] = _ops.mlp_weight_grad_all(pk, act8, grads8, n, shapes, s8=True)

        def timed(f, reps=5):
            f(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                f()
            b.record(); torch.cuda.synchronize()
            return a.elapsed_time(b) / reps * 1e3
        t16 = timed(lambda: _ops.mlp_weight_grad_all(pk, act, grads, n, shapes))
        t8 = timed(lambda: _ops.mlp_weight_grad_all(pk, act8, grads8, n, shapes, s8=True))
        print(f"weight-gradient kernel, {n} points: bf16 buffers {t16:.0f} us ({(act.numel() + grads.numel()) / t16 / 1e6:.2f} TB/s), "
              f"8-bit buffers {t8:.0f} us ({(act8.numel() + grads8.numel()) / t8 / 1e6:.2f} TB/s)")
nerf.set_precision("fp32")
names = [nm for nm, _ in m.named_modules() if isinstance(_, torch.nn.Linear)]
def cos(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float(a @ b / (a.norm() * b.norm() + 1e-300))
print(f"{'layer':16s} {'cos(bf16,fp32) dW':>18s} {'cos(s8,fp32) dW':>16s} {'cos(s8,bf16) dW':>16s} {'cos db bf16':>12s} {'cos db s8':>10s}  |dW| fp32")
for i, nm in enumerate(names):
    w32, b32 = res["fp32"][i]; w16, b16 = res["bf16"][i]; w8, b8 = res["s8"][i]
    print(f"{nm:16s} {cos(w16, w32):18.4f} {cos(w8, w32):16.4f} {cos(w8, w16):16.4f} {cos(b16, b32):12.4f} {cos(b8, b32):10.4f}  {float(w32.norm()):.3e}")
