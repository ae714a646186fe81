"""Developer probe: time dn_mlp_weight_grad for one trunk layer (256 x 256) at the fine and coarse point counts."""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd"))
import nerf
from nerf import _ops, _train, _hip
import bench

dev = torch.device("cuda:0")
nerf.set_precision("bf16")
models, cfg, ro, rd, ex, ed = bench.build_scene(dev, 0)
m = models[1]
pk = m.packed()
slots, gslots, kh = _train._slots(m, pk.precision)
for n in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["786432", "262144", "32768"])]:
    a, mk, g = _ops.train_sizes(pk, n)
    act = (torch.randn(a // 2, device=dev) * 0.5).to(torch.bfloat16).view(torch.uint8)
    grads = (torch.randn(g // 2, device=dev) * 0.5).to(torch.bfloat16).view(torch.uint8)
    for name, args in (("trunk 256x256", (gslots["trunk0"] + kh, 256, slots["trunk0"], 256, 0)),
                       ("skip 256x320", (gslots["trunk0"] + 4 * kh, 256, slots["trunk0"] + 3 * kh, 256, 1)),
                       ("fc_rgb 3x128", (gslots["out"], 3, slots["dirout"], 128, 0))):
        dw = torch.zeros(args[1], 320, device=dev); db = torch.zeros(args[1], device=dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for rep in range(3):
            dw.zero_(); db.zero_()
            ev[0].record()
            _ops.mlp_weight_grad(pk, act, grads, n, *args, dw, db)
            ev[1].record()
            torch.cuda.synchronize()
        t = ev[0].elapsed_time(ev[1]) * 1e3
        bytes_ = n * (args[1] if args[1] >= 32 else 8) * 2 + n * (args[3] + (64 if args[4] == 1 else 0)) * 2
        print(f"n={n} {name}: {t:.1f} us  {bytes_ / t / 1e6:.2f} TB/s  checksum {dw.double().sum().item():.6e}", flush=True)

mods = m.linear_modules()
shapes = [tuple(x.weight.shape) for x in mods]
for n in [786432, 262144, 32768]:
    a, mk, g = _ops.train_sizes(pk, n)
    act = (torch.randn(a // 2, device=dev) * 0.5).to(torch.bfloat16).view(torch.uint8)
    grads = (torch.randn(g // 2, device=dev) * 0.5).to(torch.bfloat16).view(torch.uint8)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for rep in range(3):
        ev[0].record()
        res = _ops.mlp_weight_grad_all(pk, act, grads, n, shapes)
        ev[1].record()
        torch.cuda.synchronize()
    t = ev[0].elapsed_time(ev[1]) * 1e3
    bytes_ = n * 10592
    print(f"n={n} ALL layers: {t:.1f} us  {bytes_ / t / 1e6:.2f} TB/s (incl. zero-fill)", flush=True)
