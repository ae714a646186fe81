"""Developer check: dn_render_rays with the network launches compositing their own rays (DEXNERF_FUSED_COMPOSITE=1; off by default:
it is slower) against the two-kernel path - every map bit for bit; times both."""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "dex-nerf_amd")]
import nerf
from nerf import _hip, _ops, synthetic as syn
dev = torch.device("cuda:0")
thres = [float(m) for m in range(5, 105, 5)]
ok = True
nerf.set_precision("bf16")
for width, layers, biases in ((256, 8, (-150.0, -20.0)), (128, 4, (-15.0, -2.0))):
    kw = dict(num_layers=layers, hidden_size=width, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True)
    nets = []
    for seed, b in zip((42, 43), biases):
        m = nerf.models.FlexibleNeRFModel(**kw)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.synth_state_dict(seed, sigma_bias=b, **kw).items()})
        nets.append(m.to(dev))
    for prec in (_hip.PREC_F16, _hip.PREC_BF16):
        pc, pf = nets[0].packed(precision=prec), nets[1].packed(precision=prec)
        for n, nc, nf, white in ((1, 64, 128, False), (777, 64, 128, True), (5000, 64, 64, False), (2000, 64, 192, False), (300, 32, 0, False), (40000, 64, 128, False)):
            g = torch.Generator(device=dev).manual_seed(n + nc)
            rd = torch.nn.functional.normalize(torch.tensor([0.0, 0.0, -1.0], device=dev) + 0.3 * torch.randn(n, 3, device=dev, generator=g), dim=-1) * 1.3
            ro = torch.tensor([0.0, 0.0, 4.0], device=dev) + 0.05 * torch.randn(n, 3, device=dev, generator=g)
            rays = torch.cat([ro, rd, torch.full((n, 1), 2.0, device=dev), torch.full((n, 1), 6.0, device=dev), torch.nn.functional.normalize(rd, dim=-1)], -1).contiguous()
            outs, times = {}, {}
            for mode in ("fused", "two-kernel"):
                if mode == "fused":
                    os.environ["DEXNERF_FUSED_COMPOSITE"] = "1"
                else:
                    os.environ.pop("DEXNERF_FUSED_COMPOSITE", None)
                for rep in range(3):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    out = _ops.render_rays(pc, pf if nf else None, rays, nc, nf, False, 0.0, white, thres)
                    torch.cuda.synchronize(); times[mode] = time.perf_counter() - t0
                outs[mode] = [o for o in out if o is not None]
            same = all(torch.equal(a, b) for a, b in zip(outs["fused"], outs["two-kernel"]))
            fin = all(bool(torch.isfinite(a).all()) for a in outs["fused"])
            ok = ok and same
            print(f"W{width} prec {prec} rays {n} {nc}+{nf} white={int(white)}: identical={same} finite={fin}  fused {times['fused']*1e3:.3f} ms  two-kernel {times['two-kernel']*1e3:.3f} ms", flush=True)
os.environ.pop("DEXNERF_FUSED_COMPOSITE", None)
print("ALL IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
