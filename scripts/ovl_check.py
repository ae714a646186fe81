"""Developer check: the as-shipped nets' forward instance that encodes tile t + 1 inside tile t (mlp_forward48_kernel<128, F, 4, 0, 1, 0, 1>)
against the plain fixed-shape instance (DEXNERF_G48_NO_OVERLAP=1) on the same rays + depths: bit-identical, several sizes; times both."""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd"))
import nerf
from nerf import _hip, _ops, synthetic as syn
dev = torch.device("cuda:0")
kw = dict(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True)
ok = True
for prec in ("bf16", "fp16"):
    nerf.set_precision("bf16")
    m = nerf.models.FlexibleNeRFModel(**kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.synth_state_dict(43, sigma_bias=-2.0, **kw).items()})
    m = m.to(dev)
    pk = m.packed(precision={"bf16": _hip.PREC_BF16, "fp16": _hip.PREC_F16}[prec])
    for n, s in [(1, 1), (3, 16), (7, 55), (1000, 64), (1024, 128), (129600, 128), (200001, 64)]:
        g = torch.Generator(device=dev).manual_seed(n + s)
        rd = torch.nn.functional.normalize(torch.randn(n, 3, device=dev, generator=g), dim=-1)
        rays = torch.cat([torch.randn(n, 3, device=dev, generator=g), rd, torch.full((n, 1), 0.3, device=dev), torch.full((n, 1), 4.0, device=dev), rd], -1).contiguous()
        z = torch.sort(torch.rand(n, s, device=dev, generator=g) * 3.7 + 0.3, -1)[0].contiguous()
        outs, times = {}, {}
        for mode in ("overlap", "plain"):
            if mode == "plain":
                os.environ["DEXNERF_G48_NO_OVERLAP"] = "1"
            else:
                os.environ.pop("DEXNERF_G48_NO_OVERLAP", None)
            reps = []
            for rep in range(12):   # device time of the launch (events on the current stream), median of the last ten
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = _ops.run_network_rays(pk, rays, z)
                e1.record(); torch.cuda.synchronize()
                reps.append(e0.elapsed_time(e1) * 1e-3)
            times[mode] = sorted(reps[2:])[5]
            outs[mode] = out
        same = torch.equal(outs["overlap"], outs["plain"])
        ok = ok and same
        print(f"{prec} rays {n} x {s}: identical={same} finite={bool(torch.isfinite(outs['overlap']).all())}  overlap {times['overlap']*1e3:.3f} ms  plain {times['plain']*1e3:.3f} ms", flush=True)
os.environ.pop("DEXNERF_G48_NO_OVERLAP", None)
print("ALL IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
