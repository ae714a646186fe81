"""Developer stress run: random W=128 / 256 nets / point counts / input forms through the 48-points-per-wave bf16 inference kernel,
against the 32-point bf16 kernel (mean difference: layout check) and the exact-fp32 kernel (bf16-level error, equal for both)."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd"))
import nerf
from nerf import _ops, synthetic as syn
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for it in range(n_cfg):
    D = int(rng.integers(2, 10)); view = bool(rng.integers(0, 2)); skip = int(rng.choice([2, 3, 4, 5, 100]))
    lx = int(rng.choice([10, 6])); logs = bool(rng.integers(0, 2)); width = int(rng.choice([128, 256]))
    kw = dict(num_layers=D, hidden_size=width, skip_connect_every=skip, num_encoding_fn_xyz=lx, num_encoding_fn_dir=4, use_viewdirs=view)
    sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(int(rng.integers(1, 1000)), sigma_gain=5.0, sigma_bias=0.0, **kw).items()}
    packed = {}
    for prec in ("fp32", "bf16"):
        nerf.set_precision(prec)
        m = nerf.models.FlexibleNeRFModel(**kw)
        m.load_state_dict(sd); m = m.to(dev)
        packed[prec] = m.packed(log_sampling_xyz=logs, log_sampling_dir=logs)
    n_rays = int(rng.integers(1, 300)); s = int(rng.integers(1, 260))
    pts = torch.randn(n_rays, s, 3, device=dev) * float(rng.choice([0.5, 1.0, 3.0]))
    vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1)
    rays = torch.cat([torch.randn(n_rays, 3, device=dev), vd * 1.3, torch.zeros(n_rays, 2, device=dev), vd], -1).contiguous()
    z = torch.sort(torch.rand(n_rays, s, device=dev) * 4 + 2, -1)[0].contiguous()
    msgs = []
    for form in ("pts", "rays"):
        def run(prec):
            with torch.no_grad():
                if form == "pts":
                    return _ops.run_network_pts(packed[prec], pts.reshape(-1, 3), vd if view else None, s)
                return _ops.run_network_rays(packed[prec], rays, z)
        ref = run("fp32")
        os.environ["DEXNERF_BF16_GEOM"] = "32"; o32 = run("bf16"); del os.environ["DEXNERF_BF16_GEOM"]
        o48 = run("bf16")
        sc = float(ref.abs().max()) + 1e-6
        d = (o48 - o32).abs() / sc; e48 = (o48 - ref).abs() / sc; e32 = (o32 - ref).abs() / sc
        big = ref.numel() >= 2000
        if not torch.isfinite(o48).all(): msgs.append(f"{form}: non-finite")
        # (linear frequency sampling: the 48-point kernel multiplies by f / 2 pi, the 32-point one by f and then by 1 / 2 pi - for the
        # default power-of-two frequencies the two round identically, for others the sine's argument differs by an ulp: ~1e-4 rad at 2^9 x)
        if float(d.max()) > 8e-2 or (big and float(d.mean()) > (1e-4 if logs else 5e-4)): msgs.append(f"{form}: 48 vs 32 mean {float(d.mean()):.2e} max {float(d.max()):.2e}")
        if big and abs(float(e48.mean()) - float(e32.mean())) > 0.15 * float(e32.mean()) + 1e-5: msgs.append(f"{form}: vs fp32 mean 48 {float(e48.mean()):.2e} / 32 {float(e32.mean()):.2e}")
    tag = f"W{width} D{D} skip{skip} view{int(view)} LX{lx} log{int(logs)} rays{n_rays}x{s}"
    print(("BAD  " if msgs else "ok   ") + tag + ("  " + "; ".join(msgs) if msgs else ""), flush=True)
    bad += bool(msgs)
nerf.set_precision("fp32")
print(f"{n_cfg - bad} / {n_cfg} configurations clean")
