"""Developer soak (GPU): the kernels whose prologue starts on ring phases 0 and 1 only (csrc/mlp_stage48.h g48_prologue_wait) - the
one-phase-period instances: W = 128 inference, every training forward / backward - on launches of ONE or TWO tiles per workgroup, where
the first tile starts while phases 2 and 3 are still in flight.  Many repetitions with fresh inputs, each compared bit for bit with a
launch that cannot share the hazard: the 32-point kernel for inference (DEXNERF_BF16_GEOM=32 is another pipeline) is NOT bit-identical,
so the reference here is the same kernel run on a grid padded to MANY tiles per workgroup (the first tiles of those workgroups are other
points) - plus the training kernels' two-group vs three-group instances.  usage: scripts/soak_prologue.py [reps]"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd"))
import nerf
from nerf import _hip, _ops
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for width, depth in ((128, 4), (256, 8)):
    nerf.set_precision("bf16")
    torch.manual_seed(width)
    m = nerf.models.FlexibleNeRFModel(num_layers=depth, hidden_size=width, skip_connect_every=4, num_encoding_fn_xyz=10,
                                      num_encoding_fn_dir=4, use_viewdirs=True).to(dev)
    pk = m.packed()
    S8 = _hip.PREC_BF16_S8
    _ops.pack_backward(pk, [x.weight for x in m.linear_modules()], S8)
    shapes = [tuple(x.weight.shape) for x in m.linear_modules()]
    for rep in range(reps):
        g = torch.Generator(device=dev).manual_seed(1000 * width + rep)
        n_rays, s = (256 + 37 * rep) % 1500 + 64, 64          # 4 k .. 100 k points: one tile per workgroup, or a ragged second round
        n = n_rays * s
        pts = torch.rand(n, 3, device=dev, generator=g) * 2 - 1
        vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev, generator=g), dim=-1)
        g_out = torch.randn(n, 4, device=dev, generator=g) * 1e-4
        # inference (W = 128: the one-phase-period instance): the launch alone vs the same points inside a 40x larger launch
        with torch.no_grad():
            small = _ops.run_network_pts(pk, pts, vd, s)
            big_pts = torch.cat([pts, torch.rand(39 * n, 3, device=dev, generator=g) * 2 - 1])
            big_vd = torch.cat([vd, torch.nn.functional.normalize(torch.randn(39 * n_rays, 3, device=dev, generator=g), dim=-1)])
            big = _ops.run_network_pts(pk, big_pts, big_vd, s)[:n]
        ok = torch.equal(small, big)
        res = {}
        for groups in ("2", "3"):
            os.environ["DEXNERF_G48_TRAIN_GROUPS"] = groups
            out, act, masks = _ops.run_network_train(pk, pts, vd, s, prec=S8)
            grads = _ops.mlp_backward_data(pk, g_out, masks, n, prec=S8)
            wg = _ops.mlp_weight_grad_all(pk, act, grads, n, shapes, prec=S8)
            res[groups] = (out, [t for pair in wg for t in pair])
        os.environ.pop("DEXNERF_G48_TRAIN_GROUPS")
        ok = ok and torch.equal(res["2"][0], res["3"][0]) and torch.equal(res["2"][0], small)
        ok = ok and all(torch.equal(a, b) for a, b in zip(res["2"][1], res["3"][1]))
        if not ok:
            bad += 1
            print(f"MISMATCH W{width} rep {rep}: rays {n_rays} x {s}", flush=True)
    print(f"W{width} D{depth}: {reps} repetitions done", flush=True)
nerf.set_precision("fp32")
print("ALL IDENTICAL" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
