"""Developer stress run: random network shapes / point counts through the fused training kernels.
Checks per configuration: (1) training forward == inference forward bit for bit (fp32, bf16); (2) fp32-mode parameter
gradients == PyTorch autograd over the nn.Linear composition (1e-3); (3) bf16 gradients vs fp32 ones: cosine > 0.9 for
every tensor with a non-negligible norm; (4) the backward chain stages vs matmuls (bf16); (5) the 8-bit saved-tensor mode: same
forward bits as the 48-point inference kernel, weight gradients vs the bf16 kernel's (cosine)."""
import os, sys, itertools
os.environ["DEXNERF_BF16_GEOM"] = "32"   # check (1) is bit-for-bit: compare against the inference kernel of the same tile shape
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd"))
import nerf
from nerf import _ops, synthetic as syn
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 24
bad = 0
for it in range(n_cfg):
    width = int(rng.choice([128, 256])); depth = int(rng.integers(2, 9)); view = bool(rng.integers(0, 2))
    skip = int(rng.choice([2, 3, 4, 100]))
    if os.environ.get("FUZZ_FOCUS"):   # the shape family of the one unexplained mismatch
        width, depth, view, skip = 256, int(rng.integers(6, 9)), True, 4
    kw = dict(num_layers=depth, hidden_size=width, skip_connect_every=skip, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=view)
    n_rays = int(rng.integers(1, 90)); s = int(rng.integers(2, 40))
    sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(int(rng.integers(1, 1000)), sigma_gain=5.0, sigma_bias=0.0, **kw).items()}
    pts = torch.randn(n_rays, s, 3, device=dev)
    vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1)
    rays = torch.cat([torch.zeros(n_rays, 8, device=dev), vd], -1)
    g_up = torch.randn(n_rays, s, 4, device=dev)
    ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
    grads = {}
    msg, notes = [], []
    for prec in ("fp32", "bf16-s16"):
        nerf.set_precision(prec)
        m = nerf.models.FlexibleNeRFModel(**kw); m.load_state_dict(sd); m = m.to(dev)
        out = nerf.run_network(m, pts, rays, 1 << 20, ex, ed if view else None)
        (out * g_up).sum().backward()
        grads[prec] = {k: p.grad.detach().double().reshape(-1).cpu().numpy() for k, p in m.named_parameters()}
        with torch.no_grad():
            inf = _ops.run_network_pts(m.packed(), pts.reshape(-1, 3), vd if view else None, s)
        if not torch.equal(out.detach().reshape(-1, 4), inf):
            msg.append(f"{prec}: training forward != inference forward")
    # kernel-level consistency on the training buffers themselves (both precisions): every trunk stage's stored gradient
    # = (next stage's stored gradient) @ W masked by the saved activation pattern
    from nerf import _train
    for prec in ("fp32", "bf16-s16"):
        nerf.set_precision(prec)
        m = nerf.models.FlexibleNeRFModel(**kw); m.load_state_dict(sd); m = m.to(dev)
        pk = m.packed(); _ops.pack_backward(pk, [x.weight for x in m.linear_modules()])
        o2, act, masks = _ops.run_network_train(pk, pts.reshape(-1, 3), vd if view else None, s)
        npts = o2.shape[0]
        gbuf = _ops.mlp_backward_data(pk, g_up.reshape(-1, 4), masks, npts)
        slots, gslots, kh = _train._slots(m, pk.precision)
        rows = lambda which, buf, slot: _ops.mlp_unpack(pk, which, buf, npts, slot, width, 0, torch.empty((npts, width), device=dev))
        lowp = (lambda t: t.to(torch.bfloat16).float()) if prec == "bf16-s16" else (lambda t: t)
        tol = 2e-2 if prec == "bf16-s16" else 2e-5
        d_next = rows(1, gbuf, gslots["trunk0"] + (depth - 2) * kh) if depth >= 2 else None
        for i in range(depth - 2, -1, -1):
            d_x = d_next @ lowp(m.layers_xyz[i].weight.detach()[:, :width])
            if i > 0:
                expect = d_x * (rows(0, act, slots["trunk0"] + (i - 1) * kh) > 0)
                got = rows(1, gbuf, gslots["trunk0"] + (i - 1) * kh)
            else:
                expect, got = d_x, rows(1, gbuf, gslots["layer1"])
            err = float((got - expect).abs().max() / expect.abs().max().clamp_min(1e-30))
            if err > tol:
                msg.append(f"{prec} chain stage i={i}: rel err {err:.2e}")
            d_next = got
    # 8-bit saved tensors (the 48-point training kernels, networks dn_mlp_train_sizes accepts): forward == the 48-point inference
    # forward bit for bit; weight gradients from the 8-bit buffers against the bf16 kernel's on the bf16 buffers (cosine per tensor)
    from nerf import _hip
    nerf.set_precision("bf16-s16")
    m = nerf.models.FlexibleNeRFModel(**kw); m.load_state_dict(sd); m = m.to(dev)
    pk = m.packed(); _ops.pack_backward(pk, [x.weight for x in m.linear_modules()])
    if not _ops.s8_supported(pk):
        notes.append("no 8-bit kernels for this network")
        w8 = w16 = []
    else:
        _ops.pack_backward(pk, [x.weight for x in m.linear_modules()], _hip.PREC_BF16_S8)
        o16, act16, masks16 = _ops.run_network_train(pk, pts.reshape(-1, 3), vd if view else None, s)
        o8, act8, masks8 = _ops.run_network_train(pk, pts.reshape(-1, 3), vd if view else None, s, prec=_hip.PREC_BF16_S8)
        npts = o16.shape[0]
        g16 = _ops.mlp_backward_data(pk, g_up.reshape(-1, 4) * 1e-4, masks16, npts)
        g8 = _ops.mlp_backward_data(pk, g_up.reshape(-1, 4) * 1e-4, masks8, npts, prec=_hip.PREC_BF16_S8)
        os.environ.pop("DEXNERF_BF16_GEOM", None)
        with torch.no_grad():
            inf48 = _ops.run_network_pts(pk, pts.reshape(-1, 3), vd if view else None, s)
        os.environ["DEXNERF_BF16_GEOM"] = "32"
        if not torch.equal(o8, inf48):
            msg.append("bf16-s8: training forward != 48-point inference forward")
    shapes = [tuple(x.weight.shape) for x in m.linear_modules()]
    if _ops.s8_supported(pk):
        w16 = _ops.mlp_weight_grad_all(pk, act16, g16, npts, shapes)
        w8 = _ops.mlp_weight_grad_all(pk, act8, g8, npts, shapes, prec=_hip.PREC_BF16_S8)
    for (a_w, a_b), (b_w, b_b), shp in zip(w8, w16, shapes):
        for a, b, what in ((a_w, b_w, "dW"), (a_b, b_b, "db")):
            a, b = a.double().reshape(-1).cpu().numpy(), b.double().reshape(-1).cpu().numpy()
            if a.size == 1:
                continue   # (one number - fc_alpha's bias gradient, a sum over the points that may sit near zero: a cosine is only its sign)
            if np.linalg.norm(b) > 0:
                cos = float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))
                # (a few hundred points here: the 8-bit rounding of e5m2 x e4m3 products does not average out the way it does over the
                #  10^5 .. 10^6 points of a training step - 0.975 was seen for a 1 x 256 layer on 590 points)
                if cos < 0.95:
                    msg.append(f"bf16-s8 {what} {shp}: cos vs bf16 kernel {cos:.4f}")
    nerf.set_precision("fp32")
    mref = nerf.models.FlexibleNeRFModel(**kw); mref.load_state_dict(sd); mref = mref.to(dev)
    parts = [ex(pts.reshape(-1, 3))] + ([ed(vd[:, None, :].expand(n_rays, s, 3).reshape(-1, 3))] if view else [])
    (mref._forward_modules(torch.cat(parts, -1)).reshape(n_rays, s, 4) * g_up).sum().backward()
    for k, p in mref.named_parameters():
        ref = p.grad.detach().double().reshape(-1).cpu().numpy()
        scale = max(np.abs(ref).max(), 1e-30)
        e32 = np.abs(grads["fp32"][k] - ref).max() / scale
        if e32 > 1e-3:
            msg.append(f"fp32 grad {k}: rel err {e32:.2e}")
        a = grads["bf16-s16"][k]
        if np.linalg.norm(ref) > 1e-6 * np.sqrt(ref.size):
            cos = float(a @ ref / max(np.linalg.norm(a) * np.linalg.norm(ref), 1e-30))
            if cos < 0.9:
                msg.append(f"bf16 grad {k}: cos {cos:.3f}")
    if any(x.startswith("fp32 grad") for x in msg):
        # who is off? redo the reference in float64 on the CPU and measure both fp32 results against it
        m64 = nerf.models.FlexibleNeRFModel(**kw); m64.load_state_dict(sd); m64 = m64.double()
        emb64 = torch.cat(parts, -1).detach().cpu().double()
        (m64._forward_modules(emb64).reshape(n_rays, s, 4) * g_up.cpu().double()).sum().backward()
        worst_hip = worst_torch = 0.0
        for (k, p64), (_, p32) in zip(m64.named_parameters(), mref.named_parameters()):
            r = p64.grad.reshape(-1).numpy(); sc = max(np.abs(r).max(), 1e-30)
            worst_hip = max(worst_hip, np.abs(grads["fp32"][k] - r).max() / sc)
            worst_torch = max(worst_torch, np.abs(p32.grad.detach().double().reshape(-1).cpu().numpy() - r).max() / sc)
        # A ReLU unit whose pre-activation is within rounding of zero flips between two fp32 evaluation orders (and
        # float64): the gradient is discontinuous there, so 1e-3-level differences between ANY two of the three are
        # expected now and then in deep, wide nets.  The kernels' own consistency is what the chain check above pins; a
        # gradient mismatch with a clean chain check is therefore reported as a note, not a failure.
        note = f"relu-boundary note: vs float64 HIP fp32 {worst_hip:.1e}, torch fp32 {worst_torch:.1e}"
        if not any("chain stage" in x or "forward" in x for x in msg):
            msg = [x for x in msg if not x.startswith("fp32 grad")]
            notes.append(note)
        else:
            msg.append(note)
    tag = f"W{width} D{depth} skip{skip} view{int(view)} rays{n_rays}x{s}"
    print(("FAIL " if msg else "ok   ") + tag + ("  " + "; ".join(msg) if msg else "") + ("  (" + notes[0] + ")" if notes else ""), flush=True)
    bad += bool(msg)
print(f"{n_cfg - bad} / {n_cfg} configurations clean")
sys.exit(1 if bad else 0)
