"""Developer probe (GPU): the training kernels storing their saved tensors at 8 bits themselves (DN_PREC_BF16_S8) against
dn_mlp_convert_saved_s8 of the bf16 buffers (must agree bit for bit), the parameter gradients of a whole training step in the
'bf16-s8' mode against the bf16 and fp32 modes (cosine per tensor), and the step time of both."""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import nerf
from nerf import _hip, _ops, synthetic as syn
import bench

dev = torch.device("cuda:0")
S = 64
n_rays = 1000   # ragged: 64,000 points = 1,999.x tiles of 32


def direct_vs_converter(**kw):
    sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(43, sigma_bias=-2.0, **kw).items()}
    nerf.set_precision("bf16")
    m = nerf.models.FlexibleNeRFModel(**kw); m.load_state_dict(sd); m = m.to(dev)
    pk = m.packed()
    _ops.pack_backward(pk, [x.weight for x in m.linear_modules()])
    gen = torch.Generator(device="cpu").manual_seed(1)
    n = n_rays * S
    pts = (torch.rand(n, 3, generator=gen) * 2 - 1).to(dev)
    vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=gen), dim=-1).to(dev) if kw["use_viewdirs"] else None
    g_out = (torch.randn(n, 4, generator=gen) * 1e-4).to(dev)
    out, act, masks = _ops.run_network_train(pk, pts, vd, S)
    out8, act8, masks8 = _ops.run_network_train(pk, pts, vd, S, prec=_hip.PREC_BF16_S8)
    ref8 = _ops.convert_saved_s8(pk, 0, act, n)
    grads = _ops.mlp_backward_data(pk, g_out, masks, n)
    grads8 = _ops.mlp_backward_data(pk, g_out, masks8, n, prec=_hip.PREC_BF16_S8)
    gref8 = _ops.convert_saved_s8(pk, 1, grads, n)
    torch.cuda.synchronize()
    tag = f"W={kw['hidden_size']} D={kw['num_layers']} viewdirs={kw['use_viewdirs']}"
    print(f"{tag}: out equal {torch.equal(out, out8)}, masks equal {torch.equal(masks, masks8)}, "
          f"act bytes {act.numel()} -> {act8.numel()} equal-to-converter {torch.equal(act8, ref8)} ({int((act8 != ref8).sum())} differ), "
          f"grad bytes {grads.numel()} -> {grads8.numel()} equal-to-converter {torch.equal(grads8, gref8)} ({int((grads8 != gref8).sum())} differ)",
          flush=True)
    if not torch.equal(act8, ref8):
        units = act8.numel() // 1024
        bad = (act8 != ref8).view(units, 1024).any(1).nonzero().flatten()
        per_tile = (act8.numel() // 1024) // ((n + 31) // 32) if n else 0
        print("   first differing act units (unit index within its tile):", [int(b) % max(per_tile, 1) for b in bad[:16]], "units per tile", per_tile)
    if not torch.equal(grads8, gref8):
        units = grads8.numel() // 1024
        bad = (grads8 != gref8).view(units, 1024).any(1).nonzero().flatten()
        per_tile = units // ((n + 31) // 32)
        print("   first differing grad units:", [int(b) % max(per_tile, 1) for b in bad[:16]], "units per tile", per_tile)


direct_vs_converter(num_layers=8, hidden_size=256, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True)
direct_vs_converter(num_layers=4, hidden_size=128, skip_connect_every=3, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True)
direct_vs_converter(num_layers=8, hidden_size=256, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=False)


# ---- a whole training step in the three modes: gradients and time ------------------------------------------------------------
def step_grads(prec, n_rays=4096, time_it=True):
    nerf.set_precision(prec)
    models, cfg, ro, rd, ex, ed = bench.build_scene(dev, 0)
    cfg.nerf.train.perturb = True
    cfg.nerf.train.radiance_field_noise_std = 0.0
    cfg.nerf.train.chunksize = n_rays
    torch.manual_seed(5)
    image = torch.rand(bench.H, bench.W, 3, device=dev)
    selector = nerf.RaySelector(bench.H, bench.W, torch.from_numpy(syn.scene_pose(7)), torch.from_numpy(syn.intrinsic(bench.H, bench.W)), 2.0, 6.0, device=dev)
    pix = selector.random_pixels(n_rays)
    params = [p for m in models for p in m.parameters()]

    def fwd_bwd():
        rays, target = selector.select(pix, image)
        torch.manual_seed(11)
        out = nerf.predict_and_render_radiance(rays, models[0], models[1], cfg, mode="train", encode_position_fn=ex,
                                               encode_direction_fn=ed, m_thres_cand=bench.M_THRES)
        loss = nerf.img2mse(out[0], target) + nerf.img2mse(out[3], target)
        for p in params:
            p.grad = None
        loss.backward()
        return loss
    loss = fwd_bwd()
    grads = [p.grad.clone() for p in params]
    dt = None
    if time_it:
        for _ in range(3):
            fwd_bwd()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fwd_bwd()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
    names = [f"{t}.{n}" for t, m in zip(("coarse", "fine"), models) for n, _ in m.named_parameters()]
    return float(loss), grads, dt, names


def cos(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float(a @ b / (a.norm() * b.norm() + 1e-300))


res = {p: step_grads(p, time_it=(p != "fp32")) for p in ("fp32", "bf16", "bf16-s8")}
print(f"loss fp32 {res['fp32'][0]:.6f} bf16 {res['bf16'][0]:.6f} bf16-s8 {res['bf16-s8'][0]:.6f}")
print(f"forward + backward, 4096 rays x (64 + 192) points: bf16 {res['bf16'][2] * 1e3:.2f} ms, bf16-s8 {res['bf16-s8'][2] * 1e3:.2f} ms")
names = res["fp32"][3]
worst = [1.0, 1.0, 1.0]
for i, nm in enumerate(names):
    c = (cos(res["bf16"][1][i], res["fp32"][1][i]), cos(res["bf16-s8"][1][i], res["fp32"][1][i]), cos(res["bf16-s8"][1][i], res["bf16"][1][i]))
    worst = [min(a, b) for a, b in zip(worst, c)]
    if nm.endswith("weight"):
        print(f"{nm:32s} cos(bf16,fp32) {c[0]:.4f}  cos(s8,fp32) {c[1]:.4f}  cos(s8,bf16) {c[2]:.4f}")
print(f"worst over all {len(names)} tensors: cos(bf16,fp32) {worst[0]:.4f}  cos(s8,fp32) {worst[1]:.4f}  cos(s8,bf16) {worst[2]:.4f}")
allg = {p: torch.cat([g.reshape(-1) for g in res[p][1]]) for p in res}
print(f"whole gradient vector: cos(bf16,fp32) {cos(allg['bf16'], allg['fp32']):.5f}  cos(s8,fp32) {cos(allg['bf16-s8'], allg['fp32']):.5f}")
nerf.set_precision("fp32")
