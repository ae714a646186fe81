"""training PSNR at matched iteration counts, HIP kernels vs the CPU oracle.
Both learn the same synthetic teacher images (rendered once by the HIP path) from the same initial weights with the same
loss / optimizer / schedule; the ray draws come from each side's own generator, so the comparison is statistical.
Lives under tests/ because it uses oracle/ (test infrastructure): the product package never does."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/ -> repo root
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd"))
import nerf
from nerf import synthetic as syn
from oracle import nerf_oracle as oc

dev = torch.device("cuda:0")
torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 16)))
H = W = 32; V = 6; N = 256; NC = NF = 32; ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
kw = dict(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True)
mode = dict(chunksize=H * W, lindisp=False, num_coarse=NC, num_fine=NF, perturb=False, radiance_field_noise_std=0.0, white_background=False)
cfg = nerf.CfgNode(dict(dataset=dict(near=2.0, far=6.0, no_ndc=True), nerf=dict(use_viewdirs=True, train=dict(mode, perturb=True), validation=dict(mode))))
ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
k_mat = torch.from_numpy(syn.intrinsic(H, W)); poses = [torch.from_numpy(syn.scene_pose(i, n_views=V)) for i in range(V)]
nerf.set_precision("fp32")
teacher = []
for seed, bias in ((42, -150.0), (43, -20.0)):
    m = nerf.models.FlexibleNeRFModel(**kw); m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.synth_state_dict(seed, sigma_bias=bias, **kw).items()}); teacher.append(m.to(dev))
images = []
for p in poses:
    ro, rd = nerf.get_ray_bundle(H, W, float(k_mat[0, 0]), p.to(dev), k_mat.to(dev))
    with torch.no_grad():
        out = nerf.run_one_iter_of_nerf(H, W, 1.0, teacher[0], teacher[1], ro, rd, cfg, mode="validation", encode_position_fn=ex, encode_direction_fn=ed)
    images.append(out[3].reshape(-1, 3).cpu())
torch.manual_seed(1)
init = [nerf.models.FlexibleNeRFModel(**kw).state_dict() for _ in range(2)]
marks = [0, 50, 100, 150, 200, 250, ITERS - 1]

def lr_at(it): return 5e-4 * 0.1 ** (it / 250000)

# ---- HIP (fp32 parity mode and bf16) ----
curves = {}
for prec in ("fp32", "bf16"):
    nerf.set_precision(prec)
    models = []
    for sd in init:
        m = nerf.models.FlexibleNeRFModel(**kw); m.load_state_dict(sd); models.append(m.to(dev))
    opt = torch.optim.Adam([p for m in models for p in m.parameters()], lr=5e-4)
    sels = [nerf.RaySelector(H, W, p, k_mat, 2.0, 6.0, device=dev) for p in poses]
    gen = np.random.default_rng(0); curve = []
    for it in range(ITERS):
        v = int(gen.integers(V)); pix = torch.from_numpy(gen.choice(H * W, N, replace=False)).to(dev)
        rays, target = sels[v].select(pix, images[v].reshape(H, W, 3).to(dev))
        out = nerf.predict_and_render_radiance(rays, models[0], models[1], cfg, mode="train", encode_position_fn=ex, encode_direction_fn=ed)
        loss = nerf.img2mse(out[0], target) + nerf.img2mse(out[3], target)
        opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
        for g in opt.param_groups: g["lr"] = lr_at(it)
        if it in marks: curve.append(nerf.mse2psnr(loss.item()))
    curves["HIP " + prec] = curve
nerf.set_precision("fp32")
# ---- CPU oracle ----
mc = oc.ModelCfg(**kw); rcfg = oc.RenderCfg(num_coarse=NC, num_fine=NF, near=2.0, far=6.0, perturb=True, m_thres=())
sds = [oc.to_torch_sd({k: v.numpy() for k, v in sd.items()}, requires_grad=True) for sd in init]
opt = torch.optim.Adam([t for sd in sds for t in sd.values()], lr=5e-4)
gen = np.random.default_rng(0); curve = []; t0 = time.perf_counter()
for it in range(ITERS):
    v = int(gen.integers(V)); pix = gen.choice(H * W, N, replace=False)
    ro, rd = oc.get_ray_bundle(H, W, poses[v], k_mat)
    out = oc.run_one_iter(ro.reshape(-1, 3)[pix], rd.reshape(-1, 3)[pix], sds[0], sds[1], mc, mc, rcfg)
    loss = oc.nerf_loss(out, images[v][pix])
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
    for g in opt.param_groups: g["lr"] = lr_at(it)
    if it in marks: curve.append(oc.mse2psnr(loss.item()))
curves[f"CPU oracle ({time.perf_counter() - t0:.0f} s)"] = curve
print("| iteration | " + " | ".join(curves) + " |"); print("|---|" + "---|" * len(curves))
for i, it in enumerate(marks): print(f"| {it} | " + " | ".join(f"{c[i]:.2f}" for c in curves.values()) + " |")
