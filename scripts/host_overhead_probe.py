"""Developer probe: host-side enqueue time per training step (no GPU sync inside) for a net shape."""
import os, sys, time, cProfile, pstats
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd"))
import nerf
from nerf import synthetic as syn
import bench
dev = torch.device("cuda:0")
nerf.set_precision("bf16")
width, layers = int(sys.argv[1]), int(sys.argv[2])
kw = dict(num_layers=layers, hidden_size=width, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True)
models = []
for seed in (42, 43):
    m = nerf.models.FlexibleNeRFModel(**kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.synth_state_dict(seed, **kw).items()})
    models.append(m.to(dev))
_, cfg, ro, rd, ex, ed = bench.build_scene(dev, 0)
n_rays = 1024
cfg.nerf.train.perturb = True; cfg.nerf.train.radiance_field_noise_std = 0.2; cfg.nerf.train.chunksize = n_rays; cfg.nerf.train.num_fine = 64
params = list(models[0].parameters()) + list(models[1].parameters())
opt = torch.optim.Adam(params, lr=5e-4, fused=True)
image = torch.rand(bench.H, bench.W, 3, device=dev)
selector = nerf.RaySelector(bench.H, bench.W, torch.from_numpy(syn.scene_pose(7)), torch.from_numpy(syn.intrinsic(bench.H, bench.W)), 2.0, 6.0, device=dev)
def step():
    rays, target = selector.select(selector.random_pixels(n_rays), image)
    out = nerf.predict_and_render_radiance(rays, models[0], models[1], cfg, mode="train", encode_position_fn=ex,
                                           encode_direction_fn=ed, m_thres_cand=bench.M_THRES)
    loss = nerf.img2mse(out[0], target) + nerf.img2mse(out[3], target)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
t_host = (time.perf_counter() - t0) / 20
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 20
print(f"W={width} D={layers}: host enqueue {t_host*1e3:.2f} ms/step, incl. GPU drain {t_all*1e3:.2f} ms/step")
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
