"""Developer probe: time one training iteration (forward + backward + Adam) of the C2 nets on random rays."""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd"))
import nerf
import bench

dev = torch.device("cuda:0")
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
for prec in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["bf16", "fp32"]):
    nerf.set_precision(prec)
    models, cfg, ro, rd, ex, ed = bench.build_scene(dev, 0)
    cfg.nerf.train.perturb = True
    cfg.nerf.train.radiance_field_noise_std = 0.2
    cfg.nerf.train.chunksize = n_rays
    params = list(models[0].parameters()) + list(models[1].parameters())
    opt = torch.optim.Adam(params, lr=5e-4, fused=(os.environ.get("ADAM_FUSED", "1") == "1"))
    ro_f, rd_f = ro.reshape(-1, 3), rd.reshape(-1, 3)
    target_img = torch.rand(bench.H * bench.W, 3, device=dev)

    def step():
        sel = torch.randint(0, bench.H * bench.W, (n_rays,), device=dev)
        out = nerf.run_one_iter_of_nerf(bench.H, bench.W, 1.0, models[0], models[1], ro_f[sel], rd_f[sel], cfg, mode="train",
                                        encode_position_fn=ex, encode_direction_fn=ed, m_thres_cand=bench.M_THRES)
        loss = nerf.img2mse(out[0], target_img[sel]) + nerf.img2mse(out[3], target_img[sel])
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 10
    for _ in range(k):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k
    print(f"{prec}: {n_rays} rays/step, {dt*1e3:.2f} ms/step, {n_rays/dt:.0f} rays/s (train), loss {loss.item():.4f}", flush=True)
