"""Developer probe: render rate of the other BASELINE.json configurations (C3 Dex-NeRF scenes on the as-shipped 4x128
nets, C4 800x800 64+192, C5 128+256 fp32) next to the C2 bench workload.  Synthetic nets and poses."""
import os, sys, time
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd"))
import nerf
from nerf import synthetic as syn

dev = torch.device("cuda:0")
M_THRES = np.arange(5, 105, 5)

def macs(layers, width):
    skip = 63 if layers > 5 else 0   # skip_connect_every=4 fires at i=4 only when the trunk is that deep
    return 63 * width + (layers - 1) * width * width + skip * width + width * width + width + (width + 27) * (width // 2) + (width // 2) * 3

def run(tag, h, w, nc, nf, layers, width, prec, near=2.0, far=6.0, reps=5):
    kw = dict(num_layers=layers, hidden_size=width, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True)
    nerf.set_precision(prec)
    models = []
    for seed, bias in ((42, -150.0), (43, -20.0)):
        m = nerf.models.FlexibleNeRFModel(**kw)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.synth_state_dict(seed, sigma_bias=bias, **kw).items()})
        models.append(m.to(dev))
    mode = dict(chunksize=h * w, lindisp=False, num_coarse=nc, num_fine=nf, perturb=False, radiance_field_noise_std=0.0, white_background=False)
    cfg = nerf.CfgNode(dict(dataset=dict(near=near, far=far, no_ndc=True), nerf=dict(use_viewdirs=True, train=dict(mode), validation=dict(mode))))
    k_mat = torch.from_numpy(syn.intrinsic(h, w)).to(dev)
    e_mat = torch.from_numpy(syn.scene_pose(7)).to(dev)
    ro, rd = nerf.get_ray_bundle(h, w, float(k_mat[0, 0]), e_mat, k_mat)
    ex, ed = nerf.get_embedding_function(10, True, True), nerf.get_embedding_function(4, True, True)
    def render():
        with torch.no_grad():
            return nerf.run_one_iter_of_nerf(h, w, 1.0, models[0], models[1], ro, rd, cfg, mode="validation", encode_position_fn=ex,
                                             encode_direction_fn=ed, m_thres_cand=M_THRES)
    for _ in range(2): render()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): render()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    flop_ray = 2 * macs(layers, width) * (nc + nc + nf)
    print(f"{tag}: {h}x{w}, {nc}+{nf}, D{layers}/W{width}, {prec}: {dt * 1e3:.2f} ms/image  {h * w / dt / 1e6:.3f} M rays/s  "
          f"{h * w / dt * flop_ray / 1e12:.0f} TFLOP/s (unpadded)", flush=True)

run("C2", 400, 400, 64, 128, 8, 256, "bf16")
run("C3a", 270, 480, 64, 64, 4, 128, "bf16", 0.3, 4.0)
run("C3b", 270, 480, 128, 192, 4, 128, "bf16", 0.3, 4.0)
run("C3b-fp32", 270, 480, 128, 192, 4, 128, "fp32", 0.3, 4.0)
run("C4", 800, 800, 64, 192, 8, 256, "bf16", reps=3)
run("C5", 400, 400, 128, 256, 8, 256, "fp32", reps=2)
run("C5-bf16", 400, 400, 128, 256, 8, 256, "bf16", reps=3)
