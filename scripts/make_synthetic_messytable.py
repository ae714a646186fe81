"""Write a scene directory in the reference's MessyTable / Dex-NeRF layout (nerf/load_messytable.py) from a synthetic
"teacher" radiance field rendered on the GPU: <root>/<split>/<scene>/{meta.pkl, 0128_irL_kuafu_half.png, depthL.png}.

Sizes follow the fork's conventions: PNGs are 540 x 960 ("_half"), meta.pkl holds the 1080 x 1920 intrinsic and the
world->camera extrinsic, depth PNGs are uint16 millimetres; the loader halves the maps to 270 x 480 and (half_res)
divides the intrinsic by 4 with the principal point at (240, 135).

    python scripts/make_synthetic_messytable.py OUT_DIR [n_train]
"""
import os
import pickle
import sys

import numpy as np
import torch
from PIL import Image

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd"))
import nerf  # noqa: E402
from nerf import synthetic as syn  # noqa: E402


def write_scene(root, n_train=6, n_val=1, n_test=1, near=0.3, far=1.3, radius=0.8, focal_full=1600.0, device="cuda:0"):
    dev = torch.device(device)
    kw = dict(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True)
    nets = []
    for seed, bias in ((42, -150.0), (43, -20.0)):
        m = nerf.models.FlexibleNeRFModel(**kw)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.synth_state_dict(seed, sigma_bias=bias, **kw).items()})
        nets.append(m.to(dev))
    h, w = 540, 960
    k_full = np.array([[focal_full, 0, 960.0], [0, focal_full, 540.0], [0, 0, 1.0]])
    k_png = torch.tensor([[focal_full / 2, 0, w / 2.0], [0, focal_full / 2, h / 2.0], [0, 0, 1.0]], dtype=torch.float32, device=dev)
    mode = dict(chunksize=h * w, lindisp=False, num_coarse=64, num_fine=64, perturb=False, radiance_field_noise_std=0.0, white_background=False)
    cfg = nerf.CfgNode(dict(dataset=dict(near=near, far=far, no_ndc=True), nerf=dict(use_viewdirs=True, train=dict(mode), validation=dict(mode))))
    ex, ed = nerf.get_embedding_function(10, True, True), nerf.get_embedding_function(4, True, True)
    total = n_train + n_val + n_test
    k = 0
    nerf.set_precision("fp16")
    try:
        for split, count in (("train", n_train), ("val", n_val), ("test", n_test)):
            for i in range(count):
                pose = syn.scene_pose(k, n_views=total, radius=radius)
                e_mat = torch.from_numpy(pose).to(dev)
                ro, rd = nerf.get_ray_bundle(h, w, float(k_png[0, 0]), e_mat, k_png)
                with torch.no_grad():
                    out = nerf.run_one_iter_of_nerf(h, w, float(k_png[0, 0]), nets[0], nets[1], ro, rd, cfg, mode="validation",
                                                    encode_position_fn=ex, encode_direction_fn=ed)
                d = os.path.join(root, split, f"scene-{k:03d}")
                os.makedirs(d, exist_ok=True)
                rgb = (out[3].clamp(0, 1) * 255).round().byte().cpu().numpy()
                Image.fromarray(rgb, "RGB").save(os.path.join(d, "0128_irL_kuafu_half.png"))
                depth_mm = (out[4].clamp(0, 65.0) * 1000).round().cpu().numpy().astype(np.uint16)
                Image.fromarray(depth_mm).save(os.path.join(d, "depthL.png"))
                with open(os.path.join(d, "meta.pkl"), "wb") as f:
                    pickle.dump({"extrinsic_l": pose.astype(np.float64), "intrinsic_l": k_full.copy()}, f)
                k += 1
    finally:
        nerf.set_precision("fp32")
    return dict(near=near, far=far, views=total)


if __name__ == "__main__":
    info = write_scene(sys.argv[1], n_train=int(sys.argv[2]) if len(sys.argv) > 2 else 6)
    print("wrote", sys.argv[1], info)
