#!/usr/bin/env python3
"""Build-owned counterpart of the reference's eval_nerf.py (:116-206): load a checkpoint (the reference's dict format:
model_coarse_state_dict / model_fine_state_dict, optional height / width / focal_length), render a sweep of spherical
poses in validation mode, optionally save RGB / disparity-style depth PNGs, and print the average time per image.

    python dex-nerf_amd/eval_nerf.py --checkpoint ckpt.ckpt --size 400 --views 8 --precision fp16 --savedir out/

The network shape (layers / width) is read off the checkpoint tensors, so the reference's shipped 4x128
`pretrained/*/checkpoint*.ckpt` files and checkpoints written by train_dexnerf.py both load unchanged.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import nerf  # noqa: E402
from nerf import synthetic as syn  # noqa: E402


def model_from_state_dict(sd, dev):
    """FlexibleNeRFModel whose constructor arguments are inferred from the tensor shapes (reference key names)."""
    width = sd["layer1.weight"].shape[0]
    dim_xyz = sd["layer1.weight"].shape[1]
    n_trunk = len([k for k in sd if k.startswith("layers_xyz.") and k.endswith(".weight")])
    wide = [i for i in range(n_trunk) if sd[f"layers_xyz.{i}.weight"].shape[1] != width]
    use_viewdirs = "fc_rgb.weight" in sd
    dim_dir = sd["layers_dir.0.weight"].shape[1] - width if use_viewdirs else 0
    skip = wide[0] if wide else n_trunk + 1  # a skip that never fires reproduces a skip-free checkpoint
    m = nerf.models.FlexibleNeRFModel(num_layers=n_trunk + 1, hidden_size=width, skip_connect_every=max(skip, 1),
                                      num_encoding_fn_xyz=(dim_xyz - 3) // 6, num_encoding_fn_dir=max((dim_dir - 3) // 6, 0),
                                      use_viewdirs=use_viewdirs)
    m.load_state_dict(sd)
    return m.to(dev)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoint", required=True)
    ap.add_argument("--size", type=int, default=0, help="image size (default: checkpoint's height, else 100)")
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--num-coarse", type=int, default=64)
    ap.add_argument("--num-fine", type=int, default=128)
    ap.add_argument("--near", type=float, default=2.0)
    ap.add_argument("--far", type=float, default=6.0)
    ap.add_argument("--white-background", action="store_true")
    ap.add_argument("--m-thres", type=int, default=0, help="> 0: also produce the Dex depth maps for thresholds 5..m")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--savedir", default="")
    ap.add_argument("--quiet", action="store_true")
    args = ap.parse_args(argv)

    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    nerf.set_precision(args.precision)
    ck = torch.load(args.checkpoint, map_location="cpu")
    coarse = model_from_state_dict(ck["model_coarse_state_dict"], dev)
    fine = model_from_state_dict(ck["model_fine_state_dict"], dev) if ck.get("model_fine_state_dict") else None
    size = args.size or int(ck.get("height", 100))
    mode = dict(chunksize=size * size, lindisp=False, num_coarse=args.num_coarse, num_fine=args.num_fine if fine else 0,
                perturb=False, radiance_field_noise_std=0.0, white_background=args.white_background)
    cfg = nerf.CfgNode(dict(dataset=dict(near=args.near, far=args.far, no_ndc=True),
                            nerf=dict(use_viewdirs=coarse.use_viewdirs, train=dict(mode), validation=dict(mode))))
    ex = nerf.get_embedding_function(coarse.num_encoding_fn_xyz, True, True)
    ed = nerf.get_embedding_function(coarse.num_encoding_fn_dir, True, True) if coarse.use_viewdirs else None
    thres = np.arange(5, args.m_thres + 5, 5) if args.m_thres > 0 else None
    k_mat = torch.from_numpy(syn.intrinsic(size, size)).to(dev)
    if "focal_length" in ck:
        k_mat[0, 0] = k_mat[1, 1] = float(ck["focal_length"])
    if args.savedir:
        os.makedirs(args.savedir, exist_ok=True)
    times, frames = [], []
    for i in range(args.views):
        pose = torch.from_numpy(syn.scene_pose(i, n_views=args.views)).to(dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            ro, rd = nerf.get_ray_bundle(size, size, float(k_mat[0, 0]), pose, k_mat)
            out = nerf.run_one_iter_of_nerf(size, size, float(k_mat[0, 0]), coarse, fine, ro, rd, cfg, mode="validation",
                                            encode_position_fn=ex, encode_direction_fn=ed, m_thres_cand=thres)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        rgb = out[3] if fine is not None else out[0]
        depth = out[4] if fine is not None else out[1]
        frames.append((rgb, depth))
        if args.savedir:
            from PIL import Image
            img = (rgb.clamp(0, 1) * 255).byte().cpu().numpy()
            Image.fromarray(img).save(os.path.join(args.savedir, f"{i:04d}.png"))
            d = depth.cpu().numpy()
            d = (255 * (d - d.min()) / max(d.max() - d.min(), 1e-8)).astype(np.uint8)
            Image.fromarray(d).save(os.path.join(args.savedir, f"depth_{i:04d}.png"))
    avg = float(np.mean(times[1:])) if len(times) > 1 else float(times[0])
    if not args.quiet:
        print(f"Avg time per image: {avg:.4f} s ({size}x{size}, {size * size / avg:.0f} rays/s, {args.precision})", flush=True)
    return dict(avg_seconds=avg, frames=frames, times=times)


if __name__ == "__main__":
    main()
