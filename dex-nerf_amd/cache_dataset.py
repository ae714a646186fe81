#!/usr/bin/env python3
"""Build-owned counterpart of the reference's cache_dataset.py (:16-135): pre-compute ray bundles + targets of a Blender
scene and write them as `.data` files (torch.save dicts) that a training loop can read back instead of regenerating
rays every iteration.

    python dex-nerf_amd/cache_dataset.py --datapath data/lego --savedir cache/lego --num-random-rays 1024 --num-variations 4

Layout, as the reference writes it: <savedir>/train/<img:04d>.data with a stacked `ray_bundle` (2, ...) and `target`
(the reference overwrites the same file for every variation, :100-103 - kept: the last variation wins unless
--keep-variations, which appends _<j>), <savedir>/val/<img:04d>.data with `ray_origins` / `ray_directions` / `target`.
Only --type blender: the LLFF loader is not part of this build.  The upstream 4-argument camera-to-world convention of
get_ray_bundle is used, as in the reference script (:73).
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import nerf  # noqa: E402


def cache_nerf_dataset(args):
    if args.type != "blender":
        raise SystemExit("cache_dataset: only --type blender is supported by this build")
    images, poses, _, hwf, i_split = nerf.load_blender_data(args.datapath, half_res=args.blender_half_res, testskip=args.blender_stride)
    i_train, i_val, _ = i_split
    H, W, focal = int(hwf[0]), int(hwf[1]), hwf[2]
    device = "cuda" if torch.cuda.is_available() else "cpu"
    for split in ("train", "val", "test"):
        os.makedirs(os.path.join(args.savedir, split), exist_ok=True)
    np.random.seed(args.randomseed)
    written = []
    for img_idx in i_train:
        for j in range(args.num_variations):
            img_target = images[img_idx].to(device)
            ray_origins, ray_directions = nerf.get_ray_bundle(H, W, focal, poses[img_idx, :3, :4].to(device))
            if args.sample_all:
                target = img_target
            else:
                # the reference's coordinate grid enumerates pixels column-major: draw f -> pixel (f % H, f // H)
                f = torch.from_numpy(np.random.choice(H * W, size=(args.num_random_rays), replace=False))
                rows, cols = (f % H).to(device), (f // H).to(device)
                ray_origins, ray_directions = ray_origins[rows, cols, :], ray_directions[rows, cols, :]
                target = img_target[rows, cols, :]
            name = str(int(img_idx)).zfill(4) + (f"_{j}" if (args.keep_variations and j) else "") + ".data"
            path = os.path.join(args.savedir, "train", name)
            nerf.save_ray_cache(path, H, W, focal, ray_origins, ray_directions, target, train=True)
            written.append(path)
            if args.sample_all:
                break
    for img_idx in i_val:
        ray_origins, ray_directions = nerf.get_ray_bundle(H, W, focal, poses[img_idx, :3, :4].to(device))
        path = os.path.join(args.savedir, "val", str(int(img_idx)).zfill(4) + ".data")
        nerf.save_ray_cache(path, H, W, focal, ray_origins, ray_directions, images[img_idx].to(device), train=False)
        written.append(path)
    return written


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--datapath", required=True)
    ap.add_argument("--type", type=str.lower, default="blender", choices=["blender", "llff"])
    ap.add_argument("--blender-half-res", type=bool, default=True)
    ap.add_argument("--blender-stride", type=int, default=1)
    ap.add_argument("--savedir", required=True)
    ap.add_argument("--num-random-rays", type=int, default=8)
    ap.add_argument("--num-variations", type=int, default=1)
    ap.add_argument("--sample-all", action="store_true")
    ap.add_argument("--keep-variations", action="store_true", help="do not overwrite: write <img>_<j>.data for variations j > 0")
    ap.add_argument("--randomseed", type=int, default=3920)
    return cache_nerf_dataset(ap.parse_args(argv))


if __name__ == "__main__":
    main()
