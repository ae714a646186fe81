"""the reference's on-disk formats, read by nerf/datasets.py.  Host-only tests on hand-built scenes
(the reference's own loaders need cv2 / imageio, absent here: only pose_spherical is pinned to a reference golden)."""
import json
import os
import pickle

import numpy as np
import pytest
import torch
from PIL import Image


def test_pose_spherical_matches_reference_golden(golden):
    import nerf
    g = golden("val_extras")
    out = np.stack([nerf.pose_spherical(*a) for a in g["pose_angles"]])
    assert out.dtype == g["pose_out"].dtype
    np.testing.assert_array_equal(out, g["pose_out"])


def test_resamplers():
    from nerf import datasets as D
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 1, size=(12, 16, 3)).astype(np.float32)
    half = D.resize_area(img, 6, 8)                       # integer factor: exact 2x2 block mean
    np.testing.assert_allclose(half, (img[0::2, 0::2] + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2]) / 4, rtol=1e-6)
    # non-integer shrink: area-weighted; a constant stays constant and the mean is preserved
    np.testing.assert_allclose(D.resize_area(np.full((10, 10), 0.25, np.float32), 4, 4), 0.25, rtol=1e-6)
    small = D.resize_area(img[..., 0], 5, 7)
    assert small.shape == (5, 7) and abs(float(small.mean()) - float(img[..., 0].mean())) < 1e-6
    depth = np.arange(12 * 16, dtype=np.float32).reshape(12, 16)
    np.testing.assert_array_equal(D.resize_nearest(depth, 6, 8), depth[0::2, 0::2])   # floor(dst * 2)
    np.testing.assert_array_equal(D.resize_nearest(depth, 4, 4), depth[[0, 3, 6, 9]][:, [0, 4, 8, 12]])


def test_inter_area_integer_factor_follows_opencv_float_rule():
    """cv2.resize(float32 image, INTER_AREA) at an integer factor (the reference's only non-debug use:
    load_messytable.py:148-157 /2, load_blender.py:107-119 /4, on images already divided by 255): the f x f block summed in
    float32, rows outer / columns inner, times float32(1 / f^2) - no 8-bit rounding anywhere (the inputs are float).
    A hand-computed 4x4 -> 2x2 case whose fp32 result depends on that order (1e8 + 1 - 1e8 is 0 in fp32, 1 in exact
    arithmetic), a .5 tie of 8-bit pixel values that must NOT be rounded, and the /4 path.  The resamplers stay "parity
    unpinned": the reference's loaders cannot run here (cv2 / imageio absent) and the reference holds no fixtures for them."""
    from nerf import datasets as D
    big = np.float32(1e8)
    img = np.array([[big, 1.0, 0.0, 0.0],
                    [-big, 0.0, 0.0, 2.0],
                    [3.0, 4.0, 10 / 255, 11 / 255],
                    [5.0, 6.0, 11 / 255, 11 / 255]], np.float32)
    out = D.resize_area(img, 2, 2)
    f = np.float32
    expect = np.array([[((f(big) + f(1.0)) + f(-big) + f(0.0)) * f(0.25), f(2.0) * f(0.25)],
                       [f(18.0) * f(0.25), (((f(10 / 255) + f(11 / 255)) + f(11 / 255)) + f(11 / 255)) * f(0.25)]], np.float32)
    assert float(expect[0, 0]) == 0.0            # the sequential fp32 sum loses the 1 (a pairwise / float64 sum would give 0.25)
    np.testing.assert_array_equal(out, expect)
    assert abs(float(out[1, 1]) * 255 - 10.75) < 1e-5   # an 8-bit .75 (or .5) mean stays fractional: float path
    img16 = np.arange(8 * 8, dtype=np.float32).reshape(8, 8) / 7
    out4 = D.resize_area(img16, 2, 2)
    acc = np.zeros((2, 2), np.float32)
    blocks = img16.reshape(2, 4, 2, 4)
    for dy in range(4):
        for dx in range(4):
            acc = acc + blocks[:, dy, :, dx]
    np.testing.assert_array_equal(out4, acc * np.float32(1 / 16))
    # three channels ride along unchanged
    rgb = np.stack([img, img * 2, img * 0.5], -1)
    np.testing.assert_array_equal(D.resize_area(rgb, 2, 2)[..., 0], expect)


def _write_blender(root, n_frames=(3, 2, 4), size=8):
    rng = np.random.default_rng(1)
    frames = {}
    for split, n in zip(("train", "val", "test"), n_frames):
        os.makedirs(os.path.join(root, split), exist_ok=True)
        meta = {"camera_angle_x": 0.6911112070083618, "frames": []}
        frames[split] = []
        for i in range(n):
            px = rng.integers(0, 256, size=(size, size, 4), dtype=np.uint8)
            Image.fromarray(px, "RGBA").save(os.path.join(root, split, f"r_{i}.png"))
            pose = np.eye(4)
            pose[:3, 3] = rng.normal(size=3)
            meta["frames"].append({"file_path": f"./{split}/r_{i}", "transform_matrix": pose.tolist()})
            frames[split].append((px, pose))
        with open(os.path.join(root, f"transforms_{split}.json"), "w") as fp:
            json.dump(meta, fp)
    return frames


def test_load_blender_data(tmp_path):
    import nerf
    frames = _write_blender(str(tmp_path))
    imgs, poses, render_poses, hwf, i_split = nerf.load_blender_data(str(tmp_path), half_res=False, testskip=2)
    # train keeps every frame, val / test every second one
    assert [len(s) for s in i_split] == [3, 1, 2] and imgs.shape == (6, 8, 8, 4) and imgs.dtype == torch.float32
    np.testing.assert_array_equal(imgs[0].numpy(), (frames["train"][0][0] / 255.0).astype(np.float32))
    np.testing.assert_array_equal(imgs[3].numpy(), (frames["val"][0][0] / 255.0).astype(np.float32))
    np.testing.assert_array_equal(imgs[5].numpy(), (frames["test"][2][0] / 255.0).astype(np.float32))
    np.testing.assert_array_equal(poses[4].numpy(), frames["test"][0][1].astype(np.float32))
    assert hwf[:2] == [8, 8] and abs(hwf[2] - 0.5 * 8 / np.tan(0.5 * 0.6911112070083618)) < 1e-9
    assert render_poses.shape == (40, 4, 4) and render_poses.dtype == torch.float64
    np.testing.assert_array_equal(render_poses[0].numpy(), nerf.pose_spherical(-180.0, -30.0, 4.0))
    # the fork's half_res is a QUARTER: size and focal / 4, box-averaged pixels
    q_imgs, _, _, q_hwf, _ = nerf.load_blender_data(str(tmp_path), half_res=True, testskip=0)
    assert q_imgs.shape == (9, 2, 2, 4) and q_hwf[:2] == [2, 2] and abs(q_hwf[2] - hwf[2] / 4) < 1e-12
    full = (frames["train"][1][0] / 255.0).astype(np.float32)
    np.testing.assert_allclose(q_imgs[1].numpy(), full.reshape(2, 4, 2, 4, 4).mean(axis=(1, 3)), rtol=1e-6)


def test_load_blender_debug_thumbnails(tmp_path):
    """debug=True: 25x25 INTER_AREA thumbnails (non-integer shrink) and H, W, focal / 32 (reference :93-104)."""
    import nerf
    _write_blender(str(tmp_path), n_frames=(1, 1, 1), size=64)
    imgs, _, _, hwf, _ = nerf.load_blender_data(str(tmp_path), debug=True)
    full, _, _, hwf_full, _ = nerf.load_blender_data(str(tmp_path))
    assert imgs.shape == (3, 25, 25, 4) and hwf[:2] == [2, 2] and abs(hwf[2] - hwf_full[2] / 32) < 1e-12
    assert abs(float(imgs[0].mean()) - float(full[0].mean())) < 1e-5   # area weighting preserves the mean


def _write_messytable(root, real_rgb=False):
    rng = np.random.default_rng(2)
    scenes = {}
    key_e, key_k, depth_n = ("extrinsic", "intrinsic", "depth.png") if real_rgb else ("extrinsic_l", "intrinsic_l", "depthL.png")
    for split, n in (("train", 2), ("val", 1), ("test", 1)):
        for i in range(n):
            d = os.path.join(root, split, f"{split}-{i}")
            os.makedirs(d)
            grey = rng.integers(0, 256, size=(12, 16), dtype=np.uint8)
            Image.fromarray(grey, "L").save(os.path.join(d, "0128_irL_kuafu_half.png"))
            depth_mm = rng.integers(300, 4000, size=(12, 16)).astype(np.uint16)
            Image.fromarray(depth_mm).save(os.path.join(d, depth_n))
            meta = {key_e: np.eye(4) + rng.normal(size=(4, 4)) * 0.01, key_k: np.array([[1400.0, 0, 960.0], [0, 1400.0, 540.0], [0, 0, 1.0]])}
            with open(os.path.join(d, "meta.pkl"), "wb") as f:
                pickle.dump(meta, f)
            scenes[(split, f"{split}-{i}")] = (grey, depth_mm, meta)
    return scenes


@pytest.mark.parametrize("real_rgb", [False, True])
def test_load_messytable_data(tmp_path, real_rgb):
    import nerf
    scenes = _write_messytable(str(tmp_path), real_rgb)
    imgs, poses, render_poses, hwf, i_split, intrinsics, depths = nerf.load_messytable_data(str(tmp_path), half_res=True, is_real_rgb=real_rgb)
    assert [len(s) for s in i_split] == [2, 1, 1]
    assert imgs.shape == (4, 6, 8, 3) and depths.shape == (4, 6, 8) and intrinsics.shape == (4, 3, 3) and poses.shape == (4, 4, 4)
    assert hwf[:2] == [6, 8] and hwf[2] == 1400.0 / 4        # always halved maps, focal / 4 (fork quirk)
    order = {"train": os.listdir(os.path.join(str(tmp_path), "train"))}
    grey, depth_mm, meta = scenes[("train", order["train"][0])]
    full = np.repeat((grey / 255.0).astype(np.float32)[..., None], 3, axis=-1)
    np.testing.assert_allclose(imgs[0].numpy(), full.reshape(6, 2, 8, 2, 3).mean(axis=(1, 3)), rtol=1e-6)
    np.testing.assert_array_equal(depths[0].numpy(), (depth_mm / 1000).astype(np.float32)[0::2, 0::2])   # mm -> m, nearest
    k = intrinsics[0].numpy()
    assert k[0, 0] == 350.0 and k[1, 1] == 350.0 and k[0, 2] == 240.0 and k[1, 2] == 135.0 and k[2, 2] == 1.0
    key_e = "extrinsic" if real_rgb else "extrinsic_l"
    np.testing.assert_array_equal(poses[0].numpy(), np.asarray(meta[key_e]).astype(np.float32))
    _, _, _, _, _, k_full, _ = nerf.load_messytable_data(str(tmp_path), half_res=False, is_real_rgb=real_rgb)
    assert k_full[0, 0, 0] == 1400.0 and k_full[0, 0, 2] == 960.0


def test_ray_cache_round_trip(tmp_path):
    import nerf
    ro, rd, tgt = torch.randn(5, 7, 3), torch.randn(5, 7, 3), torch.rand(5, 7, 4)
    for train in (False, True):
        path = os.path.join(str(tmp_path), "val" if not train else "train", "0003.data")
        nerf.save_ray_cache(path, 5, 7, 12.5, ro, rd, tgt, train=train)
        raw = torch.load(path)
        assert set(raw) == ({"height", "width", "focal_length", "target", "ray_bundle"} if train else
                            {"height", "width", "focal_length", "target", "ray_origins", "ray_directions"})
        h, w, f, ro2, rd2, t2 = nerf.load_ray_cache(path)
        assert (h, w, f) == (5, 7, 12.5) and torch.equal(ro2, ro) and torch.equal(rd2, rd) and torch.equal(t2, tgt)


def test_llff_loader_is_explicitly_absent():
    import nerf
    with pytest.raises(NotImplementedError):
        nerf.load_llff_data("/nonexistent")


def test_cache_dataset_cli(tmp_path):
    """The cache_dataset.py counterpart on a hand-built Blender scene: file names, dict dialects, ray subsets."""
    import sys
    import nerf
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dex-nerf_amd"))
    import cache_dataset
    src, dst = str(tmp_path / "scene"), str(tmp_path / "cache")
    os.makedirs(src)
    frames = _write_blender(src, n_frames=(2, 1, 1), size=16)
    if torch.cuda.is_available():
        pytest.skip("host-path test")
    written = cache_dataset.main(["--datapath", src, "--savedir", dst, "--num-random-rays", "5", "--num-variations", "2"])
    assert sorted(os.path.relpath(w, dst) for w in set(written)) == ["train/0000.data", "train/0001.data", "val/0002.data"]
    h, w, focal, ro, rd, target = nerf.load_ray_cache(os.path.join(dst, "train", "0001.data"))
    assert (h, w) == (4, 4) and ro.shape == (5, 3) and rd.shape == (5, 3) and target.shape == (5, 4)     # quarter res (fork quirk)
    h, w, focal, ro, rd, target = nerf.load_ray_cache(os.path.join(dst, "val", "0002.data"))
    assert ro.shape == (4, 4, 3) and target.shape == (4, 4, 4)
    full, _, _, _, _ = nerf.load_blender_data(src, half_res=True)
    assert torch.equal(target, full[2])
