"""On-disk formats of the reference: Blender `transforms_*.json` scenes, the MessyTable /
Dex-NeRF layout, and the cached-ray `.data` files.  Host-side I/O only (numpy + PIL); nothing here touches the GPU.

The reference reads images with imageio and resizes with OpenCV (`cv2.resize`, INTER_AREA for images, INTER_NEAREST
for depth); neither package is in this image, so the two resamplers are restated here in numpy:
  * INTER_AREA on a shrink = area-weighted box average (exact 1/f^2 block mean for integer factors - the only case the
    non-debug paths use: /2 MessyTable, /4 Blender half_res);
  * INTER_NEAREST = source index floor(dst * scale).
Parity status: pose_spherical is pinned to a golden captured from the reference; the loaders are checked against
hand-built scenes (tests/test_host_side.py) but NOT against the reference's own loaders, which cannot run here
(cv2 / imageio absent) - "parity unpinned" for the resampling arithmetic.  The integer-factor INTER_AREA path follows OpenCV's
documented float rule in its own summation order (tests/test_datasets.py checks a hand-computed 4x4 -> 2x2 case whose fp32 result
depends on that order); it stays unpinned until fixtures produced by the reference's loaders exist.
"""
import json
import os
import pickle

import numpy as np
import torch
from PIL import Image


# ---- camera path (reference nerf/load_blender.py:11-38) ---------------------------------------------------------
def _rotation(axis, angle):
    """4x4 float32 rotation about x (axis 0) or y (axis 1), signs as the reference's two helpers."""
    i, j = ((1, 2), (0, 2))[axis]
    m = np.eye(4, dtype=np.float32)
    m[i, i] = m[j, j] = np.cos(angle)
    m[i, j] = -np.sin(angle)
    m[j, i] = -m[i, j]
    return m


_BLENDER_AXES = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]])   # int64: promotes the product to float64


def pose_spherical(theta, phi, radius):
    """Camera-to-world pose on a sphere: (azimuth deg, elevation deg, radius) -> 4x4 float64 (float32 factors times the
    reference's integer axis swap)."""
    lift = np.eye(4, dtype=np.float32)
    lift[2, 3] = radius
    return _BLENDER_AXES @ (_rotation(1, theta / 180 * np.pi) @ (_rotation(0, phi / 180.0 * np.pi) @ lift))


def _render_poses():
    return torch.stack([torch.from_numpy(pose_spherical(a, -30.0, 4.0)) for a in np.linspace(-180, 180, 40 + 1)[:-1]], 0)


# ---- resamplers -----------------------------------------------------------------------------------------------------
def _area_weights(src, dst):
    """(dst, src) matrix of the fraction of each source cell covered by each destination cell, rows normalised."""
    scale = src / dst
    w = np.zeros((dst, src), np.float64)
    for d in range(dst):
        lo, hi = d * scale, (d + 1) * scale
        for s in range(int(np.floor(lo)), min(int(np.ceil(hi)), src)):
            w[d, s] = max(0.0, min(hi, s + 1) - max(lo, s))
    return (w / w.sum(axis=1, keepdims=True)).astype(np.float32)


def resize_area(img, height, width):
    """cv2.resize(img, dsize=(width, height), interpolation=cv2.INTER_AREA) for a shrink, float32 HxW[xC]."""
    img = np.asarray(img, np.float32)
    h, w = img.shape[:2]
    if h % height == 0 and w % width == 0:
        # integer factors - the only case the reference's non-debug paths use (load_messytable.py:148-157: /2,
        # load_blender.py:107-119: /4), on float32 images already divided by 255 (:83 / :66).  OpenCV's area-fast path for
        # float data: the fy x fx block summed sequentially in float (rows outer, columns inner), times the float scale
        # 1 / (fx * fy) - restated in that order so that fp32 rounding follows it too (no uint8 rounding is involved:
        # saturate_cast<float> is the identity).
        fy, fx = h // height, w // width
        blocks = img.reshape((height, fy, width, fx) + img.shape[2:])
        acc = np.zeros((height, width) + img.shape[2:], np.float32)
        for dy in range(fy):
            for dx in range(fx):
                acc = acc + blocks[:, dy, :, dx]
        return acc * np.float32(1.0 / (fx * fy))
    out = np.tensordot(_area_weights(h, height), img, axes=(1, 0))
    return np.moveaxis(np.tensordot(_area_weights(w, width), out, axes=(1, 1)), 0, 1).astype(np.float32)


def resize_nearest(img, height, width):
    """cv2.resize(..., interpolation=cv2.INTER_NEAREST): source index floor(dst * src/dst)."""
    img = np.asarray(img)
    h, w = img.shape[:2]
    ys = np.minimum((np.arange(height) * (h / height)).astype(np.int64), h - 1)
    xs = np.minimum((np.arange(width) * (w / width)).astype(np.int64), w - 1)
    return img[ys][:, xs]


def _imread(path):
    return np.array(Image.open(path))


def _join_splits(per_split):
    """[(array per field) per split] -> concatenated fields + the index ranges of train / val / test."""
    sizes = [fields[0].shape[0] for fields in per_split]
    bounds = np.concatenate([[0], np.cumsum(sizes)])
    joined = [np.concatenate([fields[k] for fields in per_split], 0) for k in range(len(per_split[0]))]
    return joined, [np.arange(bounds[i], bounds[i + 1]) for i in range(len(per_split))]


def _shrink_all(maps, size, how):
    return torch.stack([torch.from_numpy(np.ascontiguousarray(how(m, *size))) for m in maps], 0)


_SPLITS = ("train", "val", "test")


# ---- Blender (reference nerf/load_blender.py:41-127) ------------------------------------------------------------
def load_blender_data(basedir, half_res=False, testskip=1, debug=False):
    """`transforms_{train,val,test}.json` + PNG frames -> (imgs (N,H,W,C) float32 in [0,1], poses (N,4,4) float32,
    render_poses (40,4,4), [H, W, focal], i_split).  val / test keep every `testskip`-th frame (0 = all).
    Fork quirks kept: `half_res` divides the size and the focal length by FOUR (:106-110); `debug` returns 25x25
    thumbnails with H, W, focal divided by 32 (:93-104)."""
    per_split, angle = [], None
    for name in _SPLITS:
        with open(os.path.join(basedir, f"transforms_{name}.json"), "r") as fp:
            meta = json.load(fp)
        stride = 1 if (name == "train" or testskip == 0) else testskip
        frames = meta["frames"][::stride]
        pixels = np.array([_imread(os.path.join(basedir, fr["file_path"] + ".png")) for fr in frames])
        per_split.append(((pixels / 255.0).astype(np.float32), np.array([fr["transform_matrix"] for fr in frames]).astype(np.float32)))
        angle = float(meta["camera_angle_x"])   # the reference reads it from the last split
    (imgs, poses), i_split = _join_splits(per_split)
    H, W = imgs[0].shape[:2]
    focal = 0.5 * W / np.tan(0.5 * angle)
    if debug:
        return _shrink_all(imgs, (25, 25), resize_area), torch.from_numpy(poses), _render_poses(), [H // 32, W // 32, focal / 32.0], i_split
    if half_res:
        H, W, focal = H // 4, W // 4, focal / 4.0
    return _shrink_all(imgs, (H, W), resize_area), torch.from_numpy(poses), _render_poses(), [H, W, focal], i_split


# ---- MessyTable / Dex-NeRF scenes (reference nerf/load_messytable.py:18-176) ------------------------------------
def load_messytable_data(basedir, half_res=False, testskip=1, debug=False, imgname="0128_irL_kuafu_half.png", is_real_rgb=False):
    """Layout: <basedir>/<split>/<scene>/{meta.pkl, <imgname>, depthL.png | depth.png}; meta.pkl holds the world->camera
    extrinsic and the 3x3 intrinsic (`extrinsic_l` / `intrinsic_l`, or `extrinsic` / `intrinsic` for real RGB); depth PNGs
    are millimetres; a grey (IR) image is replicated to three channels.
    Returns (imgs (N,H/2,W/2,3), poses (N,4,4), render_poses, [H/2, W/2, focal/4], i_split, intrinsics (N,3,3),
    depths (N,H/2,W/2) metres).  Fork quirks kept: the maps are ALWAYS halved (:148-165) while the returned focal length
    is divided by four; `half_res` divides the first two intrinsic rows by 4 and pins the principal point to
    (240, 135) (:70-75); `testskip` is accepted and unused; scenes are enumerated in os.listdir order."""
    depth_name, key_e, key_k = ("depth.png", "extrinsic", "intrinsic") if is_real_rgb else ("depthL.png", "extrinsic_l", "intrinsic_l")
    per_split, meta = [], None
    for name in _SPLITS:
        root = os.path.join(basedir, name)
        frames, extr, intr, depth = [], [], [], []
        for scene in os.listdir(root):
            with open(os.path.join(root, scene, "meta.pkl"), "rb") as f:
                meta = pickle.load(f)
            px = _imread(os.path.join(root, scene, imgname))
            frames.append(px if px.ndim == 3 else np.repeat(px[..., None], 3, axis=-1))
            depth.append(np.array(Image.open(os.path.join(root, scene, depth_name))) / 1000)
            extr.append(np.array(meta[key_e]))
            k = np.array(meta[key_k])
            if half_res:
                k[:2, :] = k[:2, :] / 4
                k[0, 2], k[1, 2] = 240., 135.
            intr.append(k)
        per_split.append(((np.array(frames) / 255.0).astype(np.float32), np.array(extr).astype(np.float32),
                          np.array(intr).astype(np.float32), np.array(depth).astype(np.float32)))
    (imgs, poses, intrinsics, depths), i_split = _join_splits(per_split)
    H, W = imgs[0].shape[:2]
    focal = meta[key_k][0, 0]
    if debug:
        hwf, size = [H // 32, W // 32, focal / 32.0], (25, 25)
    else:
        hwf, size = [H // 2, W // 2, focal / 4.0], (H // 2, W // 2)
    return (_shrink_all(imgs, size, resize_area), torch.from_numpy(poses), _render_poses(), hwf, i_split,
            torch.from_numpy(intrinsics), _shrink_all(depths, size, resize_nearest))


from .llff import load_llff_data  # noqa: E402,F401  (LLFF captures: nerf/llff.py)


# ---- cached rays (reference cache_dataset.py:62-135) ------------------------------------------------------------
def save_ray_cache(path, height, width, focal_length, ray_origins, ray_directions, target, train=False):
    """One `.data` file in the reference's two dialects: validation files keep `ray_origins` / `ray_directions`
    (cache_dataset.py:121-128), training files a stacked `ray_bundle` (2, ...) (:104-111)."""
    d = {"height": int(height), "width": int(width), "focal_length": focal_length, "target": target.detach().cpu()}
    if train:
        d["ray_bundle"] = torch.stack([ray_origins, ray_directions], dim=0).detach().cpu()
    else:
        d["ray_origins"] = ray_origins.detach().cpu()
        d["ray_directions"] = ray_directions.detach().cpu()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(d, path)


def load_ray_cache(path, device=None):
    """Reads either dialect; returns (height, width, focal_length, ray_origins, ray_directions, target)."""
    d = torch.load(path, map_location="cpu", weights_only=False)   # plain dicts written by save_ray_cache / cache_dataset.py (may hold numpy scalars)
    if "ray_bundle" in d:
        ro, rd = d["ray_bundle"][0], d["ray_bundle"][1]
    else:
        ro, rd = d["ray_origins"], d["ray_directions"]
    out = [ro, rd, d["target"]]
    if device is not None:
        out = [t.to(device) for t in out]
    return (d["height"], d["width"], d["focal_length"], *out)
