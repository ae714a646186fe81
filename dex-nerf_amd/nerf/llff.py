"""LLFF (forward-facing / 360 real captures) scenes: `poses_bounds.npy` + an `images[_<factor>]` directory -> images, camera
poses with the intrinsics column, depth bounds, a render path and the held-out view (the behaviour
of reference nerf/load_llff.py:68-354, which the LLFF configs of train_nerf_*.py / eval_nerf.py reach through
`load_llff_data`).  Host-side numpy + PIL; nothing here touches the GPU.

What the format holds (LLFF's imgs2poses convention): one row of 17 numbers per view = a 3 x 5 block [R | t | (h, w, f)] whose
rotation columns are ordered (down, right, backwards), then the near / far depth of the view.  The loader
  1. reads the rows, takes h, w from the images it actually loads and divides f by the downsampling factor;
  2. reorders the rotation columns to (right, up, backwards) - NeRF's camera frame;
  3. scales translations and bounds so that the nearest bound sits at 1 / bd_factor;
  4. re-expresses every pose in the frame of the "average camera" (mean position, summed viewing axis, summed up axis);
  5. builds the render path: a two-turn spiral around the average camera, or, for inward-facing 360 captures (`spherify`), a
     circle on the sphere the cameras were rescaled to;
  6. names the view closest to the average position as the hold-out.

Differences from the reference, on purpose:
  * the reference shells out to ImageMagick (`mogrify -resize`) to create a missing `images_<factor>` directory
    (load_llff.py:12-65).  No subprocess here: a missing directory is produced in memory from `images/` with the area-average
    shrink of nerf/datasets.py (integer factors only).  Pixels therefore differ from ImageMagick's filter; a directory that
    exists is read as it is.
  * `path_zflat=True` halves the view count with integer division (the reference's `N_views /= 2` hands np.linspace a float,
    which current numpy refuses).
Parity status: UNPINNED - the reference module needs imageio, absent from this image, and the reference holds no fixtures for
it.  tests/test_datasets.py checks the geometry on hand-built captures through properties the construction guarantees (the
average camera of the recentred poses is the identity, spherified cameras sit on the unit sphere around the focus point, the
spiral's poses look at the focus depth) and against closed-form cases.
"""
import os

import numpy as np
from PIL import Image

_IMAGE_SUFFIXES = ("JPG", "jpg", "png")


def _unit(v):
    return v / np.linalg.norm(v)


def _frame(forward, up_hint, position):
    """3 x 4 camera-to-world block with third axis along `forward`, second as close to `up_hint` as orthogonality allows."""
    z = _unit(forward)
    x = _unit(np.cross(up_hint, z))
    y = _unit(np.cross(z, x))
    return np.stack([x, y, z, position], axis=1)


def average_pose(poses):
    """The "average camera" of (N, 3, 5) poses as a 3 x 5 block: mean position, summed third / second axes, the first view's
    intrinsics column (reference load_llff.py:159-169)."""
    centre = poses[:, :3, 3].mean(axis=0)
    block = _frame(poses[:, :3, 2].sum(axis=0), poses[:, :3, 1].sum(axis=0), centre)
    return np.concatenate([block, poses[0, :3, 4:5]], axis=1)


def _as_4x4(blocks):
    """(..., 3, 4) -> (..., 4, 4) with the homogeneous row."""
    blocks = np.asarray(blocks)
    row = np.broadcast_to(np.array([0.0, 0.0, 0.0, 1.0]), blocks.shape[:-2] + (1, 4))
    return np.concatenate([blocks, row], axis=-2)


def recentred(poses):
    """Poses expressed in the average camera's frame (reference load_llff.py:189-202): the average pose of the result is the
    identity.  Intrinsics column untouched."""
    out = np.array(poses, copy=True)
    to_avg = np.linalg.inv(_as_4x4(average_pose(poses)[:3, :4]))
    out[:, :3, :4] = (to_avg @ _as_4x4(poses[:, :3, :4]))[:, :3, :4]
    return out


def spiral_path(avg_pose, up, radii, focus_depth, z_rate, turns, n_views):
    """n_views poses on a spiral around `avg_pose` (3 x 5), each looking at the point `focus_depth` in front of the average
    camera (reference load_llff.py:172-186): position = avg frame applied to (cos t, -sin t, -sin(t z_rate)) * radii."""
    radii4 = np.append(np.asarray(radii, dtype=np.float64), 1.0)
    frame = avg_pose[:3, :4]
    focus = frame @ np.array([0.0, 0.0, -focus_depth, 1.0])
    out = []
    for t in np.linspace(0.0, 2.0 * np.pi * turns, n_views + 1)[:-1]:
        position = frame @ (np.array([np.cos(t), -np.sin(t), -np.sin(t * z_rate), 1.0]) * radii4)
        out.append(np.concatenate([_frame(position - focus, up, position), avg_pose[:, 4:5]], axis=1))
    return out


def spherified(poses, bounds, n_views=120):
    """Inward-facing captures (reference load_llff.py:205-287): move the origin to the point closest to all optical axes,
    align the third world axis with the mean camera offset, scale so that the RMS camera distance is 1, and return
    (poses, circular render path at the cameras' mean height, scaled bounds)."""
    axes = poses[:, :3, 2:3]
    origins = poses[:, :3, 3:4]
    # least-squares point nearest to the lines origin + s * axis: sum (I - a a^T) (x - o) = 0
    proj = np.eye(3) - axes * np.transpose(axes, (0, 2, 1))
    lhs = (np.transpose(proj, (0, 2, 1)) @ proj).mean(axis=0)
    rhs = (proj @ origins).mean(axis=0)
    focus = np.squeeze(np.linalg.inv(lhs) @ rhs)

    up = _unit((poses[:, :3, 3] - focus).mean(axis=0))
    side = _unit(np.cross([0.1, 0.2, 0.3], up))
    third = _unit(np.cross(up, side))
    world = np.stack([side, third, up, focus], axis=1)
    moved = np.linalg.inv(_as_4x4(world[None])) @ _as_4x4(poses[:, :3, :4])

    radius = np.sqrt(np.mean(np.sum(np.square(moved[:, :3, 3]), axis=-1)))
    scale = 1.0 / radius
    moved[:, :3, 3] *= scale
    bounds = bounds * scale
    radius = radius * scale

    height = moved[:, :3, 3].mean(axis=0)[2]
    ring = np.sqrt(radius ** 2 - height ** 2)
    path = []
    for t in np.linspace(0.0, 2.0 * np.pi, n_views):
        position = np.array([ring * np.cos(t), ring * np.sin(t), height])
        z = _unit(position)
        x = _unit(np.cross(z, np.array([0.0, 0.0, -1.0])))
        y = _unit(np.cross(z, x))
        path.append(np.stack([x, y, z, position], axis=1))
    path = np.stack(path, axis=0)
    hwf = poses[0, :3, 4:5]
    path = np.concatenate([path, np.broadcast_to(hwf, path[:, :3, :1].shape)], axis=-1)
    moved = np.concatenate([moved[:, :3, :4], np.broadcast_to(hwf, moved[:, :3, :1].shape)], axis=-1)
    return moved, path, bounds


def _image_files(directory):
    return [os.path.join(directory, f) for f in sorted(os.listdir(directory)) if f.endswith(_IMAGE_SUFFIXES)]


def _read_rgb(path):
    return np.asarray(Image.open(path).convert("RGB"), dtype=np.float64) / 255.0


def read_capture(basedir, factor=None):
    """(pose blocks (3, 5, N), bounds (2, N), images (H, W, 3, N)) as stored, h / w / f already those of the images returned
    (reference load_llff.py:68-135 with `factor`; the width= / height= forms are not used by any script)."""
    rows = np.load(os.path.join(basedir, "poses_bounds.npy"))
    if rows.ndim != 2 or rows.shape[1] != 17:
        raise ValueError(f"poses_bounds.npy: expected (N, 17) rows, got {rows.shape}")
    blocks = rows[:, :15].reshape(-1, 3, 5).transpose(1, 2, 0).copy()
    bounds = rows[:, 15:].transpose(1, 0).copy()
    factor = 1 if factor is None else factor
    directory = os.path.join(basedir, "images" if factor == 1 else f"images_{factor}")
    if os.path.isdir(directory):
        images = [_read_rgb(f) for f in _image_files(directory)]
    else:
        if int(factor) != factor or factor < 1:
            raise FileNotFoundError(f"{directory} does not exist and factor {factor} is not an integer shrink of images/")
        from .datasets import resize_area
        full = [_read_rgb(f) for f in _image_files(os.path.join(basedir, "images"))]
        images = [resize_area(im.astype(np.float32), im.shape[0] // int(factor), im.shape[1] // int(factor)).astype(np.float64)
                  for im in full]
    if len(images) != blocks.shape[-1]:
        raise ValueError(f"{directory}: {len(images)} images for {blocks.shape[-1]} poses")
    blocks[:2, 4, :] = np.array(images[0].shape[:2], dtype=np.float64).reshape(2, 1)
    blocks[2, 4, :] = blocks[2, 4, :] / factor
    return blocks, bounds, np.stack(images, axis=-1)


def load_llff_data(basedir, factor=8, recenter=True, bd_factor=0.75, spherify=False, path_zflat=False):   # noqa: A002 (reference keyword names)
    """-> (images (N, H, W, 3) f32, poses (N, 3, 5) f32 [R | t | h, w, f], bounds (N, 2) f32, render_poses (M, 3, 5) f32,
    i_test) - the reference's signature and return order (load_llff.py:290-354)."""
    blocks, bounds, images = read_capture(basedir, factor=factor)
    # (down, right, back) -> (right, up, back)
    blocks = np.concatenate([blocks[:, 1:2, :], -blocks[:, 0:1, :], blocks[:, 2:, :]], axis=1)
    poses = np.moveaxis(blocks, -1, 0).astype(np.float32)
    images = np.moveaxis(images, -1, 0).astype(np.float32)
    bounds = np.moveaxis(bounds, -1, 0).astype(np.float32)

    scale = 1.0 if bd_factor is None else 1.0 / (bounds.min() * bd_factor)
    poses[:, :3, 3] *= scale
    bounds = bounds * scale

    if recenter:
        poses = recentred(poses)
    if spherify:
        poses, render_poses, bounds = spherified(poses, bounds)
    else:
        avg = average_pose(poses)
        up = _unit(poses[:, :3, 1].sum(axis=0))
        near, far = bounds.min() * 0.9, bounds.max() * 5.0
        focus_depth = 1.0 / (0.25 / near + 0.75 / far)     # inverse-depth blend, 3/4 of the way to the far end
        radii = np.percentile(np.abs(poses[:, :3, 3]), 90, axis=0)
        n_views, turns = 120, 2
        if path_zflat:
            avg[:3, 3] = avg[:3, 3] + (-near * 0.1) * avg[:3, 2]
            radii[2] = 0.0
            n_views, turns = 60, 1
        render_poses = spiral_path(avg, up, radii, focus_depth, z_rate=0.5, turns=turns, n_views=n_views)
    render_poses = np.asarray(render_poses, dtype=np.float32)

    centre = average_pose(poses)[:3, 3]
    i_test = int(np.argmin(np.sum(np.square(centre - poses[:, :3, 3]), axis=-1)))
    return images.astype(np.float32), poses.astype(np.float32), bounds.astype(np.float32), render_poses, i_test
