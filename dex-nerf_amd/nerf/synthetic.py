"""Synthetic scenes and weights for parity tests and bench (no dataset ships with the reference).

Everything here is numpy-PCG64 seeded so the same arrays are produced in the golden-capture
container and on the GPU box, independent of torch's RNG or the host ISA.

Camera model follows the fork's convention (reference nerf_helpers.py:67-112): a world->camera
extrinsic E (OpenCV axes, +z forward) and a 3x3 intrinsic K.  Poses are the Blender-style
spherical poses of the reference loader (load_blender.py:11-38) converted with
E = inv(c2w @ diag(1,-1,-1,1)).
"""
import math

import numpy as np


def pose_spherical_c2w(theta_deg, phi_deg, radius):
    """Camera-to-world (OpenGL axes) on a sphere; same math as reference load_blender.py:11-38."""
    t = np.eye(4, dtype=np.float32)
    t[2, 3] = radius
    phi = phi_deg / 180.0 * np.pi
    rx = np.eye(4, dtype=np.float32)
    rx[1, 1] = rx[2, 2] = np.cos(phi)
    rx[1, 2] = -np.sin(phi)
    rx[2, 1] = np.sin(phi)
    th = theta_deg / 180.0 * np.pi
    ry = np.eye(4, dtype=np.float32)
    ry[0, 0] = ry[2, 2] = np.cos(th)
    ry[0, 2] = -np.sin(th)
    ry[2, 0] = np.sin(th)
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.float32)
    return (flip @ (ry @ (rx @ t))).astype(np.float32)


def extrinsic_from_c2w(c2w):
    """World->camera extrinsic in the fork's OpenCV convention."""
    m = c2w.astype(np.float64) @ np.diag([1.0, -1.0, -1.0, 1.0])
    return np.linalg.inv(m).astype(np.float32)


def intrinsic(height, width, camera_angle_x=0.6911112070083618):
    """Blender-style pinhole intrinsic (focal from the horizontal FoV, load_blender.py:75-77)."""
    focal = 0.5 * width / math.tan(0.5 * camera_angle_x)
    k = np.array([[focal, 0, width / 2.0], [0, focal, height / 2.0], [0, 0, 1]], dtype=np.float32)
    return k


def scene_pose(index, n_views=40, phi_deg=-30.0, radius=4.0):
    theta = np.linspace(-180.0, 180.0, n_views + 1)[:-1][index % n_views]
    return extrinsic_from_c2w(pose_spherical_c2w(float(theta), phi_deg, radius))


def flexible_param_shapes(num_layers=8, hidden_size=256, skip_connect_every=4,
                          num_encoding_fn_xyz=10, num_encoding_fn_dir=4,
                          include_input_xyz=True, include_input_dir=True, use_viewdirs=True):
    """(name, shape) list in the reference's parameter creation order (models.py:207-229)."""
    dim_xyz = (3 if include_input_xyz else 0) + 6 * num_encoding_fn_xyz
    dim_dir = ((3 if include_input_dir else 0) + 6 * num_encoding_fn_dir) if use_viewdirs else 0
    w = hidden_size
    out = [("layer1", (w, dim_xyz))]
    for i in range(num_layers - 1):
        wide = (i % skip_connect_every == 0) and i > 0 and i != num_layers - 1
        out.append((f"layers_xyz.{i}", (w, dim_xyz + w if wide else w)))
    if use_viewdirs:
        out.append(("layers_dir.0", (w // 2, dim_dir + w)))
        out.append(("fc_alpha", (1, w)))
        out.append(("fc_rgb", (3, w // 2)))
        out.append(("fc_feat", (w, w)))
    else:
        out.append(("fc_out", (4, w)))
    return out


def synth_state_dict(seed, sigma_gain=200.0, sigma_bias=-20.0, gain=2.45, freq_damping=0.8, **model_kwargs):
    """Deterministic nn.Linear-shaped weights (numpy float32 dict, `name.weight` / `name.bias`).

    Uniform(-b, b) with b = gain/sqrt(fan_in) (PyTorch's default is gain=1); gain=sqrt(6) keeps the
    activation variance constant through the ReLU layers so the field has structure.  The weight columns
    that read the positional encoding of frequency 2^f are damped by 2^(-freq_damping*f) - the spectral
    decay a trained NeRF has - otherwise the random field is chaotic at the 2^9 band and an fp32 ulp in a
    sample depth changes the rendered colour by percents (measured: fp32-vs-fp64 runs of the same
    reference code differ by 5e-2 without damping, 1e-5 with it).  The density head is scaled/shifted
    (`sigma_gain`, `sigma_bias`) so raw sigma is mostly negative (empty space) with peaks that cross the
    Dex thresholds (5..100) on many rays.
    """
    rng = np.random.default_rng(seed)
    sd = {}
    lx = model_kwargs.get("num_encoding_fn_xyz", 10)
    dim_xyz = (3 if model_kwargs.get("include_input_xyz", True) else 0) + 6 * lx
    off = dim_xyz - 6 * lx
    damp = np.ones(dim_xyz, np.float32)
    for f in range(lx):
        damp[off + 6 * f: off + 6 * f + 6] = 2.0 ** (-freq_damping * f)
    hidden = model_kwargs.get("hidden_size", 256)
    for name, (fan_out, fan_in) in flexible_param_shapes(**model_kwargs):
        b = gain / math.sqrt(fan_in)
        w = rng.uniform(-b, b, size=(fan_out, fan_in)).astype(np.float32)
        bias = rng.uniform(-b, b, size=(fan_out,)).astype(np.float32)
        if name == "layer1":
            w = (w * damp).astype(np.float32)
        elif name.startswith("layers_xyz.") and fan_in == hidden + dim_xyz:
            w[:, hidden:] = w[:, hidden:] * damp
        if name in ("fc_alpha",):
            w = (w * sigma_gain).astype(np.float32)
            bias = (bias * sigma_gain + sigma_bias).astype(np.float32)
        sd[name + ".weight"] = w
        sd[name + ".bias"] = bias
    return sd


def select_rays(height, width, count, seed=0):
    """Flat pixel indices of a reproducible ray subset."""
    return np.sort(np.random.default_rng(seed).choice(height * width, count, replace=False))
