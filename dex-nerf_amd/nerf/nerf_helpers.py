"""Ray / encoding / sampling helpers with the reference's names and signatures
(reference nerf/nerf_helpers.py).  Device tensors go to the HIP kernels; the few helpers that
tiny_nerf.py (BASELINE config 1, CPU plumbing) imports also accept host tensors."""
import math
from typing import Optional

import torch

from . import _ops


def img2mse(img_src, img_tgt):
    """Reference nerf_helpers.py:9-10."""
    return torch.nn.functional.mse_loss(img_src, img_tgt)


def mse2psnr(mse):
    """Reference nerf_helpers.py:13-17 (0 -> 1e-5 guard)."""
    if mse == 0:
        mse = 1e-5
    return -10.0 * math.log10(mse)


def get_minibatches(inputs: torch.Tensor, chunksize: Optional[int] = 1024 * 8):
    """Reference nerf_helpers.py:20-25."""
    return [inputs[i: i + chunksize] for i in range(0, inputs.shape[0], chunksize)]


def meshgrid_xy(tensor1: torch.Tensor, tensor2: torch.Tensor):
    """np.meshgrid(..., indexing='xy') (reference nerf_helpers.py:28-40)."""
    ii, jj = torch.meshgrid(tensor1, tensor2, indexing="ij")
    return ii.transpose(-1, -2), jj.transpose(-1, -2)


def cumprod_exclusive(tensor: torch.Tensor) -> torch.Tensor:
    """tf.math.cumprod(..., exclusive=True) along the last dim (reference nerf_helpers.py:43-64)."""
    cumprod = torch.cumprod(tensor, -1)
    cumprod = torch.roll(cumprod, 1, -1)
    cumprod[..., 0] = 1.0
    return cumprod


def get_ray_bundle(height: int, width: int, focal_length, tform_cam2world: torch.Tensor, intrinsic=None):
    """Ray origins/directions (H, W, 3) for every pixel (reference nerf_helpers.py:67-112).

    Fork convention (5 args): `tform_cam2world` is really a world->camera extrinsic E (OpenCV axes) and
    `intrinsic` a 3x3 K; dir = [(i-cx)/fx, (j-cy)/fx, 1], rd = inv(E[:3,:3]) dir, ro = inv(E)[:3,3].
    `intrinsic=None` (the stale 4-arg callers: eval_nerf.py:174, tiny_nerf.py:127, cache_dataset.py:73)
    selects the upstream pinhole camera-to-world convention - an extension with no oracle in the fork.
    The two 4x4/3x3 inverses are taken on the host (LAPACK, like the reference on CPU); the per-pixel
    work runs in the HIP kernel for device poses.
    """
    pose = tform_cam2world
    host = pose.detach().to("cpu", torch.float32)
    if intrinsic is not None:
        k = intrinsic.detach().to("cpu", torch.float32)
        rinv = torch.inverse(host[:3, :3])
        origin = torch.inverse(host)[:3, -1]
        fx, cx, cy = float(k[0, 0]), float(k[0, 2]), float(k[1, 2])
    else:
        # dirs = [(i-W/2)/f, -(j-H/2)/f, -1]; rd = dirs . c2w[:3,:3]^T : fold the two sign flips into the matrix
        rinv = host[:3, :3].clone()
        rinv[:, 1] = -rinv[:, 1]
        rinv[:, 2] = -rinv[:, 2]
        origin = host[:3, -1]
        fx, cx, cy = float(focal_length), width * 0.5, height * 0.5
    if pose.is_cuda:
        return _ops.ray_bundle(height, width, rinv.reshape(-1).tolist(), origin.tolist(), fx, cx, cy, pose.device)
    ii, jj = meshgrid_xy(torch.arange(width, dtype=torch.float32), torch.arange(height, dtype=torch.float32))
    directions = torch.stack([(ii - cx) / fx, (jj - cy) / fx, torch.ones_like(ii)], dim=-1)
    ray_directions = torch.sum(directions[..., None, :] * rinv, dim=-1)
    ray_origins = origin.expand(ray_directions.shape)
    return ray_origins, ray_directions


def _frequency_bands(num_encoding_functions, log_sampling, dtype, device):
    if log_sampling:
        return 2.0 ** torch.linspace(0.0, num_encoding_functions - 1, num_encoding_functions, dtype=dtype, device=device)
    return torch.linspace(2.0 ** 0.0, 2.0 ** (num_encoding_functions - 1), num_encoding_functions, dtype=dtype,
                          device=device)


def positional_encoding(tensor, num_encoding_functions=6, include_input=True, log_sampling=True) -> torch.Tensor:
    """[x, sin(f0 x), cos(f0 x), ...] (reference nerf_helpers.py:115-159)."""
    if num_encoding_functions == 0 and include_input:
        return tensor
    if tensor.is_cuda:
        return _ops.positional_encoding(tensor, num_encoding_functions, include_input, log_sampling)
    encoding = [tensor] if include_input else []
    for freq in _frequency_bands(num_encoding_functions, log_sampling, tensor.dtype, tensor.device):
        encoding.append(torch.sin(tensor * freq))
        encoding.append(torch.cos(tensor * freq))
    return encoding[0] if len(encoding) == 1 else torch.cat(encoding, dim=-1)


class Embedder:
    """Callable returned by get_embedding_function.  The reference returns an opaque lambda
    (nerf_helpers.py:167-169); this object behaves the same when called but carries its parameters so
    run_network can fuse the encoding into the MLP kernel."""

    def __init__(self, num_encoding_functions=6, include_input=True, log_sampling=True):
        self.num_encoding_functions = int(num_encoding_functions)
        self.include_input = bool(include_input)
        self.log_sampling = bool(log_sampling)

    def __call__(self, x):
        return positional_encoding(x, self.num_encoding_functions, self.include_input, self.log_sampling)


class RaySelector:
    """Training-ray selection on the device (reference train_dexnerf_rgb.py:229-242 builds the
    full-image bundle, a coordinate grid and three gathers per step, then run_one_iter_of_nerf normalises and packs).

    One camera = one object: the two matrix inverses are taken once on the host (like get_ray_bundle), every
    `select` is a single kernel that writes the packed (N,11) ray rows [ro, rd, near, far, viewdir] and the target
    pixels.  `pixel_index` is a device int64 tensor of row-major indices h*W + w; `from_reference_choice` converts
    the reference's `np.random.choice(H*W)` draws (its coordinate grid enumerates pixels column-major: f = w*H + h).
    """

    def __init__(self, height, width, extrinsic, intrinsic, near, far, device=None):
        host = extrinsic.detach().to("cpu", torch.float32)
        k = intrinsic.detach().to("cpu", torch.float32)
        self.height, self.width = int(height), int(width)
        self.rinv = torch.inverse(host[:3, :3]).reshape(-1).tolist()
        self.origin = torch.inverse(host)[:3, -1].tolist()
        self.fx, self.cx, self.cy = float(k[0, 0]), float(k[0, 2]), float(k[1, 2])
        self.near, self.far = float(near), float(far)
        self.device = torch.device(device) if device is not None else extrinsic.device

    def from_reference_choice(self, select_inds):
        f = torch.as_tensor(select_inds, dtype=torch.int64)
        return ((f % self.height) * self.width + f // self.height).to(self.device)

    def random_pixels(self, n, generator=None):
        """n distinct pixels, drawn on the device (no host round trip)."""
        total = self.height * self.width
        return torch.randperm(total, device=self.device, generator=generator)[:min(n, total)]

    def select(self, pixel_index, image=None):
        return _ops.select_rays(self.height, self.width, self.rinv, self.origin, self.fx, self.cx, self.cy, self.near, self.far,
                                pixel_index, image)


class MultiViewRaySelector:
    """RaySelector for a set of training cameras whose index is a DEVICE scalar: the camera records (two host inverses per
    view, once) and the stacked images live on the device, so `select` contains no per-view host constant and a captured
    HIP graph of the whole training iteration can be replayed for any view (`view.fill_(k)` before the replay)."""

    def __init__(self, height, width, extrinsics, intrinsics, near, far, images=None, device=None):
        self.height, self.width = int(height), int(width)
        self.near, self.far = float(near), float(far)
        self.device = torch.device(device) if device is not None else extrinsics[0].device
        recs = []
        for k, e in enumerate(extrinsics):
            host = e.detach().to("cpu", torch.float32)
            kmat = (intrinsics[k] if (torch.is_tensor(intrinsics) and intrinsics.dim() == 3) or isinstance(intrinsics, (list, tuple))
                    else intrinsics).detach().to("cpu", torch.float32)
            rec = torch.zeros(16)
            rec[:9] = torch.inverse(host[:3, :3]).reshape(-1)
            rec[9:12] = torch.inverse(host)[:3, -1]
            rec[12], rec[13], rec[14] = kmat[0, 0], kmat[0, 2], kmat[1, 2]
            recs.append(rec)
        self.cams = torch.stack(recs).to(self.device).contiguous()
        self.images = None if images is None else images.to(self.device, torch.float32).contiguous()   # (V, H, W, C)
        self.view = torch.zeros((), dtype=torch.int32, device=self.device)

    def random_pixels(self, n, generator=None):
        total = self.height * self.width
        return torch.randperm(total, device=self.device, generator=generator)[:min(n, total)]

    def select(self, pixel_index, view=None):
        """Rows + target pixels for the view held in `view` (default: self.view, set with `self.view.fill_(k)`)."""
        return _ops.select_rays_indirect(self.height, self.width, self.cams, self.view if view is None else view, self.near,
                                         self.far, pixel_index, self.images)


def get_embedding_function(num_encoding_functions=6, include_input=True, log_sampling=True):
    """Reference nerf_helpers.py:162-169."""
    return Embedder(num_encoding_functions, include_input, log_sampling)


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """Forward-facing NDC warp (reference nerf_helpers.py:172-199): one HIP kernel for device rays, the reference's
    elementwise composition for host tensors."""
    if rays_o.is_cuda and not (rays_o.requires_grad or rays_d.requires_grad):
        o, d = _ops.ndc_rays(H, W, float(focal), float(near), rays_o.reshape(-1, 3), rays_d.reshape(-1, 3))
        return o.reshape(rays_o.shape), d.reshape(rays_d.shape)
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    rays_o = rays_o + t[..., None] * rays_d
    o0 = -1.0 / (W / (2.0 * focal)) * rays_o[..., 0] / rays_o[..., 2]
    o1 = -1.0 / (H / (2.0 * focal)) * rays_o[..., 1] / rays_o[..., 2]
    o2 = 1.0 + 2.0 * near / rays_o[..., 2]
    d0 = -1.0 / (W / (2.0 * focal)) * (rays_d[..., 0] / rays_d[..., 2] - rays_o[..., 0] / rays_o[..., 2])
    d1 = -1.0 / (H / (2.0 * focal)) * (rays_d[..., 1] / rays_d[..., 2] - rays_o[..., 1] / rays_o[..., 2])
    d2 = -2.0 * near / rays_o[..., 2]
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)


def _require_device(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: this build runs the hot path on the ROCm device only (got a host tensor); "
                           "move the inputs to 'cuda'")


def sample_pdf_2(bins, weights, num_samples, det=False):
    """Inverse-CDF sampling (reference nerf_helpers.py:262-304 + torchsearchsorted.searchsorted side='right').
    `det=False` draws u with torch.rand on the device, like the reference."""
    _require_device(bins, "sample_pdf")
    lead = bins.shape[:-1]
    b2 = bins.reshape(-1, bins.shape[-1])
    w2 = weights.reshape(-1, weights.shape[-1])
    u = None
    if not det:
        u = torch.rand(list(b2.shape[:-1]) + [num_samples], dtype=torch.float32, device=bins.device)
    out = _ops.sample_pdf(b2, w2, num_samples, u)
    return out.reshape(*lead, num_samples)


sample_pdf = sample_pdf_2  # the legacy sample_pdf (nerf_helpers.py:224-259) computes the same values
