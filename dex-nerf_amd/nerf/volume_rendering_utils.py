"""volume_render_radiance_field with the Dex-NeRF fixed-sigma depth list
(reference nerf/volume_rendering_utils.py:6-70) on the HIP compositing kernel."""
import torch

from . import _ops
from .nerf_helpers import _require_device


def _thresholds(m_thres_cand):
    # the fork iterates m_thres_cand unconditionally (TypeError on None, :53); None -> no Dex outputs
    return [] if m_thres_cand is None else [float(m) for m in m_thres_cand]


def volume_render_radiance_field(radiance_field, depth_values, ray_directions, radiance_field_noise_std=0.0,
                                 white_background=False, m_thres_cand=None):
    """Returns (rgb_map, disp_map, acc_map, weights, depth_map, *depth_map_dex[K]) like the reference.

    radiance_field (..., S, 4) raw [r,g,b,sigma]; depth_values (..., S); ray_directions (..., 3).
    Train-time noise is drawn with torch.randn on the device (reference :31-39).  Differentiable w.r.t.
    radiance_field (the Dex depths are gathers of grad-free depths, as in the reference).
    """
    _require_device(radiance_field, "volume_render_radiance_field")
    lead = depth_values.shape[:-1]
    s = depth_values.shape[-1]
    rf = radiance_field.reshape(-1, s, 4)
    z = depth_values.reshape(-1, s)
    rd = ray_directions.reshape(-1, 3)
    thres = _thresholds(m_thres_cand)
    noise = None
    std = float(radiance_field_noise_std)
    if std > 0.0:
        noise = torch.randn(z.shape, dtype=torch.float32, device=z.device)
    if torch.is_grad_enabled() and rf.requires_grad:
        outs = _ops.VolumeRenderFn.apply(rf, z, rd, noise, std, bool(white_background), thres)
        rgb, disp, acc, weights, depth = outs[:5]
        dex = outs[5] if len(outs) > 5 else None
    else:
        rgb, disp, acc, weights, depth, dex = _ops.volume_render_fwd(rf, z, rd, noise, std, bool(white_background), thres)
    out = [rgb.reshape(*lead, 3), disp.reshape(lead), acc.reshape(lead), weights.reshape(*lead, s), depth.reshape(lead)]
    if dex is not None:
        out += [dex[k].reshape(lead) for k in range(dex.shape[0])]
    return tuple(out)
