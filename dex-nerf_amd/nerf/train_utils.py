"""run_network / predict_and_render_radiance / run_one_iter_of_nerf with the reference's call surface
(reference nerf/train_utils.py:72-288), plus the Dex depth-error helpers (:9-70)."""
import os

import numpy as np
import torch

from . import _ops
from ._train import inputs_need_grad, needs_grad, render_rays_train, run_network_fused, run_network_fused_rays, train_fused_ok
from .models import FlexibleNeRFModel
from .nerf_helpers import Embedder, _require_device, get_minibatches, ndc_rays
from .nerf_helpers import sample_pdf_2 as sample_pdf  # noqa: F401  (reference train_utils.py:6 alias)
from .volume_rendering_utils import _thresholds, volume_render_radiance_field


# ---- Dex depth metrics (reference train_utils.py:9-70; validation-time logging, numpy/torch host code) ----
def _err_dict(row):
    n = row[4]
    return {"depth_abs_err": row[0] / n, "depth_err2": row[1] / n, "depth_err4": row[2] / n, "depth_err8": row[3] / n}


def compute_err_metric(depth_gt, depth_pred, mask):
    """Masked depth errors (reference :9-30): mean |err| in millimetres (inputs are metres) and the fraction
    of masked pixels whose error exceeds 2 / 4 / 8 mm.  `mask` is a boolean selector.  Device inputs are reduced by
    one HIP kernel and a single 40-byte copy; host inputs by the reference's torch composition."""
    if depth_gt.is_cuda:
        out = _ops.dex_error_sweep(depth_gt, depth_pred.reshape(1, -1), mask.to(depth_gt.device))
        return _err_dict(out[0].tolist())
    gt, pred = depth_gt[mask], depth_pred[mask]
    diff = torch.abs(gt - pred)
    count = diff.numel()
    return {"depth_abs_err": torch.mean(torch.abs(pred * 1000 - gt * 1000)).item(),
            "depth_err2": int((diff > 2e-3).sum()) / count,
            "depth_err4": int((diff > 4e-3).sum()) / count,
            "depth_err8": int((diff > 8e-3).sum()) / count}


def dex_error_sweep(depth_gt, depth_fine_dex, mask=None, gt_lo=0.0, gt_hi=1.25):
    """The reference's validation loop over the Dex threshold candidates (train_dexnerf_rgb.py:391-408) on the device:
    every candidate depth map against the ground truth in ONE kernel and ONE device->host copy (the reference moves
    each of the K maps to the CPU).  `mask=None` is the reference's ground mask (0 < gt < 1.25 m).
    Returns (best_index, [err dict per candidate]); best = the first candidate whose mean abs error undercuts the
    running minimum, starting from 1000 mm as the reference does (None if none does)."""
    maps = depth_fine_dex if torch.is_tensor(depth_fine_dex) else torch.stack([d.reshape(-1) for d in depth_fine_dex])
    rows = _ops.dex_error_sweep(depth_gt, maps, None if mask is None else mask.to(depth_gt.device), gt_lo, gt_hi).tolist()
    errs = [_err_dict(r) for r in rows]
    best, min_abs = None, 1000.0
    for i, e in enumerate(errs):
        if e["depth_abs_err"] < min_abs:
            best, min_abs = i, e["depth_abs_err"]
    return best, errs


def gen_error_colormap_depth():
    """Piecewise-constant blue->red colormap rows [lo, hi, r, g, b] (reference :31-48)."""
    cols = np.array(
        [[0, 0.00001, 0, 0, 0], [0.00001, 2000. / (2 ** 10), 49, 54, 149], [2000. / (2 ** 10), 2000. / (2 ** 9), 69, 117, 180],
         [2000. / (2 ** 9), 2000. / (2 ** 8), 116, 173, 209], [2000. / (2 ** 8), 2000. / (2 ** 7), 171, 217, 233],
         [2000. / (2 ** 7), 2000. / (2 ** 6), 224, 243, 248], [2000. / (2 ** 6), 2000. / (2 ** 5), 254, 224, 144],
         [2000. / (2 ** 5), 2000. / (2 ** 4), 253, 174, 97], [2000. / (2 ** 4), 2000. / (2 ** 3), 244, 109, 67],
         [2000. / (2 ** 3), 2000. / (2 ** 2), 215, 48, 39], [2000. / (2 ** 2), np.inf, 165, 0, 38]], dtype=np.float32)
    cols[:, 2:5] /= 255.
    return cols


def depth_error_img(D_est_tensor, D_gt_tensor, mask, abs_thres=1., dilate_radius=1):
    """Colour-coded |gt - est| / abs_thres image for logging: inputs (B, H, W), returns the first image
    (H, W, 3) as numpy with the colour legend painted in its top-left corner (reference :46-70).  Device inputs
    are coloured by a HIP kernel (one (H,W,3) copy back instead of three input copies)."""
    if D_gt_tensor.is_cuda:
        img = _ops.depth_error_image(D_est_tensor.detach()[0], D_gt_tensor.detach()[0], mask.detach()[0].to(D_gt_tensor.device),
                                     abs_thres)
        return img.cpu().numpy()
    gt = D_gt_tensor.detach().cpu().numpy()
    est = D_est_tensor.detach().cpu().numpy()
    valid = mask.detach().cpu().numpy().astype(bool)
    batch, height, width = gt.shape
    err = np.abs(gt - est)
    err[~valid] = 0
    err[valid] = err[valid] / abs_thres
    cols = gen_error_colormap_depth()
    img = np.zeros([batch, height, width, 3], dtype=np.float32)
    for row in cols:
        img[np.logical_and(err >= row[0], err < row[1])] = row[2:]
    img[~valid] = 0.
    for i, row in enumerate(cols):  # legend: 20-pixel-wide swatches along the top edge
        img[:, :10, i * 20:(i + 1) * 20, :] = row[2:]
    return img[0]


# ---- hot path ------------------------------------------------------------------------------------------
import os as _os
_FP16_RENDER_DISABLED = [False]   # set once an fp16 render overflowed (see predict_and_render_radiance)
_STAGEWISE_TRAINING = [bool(int(_os.environ.get("DEXNERF_STAGEWISE_TRAINING", "0")))]   # tests flip this to compare the two routes


def _fusable(network_fn, embed_fn, embeddirs_fn):
    if not (isinstance(network_fn, FlexibleNeRFModel) and network_fn.fused_ok()):
        return False
    if not isinstance(embed_fn, Embedder):
        return False
    if embed_fn.num_encoding_functions != network_fn.num_encoding_fn_xyz or not embed_fn.include_input:
        return False
    if network_fn.use_viewdirs:
        if not isinstance(embeddirs_fn, Embedder):
            return False
        if embeddirs_fn.num_encoding_functions != network_fn.num_encoding_fn_dir or not embeddirs_fn.include_input:
            return False
    return True


def run_network(network_fn, pts, ray_batch, chunksize, embed_fn, embeddirs_fn):
    """Embed points (+ per-ray view directions) and evaluate the network: (..., S, 3) -> (..., S, 4)
    (reference train_utils.py:72-89).  A FlexibleNeRFModel with this package's embedders runs as one fused
    HIP kernel (encoding never materialised); any other callable gets the generic composition:
    HIP encoding -> network_fn minibatches -> concat."""
    _require_device(pts, "run_network")
    if _fusable(network_fn, embed_fn, embeddirs_fn) and (train_fused_ok(network_fn) or not needs_grad(network_fn, pts)):  # noqa: E501
        s = pts.shape[-2] if pts.dim() >= 2 else 1
        viewdirs = ray_batch[..., -3:] if network_fn.use_viewdirs else None
        log_dir = embeddirs_fn.log_sampling if network_fn.use_viewdirs else True
        out = run_network_fused(network_fn, pts, viewdirs, s, embed_fn.log_sampling, log_dir)
        return out.reshape(list(pts.shape[:-1]) + [4])
    pts_flat = pts.reshape((-1, pts.shape[-1]))
    embedded = embed_fn(pts_flat)
    if embeddirs_fn is not None:
        viewdirs = ray_batch[..., None, -3:]
        input_dirs = viewdirs.expand(pts.shape).reshape((-1, 3))
        embedded = torch.cat((embedded, embeddirs_fn(input_dirs)), dim=-1)
    preds = [network_fn(batch) for batch in get_minibatches(embedded, chunksize=chunksize)]
    radiance_field = torch.cat(preds, dim=0)
    return radiance_field.reshape(list(pts.shape[:-1]) + [radiance_field.shape[-1]])


def _wants_grad(*models):
    if not torch.is_grad_enabled():
        return False
    return any(m is not None and isinstance(m, torch.nn.Module) and any(p.requires_grad for p in m.parameters())
               for m in models)


def predict_and_render_radiance(ray_batch, model_coarse, model_fine, options, mode="train",
                                encode_position_fn=None, encode_direction_fn=None, m_thres_cand=None):
    """One ray chunk through coarse sampling -> coarse net -> composite -> inverse-CDF resampling -> fine
    net -> composite (reference train_utils.py:92-202).

    Returns (rgb_coarse, depth_coarse, acc_coarse, rgb_fine, depth_fine, acc_fine, *depth_fine_dex[K]).
    Supersets of the fork: m_thres_cand=None gives exactly six outputs (eval_nerf.py:175-187 unpacks six);
    num_fine == 0 / model_fine None returns None for the fine maps and the coarse Dex depths instead of the
    fork's NameError (:201).  RNG draw order matches the reference (rand, randn, rand, randn).
    """
    _require_device(ray_batch, "predict_and_render_radiance")
    opt = getattr(options.nerf, mode)
    thres = _thresholds(m_thres_cand)
    n = ray_batch.shape[0]
    nc, nf = int(opt.num_coarse), int(opt.num_fine)
    fine = nf > 0 and model_fine is not None
    perturb = bool(opt.perturb)
    std = float(opt.radiance_field_noise_std)
    white = bool(opt.white_background)
    lindisp = bool(opt.lindisp)
    dev = ray_batch.device
    use_viewdirs = ray_batch.shape[-1] > 8

    def rand(*shape):
        return torch.rand(shape, dtype=torch.float32, device=dev)

    def randn(*shape):
        return torch.randn(shape, dtype=torch.float32, device=dev)

    fused_models = (_fusable(model_coarse, encode_position_fn, encode_direction_fn)
                    and (not fine or _fusable(model_fine, encode_position_fn, encode_direction_fn))
                    and model_coarse.use_viewdirs == use_viewdirs)
    if fused_models and not _wants_grad(model_coarse, model_fine):
        # whole chunk in one C-ABI call (dn_render_rays); draws generated in the reference's order
        draws = {}
        if perturb:
            draws["t_rand"] = rand(n, nc)
        if std > 0.0:
            draws["noise_c"] = randn(n, nc)
        if fine and perturb:
            draws["u"] = rand(n, nf)
        if fine and std > 0.0:
            draws["noise_f"] = randn(n, nc + nf)
        lx = encode_position_fn.log_sampling
        ld = encode_direction_fn.log_sampling if use_viewdirs else True
        # bf16 modes: a no-grad render runs the fp16 instance of the kernel (_ops.set_render_policy), guarded against fp16's
        # range by the non-finite count the compositing passes leave in the workspace; stream capture cannot read it back
        # (no synchronisation inside a capture), so captured renders stay in the configured precision
        prec = _ops.render_precision()
        guarded = (prec != _ops._precision and not torch.cuda.is_current_stream_capturing() and not _FP16_RENDER_DISABLED[0]
                   and _ops.fp16_range_guard(model_coarse) and (not fine or _ops.fp16_range_guard(model_fine)))
        if not guarded:
            prec = _ops._precision
        pc = model_coarse.packed(lx, ld, precision=prec)
        pf = model_fine.packed(lx, ld, precision=prec) if fine else None
        maps = _ops.render_rays(pc, pf, ray_batch, nc, nf if fine else 0, lindisp, std, white, thres, draws)
        if guarded and _DEFERRED_GUARD[0] is not None:
            # called from run_one_iter_of_nerf: the status words of every chunk are read ONCE per image, there (no host
            # synchronisation per chunk); a copy, because the workspace may be reused by the next chunk
            _DEFERRED_GUARD[0].append(_ops.render_status_words())
        elif guarded and _ops.render_nonfinite_count() > 0:
            _warn_fp16_range()
            pc = model_coarse.packed(lx, ld)
            pf = model_fine.packed(lx, ld) if fine else None
            maps = _ops.render_rays(pc, pf, ray_batch, nc, nf if fine else 0, lindisp, std, white, thres, draws)
        rgb_c, depth_c, acc_c, rgb_f, depth_f, acc_f, dex = maps
        dex_list = [] if dex is None else [dex[k] for k in range(dex.shape[0])]
        return tuple([rgb_c, depth_c, acc_c, rgb_f, depth_f, acc_f] + dex_list)

    if (fused_models and train_fused_ok(model_coarse) and (not fine or train_fused_ok(model_fine))
            and not inputs_need_grad(ray_batch) and not _STAGEWISE_TRAINING[0]):
        # training: the whole chunk as one differentiable op - one C-ABI call forward (dn_render_rays_train), one backward
        # (dn_render_rays_backward); draws in the reference's order
        draws = {}
        if perturb:
            draws["t_rand"] = rand(n, nc)
        if std > 0.0:
            draws["noise_c"] = randn(n, nc)
        if fine and perturb:
            draws["u"] = rand(n, nf)
        if fine and std > 0.0:
            draws["noise_f"] = randn(n, nc + nf)
        logs = (encode_position_fn.log_sampling, encode_direction_fn.log_sampling if use_viewdirs else True)
        rgb_c, depth_c, acc_c, rgb_f, depth_f, acc_f, dex = render_rays_train(
            model_coarse, model_fine if fine else None, _ops.f32c(ray_batch), (nc, nf if fine else 0, lindisp, std, white), draws,
            thres, logs)
        dex_list = [] if dex is None else [dex[k] for k in range(dex.shape[0])]
        return tuple([rgb_c, depth_c, acc_c, rgb_f, depth_f, acc_f] + dex_list)

    # stage-by-stage composition (autograd through a network the fused training kernels do not cover, inputs that
    # require grad, or DEXNERF_STAGEWISE_TRAINING=1)
    rays = _ops.f32c(ray_batch)
    ro, rd = rays[..., :3], rays[..., 3:6]
    z_vals = _ops.coarse_depths(rays, nc, lindisp, rand(n, nc) if perturb else None)

    def network(model, z):
        if fused_models and (train_fused_ok(model) or not needs_grad(model)):
            # ray rows + depths straight into the fused kernel (the points are formed there)
            lx = encode_position_fn.log_sampling
            ld = encode_direction_fn.log_sampling if use_viewdirs else True
            return run_network_fused_rays(model, rays, z, lx, ld)
        pts = ro[..., None, :] + rd[..., None, :] * z[..., :, None]
        return run_network(model, pts, rays, opt.chunksize, encode_position_fn, encode_direction_fn)

    rf = network(model_coarse, z_vals)
    coarse = volume_render_radiance_field(rf, z_vals, rd, radiance_field_noise_std=std, white_background=white,
                                          m_thres_cand=thres)
    rgb_c, acc_c, weights, depth_c = coarse[0], coarse[2], coarse[3], coarse[4]
    if not fine:
        return tuple([rgb_c, depth_c, acc_c, None, None, None] + list(coarse[5:]))
    u = rand(n, nf) if perturb else None
    z_fine = _ops.fine_depths(z_vals, weights.detach(), nf, u)
    rf = network(model_fine, z_fine)
    fine_out = volume_render_radiance_field(rf, z_fine, rd, radiance_field_noise_std=std, white_background=white,
                                            m_thres_cand=thres)
    return tuple([rgb_c, depth_c, acc_c, fine_out[0], fine_out[4], fine_out[2]] + list(fine_out[5:]))


_DEFERRED_GUARD = [None]   # a list while run_one_iter_of_nerf collects its chunks' fp16 status words, else None


def _warn_fp16_range():
    import warnings
    warnings.warn("nerf: an fp16 render produced non-finite raw radiance-field values (a hidden activation beyond fp16's "
                  "range, 65504); this render is repeated in bf16 and every later one in this process runs in bf16 "
                  "(nerf.set_render_policy)", RuntimeWarning, stacklevel=3)
    _FP16_RENDER_DISABLED[0] = True


def run_one_iter_of_nerf(height, width, focal_length, model_coarse, model_fine, ray_origins, ray_directions, options,
                         mode="train", encode_position_fn=None, encode_direction_fn=None, m_thres_cand=None):
    """Pack rays, chunk, render, concatenate (reference train_utils.py:205-288).

    Output order: rgb_coarse, depth_coarse, acc_coarse, rgb_fine, depth_fine, acc_fine, *dex_fine[K]; flat
    (N,3)/(N,) in train mode, reshaped to the image in validation mode (slots 1/4 are depth, not disparity).
    """
    _require_device(ray_directions, "run_one_iter_of_nerf")
    thres = _thresholds(m_thres_cand)
    # device rays that need no gradient: the rows are packed by one kernel (dn_pack_ray_rows), op for op what follows
    one_launch = (ray_directions.is_cuda and ray_origins.is_cuda and ray_directions.dtype == torch.float32
                  and ray_origins.dtype == torch.float32 and not ray_directions.requires_grad and not ray_origins.requires_grad
                  and os.environ.get("DEXNERF_TORCH_RAY_ROWS", "") != "1")   # (developer switch: the torch composition, for A/B timing)
    viewdirs = None
    if options.nerf.use_viewdirs and not one_launch:
        viewdirs = ray_directions / ray_directions.norm(p=2, dim=-1).unsqueeze(-1)
        viewdirs = viewdirs.reshape((-1, 3))
    img_shape = ray_directions.shape
    restore_shapes = [img_shape, img_shape[:-1], img_shape[:-1]]
    if model_fine:
        restore_shapes = restore_shapes + restore_shapes
    restore_shapes = restore_shapes + [img_shape[:-1]] * len(thres)
    if options.dataset.no_ndc is False:
        ro, rd = ndc_rays(height, width, focal_length, 1.0, ray_origins, ray_directions)
    else:
        ro, rd = ray_origins, ray_directions
    ro, rd = ro.reshape((-1, 3)), rd.reshape((-1, 3))
    if one_launch:
        rays = _ops.pack_ray_rows(ro, rd, ray_directions.reshape((-1, 3)) if options.nerf.use_viewdirs else None,
                                  options.dataset.near, options.dataset.far)
    else:
        near = options.dataset.near * torch.ones_like(rd[..., :1])
        far = options.dataset.far * torch.ones_like(rd[..., :1])
        parts = [ro, rd, near, far] + ([viewdirs] if viewdirs is not None else [])
        rays = torch.cat(parts, dim=-1).float()
    def render_chunks():
        return [predict_and_render_radiance(batch, model_coarse, model_fine, options, mode=mode,
                                            encode_position_fn=encode_position_fn,
                                            encode_direction_fn=encode_direction_fn, m_thres_cand=thres)
                for batch in get_minibatches(rays, chunksize=getattr(options.nerf, mode).chunksize)]
    _DEFERRED_GUARD[0] = []
    try:
        chunks = render_chunks()
        status = _DEFERRED_GUARD[0]
    finally:
        _DEFERRED_GUARD[0] = None
    if status and int(torch.stack(status).sum().item()) > 0:
        # some chunk's fp16 render left fp16's range: the whole call again in bf16 (one host read per image, not per chunk)
        _warn_fp16_range()
        chunks = render_chunks()
    if len(chunks) == 1:
        images = list(chunks[0])   # a single chunk: nothing to concatenate (cat would copy every map)
    else:
        images = [torch.cat(col, dim=0) if col[0] is not None else None for col in zip(*chunks)]
    if mode == "validation":
        if not model_fine:
            # coarse-only: rgb, depth, acc, (None x3), dex...  -> reference returns the 3 maps + three Nones
            shapes = restore_shapes[:3] + [None, None, None] + restore_shapes[3:]
        else:
            shapes = restore_shapes
        images = [img.reshape(shape) if img is not None else None for img, shape in zip(images, shapes)]
    return tuple(images)
