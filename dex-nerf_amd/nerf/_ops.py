"""Tensor-level wrappers over the C ABI (one function per entry point) + autograd glue.

Every function here requires ROCm device tensors and calls straight into libdexnerf_hip.so; there is
no alternative implementation behind them.
"""
import ctypes
from ctypes import c_void_p

import torch

from . import _hip
from ._hip import MlpDesc, check, f32c, host_floats, lib, ptr, stream

_precision = _hip.PREC_F32
_save8 = False   # bf16 training keeps its saved activations / gradients at 8 bits (the 'bf16' mode; 'bf16-s16' keeps 16)


def set_precision(name):
    """'fp32' (exact fp32 MFMA chains, the parity mode); 'bf16' (bf16 MFMA, fp32 accumulate; render + training - the TRAINING step
    stores what it saves for the backward at 8 bits where the 48-point training kernels cover the network (widths 128 / 256, depth <= 9
    with view directions): activations as e4m3, layer gradients as e5m2 x a power-of-two scale chosen per launch from the largest
    upstream gradient, weight gradients formed with the fp8 MFMA - half the saved-tensor traffic of 16-bit saves, same forward bits;
    'bf16-s8' is the older name of this mode); 'bf16-s16' (bf16 with 16-bit saved tensors on the 32-point training kernels: what
    'bf16' meant up to round 3, and what 'bf16' falls back to for networks the 48-point training kernels do not cover); 'fp16' (fp16
    MFMA at bf16's rate with a 10-bit mantissa: ~57 dB instead of ~42 dB against fp32; render only - training in this mode
    differentiates the nn.Linear composition).  Rendering is identical in the three bf16 modes."""
    global _precision, _save8
    name = str(name).lower().replace("_", "-")
    _save8 = name in ("bf16", "bf16-s8")
    _precision = {"fp32": _hip.PREC_F32, "f32": _hip.PREC_F32, "bf16": _hip.PREC_BF16, "fp16": _hip.PREC_F16,
                  "f16": _hip.PREC_F16, "bf16-s8": _hip.PREC_BF16, "bf16-s16": _hip.PREC_BF16}[name]


_render16 = [None]   # what no-grad renders run in the bf16 modes: None = the environment decides (default "fp16")


def set_render_policy(name):
    """What `predict_and_render_radiance` WITHOUT autograd (validation images, the Dex depth sweep, eval_nerf) runs while the
    precision is 'bf16' / 'bf16-s8': "fp16" (default) = the fp16 instance of the same kernel - same matrix rate and layouts, a
    10-bit mantissa: ~58 dB / 0.997 Dex-depth agreement against fp32 where bf16 gives ~42 dB / 0.976, because the Dex readout
    is an argmax over thresholded sigma - guarded against fp16's range (65504): a render whose raw radiance field holds a
    non-finite value is repeated in bf16, with one warning; "bf16" = renders in bf16 like the training kernels.  The
    environment variable DEXNERF_BF16_RENDER sets the default.  Training is never affected."""
    name = None if name is None else str(name).lower()
    if name not in (None, "fp16", "bf16"):
        raise ValueError("render policy: 'fp16', 'bf16' or None (environment default)")
    _render16[0] = name


def get_render_policy():
    import os
    name = _render16[0] or os.environ.get("DEXNERF_BF16_RENDER", "fp16").lower()
    return name if name in ("fp16", "bf16") else "fp16"


def render_precision():
    """Precision code of no-grad renders under the current precision + render policy."""
    if _precision == _hip.PREC_BF16 and get_render_policy() == "fp16":
        return _hip.PREC_F16
    return _precision


def fp16_range_guard(model):
    """True when an fp16 render of this FlexibleNeRFModel reports every hidden activation that leaves fp16's range
    (dn_fp16_range_guard: the kernel instances that carry the tracker) - the condition for the fp16 render policy."""
    desc = MlpDesc(**{k: int(v) for k, v in model.desc_kwargs().items()})
    return bool(lib().dn_fp16_range_guard(ctypes.byref(desc)))


def get_precision():
    if _precision == _hip.PREC_BF16:
        return "bf16" if _save8 else "bf16-s16"
    return {_hip.PREC_F32: "fp32", _hip.PREC_F16: "fp16"}[_precision]


def train_precision(packed):
    """The precision code the TRAINING entry points get for this packed network: DN_PREC_BF16_S8 in the 'bf16-s8' mode where the
    48-point training kernels cover the network (other shapes train in plain bf16).  A forward records it with what it saves; the
    backward of those buffers uses the recorded code."""
    if _save8 and packed.precision == _hip.PREC_BF16 and s8_supported(packed):
        return _hip.PREC_BF16_S8
    return packed.precision


def set_s8_grad_scale(scale):
    """Power of two the saved layer gradients are multiplied by before they are rounded to e5m2, or 0 (the DEFAULT): every
    backward-data launch takes the power of two that puts ITS largest finite upstream gradient at 2^12 (one small maximum kernel per
    launch) - whatever the loss's reduction or scaling.  A fixed scale, e.g. 65536 = 2^16, saves that kernel: with 2^16, per-point
    gradients dL/d(pre-activation) between 2.3e-10 (e5m2's smallest subnormal / 2^16; smaller ones flush to zero) and 0.87
    (57344 / 2^16; larger ones saturate) are representable, 2 mantissa bits each - the range of a mean-reduced MSE loss over
    10^3 .. 10^5 rays from the first iteration to > 40 dB; s8_grad_stats tells whether a fixed scale fits."""
    check(lib().dn_set_s8_grad_scale(float(scale)), "dn_set_s8_grad_scale")


def adam_step(params, grads, exp_avg, exp_avg_sq, state, lr, lr_decay_per_step, betas, eps, zero_grads):
    """dn_adam_step on flat fp32 buffers (nerf.parallel.FlatAdam).  lr: float or device scalar."""
    lr_t = lr if torch.is_tensor(lr) else None
    check(lib().dn_adam_step(ptr(params), ptr(grads), ptr(exp_avg), ptr(exp_avg_sq), params.numel(), ptr(state), ptr(lr_t),
                             0.0 if lr_t is not None else float(lr), float(lr_decay_per_step), float(betas[0]), float(betas[1]),
                             float(eps), int(bool(zero_grads)), stream()), "dn_adam_step")


S8_RECORD_BYTES = 256     # kS8BlockBytes, csrc/mlp_geo48.h: the record behind the 8-bit saved gradients of a launch
_s8_records = []          # device views of the records of the latest backward launches (bounded; read by s8_grad_stats)


def _note_s8_record(grads, prec):
    if prec == _hip.PREC_BF16_S8 and grads is not None:
        _s8_records.append(grads[-S8_RECORD_BYTES:].view(torch.int32))   # (a view: keeps that buffer's storage until two later launches)
        del _s8_records[:-2]


def s8_grad_stats(grads=None):
    """What the 8-bit saved gradients of the latest training step lost to e5m2's range (include/dexnerf_hip.h, "8-bit saved
    tensors"): of the sampled non-zero gradient bytes (one 32-point record in sixteen, every layer), the fraction at e5m2's largest
    magnitude (saturated: clipped at 57344 / scale) and at its smallest (the edge of flushing to zero), plus the scale in use.
    grads = one saved-gradient buffer, or None = the buffers of the latest backward launches together.  Synchronises (a host
    read): call it at a logging interval, not every iteration.  None when no 8-bit backward has run."""
    recs = [grads[-S8_RECORD_BYTES:].view(torch.int32)] if grads is not None else list(_s8_records)
    if not recs:
        return None
    rows = torch.stack([r[:40] for r in recs]).cpu()
    # eight replicas of (saturated, floor, sampled) at words 8 + 4 r (csrc/mlp_geo48.h): a workgroup adds to one of them
    sat, floor, sampled = (int(rows[:, 8 + k:40:4].sum()) for k in (0, 1, 2))
    scales = sorted({float(v) for v in rows[:, 4].contiguous().view(torch.float32).tolist()})
    return {"saturated": sat / max(sampled, 1), "floor": floor / max(sampled, 1), "sampled": sampled, "scale": scales}


def _row_view(t):
    """(pointer, row stride in floats) of a (N,3)-like fp32 view whose last dim is contiguous."""
    assert t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1
    return c_void_p(t.data_ptr()), int(t.stride(0))


# ------------------------------------------------------------------------------------------------
def ray_bundle(height, width, rinv, origin, fx, cx, cy, device):
    ro = torch.empty((height, width, 3), dtype=torch.float32, device=device)
    rd = torch.empty_like(ro)
    check(lib().dn_ray_bundle(height, width, host_floats(rinv), host_floats(origin), float(fx), float(cx), float(cy),
                              ptr(ro), ptr(rd), stream()), "dn_ray_bundle")
    return ro, rd


def select_rays(height, width, rinv, origin, fx, cx, cy, near, far, pixel_index, image=None):
    """Packed ray rows (N,11) [ro, rd, near, far, viewdir] for the chosen pixels (+ their RGB from `image` (H,W,C))."""
    pix = pixel_index.contiguous()
    assert pix.dtype == torch.int64 and pix.is_cuda
    n = pix.numel()
    rays = torch.empty((n, 11), dtype=torch.float32, device=pix.device)
    target, img, channels = None, None, 0
    if image is not None:
        img = f32c(image)
        channels = img.shape[-1]
        target = torch.empty((n, 3), dtype=torch.float32, device=pix.device)
    check(lib().dn_select_rays(height, width, host_floats(rinv), host_floats(origin), float(fx), float(cx), float(cy), float(near),
                               float(far), ptr(pix), n, ptr(img), channels, ptr(rays), ptr(target), stream()), "dn_select_rays")
    return rays, target


def select_rays_indirect(height, width, cams, view, near, far, pixel_index, images=None):
    """select_rays with the camera record chosen by the device scalar `view` (int32) out of `cams` (V,16)."""
    pix = pixel_index.contiguous()
    assert pix.dtype == torch.int64 and view.dtype == torch.int32 and cams.dtype == torch.float32 and cams.is_contiguous()
    n = pix.numel()
    rays = torch.empty((n, 11), dtype=torch.float32, device=pix.device)
    target, img, channels = None, None, 0
    if images is not None:
        img = f32c(images)
        channels = img.shape[-1]
        target = torch.empty((n, 3), dtype=torch.float32, device=pix.device)
    check(lib().dn_select_rays_indirect(height, width, ptr(cams), ptr(view), float(near), float(far), ptr(pix), n, ptr(img), channels,
                                        ptr(rays), ptr(target), stream()), "dn_select_rays_indirect")
    return rays, target


def new_rng_state(seed, device, first_iteration=0):
    """Device record {seed_lo, seed_hi, cur, nxt} for the in-kernel draws of a training loop (csrc/dn_rng.h)."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    words = [seed & 0xFFFFFFFF, seed >> 32, int(first_iteration) & 0xFFFFFFFF, int(first_iteration) & 0xFFFFFFFF]
    return torch.tensor([w - (1 << 32) if w >= (1 << 31) else w for w in words], dtype=torch.int32, device=device)


def rng_fill(rng_state, stream_id, shape, normal=False):
    """The numbers the kernels draw for the current iteration from `rng_state`, stream `stream_id` (dn_rng_fill)."""
    out = torch.empty(shape, dtype=torch.float32, device=rng_state.device)
    check(lib().dn_rng_fill(ptr(rng_state), int(stream_id), out.numel(), int(bool(normal)), ptr(out), stream()), "dn_rng_fill")
    return out


def pack_ray_rows(ro, rd, view_d, near, far):
    """(N,3) origins / directions -> the (N, 8 | 11) ray rows of run_one_iter_of_nerf in one launch (dn_pack_ray_rows);
    view_d = the directions the unit view vectors come from (None: rows without them)."""
    ro, rd = f32c(ro), f32c(rd)
    vd = None if view_d is None else f32c(view_d)
    n = ro.shape[0]
    rows = torch.empty((n, 8 if vd is None else 11), dtype=torch.float32, device=ro.device)
    check(lib().dn_pack_ray_rows(ptr(ro), ptr(rd), ptr(vd), float(near), float(far), n, ptr(rows), stream()), "dn_pack_ray_rows")
    return rows


def select_rays_draw(height, width, cams, view, near, far, rng_state, n_rays, images=None, want_pixels=False):
    """select_rays_indirect with the pixels drawn on the device, without replacement, from `rng_state`'s next iteration.
    view=None: the training view is drawn in the kernel too (uniformly from the cameras in `cams`)."""
    assert (view is None or view.dtype == torch.int32) and cams.dtype == torch.float32 and cams.is_contiguous() and rng_state.dtype == torch.int32
    dev = cams.device
    rays = torch.empty((n_rays, 11), dtype=torch.float32, device=dev)
    target, img, channels = None, None, 0
    if images is not None:
        img = f32c(images)
        channels = img.shape[-1]
        target = torch.empty((n_rays, 3), dtype=torch.float32, device=dev)
    pix = torch.empty((n_rays,), dtype=torch.int64, device=dev) if want_pixels else None
    check(lib().dn_select_rays_draw(height, width, ptr(cams), ptr(view), int(cams.shape[0]), float(near), float(far), ptr(rng_state), n_rays, ptr(img), channels,
                                    ptr(rays), ptr(target), ptr(pix), stream()), "dn_select_rays_draw")
    return (rays, target, pix) if want_pixels else (rays, target)


def mse2_loss(rgb_c, rgb_f, target, luminance=False, rng_state=None):
    """(loss3 = [loss, mse_coarse, mse_fine] on the device, g_rgb_coarse, g_rgb_fine): the loss head + its upstream gradients in
    one launch (dn_mse2_loss); advances `rng_state`'s iteration counter."""
    rgb_c, target = f32c(rgb_c), f32c(target)
    n = rgb_c.shape[0]
    loss3 = torch.empty(3, dtype=torch.float32, device=rgb_c.device)
    g_c = torch.empty_like(rgb_c)
    g_f = None
    if rgb_f is not None:
        rgb_f = f32c(rgb_f)
        g_f = torch.empty_like(rgb_f)
    check(lib().dn_mse2_loss(ptr(rgb_c), ptr(rgb_f), ptr(target), n, int(bool(luminance)), ptr(loss3), ptr(g_c), ptr(g_f), ptr(rng_state),
                             stream()), "dn_mse2_loss")
    return loss3, g_c, g_f


def ndc_rays(height, width, focal, near, rays_o, rays_d):
    ro, rd = f32c(rays_o), f32c(rays_d)
    n = ro.numel() // 3
    ro_out, rd_out = torch.empty_like(ro), torch.empty_like(rd)
    check(lib().dn_ndc_rays(int(height), int(width), float(focal), float(near), ptr(ro), ptr(rd), n, ptr(ro_out), ptr(rd_out),
                            stream()), "dn_ndc_rays")
    return ro_out, rd_out


def dex_error_sweep(depth_gt, depth_pred, mask=None, gt_lo=0.0, gt_hi=1.25):
    """(K,5) float64 device tensor of [sum |err| mm, #>2mm, #>4mm, #>8mm, #masked] per candidate map."""
    gt = f32c(depth_gt).reshape(-1)
    pred = f32c(depth_pred).reshape(-1, gt.numel())
    k = pred.shape[0]
    m = None if mask is None else mask.reshape(-1).to(torch.uint8).contiguous()
    out = torch.empty((k, 5), dtype=torch.float64, device=gt.device)
    check(lib().dn_dex_error_sweep(ptr(gt), ptr(pred), k, gt.numel(), ptr(m), float(gt_lo), float(gt_hi), ptr(out), stream()),
          "dn_dex_error_sweep")
    return out


def depth_error_image(depth_est, depth_gt, mask, abs_thres=1.0):
    est, gt = f32c(depth_est), f32c(depth_gt)
    h, w = gt.shape[-2:]
    m = mask.to(torch.uint8).contiguous()
    out = torch.empty((h, w, 3), dtype=torch.float32, device=gt.device)
    check(lib().dn_depth_error_image(ptr(est), ptr(gt), ptr(m), h, w, float(abs_thres), ptr(out), stream()), "dn_depth_error_image")
    return out


def coarse_depths(rays, num_coarse, lindisp, t_rand=None):
    rays = f32c(rays)
    n = rays.shape[0]
    z = torch.empty((n, num_coarse), dtype=torch.float32, device=rays.device)
    tr = None if t_rand is None else f32c(t_rand)
    check(lib().dn_coarse_depths(ptr(rays), rays.shape[1], n, num_coarse, int(bool(lindisp)), ptr(tr), ptr(z),
                                 stream()), "dn_coarse_depths")
    return z


def positional_encoding(x, num_fns, include_input=True, log_sampling=True):
    shape = x.shape
    dim = shape[-1]
    xf = f32c(x).reshape(-1, dim)
    width = dim * ((1 if include_input else 0) + 2 * num_fns)
    out = torch.empty((xf.shape[0], width), dtype=torch.float32, device=x.device)
    check(lib().dn_positional_encoding(ptr(xf), xf.shape[0], dim, num_fns, int(bool(include_input)),
                                       int(bool(log_sampling)), ptr(out), stream()), "dn_positional_encoding")
    return out.reshape(*shape[:-1], width)


# ------------------------------------------------------------------------------------------------
class PackedMLP:
    """MFMA fragment stream of one FlexibleNeRFModel, rebuilt when the parameters change."""

    def __init__(self, desc_kwargs, device, precision=None):
        self.desc = MlpDesc(**{k: int(v) for k, v in desc_kwargs.items()})
        self.precision = _precision if precision is None else precision
        nbytes = lib().dn_mlp_packed_bytes(ctypes.byref(self.desc), self.precision)
        if nbytes == 0:
            check(-1001, "dn_mlp_packed_bytes")
        self.buffer = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.key = None          # parameter key the core stream (bias tiles + pieces: every kernel but one) was packed from
        self.key48 = None        # ... and the stream of the 48-point inference kernel: a training loop leaves it stale until a render
        self.buffers_bwd = {}    # transposed streams for the backward-data chain, per training precision code (the 8-bit-saved-tensor
        self.keys_bwd = {}       # mode runs the 48-point chain: another stream), packed on first training use

    def pack(self, weights, biases, parts=_hip.PACK_ALL):
        """weights/biases: lists of device tensors in the reference parameter order; parts: _hip.PACK_CORE | _hip.PACK_G48."""
        n = len(weights)
        ws = [f32c(w.detach()) for w in weights]
        bs = [f32c(b.detach()) for b in biases]
        wp = (c_void_p * n)(*[w.data_ptr() for w in ws])
        bp = (c_void_p * n)(*[b.data_ptr() for b in bs])
        check(lib().dn_mlp_pack_parts(ctypes.byref(self.desc), self.precision, wp, bp, ptr(self.buffer), int(parts), stream()),
              "dn_mlp_pack_parts")
        self._keep = (ws, bs)  # keep sources alive until the pack kernel has run on this stream

    def require_fresh_inference_stream(self, what):
        """The inference entry points may run the 48-point kernel, whose stream lives behind the core one in `buffer` and is
        left stale by the training entry points (FlexibleNeRFModel.packed(train=True)).  A caller that kept this object across
        optimizer steps instead of asking `model.packed()` again would render OLD weights without any error: refuse."""
        if self.key48 != self.key:
            raise RuntimeError(f"{what}: the packed network's 48-point inference stream is older than its core stream or the other "
                               "way round (it was last packed by a training entry point, which refreshes only the stream it reads) - "
                               "obtain the packed network with model.packed() (no train=True) before rendering")


def pack_backward(packed, weights, prec=None):
    """(Re)build the transposed weight stream dn_mlp_backward_data uses under training precision code `prec`."""
    prec = train_precision(packed) if prec is None else prec
    buf = packed.buffers_bwd.get(prec)
    if buf is None:
        nbytes = lib().dn_mlp_backward_packed_bytes(ctypes.byref(packed.desc), prec)
        if nbytes == 0:
            check(-1001, "dn_mlp_backward_packed_bytes")
        buf = packed.buffers_bwd[prec] = torch.empty(nbytes, dtype=torch.uint8, device=packed.buffer.device)
    ws = [f32c(w.detach()) for w in weights]
    wp = (c_void_p * len(ws))(*[w.data_ptr() for w in ws])
    check(lib().dn_mlp_pack_backward(ctypes.byref(packed.desc), prec, wp, ptr(buf), stream()), "dn_mlp_pack_backward")
    packed._keep_bwd = ws


def ensure_backward_stream(model, packed, prec=None):
    """The backward stream of `packed` for precision code `prec`, re-packed when the parameters changed (or under capture)."""
    prec = train_precision(packed) if prec is None else prec
    key = model.param_key()
    if packed.keys_bwd.get(prec) != key or torch.cuda.is_current_stream_capturing():
        pack_backward(packed, [m.weight for m in model.linear_modules()], prec)
        packed.keys_bwd[prec] = key
    return packed.buffers_bwd[prec]


def pack_train_pair(model_a, model_b, logs):
    """The streams a DN_PREC_BF16_S8 training step reads, for two networks of one architecture, in two launches (dn_mlp_pack_train_pair)
    - or, when that does not apply (different shapes, another precision, nothing stale), the per-network route.  Returns the two
    packed networks and the precision code."""
    pa, pb = model_a._packed_slot(*logs), model_b._packed_slot(*logs)
    prec = train_precision(pa)
    capturing = torch.cuda.is_current_stream_capturing()
    same = bytes(pa.desc) == bytes(pb.desc) and train_precision(pb) == prec == _hip.PREC_BF16_S8
    ka, kb = model_a.param_key(), model_b.param_key()
    stale = lambda pk, key: pk.key48 != key or pk.keys_bwd.get(prec) != key   # noqa: E731
    if same and (capturing or (stale(pa, ka) and stale(pb, kb))):
        for pk in (pa, pb):
            if pk.buffers_bwd.get(prec) is None:
                pk.buffers_bwd[prec] = torch.empty(lib().dn_mlp_backward_packed_bytes(ctypes.byref(pk.desc), prec), dtype=torch.uint8,
                                                   device=pk.buffer.device)
        arrays, keep = [], []
        for model in (model_a, model_b):
            mods = model.linear_modules()
            ws = [f32c(m.weight.detach()) for m in mods]
            bs = [f32c(m.bias.detach()) for m in mods]
            keep.append((ws, bs))
            arrays.append(((c_void_p * len(ws))(*[w.data_ptr() for w in ws]), (c_void_p * len(bs))(*[b.data_ptr() for b in bs])))
        check(lib().dn_mlp_pack_train_pair(ctypes.byref(pa.desc), arrays[0][0], arrays[0][1], ptr(pa.buffer), ptr(pa.buffers_bwd[prec]),
                                           arrays[1][0], arrays[1][1], ptr(pb.buffer), ptr(pb.buffers_bwd[prec]), stream()),
              "dn_mlp_pack_train_pair")
        pa._keep_pair = keep
        for pk, key in ((pa, ka), (pb, kb)):
            pk.key48 = key
            pk.keys_bwd[prec] = key
        return pa, pb, prec
    pa, pb = model_a.packed(*logs, train=True), model_b.packed(*logs, train=True)
    prec = train_precision(pa)
    if train_precision(pb) != prec:
        prec = pa.precision
    ensure_backward_stream(model_a, pa, prec)
    ensure_backward_stream(model_b, pb, prec)
    return pa, pb, prec


def s8_supported(packed):
    """True when the 8-bit-saved-tensor training kernels (48-point geometry) cover this network."""
    return lib().dn_mlp_backward_packed_bytes(ctypes.byref(packed.desc), _hip.PREC_BF16_S8) != 0


def train_sizes(packed, n_points, s8=False, prec=None):
    a, m, g = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
    if prec is None:
        prec = _hip.PREC_BF16_S8 if s8 else packed.precision
    check(lib().dn_mlp_train_sizes(ctypes.byref(packed.desc), prec, n_points, ctypes.byref(a), ctypes.byref(m),
                                   ctypes.byref(g)), "dn_mlp_train_sizes")
    return a.value, m.value, g.value


def run_network_train(packed, pts, viewdirs, samples_per_ray, rays=None, z_vals=None, prec=None):
    """Training forward: raw radiance field + the opaque (act, masks) buffers the backward needs.  Either explicit
    points (+ per-ray view directions) or packed ray rows + depths (the points are formed in the kernel)."""
    if rays is not None:
        rays, z_vals = f32c(rays), f32c(z_vals)
        n_rays, samples_per_ray = z_vals.shape
        n_pts = n_rays * samples_per_ray
        dev = rays.device
    else:
        pts = f32c(pts).reshape(-1, 3)
        n_pts = pts.shape[0]
        assert n_pts % samples_per_ray == 0
        dev = pts.device
    prec = packed.precision if prec is None else prec
    a_bytes, m_bytes, _ = train_sizes(packed, n_pts, prec=prec)
    out = torch.empty((n_pts, 4), dtype=torch.float32, device=dev)
    act = torch.empty(a_bytes, dtype=torch.uint8, device=dev)
    masks = torch.empty(m_bytes, dtype=torch.uint8, device=dev)
    if rays is not None:
        check(lib().dn_run_network_train(ctypes.byref(packed.desc), prec, ptr(packed.buffer), None, None, ptr(rays),
                                         rays.shape[1], ptr(z_vals), n_pts // samples_per_ray, samples_per_ray, ptr(out),
                                         ptr(act), ptr(masks), stream()), "dn_run_network_train")
        return out, act, masks
    vd = None if viewdirs is None else f32c(viewdirs).reshape(-1, 3)
    check(lib().dn_run_network_train(ctypes.byref(packed.desc), prec, ptr(packed.buffer), ptr(pts), ptr(vd), None,
                                     0, None, n_pts // samples_per_ray, samples_per_ray, ptr(out), ptr(act), ptr(masks),
                                     stream()), "dn_run_network_train")
    return out, act, masks


def mlp_backward_data(packed, g_out, masks, n_points, prec=None):
    g_out = f32c(g_out).reshape(-1, 4)
    prec = packed.precision if prec is None else prec
    _, _, g_bytes = train_sizes(packed, n_points, prec=prec)
    grads = torch.empty(g_bytes, dtype=torch.uint8, device=g_out.device)
    check(lib().dn_mlp_backward_data(ctypes.byref(packed.desc), prec, ptr(packed.buffers_bwd[prec]), ptr(g_out),
                                     ptr(masks), n_points, ptr(grads), stream()), "dn_mlp_backward_data")
    _note_s8_record(grads, prec)
    return grads


def mlp_unpack(packed, which, native, n_points, slot, width, kind, out, col0=0, prec=None):
    """native pieces -> out[:, col0:col0+width_or_pe_dim] (plain fp32 rows).  prec = PREC_BF16_S8: the 8-bit units of the
    48-point training kernels (slot in units of 64 features per 16-point group; kind 3 = the custom output-gradient unit)."""
    check(lib().dn_mlp_unpack(ctypes.byref(packed.desc), packed.precision if prec is None else prec, which, ptr(native), n_points, slot, width, kind,
                              ptr(out), out.shape[1], col0, stream()), "dn_mlp_unpack")
    return out


def mlp_weight_grad(packed, act, grads, n_points, g_slot, n_out, x_slot, x_width, pe_kind, d_w, d_b):
    """bf16 buffers: d_w (n_out, >= x_width + pe_dim) += dY^T [X | PE], d_b += sum dY (both pre-zeroed fp32)."""
    check(lib().dn_mlp_weight_grad(ctypes.byref(packed.desc), packed.precision, ptr(act), ptr(grads), n_points, g_slot, n_out,
                                   x_slot, x_width, pe_kind, ptr(d_w), d_w.shape[1], ptr(d_b), stream()), "dn_mlp_weight_grad")


_DETERMINISTIC_WGRAD = [True]


def set_deterministic_weight_gradients(on):
    """True (default): the weight-gradient launches of the training paths reduce their workgroups' partials in a fixed order (a partial
    per workgroup in a scratch buffer + a second small launch: dn_*_ws) - a training run is then a pure function of its seed, bit for
    bit, like the reference on the CPU.  False: fp32 atomics straight into the gradients (no scratch; the sum's last bits change from
    launch to launch)."""
    _DETERMINISTIC_WGRAD[0] = bool(on)


def _wgrad_scratch(packed, n_networks):
    """(tensor, bytes) of reduction scratch for this architecture, or (None, 0).  A fresh stream-ordered allocation per call (the
    caching allocator hands the same block back every step; inside a graph capture it belongs to the graph's pool)."""
    if not _DETERMINISTIC_WGRAD[0]:
        return None, 0
    nbytes = int(lib().dn_mlp_weight_grad_scratch_bytes(ctypes.byref(packed.desc), int(n_networks)))
    if nbytes <= 0:
        return None, 0
    return torch.empty(nbytes, dtype=torch.uint8, device=packed.buffer.device), nbytes


def mlp_weight_grad_all(packed, act, grads, n_points, shapes, s8=False, prec=None):
    """bf16 buffers: every layer's (dW, db) in one launch.  `shapes` = [(out, in)] in linear_modules() order; returns
    [(dW, db)] as views of ONE zero-filled fp32 buffer.  s8: the buffers are in the 8-bit unit layout (convert_saved_s8)."""
    dev = act.device
    total = sum(o * i + o for o, i in shapes)
    flat = torch.zeros(total, dtype=torch.float32, device=dev)
    out, off = [], 0
    for o, i in shapes:
        d_w = flat[off:off + o * i].view(o, i); off += o * i
        d_b = flat[off:off + o]; off += o
        out.append((d_w, d_b))
    wp = (c_void_p * len(out))(*[w.data_ptr() for w, _ in out])
    bp = (c_void_p * len(out))(*[b.data_ptr() for _, b in out])
    if prec is None:
        prec = _hip.PREC_BF16_S8 if s8 else packed.precision
    scratch, nbytes = _wgrad_scratch(packed, 1)
    check(lib().dn_mlp_weight_grad_all_ws(ctypes.byref(packed.desc), prec, ptr(act), ptr(grads), n_points, wp, bp, ptr(scratch), nbytes, stream()),
          "dn_mlp_weight_grad_all")
    return out


def mlp_weight_grad_all_into(packed, act, grads, n_points, views, prec=None):
    """The same launch accumulating into caller-owned (dW, db) tensors (dense fp32 with the nn.Linear shapes, e.g. the
    `.grad` views of a parallel.FlatGradBucket, already zeroed for this step)."""
    wp = (c_void_p * len(views))(*[w.data_ptr() for w, _ in views])
    bp = (c_void_p * len(views))(*[b.data_ptr() for _, b in views])
    scratch, nbytes = _wgrad_scratch(packed, 1)
    check(lib().dn_mlp_weight_grad_all_ws(ctypes.byref(packed.desc), packed.precision if prec is None else prec, ptr(act), ptr(grads),
                                          n_points, wp, bp, ptr(scratch), nbytes, stream()), "dn_mlp_weight_grad_all")


def run_network_pts(packed, pts, viewdirs, samples_per_ray):
    packed.require_fresh_inference_stream("run_network")
    pts = f32c(pts).reshape(-1, 3)
    n_pts = pts.shape[0]
    assert n_pts % samples_per_ray == 0
    out = torch.empty((n_pts, 4), dtype=torch.float32, device=pts.device)
    vd = None if viewdirs is None else f32c(viewdirs).reshape(-1, 3)
    check(lib().dn_run_network(ctypes.byref(packed.desc), packed.precision, ptr(packed.buffer), ptr(pts), ptr(vd), None, 0,
                               None, n_pts // samples_per_ray, samples_per_ray, ptr(out), stream()), "dn_run_network")
    return out


def run_network_rays(packed, rays, z_vals):
    packed.require_fresh_inference_stream("run_network")
    rays, z_vals = f32c(rays), f32c(z_vals)
    n, s = z_vals.shape
    out = torch.empty((n, s, 4), dtype=torch.float32, device=rays.device)
    check(lib().dn_run_network(ctypes.byref(packed.desc), packed.precision, ptr(packed.buffer), None, None, ptr(rays),
                               rays.shape[1], ptr(z_vals), n, s, ptr(out), stream()), "dn_run_network")
    return out


def mlp_forward_encoded(packed, x):
    x = f32c(x)
    shape = x.shape
    xf = x.reshape(-1, shape[-1])
    out = torch.empty((xf.shape[0], 4), dtype=torch.float32, device=x.device)
    check(lib().dn_mlp_forward_encoded(ctypes.byref(packed.desc), packed.precision, ptr(packed.buffer), ptr(xf),
                                       xf.shape[0], ptr(out), stream()), "dn_mlp_forward_encoded")
    return out.reshape(*shape[:-1], 4)


# ------------------------------------------------------------------------------------------------
def volume_render_fwd(rf, z, rd, noise, noise_std, white, m_thres, want_weights=True):
    rf, z = f32c(rf), f32c(z)
    n, s = z.shape
    dev = rf.device
    rd = rd if (rd.dtype == torch.float32 and rd.dim() == 2 and rd.stride(1) == 1) else f32c(rd).reshape(-1, 3)
    rd_ptr, rd_stride = _row_view(rd)
    k = len(m_thres)
    rgb = torch.empty((n, 3), dtype=torch.float32, device=dev)
    disp = torch.empty((n,), dtype=torch.float32, device=dev)
    acc = torch.empty_like(disp)
    depth = torch.empty_like(disp)
    weights = torch.empty((n, s), dtype=torch.float32, device=dev) if want_weights else None
    dex = torch.empty((k, n), dtype=torch.float32, device=dev) if k else None
    nz = None if (noise is None or noise_std <= 0.0) else f32c(noise)
    check(lib().dn_volume_render(ptr(rf), ptr(z), rd_ptr, rd_stride, ptr(nz), float(noise_std), int(bool(white)),
                                 host_floats(m_thres), k, n, s, ptr(rgb), ptr(disp), ptr(acc), ptr(weights), ptr(depth),
                                 ptr(dex), stream()), "dn_volume_render")
    return rgb, disp, acc, weights, depth, dex


def volume_render_bwd(rf, z, rd, noise, noise_std, white, g_rgb, g_depth, g_acc, g_disp, g_weights):
    rf, z = f32c(rf), f32c(z)
    n, s = z.shape
    rd = rd if (rd.dtype == torch.float32 and rd.dim() == 2 and rd.stride(1) == 1) else f32c(rd).reshape(-1, 3)
    rd_ptr, rd_stride = _row_view(rd)
    g_rf = torch.empty((n, s, 4), dtype=torch.float32, device=rf.device)
    nz = None if (noise is None or noise_std <= 0.0) else f32c(noise)
    gs = [None if g is None else f32c(g) for g in (g_rgb, g_depth, g_acc, g_disp, g_weights)]
    check(lib().dn_volume_render_backward(ptr(rf), ptr(z), rd_ptr, rd_stride, ptr(nz), float(noise_std), int(bool(white)),
                                          n, s, ptr(gs[0]), ptr(gs[1]), ptr(gs[2]), ptr(gs[3]), ptr(gs[4]), ptr(g_rf),
                                          stream()), "dn_volume_render_backward")
    return g_rf


class VolumeRenderFn(torch.autograd.Function):
    """Differentiable w.r.t. the radiance field only: depths and directions carry no gradient on this path
    (the reference detaches z_samples, train_utils.py:170)."""

    @staticmethod
    def forward(ctx, rf, z, rd, noise, noise_std, white, m_thres):
        rgb, disp, acc, weights, depth, dex = volume_render_fwd(rf, z, rd, noise, noise_std, white, m_thres)
        ctx.set_materialize_grads(False)   # unused outputs arrive as None (the kernel takes NULL), not as zero-filled tensors
        ctx.save_for_backward(rf, z, rd, noise if noise is not None else torch.empty(0, device=rf.device))
        ctx.cfg = (float(noise_std), bool(white), noise is not None)
        outs = (rgb, disp, acc, weights, depth) + ((dex,) if dex is not None else ())
        if dex is not None:
            ctx.mark_non_differentiable(dex)
        return outs

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_weights, g_depth, *_):
        rf, z, rd, noise = ctx.saved_tensors
        noise_std, white, has_noise = ctx.cfg

        def nz(g):
            return None if g is None else g.contiguous()
        g_rf = volume_render_bwd(rf, z, rd, noise if has_noise else None, noise_std, white, nz(g_rgb), nz(g_depth),
                                 nz(g_acc), nz(g_disp), nz(g_weights))
        return g_rf, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------
def sample_pdf(bins, weights, num_samples, u=None, want_inds=False):
    bins, weights = f32c(bins), f32c(weights)
    n, b = bins.shape
    assert weights.shape == (n, b - 1)
    samples = torch.empty((n, num_samples), dtype=torch.float32, device=bins.device)
    inds = torch.empty((n, num_samples), dtype=torch.int64, device=bins.device) if want_inds else None
    uu = None if u is None else f32c(u)
    check(lib().dn_sample_pdf(ptr(bins), ptr(weights), ptr(uu), n, b, num_samples, ptr(samples),
                              None if inds is None else c_void_p(inds.data_ptr()), stream()), "dn_sample_pdf")
    return (samples, inds) if want_inds else samples


def fine_depths(z_coarse, weights, num_fine, u=None, want_samples=False):
    z_coarse, weights = f32c(z_coarse), f32c(weights)
    n, nc = z_coarse.shape
    z_fine = torch.empty((n, nc + num_fine), dtype=torch.float32, device=z_coarse.device)
    zs = torch.empty((n, num_fine), dtype=torch.float32, device=z_coarse.device) if want_samples else None
    uu = None if u is None else f32c(u)
    check(lib().dn_fine_depths(ptr(z_coarse), ptr(weights), ptr(uu), n, nc, num_fine, ptr(z_fine), ptr(zs), stream()),
          "dn_fine_depths")
    return (z_fine, zs) if want_samples else z_fine


_ws_cache = {}


def render_rays(packed_c, packed_f, rays, num_coarse, num_fine, lindisp, noise_std, white, m_thres, draws=None):
    """dn_render_rays: the whole predict_and_render_radiance forward for one ray chunk (no autograd)."""
    packed_c.require_fresh_inference_stream("render_rays")
    if packed_f is not None:
        packed_f.require_fresh_inference_stream("render_rays")
    rays = f32c(rays)
    n = rays.shape[0]
    dev = rays.device
    draws = draws or {}
    k = len(m_thres)
    fine = num_fine > 0 and packed_f is not None
    nf = num_fine if fine else 0
    nbytes = lib().dn_render_workspace_bytes(n, num_coarse, nf)
    key = (dev, torch.cuda.current_stream().cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        _ws_cache[key] = ws
    ws = ws[:nbytes]   # (the status block is the last 256 bytes of what THIS call asked for)

    def new(*shape):
        return torch.empty(shape, dtype=torch.float32, device=dev)
    rgb_c, depth_c, acc_c = new(n, 3), new(n), new(n)
    rgb_f, depth_f, acc_f = (new(n, 3), new(n), new(n)) if fine else (None, None, None)
    dex = new(k, n) if k else None
    prec = packed_c.precision
    t = {name: (None if draws.get(name) is None else f32c(draws[name])) for name in ("t_rand", "noise_c", "u", "noise_f")}
    check(lib().dn_render_rays(
        ctypes.byref(packed_c.desc), ptr(packed_c.buffer),
        ctypes.byref(packed_f.desc) if fine else None, ptr(packed_f.buffer) if fine else None, prec,
        ptr(rays), rays.shape[1], n, num_coarse, nf, int(bool(lindisp)), float(noise_std), int(bool(white)),
        host_floats(m_thres), k, ptr(t["t_rand"]), ptr(t["noise_c"]), ptr(t["u"]), ptr(t["noise_f"]),
        ptr(rgb_c), ptr(depth_c), ptr(acc_c), ptr(rgb_f), ptr(depth_f), ptr(acc_f), ptr(dex), ptr(ws), stream()),
        "dn_render_rays")
    render_rays.last_workspace = ws
    return rgb_c, depth_c, acc_c, rgb_f, depth_f, acc_f, dex


def render_status_words(ws=None):
    """Device copy (no synchronisation) of the two status words of the last dn_render_rays call on this stream - see
    render_nonfinite_count; the caller sums the copies of several chunks and reads them back once."""
    ws = render_rays.last_workspace if ws is None else ws
    return ws[ws.numel() - 256: ws.numel() - 248].view(torch.int32).clone()


def render_nonfinite_count(ws=None):
    """Status block of the last dn_render_rays call on this stream (synchronises): non-finite raw radiance-field samples met
    by the compositing passes + waves of the fp16 network kernel that saw an activation leave fp16's range."""
    ws = render_rays.last_workspace if ws is None else ws
    words = ws[ws.numel() - 256: ws.numel() - 248].view(torch.int32).tolist()
    return int(words[0]) + int(words[1])


# ---- predict_and_render_radiance under autograd: one C call forward, one (or two halves) backward ----------------------
def render_rays_train(packed_c, packed_f, rays, num_coarse, num_fine, lindisp, noise_std, white, m_thres, draws=None, prec=None,
                      rng_state=None, perturb=False):
    """dn_render_rays_train: the training forward of a whole ray chunk.  Returns (maps, saved): maps = (rgb_c, depth_c, acc_c,
    rgb_f, depth_f, acc_f, dex), saved = what dn_render_rays_backward needs (workspace, per-network act / masks, the draws)."""
    rays = f32c(rays)
    n = rays.shape[0]
    dev = rays.device
    draws = draws or {}
    k = len(m_thres)
    fine = num_fine > 0 and packed_f is not None
    nf = num_fine if fine else 0
    ws = torch.empty(max(lib().dn_render_train_workspace_bytes(n, num_coarse, nf), 1), dtype=torch.uint8, device=dev)

    def new(*shape):
        return torch.empty(shape, dtype=torch.float32, device=dev)

    prec = train_precision(packed_c) if prec is None else prec

    def bufs(packed, n_points):
        a, m, _ = train_sizes(packed, n_points, prec=prec)
        return torch.empty(a, dtype=torch.uint8, device=dev), torch.empty(m, dtype=torch.uint8, device=dev)
    act_c, masks_c = bufs(packed_c, n * num_coarse)
    act_f, masks_f = bufs(packed_f, n * (num_coarse + nf)) if fine else (None, None)
    rgb_c, depth_c, acc_c = new(n, 3), new(n), new(n)
    rgb_f, depth_f, acc_f = (new(n, 3), new(n), new(n)) if fine else (None, None, None)
    dex = new(k, n) if k else None
    t = {name: (None if draws.get(name) is None else f32c(draws[name])) for name in ("t_rand", "noise_c", "u", "noise_f")}
    check(lib().dn_render_rays_train(
        ctypes.byref(packed_c.desc), ptr(packed_c.buffer),
        ctypes.byref(packed_f.desc) if fine else None, ptr(packed_f.buffer) if fine else None, prec,
        ptr(rays), rays.shape[1], n, num_coarse, nf, int(bool(lindisp)), float(noise_std), int(bool(white)),
        host_floats(m_thres), k, ptr(t["t_rand"]), ptr(t["noise_c"]), ptr(t["u"]), ptr(t["noise_f"]),
        ptr(rgb_c), ptr(depth_c), ptr(acc_c), ptr(rgb_f), ptr(depth_f), ptr(acc_f), ptr(dex), ptr(ws),
        ptr(act_c), ptr(masks_c), ptr(act_f), ptr(masks_f), ptr(rng_state), int(bool(perturb)), stream()), "dn_render_rays_train")
    saved = dict(rays=rays, ws=ws, act_c=act_c, masks_c=masks_c, act_f=act_f, masks_f=masks_f, noise_c=t["noise_c"],
                 noise_f=t["noise_f"], n=n, nc=num_coarse, nf=nf, noise_std=float(noise_std), white=bool(white), prec=prec,
                 rng_state=rng_state)
    return (rgb_c, depth_c, acc_c, rgb_f, depth_f, acc_f, dex), saved


def render_rays_backward(packed_c, packed_f, saved, g_c, g_f, views_c, views_f, nets=3):
    """dn_render_rays_backward: composite backward -> backward-data chain -> weight gradients for the networks selected by
    `nets` (bit 0 coarse, bit 1 fine), ACCUMULATING into views_* = [(dW, db)] in linear_modules() order.
    g_c / g_f = (g_rgb, g_depth, g_acc) upstream gradients (None = zero)."""
    dev = saved["rays"].device
    n, nc, nf = saved["n"], saved["nc"], saved["nf"]
    fine = nf > 0 and packed_f is not None

    prec = saved["prec"]

    def grads_buf(packed, n_points):
        return torch.empty(train_sizes(packed, n_points, prec=prec)[2], dtype=torch.uint8, device=dev)
    grads_c = grads_buf(packed_c, n * nc) if nets & 1 else None
    grads_f = grads_buf(packed_f, n * (nc + nf)) if (fine and nets & 2) else None

    def arrays(views):
        if views is None:
            return None, None
        return ((c_void_p * len(views))(*[w.data_ptr() for w, _ in views]), (c_void_p * len(views))(*[b.data_ptr() for _, b in views]))
    wc, bc = arrays(views_c if nets & 1 else None)
    wf, bf = arrays(views_f if (fine and nets & 2) else None)
    gs = [None if g is None else f32c(g) for g in tuple(g_c) + tuple(g_f)]
    k_nets = 2 if (fine and nets == 3) else 1
    scratch, scratch_bytes = _wgrad_scratch(packed_c, k_nets)
    if fine:   # (two architectures: the larger need)
        other = _wgrad_scratch(packed_f, k_nets)
        if other[1] > scratch_bytes:
            scratch, scratch_bytes = other
    check(lib().dn_render_rays_backward_ws(
        ctypes.byref(packed_c.desc), ptr(packed_c.buffers_bwd[prec]),
        ctypes.byref(packed_f.desc) if fine else None, ptr(packed_f.buffers_bwd[prec]) if fine else None, prec,
        ptr(saved["rays"]), saved["rays"].shape[1], n, nc, nf, saved["noise_std"], int(saved["white"]),
        ptr(saved["noise_c"]), ptr(saved["noise_f"]), ptr(gs[0]), ptr(gs[1]), ptr(gs[2]), ptr(gs[3]), ptr(gs[4]), ptr(gs[5]),
        ptr(saved["ws"]), ptr(saved["act_c"]), ptr(saved["masks_c"]), ptr(grads_c), ptr(saved["act_f"]), ptr(saved["masks_f"]),
        ptr(grads_f), wc, bc, wf, bf, int(nets), ptr(saved.get("rng_state")), ptr(scratch), scratch_bytes, stream()), "dn_render_rays_backward")
    _note_s8_record(grads_c, prec); _note_s8_record(grads_f, prec)
    return grads_c, grads_f, scratch   # (kept alive by the caller until the stream has consumed them: PyTorch's caching allocator is stream-ordered)
