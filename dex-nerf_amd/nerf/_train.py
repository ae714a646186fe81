"""Dispatch between the fused HIP network kernels and PyTorch autograd.

Inference: one fused kernel (positional encoding + MLP).  Training, for the nets the training kernels cover
(W in {128, 256}, L_xyz = 10): `FusedNetFn` - a fused forward that keeps every stage's output and ReLU masks in a
wave-native layout, and a fused backward-data chain on the transposed weight stream (dn_mlp_backward_data);
the weight/bias gradients come from one weight-gradient kernel launch per network on the saved native buffers
(bf16 MFMA, or exact-fp32 MFMA in the parity mode; DEXNERF_FP32_DW=gemm selects the older fp32 route: unpack to plain rows
+ library GEMMs).  Other configurations
differentiate the nn.Linear composition directly."""
import os

import torch

from . import _hip, _ops


def needs_grad(model, *tensors):
    if not torch.is_grad_enabled():
        return False
    return any(t is not None and t.requires_grad for t in tensors) or any(p.requires_grad for p in model.parameters())


def inputs_need_grad(*tensors):
    """True when a gradient is wanted w.r.t. points / rays / depths / view directions (pose or ray optimisation).  The fused
    training kernels differentiate w.r.t. the PARAMETERS only (the reference's own loop never asks for more:
    train_dexnerf_rgb.py:246-278), so such calls take the nn.Linear autograd path instead of silently getting no gradient."""
    return torch.is_grad_enabled() and any(t is not None and torch.is_tensor(t) and t.requires_grad for t in tensors)


def train_fused_ok(model):
    return model.fused_ok() and model.num_encoding_fn_xyz == 10 and _ops._precision != _hip.PREC_F16


def _slots(model, precision):
    """Piece slots of the saved activations / gradients (mirrors TrainLayout in csrc/mlp_layout.h)."""
    kpp = 16 if precision == _hip.PREC_BF16 else 8
    w, d = model.hidden_size, model.num_layers
    kxp = 64 // kpp   # fixed 64-wide xyz panel (kXyzPanel in csrc/mlp_layout.h)
    kdp = (((model.dim_dir + 15) // 16 * 16) // kpp) if model.use_viewdirs else 0
    kh = w // kpp
    s = {"xyz": 0, "dir": kxp, "layer1": kxp + kdp}
    s["trunk0"] = s["layer1"] + kh
    s["feat"] = s["trunk0"] + (d - 1) * kh
    s["dirout"] = s["feat"] + (kh if model.use_viewdirs else 0)
    g = {"dirout": 0}
    g["feat"] = (kh // 2) if model.use_viewdirs else 0
    g["trunk0"] = g["feat"] + (kh if model.use_viewdirs else 0)
    g["layer1"] = g["trunk0"] + (d - 1) * kh
    g["out"] = g["layer1"] + kh
    return s, g, kh


class FusedNetFn(torch.autograd.Function):
    """run_network for a FlexibleNeRFModel, differentiable w.r.t. the model parameters (points carry no gradient)."""

    @staticmethod
    def forward(ctx, model, pts, viewdirs, samples_per_ray, log_xyz, log_dir, *params):
        """`pts` (P,3) + `viewdirs` (N,3), or - when `samples_per_ray` is None - packed ray rows (N,11) + depths (N,S)."""
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:   # (inside forward the tensors themselves are detached views)
            raise RuntimeError("FusedNetFn differentiates w.r.t. the model parameters only; points / rays / view directions that "
                               "require grad must go through the nn.Linear composition (run_network does that by itself)")
        pk = model.packed(log_xyz, log_dir, train=True)
        prec = _ops.train_precision(pk)
        _ops.ensure_backward_stream(model, pk, prec)
        if samples_per_ray is None:
            out, act, masks = _ops.run_network_train(pk, None, None, None, rays=pts, z_vals=viewdirs, prec=prec)
        else:
            out, act, masks = _ops.run_network_train(pk, pts, viewdirs, samples_per_ray, prec=prec)
        ctx.model, ctx.pk, ctx.prec = model, pk, prec
        ctx.n_points = out.shape[0]
        ctx.save_for_backward(act, masks)
        ctx.sink = getattr(model, "_grad_sink", None)
        if ctx.sink is not None:
            ctx.sink.forward_issued()
        return out

    @staticmethod
    def backward(ctx, g_out):
        model, pk, n = ctx.model, ctx.pk, ctx.n_points
        act, masks = ctx.saved_tensors
        g_out = g_out.contiguous().float()
        grads = _ops.mlp_backward_data(pk, g_out, masks, n, prec=ctx.prec)
        slots, gslots, kh = _slots(model, pk.precision)
        w, d, dev = model.hidden_size, model.num_layers, g_out.device

        def rows(width):
            return torch.empty((n, width), dtype=torch.float32, device=dev)

        def act_hidden(slot, width, out=None, col0=0):
            out = rows(width) if out is None else out
            return _ops.mlp_unpack(pk, 0, act, n, slot, width, 0, out, col0)

        def grad_hidden(slot, width):
            return _ops.mlp_unpack(pk, 1, grads, n, slot, width, 0, rows(width))

        results = {}

        def put(mod, dy, x):
            results[mod] = (dy.t() @ x, dy.sum(0))

        if pk.precision == _hip.PREC_BF16 or os.environ.get("DEXNERF_FP32_DW", "kernel") != "gemm":
            # every layer's weight/bias gradient straight from the native buffers, one launch (MFMA kernel, fp32 atomics)
            mods = model.linear_modules()
            views = ctx.sink.views(model) if ctx.sink is not None else None
            if views is not None:
                # the parameters' `.grad` are views of a FlatGradBucket: accumulate straight into them (nothing goes back
                # through autograd's accumulation), then let the bucket start this network's all-reduce
                _ops.mlp_weight_grad_all_into(pk, act, grads, n, views, prec=ctx.prec)
                ctx.sink.backward_done()
                return (None,) * (6 + 2 * len(mods))
            res = _ops.mlp_weight_grad_all(pk, act, grads, n, [tuple(m.weight.shape) for m in mods], prec=ctx.prec)
            flat = []
            for d_w, d_b in res:
                flat.extend((d_w, d_b))
            return (None, None, None, None, None, None) + tuple(flat)

        pe_xyz = _ops.mlp_unpack(pk, 0, act, n, slots["xyz"], model.dim_xyz, 1, rows(model.dim_xyz))
        put(model.layer1, grad_hidden(gslots["layer1"], w), pe_xyz)
        x_prev = act_hidden(slots["layer1"], w)
        for i, layer in enumerate(model.layers_xyz):
            dy = grad_hidden(gslots["trunk0"] + i * kh, w)
            x = torch.cat((x_prev, pe_xyz), dim=-1) if i in model.skip_layers else x_prev
            put(layer, dy, x)
            x_prev = act_hidden(slots["trunk0"] + i * kh, w)
        if model.use_viewdirs:
            put(model.fc_feat, grad_hidden(gslots["feat"], w), x_prev)
            put(model.fc_alpha, g_out[:, 3:4], x_prev)
            x_dir = rows(w + model.dim_dir)
            act_hidden(slots["feat"], w, x_dir, 0)
            _ops.mlp_unpack(pk, 0, act, n, slots["dir"], model.dim_dir, 2, x_dir, w)
            put(model.layers_dir[0], grad_hidden(gslots["dirout"], w // 2), x_dir)
            put(model.fc_rgb, g_out[:, :3], act_hidden(slots["dirout"], w // 2))
        else:
            put(model.fc_out, g_out, x_prev)
        flat = []
        for m in model.linear_modules():
            flat.extend(results[m])
        return (None, None, None, None, None, None) + tuple(flat)


class RenderRaysTrainFn(torch.autograd.Function):
    """predict_and_render_radiance for one ray chunk in train mode (reference nerf/train_utils.py:92-202), differentiable
    w.r.t. the parameters of both networks: ONE C-ABI call forward (dn_render_rays_train) and one backward
    (dn_render_rays_backward) - or its two halves, fine network first, when a FlatGradBucket wants to start the fine network's
    all-reduce while the coarse half runs.  Same kernels, same order, same bits as the stage-by-stage composition."""

    @staticmethod
    def forward(ctx, model_c, model_f, rays, cfg, draws, thres, logs, *params):
        num_coarse, num_fine, lindisp, noise_std, white = cfg
        pc = model_c.packed(*logs, train=True)
        pf = model_f.packed(*logs, train=True) if model_f is not None else None
        prec = _ops.train_precision(pc)
        if pf is not None and _ops.train_precision(pf) != prec:
            prec = pc.precision   # one of the two networks is outside the 8-bit-saved-tensor kernels: both train in plain bf16
        for model, pk in ((model_c, pc), (model_f, pf)):
            if model is not None:
                _ops.ensure_backward_stream(model, pk, prec)
        maps, saved = _ops.render_rays_train(pc, pf, rays, num_coarse, num_fine, lindisp, noise_std, white, thres, draws, prec=prec)
        ctx.models, ctx.packed, ctx.saved = (model_c, model_f), (pc, pf), saved
        ctx.sinks = tuple(getattr(m, "_grad_sink", None) if m is not None else None for m in (model_c, model_f))
        for sink in ctx.sinks:
            if sink is not None:
                sink.forward_issued()
        ctx.set_materialize_grads(False)
        dex = maps[6]
        if dex is not None:
            ctx.mark_non_differentiable(dex)
        return maps

    @staticmethod
    def backward(ctx, g_rgb_c, g_depth_c, g_acc_c, g_rgb_f, g_depth_f, g_acc_f, _g_dex=None):
        model_c, model_f = ctx.models
        pc, pf = ctx.packed
        saved = ctx.saved
        fine = model_f is not None and saved["nf"] > 0
        g_c, g_f = (g_rgb_c, g_depth_c, g_acc_c), (g_rgb_f, g_depth_f, g_acc_f)
        sink_c, sink_f = ctx.sinks
        views_c = sink_c.views(model_c) if sink_c is not None else None
        views_f = sink_f.views(model_f) if (fine and sink_f is not None) else None
        n_c = 2 * len(model_c.linear_modules())
        n_f = 2 * len(model_f.linear_modules()) if model_f is not None else 0
        if views_c is not None and (views_f is not None or not fine):
            # gradients accumulate straight into the bucket's views; the fine half first so that its exchange is in flight
            # while the coarse half runs
            keep = []
            if fine:
                keep.append(_ops.render_rays_backward(pc, pf, saved, g_c, g_f, views_c, views_f, nets=2))
                sink_f.backward_done()
            keep.append(_ops.render_rays_backward(pc, pf, saved, g_c, g_f, views_c, views_f, nets=1))
            sink_c.backward_done()
            return (None,) * (7 + n_c + n_f)

        def fresh(model):
            shapes = [tuple(m.weight.shape) for m in model.linear_modules()]
            flat = torch.zeros(sum(o * i + o for o, i in shapes), dtype=torch.float32, device=saved["rays"].device)
            out, off = [], 0
            for o, i in shapes:
                out.append((flat[off:off + o * i].view(o, i), flat[off + o * i:off + o * i + o]))
                off += o * i + o
            return out
        views_c = fresh(model_c)
        views_f = fresh(model_f) if fine else None
        _ops.render_rays_backward(pc, pf, saved, g_c, g_f, views_c, views_f, nets=3)
        grads = [t for pair in views_c for t in pair]
        if model_f is not None:
            grads += [t for pair in views_f for t in pair] if fine else [None] * n_f
        return (None,) * 7 + tuple(grads)


def render_rays_train(model_c, model_f, rays, cfg, draws, thres, logs):
    """The fused training path of predict_and_render_radiance (both networks covered by the training kernels)."""
    params = []
    for model in (model_c, model_f):
        if model is not None:
            for m in model.linear_modules():
                params += [m.weight, m.bias]
    return RenderRaysTrainFn.apply(model_c, model_f, rays, cfg, draws, thres, logs, *params)


def mlp_encoded(model, x):
    """FlexibleNeRFModel.forward(x) on already-embedded device rows."""
    if needs_grad(model, x):
        return model._forward_modules(x)
    return _ops.mlp_forward_encoded(model.packed(), x)


def _torch_encoding(x, num_fns, log_sampling):
    """positional_encoding (reference nerf/nerf_helpers.py:115-159) as differentiable torch ops on the device."""
    if log_sampling:
        freqs = 2.0 ** torch.linspace(0.0, num_fns - 1, num_fns, dtype=x.dtype, device=x.device)
    else:
        freqs = torch.linspace(2.0 ** 0.0, 2.0 ** (num_fns - 1), num_fns, dtype=x.dtype, device=x.device)
    parts = [x]
    for f in freqs:
        parts += [torch.sin(x * f), torch.cos(x * f)]
    return torch.cat(parts, dim=-1)


def _modules_on_points(model, pts, viewdirs, log_xyz, log_dir):
    """(N,S,3) points + (N,3) view directions through torch encodings + the nn.Linear composition: (N*S, 4), differentiable
    w.r.t. everything."""
    n, s = pts.shape[0], pts.shape[1]
    emb = _torch_encoding(pts.reshape(-1, 3), model.num_encoding_fn_xyz, log_xyz)
    if model.use_viewdirs:
        vd = viewdirs.reshape(n, 1, 3).expand(n, s, 3).reshape(-1, 3)
        emb = torch.cat((emb, _torch_encoding(vd, model.num_encoding_fn_dir, log_dir)), dim=-1)
    return model._forward_modules(emb)


def run_network_fused_rays(model, rays, z_vals, log_xyz=True, log_dir=True):
    """run_network on packed ray rows (N, 8|11) + depths (N, S): the sample points ro + rd * z are formed inside the
    kernel (reference train_utils.py:136,177 materialise them).  Returns (N, S, 4)."""
    n, s = z_vals.shape
    if inputs_need_grad(rays, z_vals):
        ro, rd = rays[..., :3], rays[..., 3:6]
        pts = ro[..., None, :] + rd[..., None, :] * z_vals[..., :, None]
        return _modules_on_points(model, pts, rays[..., -3:] if model.use_viewdirs else None, log_xyz, log_dir).reshape(n, s, 4)
    if needs_grad(model) and train_fused_ok(model):
        params = []
        for m in model.linear_modules():
            params += [m.weight, m.bias]
        return FusedNetFn.apply(model, rays, z_vals, None, log_xyz, log_dir, *params).reshape(n, s, 4)
    return _ops.run_network_rays(model.packed(log_xyz, log_dir), rays, z_vals)


def run_network_fused(model, pts, viewdirs, samples_per_ray, log_xyz=True, log_dir=True):
    """run_network on raw points: positional encoding + MLP in one kernel; differentiable w.r.t. the parameters.  Inputs
    that require grad (pose / ray optimisation) take the differentiable torch composition instead."""
    if inputs_need_grad(pts, viewdirs):
        return _modules_on_points(model, pts.reshape(-1, samples_per_ray, 3), viewdirs, log_xyz, log_dir)
    if needs_grad(model) and train_fused_ok(model):
        params = []
        for m in model.linear_modules():
            params += [m.weight, m.bias]
        return FusedNetFn.apply(model, pts, viewdirs, samples_per_ray, log_xyz, log_dir, *params)
    return _ops.run_network_pts(model.packed(log_xyz, log_dir), pts, viewdirs, samples_per_ray)
