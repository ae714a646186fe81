"""The training iteration of the Dex-NeRF loop (reference train_dexnerf_rgb.py:229-289: pixel draw -> ray rows -> coarse + fine
render -> mse + mse -> backward) with nothing between the pixel draw and the optimizer but this library's kernels:

    dn_select_rays_draw      pixels drawn without replacement on the device + packed ray rows + target pixels      (1 kernel)
    dn_render_rays_train     coarse depths / net / composite, resampling, fine net / composite; the jitter, the resampling u and
                             the density noise drawn inside the kernels that consume them                           (7 kernels + packs)
    dn_mse2_loss             loss, both MSEs and the two upstream gradients                                          (1 kernel)
    dn_render_rays_backward  composite backward, backward-data chain, weight gradients - fine network, then coarse  (6 kernels)

No autograd graph, no ATen elementwise / reduction / RNG launches: the as-shipped configuration (4 x 128 nets, 1024 rays, 64 + 64
samples) is bound by those (profiles/r03_as_shipped_kernel_summary.md).  The gradients land in a parallel.FlatGradBucket (the
`.grad` tensors of the parameters); the caller exchanges them (world > 1) and steps its optimizer.  The explicit-draw path
(predict_and_render_radiance under autograd, draws as tensors) stays what the parity tests drive."""
import torch

from . import _ops
from ._train import train_fused_ok
from .train_utils import _fusable


class FusedTrainStep:
    def __init__(self, model_coarse, model_fine, selector, options, bucket, encode_position_fn, encode_direction_fn, num_rays, seed=0,
                 luminance=False, first_iteration=0):
        opt = options.nerf.train
        self.models = (model_coarse, model_fine)
        self.selector, self.bucket = selector, bucket
        self.num_rays = int(num_rays)
        self.nc, self.nf = int(opt.num_coarse), int(opt.num_fine)
        self.lindisp, self.perturb = bool(opt.lindisp), bool(opt.perturb)
        self.noise_std, self.white = float(opt.radiance_field_noise_std), bool(opt.white_background)
        self.luminance = bool(luminance)
        self.logs = (encode_position_fn.log_sampling, encode_direction_fn.log_sampling if model_coarse.use_viewdirs else True)
        if not self.applicable(model_coarse, model_fine, options, encode_position_fn, encode_direction_fn, num_rays):
            raise ValueError("FusedTrainStep: configuration outside the fused training kernels (see FusedTrainStep.applicable)")
        dev = next(model_coarse.parameters()).device
        self.rng_state = _ops.new_rng_state(seed, dev, first_iteration)
        self.loss3 = None

    @staticmethod
    def applicable(model_coarse, model_fine, options, encode_position_fn, encode_direction_fn, num_rays):
        """Both networks on the fused training kernels, a fine pass, one ray chunk, world-space rays."""
        opt = options.nerf.train
        return (model_fine is not None and int(opt.num_fine) > 0 and int(num_rays) <= int(opt.chunksize)
                and getattr(options.dataset, "no_ndc", True) is not False
                and _fusable(model_coarse, encode_position_fn, encode_direction_fn) and _fusable(model_fine, encode_position_fn, encode_direction_fn)
                and train_fused_ok(model_coarse) and train_fused_ok(model_fine) and model_coarse.use_viewdirs and model_fine.use_viewdirs)

    def forward_backward(self):
        """One iteration up to (and excluding) the gradient exchange and the optimizer step.  Returns the device tensor
        [loss, mse_coarse, mse_fine]; the parameter gradients are in the bucket."""
        mc, mf = self.models
        sel = self.selector
        rays, target = _ops.select_rays_draw(sel.height, sel.width, sel.cams, sel.view, sel.near, sel.far, self.rng_state, self.num_rays,
                                             sel.images)
        pc, pf, prec = _ops.pack_train_pair(mc, mf, self.logs)
        maps, saved = _ops.render_rays_train(pc, pf, rays, self.nc, self.nf, self.lindisp, self.noise_std, self.white, [], None, prec=prec,
                                             rng_state=self.rng_state, perturb=self.perturb)
        self.loss3, g_c, g_f = _ops.mse2_loss(maps[0], maps[3], target, self.luminance, self.rng_state)
        sink_c, sink_f = mc._grad_sink, mf._grad_sink
        sink_c.forward_issued(); sink_f.forward_issued()
        self.bucket.zero()
        views_c, views_f = sink_c.views(mc), sink_f.views(mf)
        if views_c is None or views_f is None:
            raise RuntimeError("FusedTrainStep: a parameter's .grad is no longer the FlatGradBucket's view")
        none3 = (None, None, None)
        keep = [_ops.render_rays_backward(pc, pf, saved, none3, (g_f, None, None), views_c, views_f, nets=2)]
        sink_f.backward_done()     # (world > 1: the fine network's all-reduce starts here, under the coarse half)
        keep.append(_ops.render_rays_backward(pc, pf, saved, (g_c, None, None), none3, views_c, views_f, nets=1))
        sink_c.backward_done()
        self._keep = (keep, saved, maps, rays, target, g_c, g_f)   # alive until the next call (stream-ordered allocator)
        return self.loss3
