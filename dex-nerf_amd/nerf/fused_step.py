"""The training iteration of the Dex-NeRF loop (reference train_dexnerf_rgb.py:223-289: view and pixel draw -> ray rows -> coarse +
fine render -> mse + mse -> backward -> Adam) with nothing in it but this library's kernels - 16 launches on the as-shipped nets:

    dn_select_rays_draw      view + pixels drawn without replacement on the device, packed ray rows, target pixels   (1 kernel)
    dn_mlp_pack_train_pair   both weight streams of both networks                                                     (2 kernels)
    dn_render_rays_train     coarse depths / net / composite, resampling, fine net / composite; the jitter, the resampling u and
                             the density noise drawn inside the kernels that consume them                             (6 kernels)
    dn_mse2_loss             loss, both MSEs and the two upstream gradients                                            (1 kernel)
    dn_render_rays_backward  composite backward + backward-data chain per network, then ONE weight-gradient launch for the
                             layers of both (one rank; with several ranks the networks are done one at a time so that the
                             fine network's all-reduce overlaps the coarse half)                                       (5 kernels)
    dn_adam_step             nerf.FlatAdam: step, learning-rate schedule and gradient clearing                         (1 kernel)

No autograd graph, no ATen elementwise / reduction / RNG launches (profiles/r03_as_shipped_kernel_summary.md).  The gradients land in
a parallel.FlatGradBucket (the `.grad` tensors of the parameters).  GraphedTrainStep replays the iteration as HIP graphs (three
around the exchange when world > 1).  The explicit-draw path (predict_and_render_radiance under autograd, draws as tensors) stays
what the parity tests drive."""
import os

import torch

from . import _ops
from ._train import train_fused_ok
from .train_utils import _fusable


_SPLIT_BACKWARD = os.environ.get("DEXNERF_SPLIT_BACKWARD", "") == "1"   # developer switch (A/B timing): the two networks' backward as two calls even with one rank


class FusedTrainStep:
    def __init__(self, model_coarse, model_fine, selector, options, bucket, encode_position_fn, encode_direction_fn, num_rays, seed=0,
                 luminance=False, first_iteration=0, draw_view=False):
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
        self.draw_view = bool(draw_view)   # True: the iteration's training view is drawn in the kernel too (else: selector.view)
        self.zero_in_step = True     # False: the optimizer leaves the gradient bucket cleared (FlatAdam(zero_grads=True))

    @staticmethod
    def applicable(model_coarse, model_fine, options, encode_position_fn, encode_direction_fn, num_rays):
        """Both networks on the fused training kernels, a fine pass, one ray chunk, world-space rays."""
        opt = options.nerf.train
        return (model_fine is not None and int(opt.num_fine) > 0 and int(num_rays) <= int(opt.chunksize)
                and getattr(options.dataset, "no_ndc", True) is not False
                and _fusable(model_coarse, encode_position_fn, encode_direction_fn) and _fusable(model_fine, encode_position_fn, encode_direction_fn)
                and train_fused_ok(model_coarse) and train_fused_ok(model_fine) and model_coarse.use_viewdirs and model_fine.use_viewdirs)

    def forward_backward(self):
        """One iteration up to (and excluding) the end of the gradient exchange and the optimizer step.  Returns the device tensor
        [loss, mse_coarse, mse_fine]; the parameter gradients are in the bucket (world > 1: each network's all-reduce is started the
        moment its backward is enqueued - FlatGradBucket.segment_ready - and finished by bucket.all_reduce_mean())."""
        from .parallel import world_info
        mc, mf = self.models
        mc._grad_sink.forward_issued(); mf._grad_sink.forward_issued()
        if self.zero_in_step:
            self.bucket.zero()
        if world_info()[1] == 1 and not _SPLIT_BACKWARD:
            self.forward_and_fine_backward(_zero=False, both=True)    # one rank: nothing to overlap, one weight-gradient launch for both networks
            mf._grad_sink.backward_done(); mc._grad_sink.backward_done()
            return self.loss3
        self.forward_and_fine_backward(_zero=False)
        mf._grad_sink.backward_done()     # (world > 1: the fine network's all-reduce starts here, under the coarse half)
        self.coarse_backward()
        mc._grad_sink.backward_done()
        return self.loss3

    # The two halves a data-parallel loop replays as separate HIP graphs around the fine network's exchange (GraphedTrainStep): only
    # this library's launches and one memset - no bucket bookkeeping, no collective.
    def forward_and_fine_backward(self, _zero=True, both=False):
        """both=True: the coarse network's backward too, in the same call - dn_render_rays_backward then forms the weight gradients
        of both networks in ONE launch (dn_mlp_weight_grad_pair); coarse_backward() must not follow."""
        mc, mf = self.models
        sel = self.selector
        rays, target = _ops.select_rays_draw(sel.height, sel.width, sel.cams, None if self.draw_view else sel.view, sel.near, sel.far,
                                             self.rng_state, self.num_rays, sel.images)
        pc, pf, prec = _ops.pack_train_pair(mc, mf, self.logs)
        maps, saved = _ops.render_rays_train(pc, pf, rays, self.nc, self.nf, self.lindisp, self.noise_std, self.white, [], None, prec=prec,
                                             rng_state=self.rng_state, perturb=self.perturb)
        self.loss3, g_c, g_f = _ops.mse2_loss(maps[0], maps[3], target, self.luminance, self.rng_state)
        if _zero and self.zero_in_step:
            self.bucket.flat.zero_()
        views_c, views_f = mc._grad_sink.views(mc), mf._grad_sink.views(mf)
        if views_c is None or views_f is None:
            raise RuntimeError("FusedTrainStep: a parameter's .grad is no longer the FlatGradBucket's view")
        none3 = (None, None, None)
        if both:
            keep = [_ops.render_rays_backward(pc, pf, saved, (g_c, None, None), (g_f, None, None), views_c, views_f, nets=3)]
        else:
            keep = [_ops.render_rays_backward(pc, pf, saved, none3, (g_f, None, None), views_c, views_f, nets=2)]
        self._half = (pc, pf, saved, g_c, views_c, views_f)
        self._keep = (keep, saved, maps, rays, target, g_c, g_f)   # alive until the next call (stream-ordered allocator)

    def coarse_backward(self):
        pc, pf, saved, g_c, views_c, views_f = self._half
        none3 = (None, None, None)
        self._keep[0].append(_ops.render_rays_backward(pc, pf, saved, (g_c, None, None), none3, views_c, views_f, nets=1))


class GraphedTrainStep:
    """A FusedTrainStep + the gradient exchange + the optimizer step, replayed as HIP graphs (reference loop:
    train_dexnerf_rgb.py:229-289 - one optimizer step per drawn batch).

    world == 1: ONE graph (draw .. Adam).  world > 1: the collective stays outside the graphs -

        graph A  pixel draw, ray rows, coarse + fine render, loss head, fine network's backward
        all-reduce of the fine segment, asynchronous on torch.distributed's stream ..........  overlapped with
        graph B  coarse network's backward
        all-reduce of the coarse segment; both waited for on the compute stream
        graph C  optimizer step (fused Adam over the flat parameter list)

    RCCL averages inside the collective (ReduceOp.AVG); gloo (CPU tests, one-GPU rehearsals) sums and one division follows.
    The first `eager_iterations` calls run eagerly (they initialise the optimizer moments and the packed weight streams); the next
    one captures and replays.  If capture fails the loop goes on eagerly and `fallback_reason` says why."""

    def __init__(self, fused, optimizer, eager_iterations=3, use_graphs=True):
        self.fused, self.opt, self.bucket = fused, optimizer, fused.bucket
        if getattr(optimizer, "zero_grads", False):
            self.bucket.flat.zero_()
            fused.zero_in_step = False      # every step ends with the bucket cleared
        self.eager_left = int(eager_iterations)
        self.use_graphs = bool(use_graphs)
        self.graphs = None
        self.fallback_reason = None

    def _eager(self):
        self.fused.forward_backward()
        self.bucket.all_reduce_mean()
        self.opt.step()

    def _capture(self):
        from .parallel import world_info
        world = world_info()[1]
        torch.cuda.synchronize()
        # thread-local capture mode: with a process group alive, torch.distributed's watchdog thread queries events while this
        # thread captures - under the default (global) mode that is a capture error
        mode = dict(capture_error_mode="thread_local")
        if world == 1:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, **mode):
                self.fused.forward_and_fine_backward(both=not _SPLIT_BACKWARD)
                if _SPLIT_BACKWARD:
                    self.fused.coarse_backward()
                self.opt.step()
            return [g]
        ga, gb, gc = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(ga, **mode):
            self.fused.forward_and_fine_backward()
        with torch.cuda.graph(gb, pool=ga.pool(), **mode):     # (reads what graph A allocated: one memory pool, replayed in capture order)
            self.fused.coarse_backward()
        with torch.cuda.graph(gc, pool=ga.pool(), **mode):
            self.opt.step()
        return [ga, gb, gc]

    def _replay(self):
        import torch.distributed as dist
        from .models import mark_parameters_updated
        from .parallel import _avg_in_collective, world_info
        if len(self.graphs) == 1:
            self.graphs[0].replay()
        else:
            ga, gb, gc = self.graphs
            avg = _avg_in_collective()
            op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
            # each network's segment through its own sink (the bucket's module order is the caller's choice)
            sinks = [m._grad_sink for m in self.fused.models]
            assert all(s.bucket is self.bucket for s in sinks), "GraphedTrainStep: both networks' gradients live in this step's bucket"
            ga.replay()
            w_fine = dist.all_reduce(self.bucket.segment(sinks[1].idx), op=op, async_op=True)
            gb.replay()
            w_coarse = dist.all_reduce(self.bucket.segment(sinks[0].idx), op=op, async_op=True)
            w_fine.wait(); w_coarse.wait()
            if not avg:
                self.bucket.flat.div_(world_info()[1])
            gc.replay()
        mark_parameters_updated()      # the replayed optimizer step ran no Python hook

    def step(self):
        """One training iteration.  The loss of the iteration is in self.fused.loss3 (device)."""
        if self.graphs is not None:
            return self._replay()
        if self.eager_left > 0 or not self.use_graphs:
            self.eager_left -= 1
            return self._eager()
        try:
            self.graphs = self._capture()          # a capture only records ...
        except Exception as exc:  # noqa: BLE001
            self.graphs, self.use_graphs = None, False
            self.fallback_reason = f"{type(exc).__name__}: {exc}"
            torch.cuda.synchronize()
            return self._eager()
        return self._replay()                      # ... this iteration's step runs here
