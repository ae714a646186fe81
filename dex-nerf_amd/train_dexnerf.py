#!/usr/bin/env python3
"""Build-owned training / evaluation driver (counterpart of the reference's train_dexnerf_rgb.py:178-457 and
eval_nerf.py:166-206) on the MI355X-native `nerf` package.

What it keeps from the reference loop: one random training view per iteration, `num_random_rays` random pixels,
loss = MSE(rgb_coarse) + MSE(rgb_fine) (or, with --ir, on the luminance of both: train_nerf_ir.py:260-263), PSNR = mse2psnr(loss), Adam with the per-iteration exponential LR
`lr0 * factor^(i / (lr_decay * 1000))`, validation renders with the Dex threshold sweep
(`m_thres_cand = arange(5, m_thres + 5, 5)`, best threshold by mean |depth error| on the (0, 1.25 m] mask-style
validity mask), and the checkpoint dict keys (iter, model_*_state_dict, optimizer_state_dict, loss, psnr).
What it adds: data-parallel training (one process per GPU under torch.distributed.run; one flat-bucket gradient
all-reduce per step) and a synthetic "teacher" scene, because no dataset ships with the reference: a fixed random
coarse/fine FlexibleNeRFModel pair is rendered from spherical poses and the student learns to reproduce it.

    python dex-nerf_amd/train_dexnerf.py --iters 2000 --size 100 --precision bf16
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 dex-nerf_amd/train_dexnerf.py ...
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import nerf  # noqa: E402
from nerf import parallel, synthetic as syn  # noqa: E402


def make_cfg(args):
    def mode(train):
        return dict(chunksize=args.chunksize, lindisp=False, num_coarse=args.num_coarse, num_fine=args.num_fine,
                    perturb=train, radiance_field_noise_std=args.noise_std if train else 0.0, white_background=False)
    return nerf.CfgNode(dict(dataset=dict(near=args.near, far=args.far, no_ndc=not getattr(args, "ndc", False)),
                             nerf=dict(use_viewdirs=True, train=mode(True), validation=mode(False))))


def build_models(kw, dev, state=None):
    out = []
    for i in range(2):
        m = nerf.models.FlexibleNeRFModel(**kw)
        if state is not None:
            m.load_state_dict({k: torch.from_numpy(v) for k, v in state[i].items()})
        out.append(m.to(dev))
    return out


def render_view(models, cfg, pose, k_mat, hw, ex, ed, thres, mode="validation"):
    h, w = hw
    ro, rd = nerf.get_ray_bundle(h, w, float(k_mat[0, 0]), pose, k_mat)
    with torch.no_grad():
        return nerf.run_one_iter_of_nerf(h, w, float(k_mat[0, 0]), models[0], models[1], ro, rd, cfg, mode=mode,
                                         encode_position_fn=ex, encode_direction_fn=ed, m_thres_cand=thres)


def dex_sweep(outputs, depth_gt, thres, gt_hi=6.0):
    """Pick the Dex threshold with the smallest mean |depth error| (reference train_dexnerf_rgb.py:391-408): all
    candidates in one kernel + one copy (nerf.dex_error_sweep).  Ground mask 0 < gt < gt_hi (the reference: 1.25 m for
    its table-top scenes; the built-in synthetic scene spans (0, 6) m)."""
    best, errs = nerf.dex_error_sweep(depth_gt, list(outputs[6:]), gt_lo=0.0, gt_hi=gt_hi)
    best = int(np.argmin([e["depth_abs_err"] for e in errs])) if best is None else best
    return thres[best], errs[best]


def synthetic_dataset(args, kw, cfg, ex, ed, thres, dev):
    """No dataset ships with the reference: a fixed random coarse/fine FlexibleNeRFModel pair ("teacher") is rendered from
    `views` + 1 spherical poses; the last pose is held out."""
    hw = (args.size, args.size)
    k_mat = torch.from_numpy(syn.intrinsic(*hw)).to(dev)
    poses = [torch.from_numpy(syn.scene_pose(i, n_views=args.views + 1)).to(dev) for i in range(args.views + 1)]
    teacher = build_models(kw, dev, (syn.synth_state_dict(42, sigma_bias=-150.0, **kw), syn.synth_state_dict(43, sigma_bias=-20.0, **kw)))
    images, depths = [], []
    for pose in poses:
        out = render_view(teacher, cfg, pose, k_mat, hw, ex, ed, thres)
        images.append(out[3].reshape(-1, 3))
        depths.append(out[4].reshape(hw))
    return dict(hw=hw, poses=poses, intrinsics=[k_mat] * len(poses), images=images, depths=depths,
                train=list(range(args.views)), val=args.views, mask_hi=6.0)


def messytable_dataset(args, dev):
    """A scene directory in the reference's MessyTable / Dex-NeRF layout (nerf/load_messytable.py; `half_res` = the
    fork's 270 x 480 training resolution with the principal point pinned to (240, 135))."""
    imgs, poses, _, hwf, i_split, intrinsics, depths = nerf.load_messytable_data(
        args.messytable, half_res=True, imgname=args.imgname, is_real_rgb=args.real_rgb)
    hw = (int(hwf[0]), int(hwf[1]))
    n = imgs.shape[0]
    val = int(i_split[1][0]) if len(i_split[1]) else int(i_split[0][-1])
    return dict(hw=hw, poses=[poses[i].to(dev) for i in range(n)], intrinsics=[intrinsics[i].to(dev) for i in range(n)],
                images=[imgs[i, ..., :3].reshape(-1, 3).to(dev) for i in range(n)], depths=[depths[i].to(dev) for i in range(n)],
                train=[int(i) for i in i_split[0]], val=val, mask_hi=1.25)


def llff_dataset(args, dev):
    """A forward-facing capture in the LLFF layout (nerf/llff.py; reference nerf/load_llff.py + the LLFF branch of
    train_nerf_rgb.py:70-95): camera-to-world poses in NeRF's (right, up, back) frame with a shared pinhole (f, W/2, H/2).
    The kernels take the fork's convention - a world-to-camera extrinsic in OpenCV axes plus a 3x3 K (nerf_helpers.py:67-112:
    dir = [(i-cx)/fx, (j-cy)/fx, 1], rd = inv(E[:3,:3]) dir, ro = inv(E)[:3,3]) - so E = inv([R diag(1,-1,-1) | t])."""
    images, poses, bds, _, i_test = nerf.load_llff_data(args.llff, factor=args.llff_factor)
    h, w, f = int(poses[0, 0, 4]), int(poses[0, 1, 4]), float(poses[0, 2, 4])
    k_mat = torch.tensor([[f, 0.0, w * 0.5], [0.0, f, h * 0.5], [0.0, 0.0, 1.0]], dtype=torch.float32, device=dev)
    flip = np.diag([1.0, -1.0, -1.0])
    extr = []
    for c2w in poses[:, :3, :4].astype(np.float64):
        m = np.eye(4)
        m[:3, :3], m[:3, 3] = c2w[:, :3] @ flip, c2w[:, 3]
        extr.append(torch.from_numpy(np.linalg.inv(m).astype(np.float32)).to(dev))
    n = images.shape[0]
    hold = list(range(n))[::args.llffhold] if args.llffhold > 0 else [int(i_test)]
    train = [i for i in range(n) if i not in hold] or [int(i_test)]
    return dict(hw=(h, w), poses=extr, intrinsics=[k_mat] * n, images=[torch.from_numpy(images[i].reshape(-1, 3)).to(dev) for i in range(n)],
                depths=[None] * n, train=train, val=hold[0], mask_hi=None, focal=f,
                bounds=(float(bds.min()) * 0.9, float(bds.max()) * 1.0))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--size", type=int, default=100, help="image height = width of the synthetic views")
    ap.add_argument("--views", type=int, default=20)
    ap.add_argument("--num-random-rays", type=int, default=2048)
    ap.add_argument("--num-coarse", type=int, default=64)
    ap.add_argument("--num-fine", type=int, default=128)
    ap.add_argument("--chunksize", type=int, default=65536)
    ap.add_argument("--noise-std", type=float, default=0.2)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--lr", type=float, default=5e-4)
    ap.add_argument("--lr-decay", type=int, default=250)
    ap.add_argument("--lr-decay-factor", type=float, default=0.1)
    ap.add_argument("--m-thres", type=int, default=100)
    ap.add_argument("--ir", action="store_true", help="IR head of train_nerf_ir.py / train_dexnerf_ir.py: MSE on the luminance "
                                                      "0.299 r + 0.587 g + 0.114 b of prediction and target (:260-263)")
    ap.add_argument("--s8-grad-scale", type=float, default=None,
                    help="bf16: power of two the 8-bit saved layer gradients are scaled by; default 0 = chosen per launch from the "
                         "largest upstream gradient (nerf.set_s8_grad_scale)")
    ap.add_argument("--atomic-weight-gradients", action="store_true",
                    help="add the weight-gradient partials with fp32 atomics (no scratch buffer, a few percent faster on small nets) instead "
                         "of the fixed-order reduction: a run is then no longer reproducible bit for bit")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "bf16-s8", "bf16-s16", "fp32"],
                    help="bf16 (= bf16-s8): bf16 kernels, the tensors saved for the backward at 8 bits; bf16-s16: at 16 bits (nerf.set_precision)")
    ap.add_argument("--validate-every", type=int, default=500)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--save", default="", help="checkpoint path (reference dict format)")
    ap.add_argument("--load-checkpoint", default="")
    ap.add_argument("--messytable", default="", help="train on a scene directory in the reference's MessyTable / Dex-NeRF "
                                                     "layout instead of the built-in synthetic scene")
    ap.add_argument("--llff", default="", help="train on a forward-facing capture in the LLFF layout (poses_bounds.npy + images[_N]/); "
                                               "rays are warped to NDC like the reference's LLFF configs unless --no-ndc")
    ap.add_argument("--llff-factor", type=int, default=8, help="image downsampling factor (images_<factor>/; the reference default)")
    ap.add_argument("--llffhold", type=int, default=8, help="every N-th view is held out (0: only the view nearest the average pose)")
    ap.add_argument("--no-ndc", action="store_true", help="LLFF: keep world-space rays with the capture's depth bounds")
    ap.add_argument("--imgname", default="0128_irL_kuafu_half.png")
    ap.add_argument("--real-rgb", action="store_true")
    ap.add_argument("--near", type=float, default=None, help="default 2 (synthetic scene) / 0.3 (MessyTable)")
    ap.add_argument("--far", type=float, default=None, help="default 6 (synthetic scene) / 4 (MessyTable)")
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--autograd-step", action="store_true",
                    help="run the iteration as torch ops + autograd over the fused kernels (torch.randperm / rand / randn draws, mse_loss, "
                         "loss.backward()) instead of nerf.FusedTrainStep")
    ap.add_argument("--torch-adam", action="store_true",
                    help="step torch.optim.Adam (fused, multi-tensor) instead of the flat one-launch Adam of this build (nerf.FlatAdam)")
    ap.add_argument("--no-hip-graph", action="store_true",
                    help="launch every kernel of an iteration from Python instead of replaying one captured HIP graph "
                         "(single-GPU runs capture by default: the as-shipped 4x128 nets at 1024 rays are launch-bound)")
    args = ap.parse_args(argv)
    user_bounds = args.near is not None or args.far is not None
    args.ndc = bool(args.llff) and not args.no_ndc
    if args.ndc:
        args.near, args.far = 0.0, 1.0        # the NDC cube (the reference's LLFF configs)
    if args.near is None:
        args.near = 0.3 if args.messytable else 2.0
    if args.far is None:
        args.far = 4.0 if args.messytable else 6.0

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)   # (rehearsals: ranks may share a GPU)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm; DEXNERF_DIST_BACKEND=gloo lets the data-parallel loop be rehearsed with several ranks on
        # one GPU (RCCL refuses two ranks on the same device)
        backend = os.environ.get("DEXNERF_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    np.random.seed(args.seed + rank)          # reference seeds np + torch from cfg.experiment.randomseed (:97-99)
    torch.manual_seed(args.seed + rank)
    nerf.set_precision(args.precision)
    if args.s8_grad_scale is not None:
        nerf.set_s8_grad_scale(args.s8_grad_scale)
    from nerf import _ops as _nerf_ops
    _nerf_ops.set_deterministic_weight_gradients(not args.atomic_weight_gradients)
    s8_warned, s8_saturated_max = False, 0.0

    kw = dict(num_layers=args.layers, hidden_size=args.width, skip_connect_every=4, num_encoding_fn_xyz=10,
              num_encoding_fn_dir=4, use_viewdirs=True)
    cfg = make_cfg(args)
    ex, ed = nerf.get_embedding_function(10, True, True), nerf.get_embedding_function(4, True, True)
    thres = np.arange(5, args.m_thres + 5, 5)
    if args.llff:
        data = llff_dataset(args, dev)
        if not args.ndc and not user_bounds:      # world-space rays: the capture's own depth bounds
            args.near, args.far = data["bounds"]
            cfg.dataset.near, cfg.dataset.far = data["bounds"]
    else:
        data = messytable_dataset(args, dev) if args.messytable else synthetic_dataset(args, kw, cfg, ex, ed, thres, dev)
    hw, poses, intrinsics, images, depths = data["hw"], data["poses"], data["intrinsics"], data["images"], data["depths"]
    held_out = data["val"]

    torch.manual_seed(args.seed)              # identical student init on every rank
    student = build_models(kw, dev)
    parallel.broadcast_parameters(student)
    torch.manual_seed(args.seed + 1000 + rank)
    bucket = parallel.FlatGradBucket(student)
    use_graph = not args.no_hip_graph         # (world > 1: graphs around the exchange, nerf.GraphedTrainStep; the autograd step only with one rank)
    # The whole iteration on this library's kernels (nerf.FusedTrainStep: pixel draw, jitter, resampling and density noise drawn
    # inside the kernels, loss head + upstream gradients in one launch, no autograd graph) wherever the fused training kernels
    # cover the configuration; --autograd-step keeps the torch composition (torch.randperm / rand / randn, autograd).
    fused_ok = (not args.autograd_step and not args.ndc
                and nerf.FusedTrainStep.applicable(student[0], student[1], cfg, ex, ed, args.num_random_rays))
    flat_adam = fused_ok and not args.torch_adam
    lr_t = torch.tensor(args.lr, dtype=torch.float32, device=dev)   # (torch's Adam) a device scalar: the schedule is applied by fill_()
    if flat_adam:
        # Adam over ONE flat parameter buffer: one launch, the learning-rate schedule evaluated inside it, the gradients cleared
        # in the same pass (reference: torch.optim.Adam + zero_grad + the schedule, train_dexnerf_rgb.py:146-148, 280-289)
        opt = nerf.FlatAdam(bucket, lr=args.lr, lr_decay_factor=args.lr_decay_factor, lr_decay_steps=args.lr_decay * 1000, zero_grads=True)
    else:
        opt = torch.optim.Adam(bucket.params, lr=lr_t, fused=True, capturable=True)
    start = 0
    if args.load_checkpoint:
        ck = torch.load(args.load_checkpoint, map_location=dev)
        student[0].load_state_dict(ck["model_coarse_state_dict"])
        student[1].load_state_dict(ck["model_fine_state_dict"])
        nerf.models.mark_parameters_updated()
        opt.load_state_dict(ck["optimizer_state_dict"])
        if not flat_adam:
            for group in opt.param_groups:
                group["lr"] = lr_t
        start = ck["iter"]
    train_ids = data["train"][rank::world] or [data["train"][rank % len(data["train"])]]
    # All training cameras + images live on the device and the view is a device scalar: from pixel draws to packed ray rows
    # + target pixels is ONE kernel with no per-view host constant (reference: full-image bundle + coordinate grid + three
    # gathers + normalise + cat per step), so the whole iteration can be captured once and replayed for any view.
    selector = nerf.MultiViewRaySelector(hw[0], hw[1], [poses[v] for v in train_ids], [intrinsics[v] for v in train_ids],
                                         args.near, args.far,
                                         images=torch.stack([images[v].reshape(hw[0], hw[1], 3) for v in train_ids]), device=dev)
    loss_t = torch.zeros((), dtype=torch.float32, device=dev)
    fused = None
    if fused_ok:
        fused = nerf.FusedTrainStep(student[0], student[1], selector, cfg, bucket, ex, ed, args.num_random_rays, seed=args.seed + 7919 * rank,
                                    luminance=args.ir, first_iteration=start, draw_view=True)
    graphed = nerf.GraphedTrainStep(fused, opt, eager_iterations=3, use_graphs=use_graph) if fused is not None else None
    use_graph = use_graph and world == 1      # (the autograd step's single graph below: one rank only)

    def iteration():
        """The torch composition (--autograd-step, NDC rays, configurations outside nerf.FusedTrainStep): select rays -> coarse +
        fine render -> loss -> backward -> (all-reduce) -> Adam; device-side state only."""
        rays, target = selector.select(selector.random_pixels(args.num_random_rays))
        if args.ndc:
            # run_one_iter_of_nerf's NDC branch (reference train_utils.py:240-262) on the selected rows: origins / directions
            # through dn_ndc_rays (near plane at 1), bounds 0 .. 1, view directions stay those of the unwarped rays
            ro, rd = nerf.ndc_rays(hw[0], hw[1], data["focal"], 1.0, rays[:, :3].contiguous(), rays[:, 3:6].contiguous())
            rays = torch.cat([ro, rd, rays[:, 6:]], dim=-1)
        chunks = [nerf.predict_and_render_radiance(batch, student[0], student[1], cfg, mode="train", encode_position_fn=ex,
                                                   encode_direction_fn=ed, m_thres_cand=thres)
                  for batch in nerf.get_minibatches(rays, chunksize=args.chunksize)]
        out = chunks[0] if len(chunks) == 1 else [torch.cat(c, dim=0) for c in zip(*chunks)]
        if args.ir:
            lum = lambda t: 0.299 * t[..., 0] + 0.587 * t[..., 1] + 0.114 * t[..., 2]  # noqa: E731
            loss = nerf.img2mse(lum(out[0]), lum(target)) + nerf.img2mse(lum(out[3]), lum(target))
        else:
            loss = nerf.img2mse(out[0][..., :3], target) + nerf.img2mse(out[3][..., :3], target)
        bucket.zero()
        loss.backward()
        bucket.all_reduce_mean()              # one flat all-reduce (RCCL over xGMI when world > 1)
        opt.step()
        loss_t.copy_(loss.detach())

    graph, graph_warned = None, False
    history = []
    t0 = time.perf_counter()
    t_steady = None
    loss_val = psnr = float("nan")
    for it in range(start, args.iters):
        if fused is None:                                                            # (the fused step draws its view in the kernel)
            selector.view.fill_(int(np.random.randint(len(train_ids))))              # one random training view per iteration
        lr = args.lr * args.lr_decay_factor ** (it / (args.lr_decay * 1000))         # train_dexnerf_rgb.py:284-289
        if not flat_adam:
            lr_t.fill_(lr)
        if it == start + 20:                   # past the eager iterations and the capture: the steady part is timed from here
            torch.cuda.synchronize()
            t_steady = time.perf_counter()
        if graphed is not None:
            graphed.step()
            if graphed.fallback_reason and rank == 0 and not args.quiet and not graph_warned:
                graph_warned = True
                print(f"[train] HIP-graph capture unavailable ({graphed.fallback_reason}); launching eagerly", flush=True)
        elif use_graph and graph is None and it >= start + 3:
            # three eager iterations have initialised every lazy state (optimizer moments, packed streams); capture the
            # fourth and replay it from here on.  Anything that cannot be captured falls back to eager launches.
            try:
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    iteration()
                graph.replay()                         # a capture only records: this iteration's step runs here
                nerf.models.mark_parameters_updated()
            except Exception as exc:  # noqa: BLE001
                graph, use_graph = None, False
                torch.cuda.synchronize()
                if rank == 0 and not args.quiet:
                    print(f"[train] HIP-graph capture unavailable ({type(exc).__name__}: {exc}); launching eagerly", flush=True)
                iteration()
        elif graph is not None:
            graph.replay()
            nerf.models.mark_parameters_updated()   # the replayed optimizer step ran no Python hook
        else:
            iteration()
        if it % 100 == 0 or it == args.iters - 1:
            loss_val = (fused.loss3[0] if fused is not None else loss_t).item()
            psnr = nerf.mse2psnr(loss_val)
            history.append((it, loss_val, psnr))
            if rank == 0 and not args.quiet:
                print(f"[train] iter {it:6d} loss {loss_val:.5f} psnr {psnr:.2f} dB lr {lr:.2e} "
                      f"{(time.perf_counter() - t0):.1f} s", flush=True)
            s8 = nerf.s8_grad_stats() if args.precision in ("bf16", "bf16-s8") else None
            if s8 is not None:
                s8_saturated_max = max(s8_saturated_max, s8["saturated"])
                if s8["saturated"] > 1e-4 and rank == 0 and not s8_warned:
                    s8_warned = True
                    print(f"[train] WARNING: {s8['saturated']:.2e} of the 8-bit saved layer gradients are clipped at e5m2's largest value "
                          f"(scale {s8['scale']}): the loss is outside the range the fixed gradient scale covers (a sum-reduced or "
                          "scaled loss?) - pass --s8-grad-scale 0 (per-launch scale) or a smaller power of two", flush=True)
        if rank == 0 and args.validate_every and (it + 1) % args.validate_every == 0:
            out = render_view(student, cfg, poses[held_out], intrinsics[held_out], hw, ex, ed, thres)
            vmse = nerf.img2mse(out[3].reshape(-1, 3), images[held_out]).item()
            if depths[held_out] is None:     # (LLFF captures carry no depth maps: no Dex threshold sweep)
                if not args.quiet:
                    print(f"[val]   iter {it + 1:6d} held-out view psnr {nerf.mse2psnr(vmse):.2f} dB", flush=True)
                continue
            m_best, err = dex_sweep(out, depths[held_out], thres, data["mask_hi"])
            if not args.quiet:
                print(f"[val]   iter {it + 1:6d} held-out view psnr {nerf.mse2psnr(vmse):.2f} dB; Dex best m={m_best} "
                      f"abs depth err {err['depth_abs_err']:.1f} mm", flush=True)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    result = dict(history=history, final_loss=loss_val, final_psnr=psnr, seconds=elapsed,
                  rays_per_s=world * args.num_random_rays * (args.iters - start) / max(elapsed, 1e-9))
    if t_steady is not None and args.iters - start > 20:
        result["steady_ms_per_iter"] = (t0 + elapsed - t_steady) * 1e3 / (args.iters - start - 20)
    result["hip_graphs"] = len(graphed.graphs) if (graphed is not None and graphed.graphs) else int(graph is not None)   # graphs replayed per iteration
    if args.precision in ("bf16", "bf16-s8"):
        result["s8_saturated_max"] = s8_saturated_max
    if args.save and rank != 0 and os.environ.get("DEXNERF_SAVE_ALL_RANKS"):   # rehearsals: compare the replicas
        torch.save({"model_coarse_state_dict": student[0].state_dict(), "model_fine_state_dict": student[1].state_dict()}, args.save)
    if rank == 0:
        out = render_view(student, cfg, poses[held_out], intrinsics[held_out], hw, ex, ed, thres)
        result["val_psnr"] = nerf.mse2psnr(nerf.img2mse(out[3].reshape(-1, 3), images[held_out]).item())
        if depths[held_out] is not None:
            m_best, err = dex_sweep(out, depths[held_out], thres, data["mask_hi"])
            result["dex_best_threshold"], result["dex_abs_err_mm"] = int(m_best), err["depth_abs_err"]
        if args.save:
            opt_state = opt.state_dict()
            for group in opt_state["param_groups"]:   # the reference stores a Python float (train_dexnerf_rgb.py:443-456), not the
                group["lr"] = float(group["lr"])     # device scalar the captured graph reads the schedule from
            torch.save({"iter": args.iters, "model_coarse_state_dict": student[0].state_dict(),
                        "model_fine_state_dict": student[1].state_dict(), "optimizer_state_dict": opt_state,
                        "loss": loss_val, "psnr": psnr}, args.save)
        if not args.quiet:
            print(f"[done] {args.iters - start} iters in {elapsed:.1f} s = {result['rays_per_s']:.0f} rays/s; "
                  f"held-out psnr {result['val_psnr']:.2f} dB", flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    # process-wide switches this run set go back to the library's defaults (main() is also called as a function: tests, bench.py)
    if args.s8_grad_scale is not None:
        nerf.set_s8_grad_scale(0.0)
    _nerf_ops.set_deterministic_weight_gradients(True)
    return result


if __name__ == "__main__":
    main()
