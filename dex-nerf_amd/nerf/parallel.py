"""Data-parallel helpers: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm;
"gloo" in the CPU tests).  The reference has no distributed code; rays are independent,
so the render path shards with no collective inside it and training needs exactly one gradient all-reduce
per step."""
import torch
import torch.distributed as dist


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n_items, rank, world):
    """Contiguous block [lo, hi) of n_items for `rank`; blocks differ by at most one item."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def render_sharded(render_fn, ray_origins, ray_directions):
    """Render an (H, W, 3) ray bundle with the rows split over the ranks; every rank gets the full images.

    render_fn(ro_block, rd_block) -> tuple of tensors whose leading dims are the block's (rows, W) (None allowed).
    One all_gather of the outputs ((10+K) floats per ray); nothing is exchanged inside the path.
    """
    rank, world = world_info()
    height = ray_directions.shape[0]
    lo, hi = shard_bounds(height, rank, world)
    outs = render_fn(ray_origins[lo:hi], ray_directions[lo:hi])
    if world == 1:
        return outs
    spans = [shard_bounds(height, r, world) for r in range(world)]
    rows_max = max(hi_ - lo_ for lo_, hi_ in spans)
    gathered = []
    for o in outs:
        if o is None:
            gathered.append(None)
            continue
        # all_gather wants equal shapes: pad the (at most one row) shorter blocks, trim after the exchange
        block = o.contiguous()
        if block.shape[0] < rows_max:
            pad = torch.zeros((rows_max - block.shape[0],) + tuple(block.shape[1:]), dtype=block.dtype, device=block.device)
            block = torch.cat((block, pad), dim=0)
        parts = [torch.empty_like(block) for _ in range(world)]
        dist.all_gather(parts, block)
        gathered.append(torch.cat([part[: hi_ - lo_] for part, (lo_, hi_) in zip(parts, spans)], dim=0))
    return tuple(gathered)


def _avg_in_collective():
    """RCCL averages inside the collective (ReduceOp.AVG); gloo has no AVG: sum, then one division."""
    return dist.get_backend() == "nccl"


def _all_reduce_avg(t, world):
    if _avg_in_collective():
        dist.all_reduce(t, op=dist.ReduceOp.AVG)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t.div_(world)


class FlatGradBucket:
    """The gradients of the coarse + fine nets as slices of ONE contiguous fp32 buffer (2 x 595,844 floats = 4.77 MB for
    D8/W256), exchanged with one all-reduce per network per step.

    * Every `.grad` is a permanent view of `flat` (parameter order, one contiguous segment per module).  The fused training
      path (`_train.FusedNetFn`) finds the module's segment through `module._grad_sink` and has the weight-gradient kernel
      accumulate straight into it - no per-step allocation, no concatenation, no autograd accumulation kernels; any other
      autograd path accumulates into the same views in place.  `zero()` is one memset of the buffer.
    * Overlap (world > 1): a segment's all-reduce is launched asynchronously the moment its last backward has been enqueued
      (`segment_ready`, called from FusedNetFn.backward; torch.distributed's RCCL stream waits on the compute stream by
      itself).  Autograd reaches the fine net first, so the fine segment (2.4 MB) crosses xGMI while the coarse net's
      backward runs; `all_reduce_mean()` then waits for what is in flight, reduces whatever was not launched (a
      configuration outside the fused kernels) and divides by the world size.  Two ~2.4 MB messages instead of one 4.8 MB
      one: both are latency-bound on xGMI's point-to-point links, and the first is hidden.
    * With one rank nothing is exchanged at all."""

    def __init__(self, modules, overlap=True):
        self.modules = [m for m in modules if m is not None]
        self.params = [p for m in self.modules for p in m.parameters() if p.requires_grad]
        self.overlap = bool(overlap)
        dev = self.params[0].device
        n_flat = sum(p.numel() for p in self.params)
        self.flat = torch.zeros((n_flat + 3) // 4 * 4, dtype=torch.float32, device=dev)   # (whole float4s: nerf.FlatAdam)
        self.segments, self._views, off = [], [], 0
        for idx, m in enumerate(self.modules):
            lo = off
            for p in m.parameters():
                if p.requires_grad:
                    self._views.append(self.flat[off: off + p.numel()].view_as(p))
                    off += p.numel()
            self.segments.append((lo, off))
            m._grad_sink = _GradSink(self, idx)
        self._pending = [0] * len(self.modules)   # fused forwards of the segment's module whose backward has not run yet
        self._works = {}
        self._attach()

    def _attach(self):
        for p, v in zip(self.params, self._views):
            if p.grad is not v:
                p.grad = v

    def flatten_parameters(self):
        """Move the parameters themselves into ONE contiguous fp32 buffer laid out like `flat` (every Parameter keeps its identity;
        its storage becomes a view of `flat_params`) - what nerf.FlatAdam steps in one elementwise pass.  Returns the buffer."""
        if getattr(self, "flat_params", None) is None:
            self.flat_params = torch.zeros_like(self.flat)
            off = 0
            for p in self.params:
                view = self.flat_params[off: off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                off += p.numel()
            from .models import mark_parameters_updated
            mark_parameters_updated()      # (the packed-weight caches are keyed on the old storages' versions)
        return self.flat_params

    def zero(self):
        """One memset; re-point any `.grad` something else dropped or replaced.  May be called before the forward or - as the
        training loops do (train_dexnerf.py iteration(), bench.py step()) - between the forward and `loss.backward()`: the count
        of forwards whose backward is still to come (`_pending`) is NOT touched here, it is the step's own bookkeeping and is
        reset where the step ends (all_reduce_mean).  An exchange still in flight would race with the memset: refuse."""
        if self._works:
            raise RuntimeError("FlatGradBucket.zero(): a segment's all-reduce is still in flight - call all_reduce_mean() first")
        self._attach()
        self.flat.zero_()

    def segment(self, idx):
        lo, hi = self.segments[idx]
        return self.flat[lo:hi]

    def segment_ready(self, idx):
        """All gradient kernels of module `idx` are enqueued: start its exchange (no-op with one rank / overlap off)."""
        rank, world = world_info()
        if world == 1 or not self.overlap or idx in self._works:
            return
        op = dist.ReduceOp.AVG if _avg_in_collective() else dist.ReduceOp.SUM
        self._works[idx] = dist.all_reduce(self.segment(idx), op=op, async_op=True)

    def all_reduce_mean(self):
        """Finish the step's exchange: afterwards every `.grad` holds the mean over the ranks."""
        rank, world = world_info()
        self._pending = [0] * len(self.modules)   # the step is over: a forward whose backward never ran must not block the next one
        if world == 1:
            return
        if not self._works:
            _all_reduce_avg(self.flat, world)                         # nothing was launched early: one flat message
        else:
            for idx in range(len(self.modules)):
                work = self._works.get(idx)
                if work is None:
                    _all_reduce_avg(self.segment(idx), world)
                else:
                    work.wait()
                    if not _avg_in_collective():
                        self.segment(idx).div_(world)
        self._works = {}


class _GradSink:
    """Handle a module carries to its segment of a FlatGradBucket (see FusedNetFn)."""

    def __init__(self, bucket, idx):
        self.bucket, self.idx = bucket, idx

    def views(self, module):
        """[(dW, db)] views of the bucket in module.linear_modules() order, or None when a `.grad` is not (any more) the
        bucket's view - the caller then returns gradients to autograd the ordinary way."""
        if self.idx in self.bucket._works:
            # gradient accumulation (a second backward before all_reduce_mean) with world > 1: this segment has already been
            # handed to the collective - adding to it now would race with the exchange and stay un-averaged
            raise RuntimeError("FlatGradBucket: a backward reached a segment whose all-reduce is already in flight; the overlapped "
                               "exchange needs exactly one loss.backward() per step (use overlap=False to accumulate micro-batches)")
        out = []
        for lin in module.linear_modules():
            w, b = lin.weight.grad, lin.bias.grad
            if w is None or b is None or w.untyped_storage().data_ptr() != self.bucket.flat.untyped_storage().data_ptr() \
                    or b.untyped_storage().data_ptr() != self.bucket.flat.untyped_storage().data_ptr():
                return None
            out.append((w, b))
        return out

    def forward_issued(self):
        self.bucket._pending[self.idx] += 1

    def backward_done(self):
        self.bucket._pending[self.idx] -= 1
        if self.bucket._pending[self.idx] <= 0:
            self.bucket.segment_ready(self.idx)


def broadcast_parameters(modules, src=0):
    """Make every rank start from rank `src`'s weights.

    The broadcast writes through `t.data`, which does not bump the tensors' versions: the packed-weight caches
    (FlexibleNeRFModel.param_key) are told explicitly, so a model that rendered or trained before the broadcast does not keep
    serving its old MFMA weight stream on the non-source ranks.  The same call - `nerf.models.mark_parameters_updated()` - is
    what any other out-of-band parameter write (`p.data.copy_`, an EMA swap, a manual sync) must be followed by."""
    rank, world = world_info()
    if world == 1:
        return
    for m in modules:
        if m is None:
            continue
        for t in list(m.parameters()) + list(m.buffers()):
            dist.broadcast(t.data, src)
    from .models import mark_parameters_updated
    mark_parameters_updated()


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam (the reference's optimizer: train_dexnerf_rgb.py:146-148, default betas / eps, no weight decay, amsgrad
    off) over a FlatGradBucket whose parameters have been flattened: ONE launch (dn_adam_step) instead of torch's multi-tensor
    kernels over 48 tensors.  The step count lives on the device and the kernel advances it, so a captured step replays correctly.

    lr: a float; a device scalar tensor (the caller writes its schedule into it); or, with lr_decay_steps, the reference's schedule
    lr * lr_decay_factor ** (step / lr_decay_steps) (train_dexnerf_rgb.py:284-289) evaluated inside the kernel - nothing is written
    from the host between iterations.  zero_grads=True clears the gradients in the same pass (optimizer.zero_grad(), :281).
    state_dict() / load_state_dict() use torch.optim.Adam's format (per-parameter step / exp_avg / exp_avg_sq), so checkpoints move
    between the two."""

    def __init__(self, bucket, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, lr_decay_factor=None, lr_decay_steps=None, zero_grads=False):
        self.bucket = bucket
        self.flat_params = bucket.flatten_parameters()
        dev = self.flat_params.device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam: the flat optimizer step is a HIP kernel (dn_adam_step); use torch.optim.Adam on the CPU")
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None, capturable=True,
                        differentiable=False, fused=True)
        super().__init__(bucket.params, defaults)
        self.exp_avg = torch.zeros_like(self.flat_params)
        self.exp_avg_sq = torch.zeros_like(self.flat_params)
        # dn_adam_step's state record: float [steps taken, ticket, last lr, -], then double [beta1^t, beta2^t, decay^t]
        self.step_state = torch.zeros(12, dtype=torch.float32, device=dev)[:10]
        self.lr_decay_per_step = float(lr_decay_factor) ** (1.0 / float(lr_decay_steps)) if lr_decay_steps else 1.0
        self._set_step(0)
        self.zero_grads = bool(zero_grads)
        off = 0
        for p in bucket.params:
            n = p.numel()
            self.state[p] = dict(step=self.step_state[0], exp_avg=self.exp_avg[off: off + n].view_as(p),
                                 exp_avg_sq=self.exp_avg_sq[off: off + n].view_as(p))
            off += n

    def _set_step(self, t):
        """Put the state record at `t` steps taken (a fresh optimizer, a loaded checkpoint): the running products in closed form."""
        b1, b2 = self.param_groups[0]["betas"]
        self.step_state[0] = float(t)
        self.step_state[1] = 0.0
        self.step_state[4:10].view(torch.float64).copy_(torch.tensor([b1 ** t, b2 ** t, self.lr_decay_per_step ** t], dtype=torch.float64))

    @torch.no_grad()
    def step(self, closure=None):
        from . import _ops
        if closure is not None:
            raise RuntimeError("FlatAdam.step: closures are not supported")
        group = self.param_groups[0]
        lr = group["lr"]
        b = self.bucket
        for p, v in zip(b.params, b._views):
            if p.grad is not v:
                raise RuntimeError("FlatAdam.step: a parameter's .grad is no longer the FlatGradBucket's view")
        _ops.adam_step(self.flat_params, b.flat, self.exp_avg, self.exp_avg_sq, self.step_state, lr, self.lr_decay_per_step,
                       group["betas"], group["eps"], self.zero_grads)
        from .models import mark_parameters_updated
        mark_parameters_updated()       # (in-place writes through the flat buffer do not bump the parameters' versions)

    def last_lr(self):
        """The learning rate the latest step used (a host read)."""
        return float(self.step_state[2])

    def load_state_dict(self, state_dict):
        """torch.optim.Adam's format: the moment estimates are copied INTO the flat buffers (the base class would replace the views)."""
        groups = state_dict["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.bucket.params):
            raise ValueError("FlatAdam.load_state_dict: one parameter group over the bucket's parameters expected")
        steps = 0
        for idx, p in zip(groups[0]["params"], self.bucket.params):
            st = state_dict["state"].get(idx)
            if st is None:
                continue
            self.state[p]["exp_avg"].copy_(st["exp_avg"])
            self.state[p]["exp_avg_sq"].copy_(st["exp_avg_sq"])
            steps = int(float(st["step"]))
        for key in ("betas", "eps"):
            self.param_groups[0][key] = groups[0][key]
        self._set_step(steps)
        # The stored lr is the value of the SAVING run's latest step: the reference rewrites param_group['lr'] every iteration
        # (train_dexnerf_rgb.py:284-289) and saves it (:449), i.e. it is already decayed.  With the in-kernel schedule
        # (lr_decay_steps) dn_adam_step multiplies lr0 by decay^t itself, so lr0 stays the constructor's; without a schedule the
        # checkpoint's value is the learning rate, as in torch.optim.Adam.
        if not torch.is_tensor(groups[0]["lr"]) and self.lr_decay_per_step == 1.0 and not torch.is_tensor(self.param_groups[0]["lr"]):
            self.param_groups[0]["lr"] = groups[0]["lr"]

    def state_dict(self):
        """torch.optim.Adam's format with INDEPENDENT tensors: every parameter its own `step` clone and cloned moments.  The live
        state shares one 0-dim step view among all parameters and slices of two flat buffers; saved as they are, torch.save /
        torch.load keep that sharing and torch.optim.Adam would then advance the one step once PER PARAMETER per iteration."""
        sd = super().state_dict()
        for st in sd["state"].values():
            for key in ("step", "exp_avg", "exp_avg_sq"):
                if torch.is_tensor(st.get(key)):
                    st[key] = st[key].detach().clone()
        group = sd["param_groups"][0]
        if self.lr_decay_per_step != 1.0 and not torch.is_tensor(group["lr"]):
            # what the reference's loop would have left in param_group['lr'] (train_dexnerf_rgb.py:284-289, :449): the decayed value
            steps = int(float(self.step_state[0]))
            group["lr"] = float(group["lr"]) * self.lr_decay_per_step ** max(steps - 1, 0) if steps else float(group["lr"])
        return sd
