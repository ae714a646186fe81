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
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev)
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
