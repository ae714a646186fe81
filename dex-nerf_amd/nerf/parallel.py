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


class FlatGradBucket:
    """The per-step gradient exchange of the coarse+fine nets as ONE all-reduce of one contiguous buffer
    (2 x 595,844 fp32 = 4.77 MB for D8/W256: latency-bound, one flat message is the right shape for xGMI's
    point-to-point links).

    `zero()` drops the gradients instead of zero-filling preassigned views: autograd then ASSIGNS what the backward
    returns (the fused training path hands back views of one buffer per network) rather than accumulating into
    existing tensors - that accumulation was 48 four-microsecond kernels per step.  `all_reduce_mean()` gathers the
    gradients into `flat` with one concatenation, reduces, and re-points every `.grad` at its slice; with a single rank
    it does nothing at all."""

    def __init__(self, modules):
        self.params = [p for m in modules if m is not None for p in m.parameters() if p.requires_grad]
        self.flat = None

    def zero(self):
        for p in self.params:
            p.grad = None

    def gather(self):
        """Concatenate the current gradients into `flat` (parameter order) and make every `.grad` a view of it."""
        self.flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.params])
        off = 0
        for p in self.params:
            p.grad = self.flat[off: off + p.numel()].view_as(p)
            off += p.numel()
        return self.flat

    def all_reduce_mean(self, async_op=False):
        rank, world = world_info()
        if world == 1:
            return None
        flat = self.gather()
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op)
        if not async_op:
            flat.div_(world)
        return work


def broadcast_parameters(modules, src=0):
    """Make every rank start from rank `src`'s weights."""
    rank, world = world_info()
    if world == 1:
        return
    for m in modules:
        if m is None:
            continue
        for t in list(m.parameters()) + list(m.buffers()):
            dist.broadcast(t.data, src)
