"""NeRF MLPs with the reference's class names, constructor signatures and state_dict keys
(reference nerf/models.py).  FlexibleNeRFModel - the only class any reference config names - runs on
the fused HIP kernel for device tensors; the other classes are kept as plain modules for name
compatibility (out of the HIP scope)."""
import torch
from torch.optim.optimizer import register_optimizer_step_post_hook as _register_post_step

from . import _hip, _ops

_optimizer_steps = [0]   # bumped by a global optimizer post-step hook (see FlexibleNeRFModel.packed)


def _count_optimizer_step(*_args, **_kwargs):
    _optimizer_steps[0] += 1


def mark_parameters_updated():
    """Tell the packed-weights caches that parameters changed behind PyTorch's back - e.g. after replaying a captured
    HIP graph that contains an optimizer step (no Python hook runs during a replay)."""
    _optimizer_steps[0] += 1


_register_post_step(_count_optimizer_step)


class FlexibleNeRFModel(torch.nn.Module):
    """Drop-in for reference nerf/models.py:185-256.

    Parameters are created in the reference's order (layer1, layers_xyz.*, layers_dir.0, fc_alpha, fc_rgb,
    fc_feat | fc_out) so checkpoints load unchanged and a given torch seed yields the same init.
    Differences, both supersets: the skip concatenation fires exactly where __init__ built the wide layer
    (the reference's forward tests an undefined attribute, models.py:243), and the scripts' missing
    num_layers/hidden_size/skip_connect_every keys are honoured when passed.
    """

    def __init__(self, num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=6,
                 num_encoding_fn_dir=4, include_input_xyz=True, include_input_dir=True, use_viewdirs=True):
        super().__init__()
        self.num_layers = num_layers
        self.hidden_size = hidden_size
        self.skip_connect_every = skip_connect_every
        self.num_encoding_fn_xyz = num_encoding_fn_xyz
        self.num_encoding_fn_dir = num_encoding_fn_dir
        self.include_input_xyz = bool(include_input_xyz)
        self.include_input_dir = bool(include_input_dir)
        self.dim_xyz = (3 if include_input_xyz else 0) + 2 * 3 * num_encoding_fn_xyz
        self.dim_dir = ((3 if include_input_dir else 0) + 2 * 3 * num_encoding_fn_dir) if use_viewdirs else 0
        self.use_viewdirs = bool(use_viewdirs)
        self.skip_layers = [i for i in range(num_layers - 1)
                            if i % skip_connect_every == 0 and i > 0 and i != num_layers - 1]
        self.layer1 = torch.nn.Linear(self.dim_xyz, hidden_size)
        self.layers_xyz = torch.nn.ModuleList()
        for i in range(num_layers - 1):
            fan_in = self.dim_xyz + hidden_size if i in self.skip_layers else hidden_size
            self.layers_xyz.append(torch.nn.Linear(fan_in, hidden_size))
        if self.use_viewdirs:
            self.layers_dir = torch.nn.ModuleList([torch.nn.Linear(self.dim_dir + hidden_size, hidden_size // 2)])
            self.fc_alpha = torch.nn.Linear(hidden_size, 1)
            self.fc_rgb = torch.nn.Linear(hidden_size // 2, 3)
            self.fc_feat = torch.nn.Linear(hidden_size, hidden_size)
        else:
            self.fc_out = torch.nn.Linear(hidden_size, 4)
        self.relu = torch.nn.functional.relu
        self._packed = {}

    # ---- HIP side ---------------------------------------------------------------------------------
    def linear_modules(self):
        mods = [self.layer1] + list(self.layers_xyz)
        if self.use_viewdirs:
            mods += [self.layers_dir[0], self.fc_alpha, self.fc_rgb, self.fc_feat]
        else:
            mods += [self.fc_out]
        return mods

    def param_key(self):
        """Changes whenever the parameters may have (see packed())."""
        return tuple((m.weight.data_ptr(), m.weight._version, m.bias.data_ptr(), m.bias._version)
                     for m in self.linear_modules()) + (_optimizer_steps[0],)

    def desc_kwargs(self, log_sampling_xyz=True, log_sampling_dir=True):
        return dict(num_layers=self.num_layers, hidden_size=self.hidden_size,
                    skip_connect_every=self.skip_connect_every, num_encoding_fn_xyz=self.num_encoding_fn_xyz,
                    num_encoding_fn_dir=self.num_encoding_fn_dir, include_input_xyz=self.include_input_xyz,
                    include_input_dir=self.include_input_dir, use_viewdirs=self.use_viewdirs,
                    log_sampling_xyz=log_sampling_xyz, log_sampling_dir=log_sampling_dir)

    def packed(self, log_sampling_xyz=True, log_sampling_dir=True, train=False, precision=None):
        """MFMA fragment streams for the current parameters (re-packed when any parameter changed).  `train=True` (the training
        entry points) refreshes only the stream the training forward reads - the core one, or the 48-point kernel's in the
        8-bit-saved-tensor mode - and leaves the other stale; the next caller without it - any render - brings both up to date.

        "Changed" = a new storage, a bumped tensor version (every ordinary in-place op), or ANY optimizer step since the
        last pack: fused optimizers (`Adam(fused=True)`) update parameters without bumping their versions."""
        mods = self.linear_modules()
        dev = mods[0].weight.device
        prec = _ops._precision if precision is None else precision   # (a render may ask for fp16 beside the bf16 training pack)
        key = self.param_key()
        pk = self._packed_slot(log_sampling_xyz, log_sampling_dir, prec)
        # under stream capture always (re)pack: a captured graph must contain the pack of the weights it runs on, whatever
        # the host-side cache believes at capture time
        capturing = dev.type == "cuda" and torch.cuda.is_current_stream_capturing()
        # a training entry point reads ONE of the two streams: the 48-point one in the 8-bit-saved-tensor mode (its forward is the
        # 48-point kernel), the core one otherwise; a render may read either
        want_core = want_48 = True
        if train:
            want_48 = _ops.train_precision(pk) == _hip.PREC_BF16_S8
            want_core = not want_48
        parts = 0
        if want_core and (pk.key != key or capturing):
            parts |= _hip.PACK_CORE
        if want_48 and (pk.key48 != key or capturing):
            parts |= _hip.PACK_G48
        if parts:
            pk.pack([m.weight for m in mods], [m.bias for m in mods], parts)
            if parts & _hip.PACK_CORE:
                pk.key = key
            if parts & _hip.PACK_G48:
                pk.key48 = key
        return pk

    def _packed_slot(self, log_sampling_xyz=True, log_sampling_dir=True, precision=None):
        """The PackedMLP object of (precision, sampling flags, device), created empty on first use (packed() fills it)."""
        dev = self.layer1.weight.device
        prec = _ops._precision if precision is None else precision
        slot = (prec, bool(log_sampling_xyz), bool(log_sampling_dir), dev)
        pk = self._packed.get(slot)
        if pk is None:
            pk = _ops.PackedMLP(self.desc_kwargs(log_sampling_xyz, log_sampling_dir), dev, prec)
            self._packed[slot] = pk
        return pk

    def fused_ok(self):
        """True when the fused HIP kernel covers this configuration."""
        return (self.hidden_size in (128, 256) and 2 <= self.num_layers <= 32 and self.include_input_xyz
                and (self.include_input_dir or not self.use_viewdirs) and self.num_encoding_fn_xyz in (6, 10)
                and (not self.use_viewdirs or self.num_encoding_fn_dir == 4))

    # ---- forward ------------------------------------------------------------------------------------
    def _forward_modules(self, x):
        """nn.Linear composition (host tensors, and the autograd path of configs outside the fused kernel)."""
        xyz = x[..., : self.dim_xyz]
        h = self.layer1(xyz)  # no activation after layer1 (reference models.py:238)
        for i, layer in enumerate(self.layers_xyz):
            if i in self.skip_layers:
                h = torch.cat((h, xyz), dim=-1)
            h = self.relu(layer(h))
        if not self.use_viewdirs:
            return self.fc_out(h)
        view = x[..., self.dim_xyz:]
        feat = self.relu(self.fc_feat(h))
        alpha = self.fc_alpha(h)
        g = self.relu(self.layers_dir[0](torch.cat((feat, view), dim=-1)))
        return torch.cat((self.fc_rgb(g), alpha), dim=-1)

    def forward(self, x):
        if x.is_cuda and self.fused_ok():
            from ._train import mlp_encoded
            return mlp_encoded(self, x)
        return self._forward_modules(x)


class VeryTinyNeRFModel(torch.nn.Module):
    """Reference nerf/models.py:4-31 (3-layer MLP on the xyz(+dir) encoding); plain module."""

    def __init__(self, filter_size=128, num_encoding_functions=6, use_viewdirs=True):
        super().__init__()
        self.xyz_encoding_dims = 3 + 3 * 2 * num_encoding_functions
        self.viewdir_encoding_dims = (3 + 3 * 2 * num_encoding_functions) if use_viewdirs else 0
        self.layer1 = torch.nn.Linear(self.xyz_encoding_dims + self.viewdir_encoding_dims, filter_size)
        self.layer2 = torch.nn.Linear(filter_size, filter_size)
        self.layer3 = torch.nn.Linear(filter_size, 4)
        self.relu = torch.nn.functional.relu

    def forward(self, x):
        return self.layer3(self.relu(self.layer2(self.relu(self.layer1(x)))))


class MultiHeadNeRFModel(torch.nn.Module):
    """Reference nerf/models.py:34-78; plain module."""

    def __init__(self, hidden_size=128, num_encoding_functions=6, use_viewdirs=True):
        super().__init__()
        self.xyz_encoding_dims = 3 + 3 * 2 * num_encoding_functions
        self.viewdir_encoding_dims = (3 + 3 * 2 * num_encoding_functions) if use_viewdirs else 0
        self.layer1 = torch.nn.Linear(self.xyz_encoding_dims, hidden_size)
        self.layer2 = torch.nn.Linear(hidden_size, hidden_size)
        self.layer3_1 = torch.nn.Linear(hidden_size, 1)
        self.layer3_2 = torch.nn.Linear(hidden_size, hidden_size)
        self.layer4 = torch.nn.Linear(self.viewdir_encoding_dims + hidden_size, hidden_size)
        self.layer5 = torch.nn.Linear(hidden_size, hidden_size)
        self.layer6 = torch.nn.Linear(hidden_size, 3)
        self.relu = torch.nn.functional.relu

    def forward(self, x):
        x, view = x[..., : self.xyz_encoding_dims], x[..., self.xyz_encoding_dims:]
        x = self.relu(self.layer2(self.relu(self.layer1(x))))
        sigma = self.layer3_1(x)
        feat = self.relu(self.layer3_2(x))
        x = self.relu(self.layer4(torch.cat((feat, view), dim=-1)))
        x = self.relu(self.layer5(x))
        return torch.cat((self.layer6(x), sigma), dim=-1)


class ReplicateNeRFModel(torch.nn.Module):
    """Reference nerf/models.py:81-120; plain module."""

    def __init__(self, hidden_size=256, num_layers=4, num_encoding_fn_xyz=6, num_encoding_fn_dir=4,
                 include_input_xyz=True, include_input_dir=True):
        super().__init__()
        self.dim_xyz = (3 if include_input_xyz else 0) + 2 * 3 * num_encoding_fn_xyz
        self.dim_dir = (3 if include_input_dir else 0) + 2 * 3 * num_encoding_fn_dir
        self.layer1 = torch.nn.Linear(self.dim_xyz, hidden_size)
        self.layer2 = torch.nn.Linear(hidden_size, hidden_size)
        self.layer3 = torch.nn.Linear(hidden_size, hidden_size)
        self.fc_alpha = torch.nn.Linear(hidden_size, 1)
        self.layer4 = torch.nn.Linear(hidden_size + self.dim_dir, hidden_size // 2)
        self.layer5 = torch.nn.Linear(hidden_size // 2, hidden_size // 2)
        self.fc_rgb = torch.nn.Linear(hidden_size // 2, 3)
        self.relu = torch.nn.functional.relu

    def forward(self, x):
        xyz, direction = x[..., : self.dim_xyz], x[..., self.dim_xyz:]
        h = self.relu(self.layer2(self.relu(self.layer1(xyz))))
        feat = self.layer3(h)
        alpha = self.fc_alpha(h)
        y = self.relu(self.layer4(torch.cat((feat, direction), dim=-1)))
        y = self.relu(self.layer5(y))
        return torch.cat((self.fc_rgb(y), alpha), dim=-1)


class PaperNeRFModel(FlexibleNeRFModel):
    """Name kept for `getattr(models, cfg.models.coarse.type)`; the reference's own forward is broken
    (models.py:163-182 feeds 90 columns to a 63-wide layer), so this aliases the paper-shaped
    FlexibleNeRFModel (8 x 256, skip 4)."""

    def __init__(self, num_layers=8, hidden_size=256, skip_connect_every=4, num_encoding_fn_xyz=6,
                 num_encoding_fn_dir=4, include_input_xyz=True, include_input_dir=True, use_viewdirs=True):
        super().__init__(num_layers, hidden_size, skip_connect_every, num_encoding_fn_xyz, num_encoding_fn_dir,
                         include_input_xyz, include_input_dir, use_viewdirs)
