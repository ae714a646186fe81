"""Autograd glue between PyTorch and the fused HIP network kernels."""
import torch

from . import _ops


def _needs_grad(model, *tensors):
    if not torch.is_grad_enabled():
        return False
    return any(t is not None and t.requires_grad for t in tensors) or any(p.requires_grad for p in model.parameters())


def mlp_encoded(model, x):
    """FlexibleNeRFModel.forward(x) on already-embedded device rows."""
    if _needs_grad(model, x):
        raise NotImplementedError("autograd through the fused MLP kernel is not wired up yet")
    return _ops.mlp_forward_encoded(model.packed(), x)


def run_network_fused(model, pts, viewdirs, samples_per_ray, log_xyz=True, log_dir=True):
    """run_network on raw points: positional encoding + MLP in one kernel."""
    if _needs_grad(model, pts):
        raise NotImplementedError("autograd through the fused MLP kernel is not wired up yet")
    return _ops.run_network_pts(model.packed(log_xyz, log_dir), pts, viewdirs, samples_per_ray)
