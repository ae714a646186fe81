"""Dispatch between the fused HIP network kernel (inference) and the autograd path (training).

Round-1 state: the fused kernel is forward-only.  When gradients w.r.t. the parameters are needed the
network is evaluated as the same nn.Linear composition on the device (rocBLAS GEMMs, differentiated by
PyTorch), fed by the HIP positional-encoding kernel and followed by the HIP compositing forward/backward
(`_ops.VolumeRenderFn`).  A fused backward chain is the planned replacement ("next")."""
import torch

from . import _ops


def needs_grad(model, *tensors):
    if not torch.is_grad_enabled():
        return False
    return any(t is not None and t.requires_grad for t in tensors) or any(p.requires_grad for p in model.parameters())


def mlp_encoded(model, x):
    """FlexibleNeRFModel.forward(x) on already-embedded device rows."""
    if needs_grad(model, x):
        return model._forward_modules(x)
    return _ops.mlp_forward_encoded(model.packed(), x)


def run_network_fused(model, pts, viewdirs, samples_per_ray, log_xyz=True, log_dir=True):
    """run_network on raw points: positional encoding + MLP in one kernel (no autograd)."""
    return _ops.run_network_pts(model.packed(log_xyz, log_dir), pts, viewdirs, samples_per_ray)
