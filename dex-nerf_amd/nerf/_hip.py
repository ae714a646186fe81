"""ctypes binding of libdexnerf_hip.so (C ABI declared in include/dexnerf_hip.h).

No torch types cross the boundary: tensors are passed as raw device pointers (`tensor.data_ptr()`) plus
sizes, and work is enqueued on PyTorch's current HIP stream.  The library is REQUIRED for any device
tensor: there is no eager/CPU fallback - `lib()` raises if the shared object is missing.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DEXNERF_HIP_LIB", os.path.join(os.path.dirname(_HERE), "lib", "libdexnerf_hip.so"))

PREC_F32 = 0
PREC_BF16 = 1
PREC_F16 = 2
PACK_CORE, PACK_G48, PACK_ALL = 1, 2, 3   # dn_mlp_pack_parts: the streams of a packed buffer
PREC_BF16_S8 = 3   # training entry points: bf16 arithmetic, 8-bit saved activations / gradients (experimental)

EXPORTS = (
    "dn_abi_version", "dn_last_error", "dn_ray_bundle", "dn_coarse_depths", "dn_positional_encoding",
    "dn_mlp_packed_bytes", "dn_mlp_pack", "dn_run_network", "dn_mlp_forward_encoded", "dn_volume_render",
    "dn_volume_render_backward", "dn_sample_pdf", "dn_fine_depths", "dn_render_workspace_bytes", "dn_render_rays",
    "dn_mlp_train_sizes", "dn_mlp_backward_packed_bytes", "dn_mlp_pack_backward", "dn_run_network_train",
    "dn_mlp_backward_data", "dn_mlp_unpack", "dn_mlp_weight_grad", "dn_mlp_weight_grad_all",
    "dn_select_rays", "dn_select_rays_indirect", "dn_ndc_rays", "dn_dex_error_sweep", "dn_depth_error_image",
    "dn_render_train_workspace_bytes", "dn_render_rays_train", "dn_render_rays_backward",
    "dn_set_s8_grad_scale", "dn_mlp_pack_parts", "dn_fp16_range_guard", "dn_select_rays_draw", "dn_mse2_loss", "dn_rng_fill", "dn_mlp_pack_train_pair",
    "dn_adam_step", "dn_pack_ray_rows", "dn_mlp_weight_grad_pair",
    "dn_mlp_weight_grad_scratch_bytes", "dn_mlp_weight_grad_all_ws", "dn_mlp_weight_grad_pair_ws", "dn_render_rays_backward_ws",
)


class MlpDesc(ctypes.Structure):
    """dn_mlp_desc (mirrors FlexibleNeRFModel.__init__, reference nerf/models.py:186-196)."""
    _fields_ = [(n, c_int32) for n in (
        "num_layers", "hidden_size", "skip_connect_every", "num_encoding_fn_xyz", "num_encoding_fn_dir",
        "include_input_xyz", "include_input_dir", "use_viewdirs", "log_sampling_xyz", "log_sampling_dir")]


_lib = None


def _declare(lib):
    fp, vp = c_void_p, c_void_p  # device pointers travel as integers
    lib.dn_abi_version.restype = c_int
    lib.dn_last_error.restype = c_char_p
    lib.dn_ray_bundle.argtypes = [c_int, c_int, POINTER(c_float), POINTER(c_float), c_float, c_float, c_float, fp, fp, vp]
    lib.dn_coarse_depths.argtypes = [fp, c_int, c_int64, c_int, c_int, fp, fp, vp]
    lib.dn_positional_encoding.argtypes = [fp, c_int64, c_int, c_int, c_int, c_int, fp, vp]
    lib.dn_mlp_packed_bytes.argtypes = [POINTER(MlpDesc), c_int]
    lib.dn_mlp_packed_bytes.restype = c_size_t
    lib.dn_mlp_pack.argtypes = [POINTER(MlpDesc), c_int, POINTER(c_void_p), POINTER(c_void_p), vp, vp]
    lib.dn_mlp_pack_parts.argtypes = [POINTER(MlpDesc), c_int, POINTER(c_void_p), POINTER(c_void_p), vp, c_int, vp]
    lib.dn_run_network.argtypes = [POINTER(MlpDesc), c_int, vp, fp, fp, fp, c_int, fp, c_int64, c_int, fp, vp]
    lib.dn_mlp_forward_encoded.argtypes = [POINTER(MlpDesc), c_int, vp, fp, c_int64, fp, vp]
    lib.dn_volume_render.argtypes = [fp, fp, fp, c_int, fp, c_float, c_int, POINTER(c_float), c_int, c_int64, c_int,
                                     fp, fp, fp, fp, fp, fp, vp]
    lib.dn_volume_render_backward.argtypes = [fp, fp, fp, c_int, fp, c_float, c_int, c_int64, c_int, fp, fp, fp, fp,
                                              fp, fp, vp]
    lib.dn_sample_pdf.argtypes = [fp, fp, fp, c_int64, c_int, c_int, fp, fp, vp]
    lib.dn_fine_depths.argtypes = [fp, fp, fp, c_int64, c_int, c_int, fp, fp, vp]
    lib.dn_render_workspace_bytes.argtypes = [c_int64, c_int, c_int]
    lib.dn_render_workspace_bytes.restype = c_size_t
    lib.dn_render_rays.argtypes = [POINTER(MlpDesc), vp, POINTER(MlpDesc), vp, c_int, fp, c_int, c_int64, c_int, c_int,
                                   c_int, c_float, c_int, POINTER(c_float), c_int, fp, fp, fp, fp, fp, fp, fp, fp, fp,
                                   fp, fp, vp, vp]
    lib.dn_mlp_train_sizes.argtypes = [POINTER(MlpDesc), c_int, c_int64, POINTER(c_size_t), POINTER(c_size_t),
                                       POINTER(c_size_t)]
    lib.dn_mlp_backward_packed_bytes.argtypes = [POINTER(MlpDesc), c_int]
    lib.dn_mlp_backward_packed_bytes.restype = c_size_t
    lib.dn_mlp_pack_backward.argtypes = [POINTER(MlpDesc), c_int, POINTER(c_void_p), vp, vp]
    lib.dn_run_network_train.argtypes = [POINTER(MlpDesc), c_int, vp, fp, fp, fp, c_int, fp, c_int64, c_int, fp, vp, vp, vp]
    lib.dn_mlp_backward_data.argtypes = [POINTER(MlpDesc), c_int, vp, fp, vp, c_int64, vp, vp]
    lib.dn_mlp_unpack.argtypes = [POINTER(MlpDesc), c_int, c_int, vp, c_int64, c_int, c_int, c_int, fp, c_int, c_int, vp]
    lib.dn_mlp_weight_grad.argtypes = [POINTER(MlpDesc), c_int, vp, vp, c_int64, c_int, c_int, c_int, c_int, c_int, fp, c_int,
                                       fp, vp]
    lib.dn_select_rays.argtypes = [c_int, c_int, POINTER(c_float), POINTER(c_float), c_float, c_float, c_float, c_float, c_float,
                                   vp, c_int64, fp, c_int, fp, fp, vp]
    lib.dn_select_rays_indirect.argtypes = [c_int, c_int, fp, vp, c_float, c_float, vp, c_int64, fp, c_int, fp, fp, vp]
    lib.dn_ndc_rays.argtypes = [c_int, c_int, ctypes.c_double, ctypes.c_double, fp, fp, c_int64, fp, fp, vp]
    lib.dn_dex_error_sweep.argtypes = [fp, fp, c_int, c_int64, vp, c_float, c_float, vp, vp]
    lib.dn_depth_error_image.argtypes = [fp, fp, vp, c_int, c_int, c_float, fp, vp]
    lib.dn_mlp_weight_grad_all.argtypes = [POINTER(MlpDesc), c_int, vp, vp, c_int64, POINTER(c_void_p), POINTER(c_void_p), vp]
    lib.dn_mlp_weight_grad_pair.argtypes = [POINTER(MlpDesc), c_int, vp, vp, c_int64, POINTER(c_void_p), POINTER(c_void_p), vp, vp, c_int64,
                                            POINTER(c_void_p), POINTER(c_void_p), vp]
    lib.dn_mlp_weight_grad_scratch_bytes.argtypes = [POINTER(MlpDesc), c_int]
    lib.dn_mlp_weight_grad_scratch_bytes.restype = c_size_t
    lib.dn_mlp_weight_grad_all_ws.argtypes = [POINTER(MlpDesc), c_int, vp, vp, c_int64, POINTER(c_void_p), POINTER(c_void_p), vp, c_size_t, vp]
    lib.dn_mlp_weight_grad_pair_ws.argtypes = [POINTER(MlpDesc), c_int, vp, vp, c_int64, POINTER(c_void_p), POINTER(c_void_p), vp, vp, c_int64,
                                               POINTER(c_void_p), POINTER(c_void_p), vp, c_size_t, vp]
    lib.dn_set_s8_grad_scale.argtypes = [c_float]
    lib.dn_fp16_range_guard.argtypes = [POINTER(MlpDesc)]
    lib.dn_render_train_workspace_bytes.argtypes = [c_int64, c_int, c_int]
    lib.dn_render_train_workspace_bytes.restype = c_size_t
    lib.dn_render_rays_train.argtypes = [POINTER(MlpDesc), vp, POINTER(MlpDesc), vp, c_int, fp, c_int, c_int64, c_int, c_int,
                                         c_int, c_float, c_int, POINTER(c_float), c_int, fp, fp, fp, fp, fp, fp, fp, fp, fp,
                                         fp, fp, vp, vp, vp, vp, vp, vp, c_int, vp]
    lib.dn_render_rays_backward.argtypes = [POINTER(MlpDesc), vp, POINTER(MlpDesc), vp, c_int, fp, c_int, c_int64, c_int, c_int,
                                            c_float, c_int, fp, fp, fp, fp, fp, fp, fp, fp, vp, vp, vp, vp, vp, vp, vp,
                                            POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), c_int, vp, vp]
    lib.dn_render_rays_backward_ws.argtypes = [POINTER(MlpDesc), vp, POINTER(MlpDesc), vp, c_int, fp, c_int, c_int64, c_int, c_int,
                                               c_float, c_int, fp, fp, fp, fp, fp, fp, fp, fp, vp, vp, vp, vp, vp, vp, vp,
                                               POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), c_int, vp, vp, c_size_t, vp]
    lib.dn_select_rays_draw.argtypes = [c_int, c_int, fp, vp, c_int, c_float, c_float, vp, c_int64, fp, c_int, fp, fp, vp, vp]
    lib.dn_mse2_loss.argtypes = [fp, fp, fp, c_int64, c_int, fp, fp, fp, vp, vp]
    lib.dn_rng_fill.argtypes = [vp, ctypes.c_uint32, c_int64, c_int, fp, vp]
    lib.dn_pack_ray_rows.argtypes = [fp, fp, fp, c_float, c_float, c_int64, fp, vp]
    dbl = ctypes.c_double
    lib.dn_adam_step.argtypes = [fp, fp, fp, fp, c_int64, fp, fp, dbl, dbl, dbl, dbl, dbl, c_int, vp]
    lib.dn_mlp_pack_train_pair.argtypes = [POINTER(MlpDesc), POINTER(c_void_p), POINTER(c_void_p), vp, vp, POINTER(c_void_p), POINTER(c_void_p),
                                           vp, vp, vp]
    for name in EXPORTS:
        if name not in ("dn_last_error", "dn_mlp_packed_bytes", "dn_render_workspace_bytes",
                        "dn_mlp_backward_packed_bytes", "dn_render_train_workspace_bytes"):
            getattr(lib, name).restype = c_int


def lib():
    """The loaded library; raises RuntimeError (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"libdexnerf_hip.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or `make -C dex-nerf_amd/csrc`).  The HIP path has no fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        _declare(handle)
        if handle.dn_abi_version() != 2:
            raise RuntimeError("libdexnerf_hip.so ABI version mismatch")
        _lib = handle
    return _lib


def available():
    return os.path.exists(LIB_PATH)


def check(rc, what=""):
    if rc != 0:
        msg = lib().dn_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what or 'dexnerf_hip'} failed (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a contiguous fp32 tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "HIP entry points need contiguous device tensors"
    return c_void_p(t.data_ptr())


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def host_floats(values):
    arr = (c_float * max(len(values), 1))(*[float(v) for v in values])
    return arr


def f32c(t):
    """Contiguous fp32 view/copy of a device tensor."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
