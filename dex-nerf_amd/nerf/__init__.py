"""Drop-in `nerf` package for the Dex-NeRF fork of nerf-pytorch, MI355X-native.

Same public names as the reference's nerf/__init__.py:1-8 star-exports, for the ray-marching hot path:
ray generation, stratified + hierarchical sampling, positional encoding, the coarse/fine MLPs and
alpha compositing with the Dex fixed-sigma depth readout all run in hand-written HIP kernels
(libdexnerf_hip.so) whenever tensors live on the ROCm device.  The Blender / MessyTable / LLFF loaders and the ray-cache format are
host-side numpy + PIL (nerf/datasets.py, nerf/llff.py).
"""
from . import models, parallel, synthetic  # noqa: F401  (scripts use getattr(models, cfg.models.coarse.type))
from ._ops import get_precision, get_render_policy, s8_grad_stats, set_precision, set_render_policy, set_s8_grad_scale  # noqa: F401
from .cfgnode import CfgNode  # noqa: F401
from .fused_step import FusedTrainStep, GraphedTrainStep  # noqa: F401
from .parallel import FlatAdam  # noqa: F401
from .models import *  # noqa: F401,F403
from .nerf_helpers import *  # noqa: F401,F403
from .train_utils import *  # noqa: F401,F403
from .volume_rendering_utils import *  # noqa: F401,F403


from .datasets import (load_blender_data, load_llff_data, load_messytable_data, load_ray_cache, pose_spherical,  # noqa: F401,E402
                       save_ray_cache)
