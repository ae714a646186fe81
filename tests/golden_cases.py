"""Shared description of the golden fixtures (which nets / render settings each one used)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd"))
from nerf import synthetic as syn  # noqa: E402

M_THRES = tuple(float(x) for x in range(5, 105, 5))
D8 = dict(num_layers=8, hidden_size=256, skip_connect_every=4, num_encoding_fn_xyz=10,
          num_encoding_fn_dir=4, use_viewdirs=True)
D4 = dict(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10,
          num_encoding_fn_dir=4, use_viewdirs=True)


def lego_weights():
    w = dict(np.load(os.path.join(HERE, "golden", "lego_weights.npz")))
    return ({k[3:]: v for k, v in w.items() if k.startswith("wc_")},
            {k[3:]: v for k, v in w.items() if k.startswith("wf_")})


def d8_weights():
    return (syn.synth_state_dict(42, sigma_bias=-150.0, **D8), syn.synth_state_dict(43, sigma_bias=-20.0, **D8))


# name -> (model kwargs, weights fn, render settings)
CASES = {
    "render_lego_val": (D4, lego_weights, dict(num_coarse=64, num_fine=64, near=2.0, far=6.0, white_background=True)),
    "render_lego_val_64_128": (D4, lego_weights, dict(num_coarse=64, num_fine=128, near=2.0, far=6.0)),
    "render_d8w256_val": (D8, d8_weights, dict(num_coarse=64, num_fine=128, near=2.0, far=6.0)),
    "render_d8w256_lindisp": (D8, d8_weights, dict(num_coarse=64, num_fine=64, near=0.3, far=4.0, lindisp=True)),
    "train_lego": (D4, lego_weights, dict(num_coarse=64, num_fine=64, near=2.0, far=6.0, white_background=True,
                                          perturb=True, noise_std=0.2)),
    "train_d8w256": (D8, d8_weights, dict(num_coarse=64, num_fine=128, near=2.0, far=6.0, perturb=True,
                                          noise_std=0.2)),
}


def draws_of(g):
    """The reference's recorded RNG draws in its call order."""
    if "draw_rand0" not in g:
        return None
    return dict(t_rand=g["draw_rand0"], noise_c=g["draw_randn0"], u=g["draw_rand1"], noise_f=g["draw_randn1"])
