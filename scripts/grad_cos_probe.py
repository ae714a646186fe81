"""Developer probe: bf16-vs-fp32 parameter-gradient cosine of the fused training kernels on identical points
(the check of tests/test_hip_parity.py::test_fused_training_kernels, all tensors printed)."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import nerf
from golden_cases import CASES
from test_hip_parity import make_models, G, C
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "train_d8w256"
g = dict(np.load(os.path.join(REPO, "tests", "golden", name + ".npz")))
mkw, wfn, _ = CASES[name]
pts = G(g["pts_fine"], dev); rd = G(g["rd"], dev)
vd = torch.nn.functional.normalize(rd, dim=-1)
rays = torch.cat([torch.zeros(len(rd), 8, device=dev), vd], -1)
g_up = G(np.random.default_rng(3).normal(size=pts.shape[:2] + (4,)).astype(np.float32), dev)
ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
grads = {}
for prec in ("fp32", "bf16"):
    nerf.set_precision(prec)
    _, mf = make_models(mkw, *wfn(), dev)
    out = nerf.run_network(mf, pts, rays, 4096, ex, ed)
    (out * g_up).sum().backward()
    grads[prec] = {k: C(p.grad).astype(np.float64).reshape(-1) for k, p in mf.named_parameters()}
for k in grads["fp32"]:
    a, b = grads["bf16"][k], grads["fp32"][k]
    print(f"{k:24s} cos {a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-30):.4f}  |fp32| {np.linalg.norm(b):.3e}")
