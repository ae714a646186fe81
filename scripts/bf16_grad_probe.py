"""Developer probe: per-parameter gradient agreement of the fused training path (fp32 / bf16) with the reference goldens."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import nerf
from golden_cases import CASES, M_THRES, draws_of
from test_hip_parity import make_models, make_cfg, G, C
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "train_d8w256"
g = dict(np.load(os.path.join(REPO, "tests", "golden", name + ".npz")))
mkw, wfn, rkw = CASES[name]
for prec in ("fp32", "bf16"):
    nerf.set_precision(prec)
    mc, mf = make_models(mkw, *wfn(), dev)
    d = draws_of(g)
    q_rand = [G(d["t_rand"], dev), G(d["u"], dev)]; q_randn = [G(d["noise_c"], dev), G(d["noise_f"], dev)]
    orand, orandn = torch.rand, torch.randn
    torch.rand = lambda *a, **k: q_rand.pop(0); torch.randn = lambda *a, **k: q_randn.pop(0)
    try:
        ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
        out = nerf.run_one_iter_of_nerf(1, len(g["ro"]), 1.0, mc, mf, G(g["ro"], dev), G(g["rd"], dev), make_cfg(rkw), mode="train",
                                        encode_position_fn=ex, encode_direction_fn=ed, m_thres_cand=list(M_THRES))
    finally:
        torch.rand, torch.randn = orand, orandn
    target = G(g["target"], dev)
    loss = nerf.img2mse(out[0][..., :3], target) + nerf.img2mse(out[3][..., :3], target)
    loss.backward()
    print(prec, "loss", loss.item(), "ref", float(g["loss"]))
    for pref, m in (("gc_", mc), ("gf_", mf)):
        for k, p in m.named_parameters():
            key = pref + k
            ours = C(p.grad).reshape(-1).astype(np.float64)
            ref = (g[key].reshape(-1) if key in g else None)
            if ref is None:
                ours = ours[::97]; ref = g[key + ".sub"]
            ref = ref.astype(np.float64)
            cos = ours @ ref / max(np.linalg.norm(ours) * np.linalg.norm(ref), 1e-30)
            print(f"  {key:28s} n={ref.size:6d} cos={cos:.4f} |ref|={np.linalg.norm(ref):.3e} |ours|={np.linalg.norm(ours):.3e}")
