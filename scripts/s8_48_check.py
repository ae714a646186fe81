#!/usr/bin/env python3
"""Developer probe: stage-by-stage comparison of the 48-point 8-bit training kernels (DN_PREC_BF16_S8) with the 32-point bf16
ones - saved activations, saved gradients (in backward-chain order), weight gradients.  Usage: s8_48_check.py [D W viewdirs skip]"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "dex-nerf_amd")]
import nerf
from nerf import _hip, _ops, _train
depth, width, viewdirs, skip = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (8, 256, 1, 4)
viewdirs = bool(viewdirs)
dev = torch.device("cuda:0")
nerf.set_precision("bf16")
torch.manual_seed(3)
m = nerf.models.FlexibleNeRFModel(num_layers=depth, hidden_size=width, skip_connect_every=skip, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=viewdirs).to(dev)
pk = m.packed()
weights = [x.weight for x in m.linear_modules()]
_ops.pack_backward(pk, weights, _hip.PREC_BF16)
_ops.pack_backward(pk, weights, _hip.PREC_BF16_S8)
n_rays, s = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) >= 7 else (37, 53)
n = n_rays * s
pts = torch.rand(n, 3, device=dev) * 2 - 1
vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1) if viewdirs else None
g_out = torch.randn(n, 4, device=dev) * 1e-4
out, act, masks = _ops.run_network_train(pk, pts, vd, s, prec=_hip.PREC_BF16)
out8, act8, masks8 = _ops.run_network_train(pk, pts, vd, s, prec=_hip.PREC_BF16_S8)
inf = _ops.run_network_pts(pk, pts, vd, s)
print("forward == 48-point inference:", torch.equal(out8, inf), " non-finite outputs:", int((~torch.isfinite(out8)).sum()), "of", out8.numel(),
      " first bad point:", (int((~torch.isfinite(out8)).any(1).nonzero()[0]) if (~torch.isfinite(out8)).any() else None),
      " mismatching points:", int((out8 != inf).any(1).sum()))
if viewdirs:
    rays_ = torch.cat([torch.zeros(n_rays, 3, device=dev), vd * 1.5, torch.zeros(n_rays, 2, device=dev), vd], -1).contiguous()
    z_ = torch.sort(torch.rand(n_rays, s, device=dev) * 4 + 2, -1)[0].contiguous()
    o_r, a_r, m_r = _ops.run_network_train(pk, None, None, None, rays=rays_, z_vals=z_, prec=_hip.PREC_BF16_S8)
    i_r = _ops.run_network_rays(pk, rays_, z_).reshape(-1, 4)
    print("rays form: forward == inference:", torch.equal(o_r, i_r), " non-finite:", int((~torch.isfinite(o_r)).sum()), " mismatching points:", int((o_r != i_r).any(1).sum()))
    if (o_r != i_r).any():
        badp = (o_r != i_r).any(1).nonzero().flatten().cpu().numpy()
        print("  mismatching points mod 384:", np.unique(badp % 384)[:40], " tiles:", np.unique(badp // 384)[:20], "...", len(np.unique(badp // 384)))
grads = _ops.mlp_backward_data(pk, g_out, masks, n, prec=_hip.PREC_BF16)
grads8 = _ops.mlp_backward_data(pk, g_out, masks8, n, prec=_hip.PREC_BF16_S8)
slots, gslots, kh = _train._slots(m, _hip.PREC_BF16)
d, w = depth, width
khu = w // 64
s8 = {"xyz": 0, "dir": 1, "layer1": 1 + (1 if viewdirs else 0)}
s8["trunk0"] = s8["layer1"] + khu; s8["feat"] = s8["trunk0"] + (d - 1) * khu; s8["dirout"] = s8["feat"] + (khu if viewdirs else 0)
g8 = {"dirout": 0, "feat": (w // 128) if viewdirs else 0}
g8["trunk0"] = g8["feat"] + (khu if viewdirs else 0); g8["layer1"] = g8["trunk0"] + (d - 1) * khu; g8["out"] = g8["layer1"] + khu
C = lambda t: t.detach().cpu().numpy().astype(np.float64)
def rows16(which, buf, slot, width_, kind=0):
    return C(_ops.mlp_unpack(pk, which, buf, n, slot, width_, kind, torch.zeros((n, width_), dtype=torch.float32, device=dev)))
def rows8(which, buf, slot, width_, kind=0, cols=None):
    return C(_ops.mlp_unpack(pk, which, buf, n, slot, width_, kind, torch.zeros((n, cols or width_), dtype=torch.float32, device=dev), prec=_hip.PREC_BF16_S8))
def rep(what, a8, a16, rel):
    ok = (np.abs(a8 - a16) <= rel * np.abs(a16) + (2.0 ** -8 if rel < 0.1 else 1e-9)).mean()
    cos = float((a8 * a16).sum() / max(np.linalg.norm(a8) * np.linalg.norm(a16), 1e-300))
    # per 16-point group / per feature-block diagnostics
    bad_pts = np.where((np.abs(a8 - a16) > rel * np.abs(a16) + 1e-3 * np.abs(a16).max()).any(1))[0]
    print(f"{what:28s} within {ok:.4f}  cosine {cos:.5f}  |max16| {np.abs(a16).max():.3e} |max8| {np.abs(a8).max():.3e}  bad points {len(bad_pts)} {bad_pts[:6]}")
rep("xyz", rows8(0, act8, s8["xyz"], m.dim_xyz, 1), rows16(0, act, slots["xyz"], m.dim_xyz, 1), 0.07)
if viewdirs: rep("dir", rows8(0, act8, s8["dir"], m.dim_dir, 2), rows16(0, act, slots["dir"], m.dim_dir, 2), 0.07)
rep("layer1", rows8(0, act8, s8["layer1"], w), rows16(0, act, slots["layer1"], w), 0.07)
for i in range(d - 1):
    rep(f"layers_xyz[{i}]", rows8(0, act8, s8["trunk0"] + i * khu, w), rows16(0, act, slots["trunk0"] + i * kh, w), 0.07)
if viewdirs:
    rep("fc_feat", rows8(0, act8, s8["feat"], w), rows16(0, act, slots["feat"], w), 0.07)
    rep("layers_dir.0", rows8(0, act8, s8["dirout"], max(w // 2, 64))[:, : w // 2], rows16(0, act, slots["dirout"], w // 2), 0.07)
print("---- gradients, backward-chain order")
custom = rows8(1, grads8, g8["out"], 8, 3, cols=8)
rep("custom unit", custom[:, [0, 1, 2, 4]] if viewdirs else custom[:, :4], C(g_out), 0.14)
if viewdirs:
    rep("d layers_dir.0", rows8(1, grads8, g8["dirout"], max(w // 2, 64))[:, : w // 2], rows16(1, grads, gslots["dirout"], w // 2), 0.14)
    rep("d fc_feat", rows8(1, grads8, g8["feat"], w), rows16(1, grads, gslots["feat"], w), 0.14)
for i in range(d - 2, -1, -1):
    rep(f"d layers_xyz[{i}]", rows8(1, grads8, g8["trunk0"] + i * khu, w), rows16(1, grads, gslots["trunk0"] + i * kh, w), 0.14)
rep("d layer1", rows8(1, grads8, g8["layer1"], w), rows16(1, grads, gslots["layer1"], w), 0.14)
print("---- weight gradients (fp8 MFMA on 8-bit buffers vs bf16 MFMA on bf16 buffers)")
shapes = [tuple(x.weight.shape) for x in m.linear_modules()]
ref = _ops.mlp_weight_grad_all(pk, act, grads, n, shapes, prec=_hip.PREC_BF16)
got = _ops.mlp_weight_grad_all(pk, act8, grads8, n, shapes, prec=_hip.PREC_BF16_S8)
names = [k[:-7] for k, _ in m.named_parameters() if k.endswith(".weight")]
for (w16, b16), (w8, b8), shp in zip(ref, got, shapes):
    cw = float((C(w8) * C(w16)).sum() / max(np.linalg.norm(C(w8)) * np.linalg.norm(C(w16)), 1e-300))
    cb = float((C(b8) * C(b16)).sum() / max(np.linalg.norm(C(b8)) * np.linalg.norm(C(b16)), 1e-300))
    print(f"{str(shp):14s} dW cosine {cw:.5f}  db cosine {cb:.5f}  finite {bool(torch.isfinite(w8).all())}")
if viewdirs:
    print("---- first backward stage against plain arithmetic")
    a_dir8 = rows8(0, act8, s8["dirout"], max(w // 2, 64))[:, : w // 2]
    a_dir16 = rows16(0, act, slots["dirout"], w // 2)
    g8r = rows8(1, grads8, g8["dirout"], max(w // 2, 64))[:, : w // 2]
    g16r = rows16(1, grads, gslots["dirout"], w // 2)
    wr = C(m.fc_rgb.weight.to(torch.bfloat16).float())
    full = C(g_out[:, :3].to(torch.bfloat16).float()) @ wr
    for nm, gg, aa in (("32-point", g16r, a_dir16), ("48-point", g8r, a_dir8)):
        exp = full * (aa > 0)
        nz_match = ((gg != 0) == (aa > 0)).mean()
        ok = (np.abs(gg - exp) <= 0.14 * np.abs(exp) + 1e-9).mean()
        print(f"{nm}: nonzero pattern == (activation > 0): {nz_match:.4f}; value within tolerance of (d rgb @ W_rgb) * mask: {ok:.4f}; unmasked-value agreement where nonzero: "
              f"{(np.abs(gg - full)[gg != 0] <= 0.14 * np.abs(full)[gg != 0] + 1e-9).mean():.4f}")
    # which features / points disagree
    bad = np.abs(g8r - full * (a_dir8 > 0)) > 0.14 * np.abs(full) + 1e-9
    print("48-point: bad fraction per feature block of 16:", np.round(bad.reshape(n, -1, 16).mean((0, 2)), 2))
    print("48-point: bad fraction per point (mod 48) block of 16:", np.round(np.array([bad[np.arange(n) % 48 // 16 == k].mean() for k in range(3)]), 2))
    print("48-point: bad fraction per feature mod 16:", np.round(bad.reshape(n, -1, 16).mean((0, 1)), 2))
if viewdirs:
    print("---- forward mask words of the layers_dir.0 stage, decoded here, against (saved activation > 0)")
    mk = masks8.cpu().numpy().view(np.uint32)
    stages = (d - 1) + 2
    nf = w // 2
    agree = np.zeros((nf,)); cnt = 0
    per_rd = np.zeros((4,))
    for p_ in range(0, n, 7):
        tile, wv, t, j = p_ // 384, (p_ % 384) // 48, (p_ % 48) // 16, p_ % 16
        base = ((tile * 8 + wv) * stages + d) * 512   # dwords: 2 KiB per stage
        for f in range(nf):
            nt, g, r = f // 16, (f % 16) // 4, f % 4
            lane = g * 16 + j
            word = mk[base + (0 if t < 2 else 256) + lane * 4 + (t % 2) * 2 + (nt >> 3)]
            bit = (word >> (((nt & 7) * 2 + r // 2) + 16 * (r % 2))) & 1
            ok = int(bit) == int(a_dir8[p_, f] > 0)
            agree[f] += ok; per_rd[r] += ok
        cnt += 1
    print("mask bit == (activation > 0), per register r of a tile:", np.round(per_rd / (cnt * nf / 4), 3), " per feature (first 32):", np.round(agree[:32] / cnt, 2))
