#!/usr/bin/env python3
"""Golden-vector capture: runs the REFERENCE library (CPU PyTorch) and records inputs/outputs.

Run in the build container only (the reference lives at /root/reference and never travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What it does:
  * pre-seeds sys.modules with a `torchsearchsorted` shim (= torch.searchsorted, right=side=="right";
    the un-vendored third-party op the reference calls at nerf_helpers.py:290) and empty `cv2` /
    `imageio` stubs (only the dataset loaders touch those), then imports /root/reference/nerf-pytorch/nerf;
  * drives the reference's *library functions* directly and records every boundary by wrapping
    them (run_network, volume_render_radiance_field, sample_pdf, searchsorted, torch.rand/randn) -
    no reference logic is restated here;
  * writes small .npz fixtures next to this file.  Fixtures are data only (arrays).

The D8/W256 skip-4 net needs the one-line harness alias documented in SURVEY.md section 2
(`m.__dict__["linear_layers"] = m.layers_xyz`): the reference class references an attribute it
never defines (models.py:243).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/nerf-pytorch"

# --- import shims -------------------------------------------------------------------------------
_search_log = []


def _searchsorted(a, v, side="left"):
    out = torch.searchsorted(a, v, right=(side == "right"))
    _search_log.append((a.detach().clone(), v.detach().clone(), out.clone()))
    return out


_ts = types.ModuleType("torchsearchsorted")
_ts.searchsorted = _searchsorted
sys.modules["torchsearchsorted"] = _ts
for _m in ("cv2", "imageio"):
    sys.modules[_m] = types.ModuleType(_m)
sys.path.insert(0, REF)
import nerf as ref  # noqa: E402  (the reference package)

_spec = importlib.util.spec_from_file_location(
    "dexnerf_synthetic", os.path.join(REPO, "dex-nerf_amd", "nerf", "synthetic.py"))
syn = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(syn)

torch.set_num_threads(8)
M_THRES = np.arange(5, 100 + 5, 5)  # train_dexnerf_rgb.py:153-154 with m_thres: 100


def npf(t):
    return t.detach().cpu().numpy()


def make_cfg(num_coarse, num_fine, near, far, perturb, noise_std, white, lindisp=False,
             chunksize=4096, use_viewdirs=True, no_ndc=True):
    mode = dict(chunksize=chunksize, lindisp=lindisp, num_coarse=num_coarse, num_fine=num_fine,
                perturb=perturb, radiance_field_noise_std=noise_std, white_background=white)
    d = dict(dataset=dict(near=near, far=far, no_ndc=no_ndc),
             nerf=dict(use_viewdirs=use_viewdirs, train=dict(mode), validation=dict(mode)))
    return ref.CfgNode(d)


def build_model(sd_np, **kw):
    m = ref.models.FlexibleNeRFModel(**kw)
    m.__dict__["linear_layers"] = m.layers_xyz  # harness alias, see module docstring
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    return m


class Recorder:
    """Wraps the reference's stage functions inside nerf.train_utils to log their I/O."""

    def __init__(self, inject_rand=None, inject_randn=None):
        self.calls = {"run_network": [], "volume_render": [], "sample_pdf": [], "rand": [], "randn": []}
        self.inject_rand = list(inject_rand) if inject_rand else None
        self.inject_randn = list(inject_randn) if inject_randn else None

    def __enter__(self):
        tu = ref.train_utils
        self._orig = (tu.run_network, tu.volume_render_radiance_field, tu.sample_pdf, torch.rand, torch.randn)
        rec = self

        def run_network(network_fn, pts, ray_batch, chunksize, embed_fn, embeddirs_fn):
            out = rec._orig[0](network_fn, pts, ray_batch, chunksize, embed_fn, embeddirs_fn)
            rec.calls["run_network"].append((pts.detach().clone(), out.detach().clone()))
            return out

        def volume_render(*a, **k):
            out = rec._orig[1](*a, **k)
            rec.calls["volume_render"].append(([x.detach().clone() for x in a[:3]], [o.detach().clone() for o in out]))
            return out

        def sample_pdf(bins, weights, n, det=False):
            out = rec._orig[2](bins, weights, n, det=det)
            rec.calls["sample_pdf"].append((bins.detach().clone(), weights.detach().clone(), out.detach().clone()))
            return out

        def rand(*a, **k):
            out = rec._orig[3](*a, **k)
            if rec.inject_rand is not None:
                out = rec.inject_rand.pop(0).to(out)
            rec.calls["rand"].append(out.clone())
            return out

        def randn(*a, **k):
            out = rec._orig[4](*a, **k)
            if rec.inject_randn is not None:
                out = rec.inject_randn.pop(0).to(out)
            rec.calls["randn"].append(out.clone())
            return out

        tu.run_network, tu.volume_render_radiance_field, tu.sample_pdf = run_network, volume_render, sample_pdf
        ref.volume_rendering_utils.torch.randn = randn  # same module object as torch
        torch.rand, torch.randn = rand, randn
        _search_log.clear()
        return self

    def __exit__(self, *exc):
        tu = ref.train_utils
        tu.run_network, tu.volume_render_radiance_field, tu.sample_pdf, torch.rand, torch.randn = self._orig


def rays_for(height, width, pose_index, count, seed):
    E = torch.from_numpy(syn.scene_pose(pose_index))
    K = torch.from_numpy(syn.intrinsic(height, width))
    ro, rd = ref.get_ray_bundle(height, width, float(K[0, 0]), E, K)
    sel = syn.select_rays(height, width, count, seed)
    return E, K, ro.reshape(-1, 3)[sel].contiguous(), rd.reshape(-1, 3)[sel].contiguous(), sel


def capture_render(tag, model_c, model_f, cfg, ro, rd, mode, l_xyz, l_dir, extra=None,
                   inject_rand=None, inject_randn=None, target=None, save_grads=None):
    ex = ref.get_embedding_function(num_encoding_functions=l_xyz, include_input=True, log_sampling=True)
    ed = ref.get_embedding_function(num_encoding_functions=l_dir, include_input=True, log_sampling=True)
    grad_ctx = torch.enable_grad() if target is not None else torch.no_grad()
    with Recorder(inject_rand, inject_randn) as rec, grad_ctx:
        out = ref.run_one_iter_of_nerf(1, ro.shape[0], 1.0, model_c, model_f, ro[None], rd[None], cfg,
                                       mode=mode, encode_position_fn=ex, encode_direction_fn=ed,
                                       m_thres_cand=M_THRES)
        loss = None
        if target is not None:
            loss = ((out[0].reshape(-1, 3) - target) ** 2).mean() + ((out[3].reshape(-1, 3) - target) ** 2).mean()
            loss.backward()
    d = dict(ro=npf(ro), rd=npf(rd), m_thres=M_THRES.astype(np.float32))
    names = ["rgb_coarse", "depth_coarse", "acc_coarse", "rgb_fine", "depth_fine", "acc_fine"]
    for n, o in zip(names, out[:6]):
        d["out_" + n] = npf(o).reshape((-1, 3) if n.startswith("rgb") else (-1,))
    d["out_dex_fine"] = np.stack([npf(o).reshape(-1) for o in out[6:]], 0)
    (pts_c, rf_c), (pts_f, rf_f) = rec.calls["run_network"]
    (vin_c, vout_c), (vin_f, vout_f) = rec.calls["volume_render"]
    d.update(pts_coarse=npf(pts_c), rf_coarse=npf(rf_c), pts_fine=npf(pts_f), rf_fine=npf(rf_f))
    d.update(z_coarse=npf(vin_c[1]), z_fine=npf(vin_f[1]))
    vnames = ["rgb", "disp", "acc", "weights", "depth"]
    for n, o in zip(vnames, vout_c[:5]):
        d["vc_" + n] = npf(o)
    d["vc_dex"] = np.stack([npf(o) for o in vout_c[5:]], 0)
    for n, o in zip(vnames, vout_f[:5]):
        d["vf_" + n] = npf(o)
    d["vf_dex"] = np.stack([npf(o) for o in vout_f[5:]], 0)
    (bins, w_in, zs), = rec.calls["sample_pdf"]
    d.update(sp_bins=npf(bins), sp_weights=npf(w_in), sp_z_samples=npf(zs))
    (cdf, u, inds), = _search_log
    d.update(sp_cdf=npf(cdf), sp_u=npf(u), sp_inds=npf(inds).astype(np.int64))
    for i, r in enumerate(rec.calls["rand"]):
        d[f"draw_rand{i}"] = npf(r)
    for i, r in enumerate(rec.calls["randn"]):
        d[f"draw_randn{i}"] = npf(r)
    # positional-encoding spot check on the first 64 coarse points / rays
    vd = rd / rd.norm(p=2, dim=-1, keepdim=True)
    d["pe_xyz_in"] = npf(pts_c.reshape(-1, 3)[:64])
    d["pe_xyz_out"] = npf(ex(pts_c.reshape(-1, 3)[:64]))
    d["pe_dir_in"] = npf(vd[:64])
    d["pe_dir_out"] = npf(ed(vd[:64]))
    if target is not None:
        d["target"] = npf(target)
        d["loss"] = np.float64(loss.item())
        for pref, m in (("gc_", model_c), ("gf_", model_f)):
            for k, p in m.named_parameters():
                if k.startswith("linear_layers"):
                    continue
                g = npf(p.grad)
                if save_grads == "full":
                    d[pref + k] = g
                else:  # strided subsample + norm, keeps the fixture small
                    d[pref + k + ".sub"] = g.reshape(-1)[::97].copy()
                    d[pref + k + ".norm"] = np.float64(np.linalg.norm(g.astype(np.float64)))
    if extra:
        d.update(extra)
    path = os.path.join(HERE, tag + ".npz")
    np.savez_compressed(path, **d)
    print(f"{tag}: {os.path.getsize(path) / 1e6:.2f} MB, rays={ro.shape[0]}")
    return d, out


# ------------------------------------------------------------------------------------------------
def kat():
    """Small known-answer vectors for each primitive."""
    d = {}
    rng = np.random.default_rng(7)
    # G1 ray bundle KAT
    E = torch.tensor([[0, 1, 0, .1], [0, 0, -1, .2], [-1, 0, 0, 2], [0, 0, 0, 1]], dtype=torch.float32)
    K = torch.tensor([[100, 0, 2], [0, 120, 1.5], [0, 0, 1]], dtype=torch.float32)
    ro, rd = ref.get_ray_bundle(3, 4, 100.0, E, K)
    d.update(rb0_E=npf(E), rb0_K=npf(K), rb0_ro=npf(ro), rb0_rd=npf(rd))
    E1 = torch.from_numpy(syn.scene_pose(3))
    K1 = torch.from_numpy(syn.intrinsic(20, 30))
    ro, rd = ref.get_ray_bundle(20, 30, float(K1[0, 0]), E1, K1)
    d.update(rb1_E=npf(E1), rb1_K=npf(K1), rb1_ro=npf(ro), rb1_rd=npf(rd))
    d["rb1_Einv"] = npf(torch.inverse(E1))
    d["rb1_Rinv"] = npf(torch.inverse(E1[:3, :3]))
    # positional encoding
    x = torch.from_numpy(rng.uniform(-6, 6, size=(97, 3)).astype(np.float32))
    d["pe_x"] = npf(x)
    for name, kw in (("pe_l10", dict(num_encoding_functions=10)),
                     ("pe_l4", dict(num_encoding_functions=4)),
                     ("pe_l6_lin", dict(num_encoding_functions=6, log_sampling=False)),
                     ("pe_l4_noinput", dict(num_encoding_functions=4, include_input=False)),
                     ("pe_l0", dict(num_encoding_functions=0))):
        d[name] = npf(ref.positional_encoding(x, **kw))
    d["pe_kat_in"] = np.array([[.1, .2, .3]], np.float32)
    d["pe_kat_out"] = npf(ref.positional_encoding(torch.tensor([[.1, .2, .3]]), num_encoding_functions=2))
    # cumprod_exclusive
    t = torch.from_numpy(rng.uniform(0.0, 1.0, size=(5, 17)).astype(np.float32))
    d["cpe_in"] = npf(t)
    d["cpe_out"] = npf(ref.cumprod_exclusive(t))
    # composite KAT (section 8a row S6) + random block with edge rows
    rf = torch.zeros(2, 4, 4)
    rf[0, :, 3] = torch.tensor([-1., 3., 50., 7.])
    rf[0, 1, :3] = torch.tensor([1., -1., .5])
    rf[1, :, 3] = -1.0
    z = torch.tensor([[2., 2.5, 3.5, 6.]]).expand(2, 4).contiguous()
    rdk = torch.tensor([[0., 0., 2.], [0., 0., 2.]])
    out = ref.volume_render_radiance_field(rf, z, rdk, m_thres_cand=[5.0, 10.0])
    d.update(vr0_rf=npf(rf), vr0_z=npf(z), vr0_rd=npf(rdk))
    for n, o in zip(["rgb", "disp", "acc", "weights", "depth", "dex5", "dex10"], out):
        d["vr0_" + n] = npf(o)
    n, s = 96, 64
    rf = torch.from_numpy(rng.normal(0, 1, size=(n, s, 4)).astype(np.float32))
    rf[..., 3] = torch.from_numpy((rng.normal(0, 30, size=(n, s))).astype(np.float32))
    rf[0, :, 3] = -5.0                      # acc == 0 ray -> NaN disp, dex = z[0]
    rf[1, :, 3] = -5.0
    rf[1, 0, 3] = 200.0                     # crossing exactly at sample 0
    rf[2, :, 3] = -5.0
    rf[2, -1, 3] = 200.0                    # crossing only at the last sample
    rf[3, :, 3] = 5.0                       # sigma == threshold is NOT a crossing (strict >)
    zz = np.sort(rng.uniform(2, 6, size=(n, s)).astype(np.float32), axis=-1)
    z = torch.from_numpy(zz)
    rdr = torch.from_numpy(rng.normal(0, 1, size=(n, 3)).astype(np.float32))
    noise = torch.from_numpy(rng.normal(0, 1, size=(n, s)).astype(np.float32))
    d.update(vr1_rf=npf(rf), vr1_z=npf(z), vr1_rd=npf(rdr), vr1_noise=npf(noise), vr1_m=M_THRES.astype(np.float32))
    for tag, std, white in (("a", 0.0, False), ("b", 0.0, True), ("c", 0.2, True)):
        with Recorder(inject_randn=[noise]):
            out = ref.volume_render_radiance_field(rf, z, rdr, radiance_field_noise_std=std,
                                                   white_background=white, m_thres_cand=M_THRES)
        for nme, o in zip(["rgb", "disp", "acc", "weights", "depth"], out[:5]):
            d[f"vr1{tag}_{nme}"] = npf(o)
        d[f"vr1{tag}_dex"] = np.stack([npf(o) for o in out[5:]], 0)
    # sample_pdf KAT + random, det and with injected u
    bins = torch.tensor([[2., 3., 4., 5.]])
    w = torch.tensor([[0., .8, .1]])
    d.update(sp0_bins=npf(bins), sp0_w=npf(w), sp0_out=npf(ref.sample_pdf_2(bins, w, 6, det=True)))
    for tag, nb, nf in (("sp1", 63, 128), ("sp2", 127, 64), ("sp3", 63, 64)):
        bz = np.sort(rng.uniform(2, 6, size=(200, nb)).astype(np.float32), axis=-1)
        ww = rng.uniform(0, 1, size=(200, nb - 1)).astype(np.float32) ** 8
        ww[:5] = 0.0                         # degenerate pdf rows
        ww[5:10] = ww[5:10] * 0 + rng.uniform(0, 1e-6, size=(5, nb - 1)).astype(np.float32)
        ww[10, :] = 0.0
        ww[10, 17] = 1.0                     # delta pdf
        u_in = torch.from_numpy(rng.uniform(0, 1, size=(200, nf)).astype(np.float32))
        _search_log.clear()
        out_det = ref.sample_pdf_2(torch.from_numpy(bz), torch.from_numpy(ww), nf, det=True)
        cdf, u_det, inds_det = _search_log[-1]
        with Recorder(inject_rand=[u_in]):
            out_rnd = ref.sample_pdf_2(torch.from_numpy(bz), torch.from_numpy(ww), nf, det=False)
            _, _, inds_rnd = _search_log[-1]
        legacy = ref.nerf_helpers.sample_pdf(torch.from_numpy(bz), torch.from_numpy(ww), nf, det=True)
        d.update({f"{tag}_bins": bz, f"{tag}_w": ww, f"{tag}_u": npf(u_in), f"{tag}_cdf": npf(cdf),
                  f"{tag}_det": npf(out_det), f"{tag}_det_inds": npf(inds_det).astype(np.int64),
                  f"{tag}_rnd": npf(out_rnd), f"{tag}_rnd_inds": npf(inds_rnd).astype(np.int64),
                  f"{tag}_legacy_det": npf(legacy)})
    # ndc_rays
    o = torch.from_numpy(rng.normal(0, 1, size=(50, 3)).astype(np.float32))
    dd = torch.from_numpy(rng.normal(0, 1, size=(50, 3)).astype(np.float32))
    dd[:, 2] = -dd[:, 2].abs() - 0.1
    no, nd = ref.ndc_rays(378, 504, 407.5, 1.0, o, dd)
    d.update(ndc_o=npf(o), ndc_d=npf(dd), ndc_out_o=npf(no), ndc_out_d=npf(nd))
    # misc helpers
    a = torch.from_numpy(rng.uniform(0, 1, size=(10, 3)).astype(np.float32))
    b = torch.from_numpy(rng.uniform(0, 1, size=(10, 3)).astype(np.float32))
    d.update(mse_a=npf(a), mse_b=npf(b), mse=np.float64(ref.img2mse(a, b).item()),
             psnr=np.float64(ref.mse2psnr(ref.img2mse(a, b).item())), psnr0=np.float64(ref.mse2psnr(0)))
    # the model alone (4x128 default and D8/W256 skip 4), 33 random 90-wide inputs
    xin = torch.from_numpy(rng.normal(0, 1, size=(33, 90)).astype(np.float32))
    d["mlp_in"] = npf(xin)
    for tag, kw in (("mlp_d4w128", dict(num_layers=4, hidden_size=128)),
                    ("mlp_d8w256", dict(num_layers=8, hidden_size=256)),
                    ("mlp_d8w256_noview", dict(num_layers=8, hidden_size=256, use_viewdirs=False))):
        full = dict(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10,
                    num_encoding_fn_dir=4, use_viewdirs=True)
        full.update(kw)
        m = build_model(syn.synth_state_dict(11, **full), **full)
        with torch.no_grad():
            d[tag] = npf(m(xin))
    path = os.path.join(HERE, "kat.npz")
    np.savez_compressed(path, **d)
    print(f"kat: {os.path.getsize(path) / 1e6:.2f} MB")


def lego():
    """As-shipped 4x128 nets with the reference's real lego-lowres weights; 64+64, white bg."""
    ck = torch.load(os.path.join(REF, "pretrained/lego-lowres/checkpoint199999.ckpt"), map_location="cpu")
    kw = dict(num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
    mc = ref.models.FlexibleNeRFModel(**kw)
    mf = ref.models.FlexibleNeRFModel(**kw)
    mc.load_state_dict(ck["model_coarse_state_dict"])
    mf.load_state_dict(ck["model_fine_state_dict"])
    w = {"wc_" + k: npf(v) for k, v in ck["model_coarse_state_dict"].items()}
    w.update({"wf_" + k: npf(v) for k, v in ck["model_fine_state_dict"].items()})
    np.savez_compressed(os.path.join(HERE, "lego_weights.npz"), **w)
    E, K, ro, rd, sel = rays_for(100, 100, 5, 256, seed=1)
    cfg = make_cfg(64, 64, 2.0, 6.0, perturb=False, noise_std=0.0, white=True)
    capture_render("render_lego_val", mc, mf, cfg, ro, rd, "validation", 10, 4,
                   extra=dict(E=npf(E), K=npf(K), sel=sel))
    # 64+128 variant (ship/hotdog configs), no white background
    cfg = make_cfg(64, 128, 2.0, 6.0, perturb=False, noise_std=0.0, white=False)
    capture_render("render_lego_val_64_128", mc, mf, cfg, ro[:128], rd[:128], "validation", 10, 4)
    # one training step: perturb + noise, injected draws, full grads (84,548 params per net)
    rng = np.random.default_rng(21)
    n = 96
    cfg = make_cfg(64, 64, 2.0, 6.0, perturb=True, noise_std=0.2, white=True)
    draws_rand = [torch.from_numpy(rng.uniform(0, 1, size=(n, 64)).astype(np.float32)),
                  torch.from_numpy(rng.uniform(0, 1, size=(n, 64)).astype(np.float32))]
    draws_randn = [torch.from_numpy(rng.normal(0, 1, size=(n, 64)).astype(np.float32)),
                   torch.from_numpy(rng.normal(0, 1, size=(n, 128)).astype(np.float32))]
    target = torch.from_numpy(rng.uniform(0, 1, size=(n, 3)).astype(np.float32))
    for m in (mc, mf):
        m.zero_grad()
    d, _ = capture_render("train_lego", mc, mf, cfg, ro[:n], rd[:n], "train", 10, 4,
                          inject_rand=draws_rand, inject_randn=draws_randn, target=target, save_grads="full")
    # Adam step on top (train_dexnerf_rgb.py:147-150,280): lr 5e-3
    params = list(mc.parameters()) + list(mf.parameters())
    opt = torch.optim.Adam(params, lr=5e-3)
    opt.step()
    post = {"pc_" + k: npf(v) for k, v in mc.state_dict().items()}
    post.update({"pf_" + k: npf(v) for k, v in mf.state_dict().items()})
    np.savez_compressed(os.path.join(HERE, "train_lego_post_adam.npz"), **post)


def d8w256():
    """North-star nets: D=8, W=256, skip 4 (needs the harness alias), 64+128."""
    full = dict(num_layers=8, hidden_size=256, skip_connect_every=4, num_encoding_fn_xyz=10,
                num_encoding_fn_dir=4, use_viewdirs=True)
    mc = build_model(syn.synth_state_dict(42, sigma_bias=-150.0, **full), **full)
    mf = build_model(syn.synth_state_dict(43, sigma_bias=-20.0, **full), **full)
    E, K, ro, rd, sel = rays_for(400, 400, 7, 192, seed=0)
    cfg = make_cfg(64, 128, 2.0, 6.0, perturb=False, noise_std=0.0, white=False)
    capture_render("render_d8w256_val", mc, mf, cfg, ro, rd, "validation", 10, 4,
                   extra=dict(E=npf(E), K=npf(K), sel=sel, seed_c=42, seed_f=43, sigma_bias_c=-150.0, sigma_bias_f=-20.0))
    # lindisp + Dex-scene bounds (C3: near .3 far 4)
    cfg = make_cfg(64, 64, 0.3, 4.0, perturb=False, noise_std=0.0, white=False, lindisp=True)
    capture_render("render_d8w256_lindisp", mc, mf, cfg, ro[:64] * 0.25, rd[:64], "validation", 10, 4)
    # training step with injected draws; subsampled grads
    rng = np.random.default_rng(22)
    n = 64
    cfg = make_cfg(64, 128, 2.0, 6.0, perturb=True, noise_std=0.2, white=False)
    draws_rand = [torch.from_numpy(rng.uniform(0, 1, size=(n, 64)).astype(np.float32)),
                  torch.from_numpy(rng.uniform(0, 1, size=(n, 128)).astype(np.float32))]
    draws_randn = [torch.from_numpy(rng.normal(0, 1, size=(n, 64)).astype(np.float32)),
                   torch.from_numpy(rng.normal(0, 1, size=(n, 192)).astype(np.float32))]
    target = torch.from_numpy(rng.uniform(0, 1, size=(n, 3)).astype(np.float32))
    capture_render("train_d8w256", mc, mf, cfg, ro[:n], rd[:n], "train", 10, 4,
                   inject_rand=draws_rand, inject_randn=draws_randn, target=target, save_grads="sub")


def val_extras():
    """the validation error metrics + error image, and the training-ray selection.

    compute_err_metric / depth_error_img are the reference's library functions.  The selection itself lives in the
    reference's *script* (train_dexnerf_rgb.py:229-242, not importable here: tensorboard / torchvision), so its
    five lines of index plumbing are spelled out below on the reference's get_ray_bundle / meshgrid_xy, and the packed
    (N,11) rows are whatever the reference's run_one_iter_of_nerf hands to predict_and_render_radiance."""
    rng = np.random.default_rng(31)
    d = {}
    H, W, K = 24, 240, 6
    gt = rng.uniform(0.2, 1.6, size=(H, W)).astype(np.float32)
    gt[rng.uniform(size=(H, W)) < 0.1] = 0.0                       # holes in the ground-truth depth
    noise = rng.normal(0, 1, size=(K, H, W)).astype(np.float32)
    scale = np.array([0.0005, 0.002, 0.004, 0.008, 0.03, 0.3], np.float32)[:, None, None]
    pred = (gt[None] + noise * scale).astype(np.float32)
    pred[3, :4] = gt[:4]                                           # exact hits (error 0 -> black bin)
    gt_t = torch.from_numpy(gt)
    mask = (gt_t > 0) & (gt_t < 1.25)                               # train_dexnerf_rgb.py:392
    errs = []
    for k in range(K):
        e = ref.train_utils.compute_err_metric(gt_t, torch.from_numpy(pred[k]), mask)
        errs.append([e["depth_abs_err"], e["depth_err2"], e["depth_err4"], e["depth_err8"]])
    d.update(err_gt=gt, err_pred=pred, err_mask=mask.numpy(), err_out=np.array(errs, np.float64))
    for k in (1, 4):
        img = ref.train_utils.depth_error_img(torch.from_numpy(pred[k])[None] * 1000, gt_t[None] * 1000, mask[None])
        d[f"err_img_{k}"] = img.astype(np.float32)
    # training-ray selection, 30 x 40 image, 64 rays
    H, W, n = 30, 40, 64
    E = torch.from_numpy(syn.scene_pose(3))
    Kmat = torch.from_numpy(syn.intrinsic(H, W))
    image = torch.from_numpy(rng.uniform(0, 1, size=(H, W, 4)).astype(np.float32))
    ro, rd = ref.get_ray_bundle(H, W, float(Kmat[0, 0]), E, Kmat)
    coords = torch.stack(ref.meshgrid_xy(torch.arange(H), torch.arange(W)), dim=-1).reshape((-1, 2))   # :230-234
    select_inds = np.random.default_rng(5).choice(coords.shape[0], size=(n), replace=False)               # :235-237
    sel = coords[select_inds]
    ro_s, rd_s = ro[sel[:, 0], sel[:, 1], :], rd[sel[:, 0], sel[:, 1], :]                                 # :239-240
    target_s = image[sel[:, 0], sel[:, 1]]                                                                # :242
    seen = {}
    orig = ref.train_utils.predict_and_render_radiance

    def spy(ray_batch, *a, **k):
        seen["rays"] = ray_batch.detach().clone()
        return orig(ray_batch, *a, **k)

    kw = dict(num_layers=2, hidden_size=32, skip_connect_every=4, num_encoding_fn_xyz=2, num_encoding_fn_dir=2, use_viewdirs=True)
    mc = ref.models.FlexibleNeRFModel(**kw)
    cfg = make_cfg(4, 0, 2.0, 6.0, perturb=False, noise_std=0.0, white=False)
    ref.train_utils.predict_and_render_radiance = spy
    try:
        with torch.no_grad():
            try:
                ref.run_one_iter_of_nerf(H, W, float(Kmat[0, 0]), mc, None, ro_s, rd_s, cfg, mode="train",
                                         encode_position_fn=ref.get_embedding_function(2), encode_direction_fn=ref.get_embedding_function(2),
                                         m_thres_cand=M_THRES)
            except Exception:   # the fork raises NameError after a coarse-only pass (train_utils.py:201); the rows are recorded by then
                pass
    finally:
        ref.train_utils.predict_and_render_radiance = orig
    d.update(sel_E=npf(E), sel_K=npf(Kmat), sel_image=npf(image), sel_inds=select_inds.astype(np.int64), sel_ro=npf(ro_s),
             sel_rd=npf(rd_s), sel_target=npf(target_s), sel_rays=npf(seen["rays"]), sel_near=2.0, sel_far=6.0)
    # camera path of the loaders (nerf/load_blender.py:33-38; importable: only the image readers need cv2 / imageio)
    angles = np.array([[-180.0, -30.0, 4.0], [37.5, -30.0, 4.0], [90.0, 15.0, 2.5], [171.0, -89.0, 0.3]])
    d.update(pose_angles=angles, pose_out=np.stack([ref.load_blender.pose_spherical(*a) for a in angles]))
    path = os.path.join(HERE, "val_extras.npz")
    np.savez_compressed(path, **d)
    print(f"val_extras: {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    torch.manual_seed(0)
    which = sys.argv[1:] or ["kat", "lego", "d8w256", "val_extras"]
    for name in which:
        {"kat": kat, "lego": lego, "d8w256": d8w256, "val_extras": val_extras}[name]()
