"""Pins the CPU oracle (oracle/) to golden vectors captured from the imported reference.

CPU-only (`-m "not gpu"`).  Tolerances: index work bit-exact; floating point `max|a-b| <= tol * max|b|`
per tensor with tol = 1e-4 at most (most checks are far tighter, the oracle runs
the same ATen kernels the reference ran).
"""
import numpy as np
import pytest
import torch

from conftest import rel_err
from golden_cases import CASES, M_THRES, draws_of
from oracle import exact_sampler as es
from oracle import nerf_oracle as oc

TOL = 1e-4


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def test_ray_bundle_kat(golden):
    g = golden("kat")
    for tag, h, w in (("rb0", 3, 4), ("rb1", 20, 30)):
        ro, rd = oc.get_ray_bundle(h, w, g[tag + "_E"], g[tag + "_K"])
        assert ro.shape == (h, w, 3) and rd.shape == (h, w, 3)
        np.testing.assert_array_equal(ro.numpy(), g[tag + "_ro"])
        np.testing.assert_array_equal(rd.numpy(), g[tag + "_rd"])
    # SURVEY.md G1 known answers
    np.testing.assert_allclose(g["rb0_ro"][0, 0], [2, -.1, .2], atol=1e-6)
    np.testing.assert_allclose(g["rb0_rd"][0, 0], [-1, -.02, .015], atol=1e-6)
    np.testing.assert_allclose(g["rb0_rd"][2, 3], [-1, .01, -.005], atol=1e-6)


def test_positional_encoding(golden):
    g = golden("kat")
    x = g["pe_x"]
    cases = (("pe_l10", dict(num_fns=10)), ("pe_l4", dict(num_fns=4)),
             ("pe_l6_lin", dict(num_fns=6, log_sampling=False)),
             ("pe_l4_noinput", dict(num_fns=4, include_input=False)), ("pe_l0", dict(num_fns=0)))
    for name, kw in cases:
        out = oc.positional_encoding(T(x), **kw).numpy()
        np.testing.assert_array_equal(out, g[name])
    out = oc.positional_encoding(T(g["pe_kat_in"]), num_fns=2).numpy()
    np.testing.assert_array_equal(out, g["pe_kat_out"])
    kat = [.1, .2, .3, .0998334, .1986693, .2955202, .9950042, .9800666, .9553365,
           .1986693, .3894184, .5646425, .9800666, .9210610, .8253356]
    np.testing.assert_allclose(out[0], kat, atol=1e-6)


def test_cumprod_exclusive(golden):
    g = golden("kat")
    np.testing.assert_array_equal(oc.cumprod_exclusive(T(g["cpe_in"])).numpy(), g["cpe_out"])


def test_volume_render_kat(golden):
    g = golden("kat")
    v = oc.volume_render(T(g["vr0_rf"]), T(g["vr0_z"]), T(g["vr0_rd"]), m_thres=(5.0, 10.0))
    for n in ("rgb", "disp", "acc", "weights", "depth"):
        np.testing.assert_array_equal(v[n].numpy(), g["vr0_" + n])
    np.testing.assert_array_equal(v["dex"][0].numpy(), g["vr0_dex5"])
    np.testing.assert_array_equal(v["dex"][1].numpy(), g["vr0_dex10"])
    # SURVEY.md S6 known answers
    np.testing.assert_allclose(g["vr0_depth"][0], 2.5024788, atol=1e-6)
    assert g["vr0_dex5"][0] == 3.5 and g["vr0_dex10"][0] == 3.5
    assert np.isnan(g["vr0_disp"][1]) and g["vr0_dex5"][1] == 2.0 and g["vr0_acc"][1] == 0


@pytest.mark.parametrize("tag,std,white", [("a", 0.0, False), ("b", 0.0, True), ("c", 0.2, True)])
def test_volume_render_random(golden, tag, std, white):
    g = golden("kat")
    v = oc.volume_render(T(g["vr1_rf"]), T(g["vr1_z"]), T(g["vr1_rd"]), noise=T(g["vr1_noise"]),
                         noise_std=std, white_background=white, m_thres=M_THRES)
    for n in ("rgb", "disp", "acc", "weights", "depth"):
        np.testing.assert_array_equal(v[n].numpy(), g[f"vr1{tag}_{n}"])
    np.testing.assert_array_equal(np.stack([d.numpy() for d in v["dex"]]), g[f"vr1{tag}_dex"])


@pytest.mark.parametrize("tag", ["sp1", "sp2", "sp3"])
def test_sample_pdf(golden, tag):
    g = golden("kat")
    nf = g[tag + "_det"].shape[1]
    s, aux = oc.sample_pdf(T(g[tag + "_bins"]), T(g[tag + "_w"]), nf, det=True, return_aux=True)
    np.testing.assert_array_equal(aux["inds"].numpy(), g[tag + "_det_inds"])
    np.testing.assert_array_equal(s.numpy(), g[tag + "_det"])
    np.testing.assert_array_equal(s.numpy(), g[tag + "_legacy_det"])  # legacy sample_pdf agrees
    s, aux = oc.sample_pdf(T(g[tag + "_bins"]), T(g[tag + "_w"]), nf, det=False, u=T(g[tag + "_u"]), return_aux=True)
    np.testing.assert_array_equal(aux["inds"].numpy(), g[tag + "_rnd_inds"])
    np.testing.assert_array_equal(s.numpy(), g[tag + "_rnd"])
    # explicit-arithmetic restatement: bit-exact cdf, indices and samples, no torch kernels involved
    r = es.sample_pdf_exact(g[tag + "_bins"], g[tag + "_w"], nf)
    np.testing.assert_array_equal(r["cdf"], g[tag + "_cdf"])
    np.testing.assert_array_equal(r["inds"], g[tag + "_det_inds"])
    np.testing.assert_array_equal(r["samples"], g[tag + "_det"])
    r = es.sample_pdf_exact(g[tag + "_bins"], g[tag + "_w"], nf, u=g[tag + "_u"])
    np.testing.assert_array_equal(r["inds"], g[tag + "_rnd_inds"])
    np.testing.assert_array_equal(r["samples"], g[tag + "_rnd"])


def test_sample_pdf_kat(golden):
    g = golden("kat")
    r = es.sample_pdf_exact(g["sp0_bins"], g["sp0_w"], 6)
    np.testing.assert_array_equal(r["samples"], g["sp0_out"])
    np.testing.assert_allclose(g["sp0_out"][0], [2.0, 3.2249923, 3.4499969, 3.6750016, 3.9000063, 5.0], atol=1e-6)


def test_exact_sum_matches_aten():
    rng = np.random.default_rng(3)
    for length in (8, 30, 62, 63, 126, 190, 254):
        x = rng.uniform(0, 1, size=(4000, length)).astype(np.float32) ** 4
        np.testing.assert_array_equal(es.aten_sum_lastdim(x), T(x).sum(-1).numpy())
    for n in (2, 6, 64, 128, 192):
        np.testing.assert_array_equal(es.linspace_f32(0.0, 1.0, n), torch.linspace(0.0, 1.0, n).numpy())


def test_ndc_and_misc(golden):
    g = golden("kat")
    o, d = oc.ndc_rays(378, 504, 407.5, 1.0, T(g["ndc_o"]), T(g["ndc_d"]))
    np.testing.assert_array_equal(o.numpy(), g["ndc_out_o"])
    np.testing.assert_array_equal(d.numpy(), g["ndc_out_d"])
    mse = torch.nn.functional.mse_loss(T(g["mse_a"]), T(g["mse_b"])).item()
    assert mse == float(g["mse"])
    assert oc.mse2psnr(mse) == float(g["psnr"]) and oc.mse2psnr(0) == float(g["psnr0"])


@pytest.mark.parametrize("tag,kw", [("mlp_d4w128", dict(num_layers=4, hidden_size=128)),
                                    ("mlp_d8w256", dict(num_layers=8, hidden_size=256)),
                                    ("mlp_d8w256_noview", dict(num_layers=8, hidden_size=256, use_viewdirs=False))])
def test_mlp(golden, tag, kw):
    from nerf import synthetic as syn
    g = golden("kat")
    full = dict(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10,
                num_encoding_fn_dir=4, use_viewdirs=True)
    full.update(kw)
    sd = oc.to_torch_sd(syn.synth_state_dict(11, **full))
    out = oc.flexible_mlp(sd, T(g["mlp_in"]), oc.ModelCfg(**full)).numpy()
    assert rel_err(out, g[tag]) < 1e-6


@pytest.mark.parametrize("name", list(CASES))
def test_render_and_train_goldens(golden, name):
    g = golden(name)
    mkw, wfn, rkw = CASES[name]
    sd_c_np, sd_f_np = wfn()
    train = "loss" in g
    sd_c, sd_f = oc.to_torch_sd(sd_c_np, train), oc.to_torch_sd(sd_f_np, train)
    mc = oc.ModelCfg(**mkw)
    cfg = oc.RenderCfg(chunksize=4096, m_thres=M_THRES, **rkw)
    draws = draws_of(g)
    if draws is not None:
        draws = {k: T(v) for k, v in draws.items()}
    rays = oc.pack_rays(T(g["ro"]), T(g["rd"]), cfg)
    with torch.set_grad_enabled(train):
        out, aux = oc.predict_and_render(rays, sd_c, sd_f, mc, mc, cfg, draws, return_aux=True)
    # stage boundaries
    np.testing.assert_array_equal(aux["z_coarse"].numpy(), g["z_coarse"])
    np.testing.assert_array_equal(aux["pts_coarse"].numpy(), g["pts_coarse"])
    assert rel_err(aux["rf_coarse"].detach().numpy(), g["rf_coarse"]) < 1e-5
    for n in ("rgb", "acc", "weights", "depth"):
        assert rel_err(aux["vc"][n].detach().numpy(), g["vc_" + n]) < TOL
    # sampler boundary on the golden inputs: indices bit-exact (explicit arithmetic)
    r = es.sample_pdf_exact(g["sp_bins"], g["sp_weights"], g["sp_u"].shape[1], u=g["sp_u"])
    np.testing.assert_array_equal(r["cdf"], g["sp_cdf"])
    np.testing.assert_array_equal(r["inds"], g["sp_inds"])
    np.testing.assert_array_equal(r["samples"], g["sp_z_samples"])
    assert rel_err(aux["z_fine"].numpy(), g["z_fine"]) < TOL
    assert rel_err(aux["rf_fine"].detach().numpy(), g["rf_fine"]) < 5e-3  # raw sigma is chaotic in z
    names = ["rgb_coarse", "depth_coarse", "acc_coarse", "rgb_fine", "depth_fine", "acc_fine"]
    for n, o in zip(names, out[:6]):
        assert rel_err(o.detach().numpy().reshape(g["out_" + n].shape), g["out_" + n]) < TOL, n
    dex = np.stack([o.numpy() for o in out[6:]])
    assert dex.shape == g["out_dex_fine"].shape
    # Dex depth is an argmax over a thresholded signal: exact given sigma; allow rare flips from ulps
    assert (dex == g["out_dex_fine"]).mean() > 0.999
    if train:
        loss = oc.nerf_loss(out, T(g["target"]))
        assert abs(loss.item() - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
        loss.backward()
        for pref, sd in (("gc_", sd_c), ("gf_", sd_f)):
            for k, p in sd.items():
                gr = p.grad.numpy()
                if pref + k in g:
                    assert rel_err(gr, g[pref + k]) < 1e-3, k
                else:
                    assert rel_err(gr.reshape(-1)[::97], g[pref + k + ".sub"]) < 1e-3, k
                    nrm = float(g[pref + k + ".norm"])
                    assert abs(np.linalg.norm(gr.astype(np.float64)) - nrm) < 1e-3 * nrm


def test_validation_extras_and_ray_selection(golden):
    """the oracle's restatements of compute_err_metric, depth_error_img and the training-ray
    selection equal the reference's recorded outputs exactly."""
    g = golden("val_extras")
    for k in range(g["err_pred"].shape[0]):
        e = oc.compute_err_metric(g["err_gt"], g["err_pred"][k], g["err_mask"])
        assert [e["depth_abs_err"], e["depth_err2"], e["depth_err4"], e["depth_err8"]] == list(g["err_out"][k])
    for k in (1, 4):
        img = oc.depth_error_img(g["err_pred"][k] * np.float32(1000), g["err_gt"] * np.float32(1000), g["err_mask"])
        np.testing.assert_array_equal(img, g[f"err_img_{k}"])
    rays, target = oc.select_training_rays(30, 40, g["sel_E"], g["sel_K"], g["sel_inds"], g["sel_image"], 2.0, 6.0)
    np.testing.assert_array_equal(rays.numpy(), g["sel_rays"])
    np.testing.assert_array_equal(target.numpy(), g["sel_target"])
    np.testing.assert_array_equal(rays.numpy()[:, :3], g["sel_ro"])
    np.testing.assert_array_equal(rays.numpy()[:, 3:6], g["sel_rd"])
