"""GPU parity tests: the HIP path (through the C ABI, via nerf._ops / the drop-in call surface) against
golden vectors captured from the reference and against the CPU oracle.

Tolerances: index / integer work bit-exact; floating point
`max|a-b| <= 1e-4 * max|b|` per output tensor in fp32 mode; bf16 mode is judged on PSNR.
"""
import os

import numpy as np
import pytest
import torch

from conftest import REPO, rel_err
from golden_cases import CASES, M_THRES, draws_of

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    import nerf
    from nerf import _hip
    _hip.lib()  # must load: no fallback
    nerf.set_precision("fp32")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _renders_in_the_named_precision():
    """Tests that name 'bf16' test the bf16 kernels: no-grad renders stay in bf16 here.  The default policy (fp16 renders under
    the bf16 modes, with the overflow guard) has its own tests, which set it explicitly."""
    import nerf
    nerf.set_render_policy("bf16")
    yield
    nerf.set_render_policy(None)


def G(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def C(t):
    return t.detach().cpu().numpy()


def make_models(mkw, sd_c, sd_f, dev):
    import nerf
    out = []
    for sd in (sd_c, sd_f):
        m = nerf.models.FlexibleNeRFModel(**mkw)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        out.append(m.to(dev))
    return out


def make_cfg(rkw, chunksize=4096):
    import nerf
    mode = dict(chunksize=chunksize, lindisp=rkw.get("lindisp", False), num_coarse=rkw["num_coarse"],
                num_fine=rkw["num_fine"], perturb=rkw.get("perturb", False),
                radiance_field_noise_std=rkw.get("noise_std", 0.0), white_background=rkw.get("white_background", False))
    return nerf.CfgNode(dict(dataset=dict(near=rkw["near"], far=rkw["far"], no_ndc=True),
                             nerf=dict(use_viewdirs=True, train=dict(mode), validation=dict(mode))))


# ------------------------------------------------------------------------------------------------
def test_ray_bundle(golden, dev):
    import nerf
    from nerf import _ops
    from oracle import nerf_oracle as oc
    g = golden("kat")
    # golden inverse (LAPACK results differ by an ulp between hosts) -> per-pixel arithmetic is bit-exact
    k = g["rb1_K"]
    ro, rd = _ops.ray_bundle(20, 30, g["rb1_Rinv"].reshape(-1).tolist(), g["rb1_Einv"][:3, 3].tolist(), float(k[0, 0]),
                             float(k[0, 2]), float(k[1, 2]), dev)
    np.testing.assert_array_equal(C(ro), g["rb1_ro"])
    np.testing.assert_array_equal(C(rd), g["rb1_rd"])
    # drop-in call (host-side inverse on this box) against the oracle on the same box: bit-exact as well
    for tag, h, w in (("rb0", 3, 4), ("rb1", 20, 30)):
        ro, rd = nerf.get_ray_bundle(h, w, 1.0, G(g[tag + "_E"], dev), G(g[tag + "_K"], dev))
        ro_o, rd_o = oc.get_ray_bundle(h, w, g[tag + "_E"], g[tag + "_K"])
        np.testing.assert_array_equal(C(ro), ro_o.numpy())
        np.testing.assert_array_equal(C(rd), rd_o.numpy())
        assert rel_err(C(rd), g[tag + "_rd"]) < 1e-6


def test_coarse_depths_bit_exact(golden, dev):
    from nerf import _ops
    from oracle import nerf_oracle as oc
    for name in ("render_d8w256_val", "render_d8w256_lindisp", "train_d8w256", "train_lego"):
        g = golden(name)
        rkw = CASES[name][2]
        cfg = oc.RenderCfg(**{k: v for k, v in rkw.items()})
        rays = oc.pack_rays(torch.from_numpy(g["ro"]), torch.from_numpy(g["rd"]), cfg)
        t_rand = G(g["draw_rand0"], dev) if "draw_rand0" in g else None
        z = _ops.coarse_depths(rays.to(dev), rkw["num_coarse"], rkw.get("lindisp", False), t_rand)
        np.testing.assert_array_equal(C(z), g["z_coarse"])


def test_positional_encoding(golden, dev):
    import nerf
    g = golden("kat")
    x = G(g["pe_x"], dev)
    cases = (("pe_l10", dict(num_encoding_functions=10)), ("pe_l4", dict(num_encoding_functions=4)),
             ("pe_l6_lin", dict(num_encoding_functions=6, log_sampling=False)),
             ("pe_l4_noinput", dict(num_encoding_functions=4, include_input=False)),
             ("pe_l0", dict(num_encoding_functions=0)))
    for name, kw in cases:
        out = C(nerf.positional_encoding(x, **kw))
        assert out.shape == g[name].shape
        assert np.abs(out - g[name]).max() < 2e-6, name  # sin/cos agree to ~1 ulp of 1.0
    e = nerf.get_embedding_function(10, True, True)
    assert np.abs(C(e(x)) - g["pe_l10"]).max() < 2e-6


@pytest.mark.parametrize("tag", ["sp1", "sp2", "sp3"])
def test_sampler_bit_exact(golden, dev, tag):
    from nerf import _ops
    g = golden("kat")
    nf = g[tag + "_det"].shape[1]
    s, inds = _ops.sample_pdf(G(g[tag + "_bins"], dev), G(g[tag + "_w"], dev), nf, None, want_inds=True)
    np.testing.assert_array_equal(C(inds), g[tag + "_det_inds"])
    np.testing.assert_array_equal(C(s), g[tag + "_det"])
    s, inds = _ops.sample_pdf(G(g[tag + "_bins"], dev), G(g[tag + "_w"], dev), nf, G(g[tag + "_u"], dev), want_inds=True)
    np.testing.assert_array_equal(C(inds), g[tag + "_rnd_inds"])
    np.testing.assert_array_equal(C(s), g[tag + "_rnd"])


@pytest.mark.parametrize("name", list(CASES))
def test_sampler_boundary_on_render_goldens(golden, dev, name):
    """Golden (bins, weights, u) -> indices bit-exact at the sampler boundary; merged depths bit-exact."""
    from nerf import _ops
    g = golden(name)
    nf = g["sp_u"].shape[1]
    u = G(g["sp_u"], dev) if "draw_rand1" in g else None
    s, inds = _ops.sample_pdf(G(g["sp_bins"], dev), G(g["sp_weights"], dev), nf, u, want_inds=True)
    np.testing.assert_array_equal(C(inds), g["sp_inds"])
    np.testing.assert_array_equal(C(s), g["sp_z_samples"])
    z_fine, zs = _ops.fine_depths(G(g["z_coarse"], dev), G(g["vc_weights"], dev), nf, u, want_samples=True)
    np.testing.assert_array_equal(C(zs), g["sp_z_samples"])
    np.testing.assert_array_equal(C(z_fine), g["z_fine"])


def test_volume_render_kat(golden, dev):
    import nerf
    g = golden("kat")
    out = nerf.volume_render_radiance_field(G(g["vr0_rf"], dev), G(g["vr0_z"], dev), G(g["vr0_rd"], dev),
                                            m_thres_cand=[5.0, 10.0])
    assert len(out) == 7
    for n, o in zip(["rgb", "disp", "acc", "weights", "depth"], out[:5]):
        assert rel_err(C(o), g["vr0_" + n]) < 1e-6, n
    np.testing.assert_array_equal(C(out[5]), g["vr0_dex5"])
    np.testing.assert_array_equal(C(out[6]), g["vr0_dex10"])
    # m_thres_cand=None -> exactly 5 outputs (superset of the fork, which raises)
    assert len(nerf.volume_render_radiance_field(G(g["vr0_rf"], dev), G(g["vr0_z"], dev), G(g["vr0_rd"], dev))) == 5


@pytest.mark.parametrize("tag,std,white", [("a", 0.0, False), ("b", 0.0, True), ("c", 0.2, True)])
def test_volume_render_random(golden, dev, tag, std, white):
    from nerf import _ops
    g = golden("kat")
    rgb, disp, acc, weights, depth, dex = _ops.volume_render_fwd(
        G(g["vr1_rf"], dev), G(g["vr1_z"], dev), G(g["vr1_rd"], dev), G(g["vr1_noise"], dev), std, white, list(M_THRES))
    for n, o in zip(["rgb", "disp", "acc", "weights", "depth"], (rgb, disp, acc, weights, depth)):
        assert rel_err(C(o), g[f"vr1{tag}_{n}"]) < 1e-5, n  # includes the NaN-disparity rows (acc == 0)
    np.testing.assert_array_equal(C(dex), g[f"vr1{tag}_dex"])  # Dex depth exact given sigma


def test_volume_render_backward_matches_autograd_of_oracle(golden, dev):
    from nerf import _ops
    from oracle import nerf_oracle as oc
    g = golden("kat")
    rng = np.random.default_rng(5)
    for std, white in ((0.0, False), (0.2, True)):
        rf = torch.from_numpy(g["vr1_rf"]).clone().requires_grad_(True)
        z, rd, noise = (torch.from_numpy(g[k]) for k in ("vr1_z", "vr1_rd", "vr1_noise"))
        v = oc.volume_render(rf, z, rd, noise, std, white, ())
        n, s = z.shape
        g_rgb = torch.from_numpy(rng.normal(size=(n, 3)).astype(np.float32))
        g_depth = torch.from_numpy(rng.normal(size=(n,)).astype(np.float32))
        g_acc = torch.from_numpy(rng.normal(size=(n,)).astype(np.float32))
        g_w = torch.from_numpy(rng.normal(size=(n, s)).astype(np.float32))
        loss = (v["rgb"] * g_rgb).sum() + (v["depth"] * g_depth).sum() + (v["acc"] * g_acc).sum() + (v["weights"] * g_w).sum()
        loss.backward()
        rfd = G(g["vr1_rf"], dev).requires_grad_(True)
        out = _ops.VolumeRenderFn.apply(rfd, G(g["vr1_z"], dev), G(g["vr1_rd"], dev), G(g["vr1_noise"], dev), std, white, [])
        lossd = ((out[0] * g_rgb.to(dev)).sum() + (out[4] * g_depth.to(dev)).sum() + (out[2] * g_acc.to(dev)).sum()
                 + (out[3] * g_w.to(dev)).sum())
        lossd.backward()
        assert rel_err(C(rfd.grad), rf.grad.numpy()) < 1e-4


@pytest.mark.parametrize("tag,kw", [("mlp_d4w128", dict(num_layers=4, hidden_size=128)),
                                    ("mlp_d8w256", dict(num_layers=8, hidden_size=256)),
                                    ("mlp_d8w256_noview", dict(num_layers=8, hidden_size=256, use_viewdirs=False))])
def test_mlp_forward_encoded(golden, dev, tag, kw):
    import nerf
    from nerf import synthetic as syn
    g = golden("kat")
    full = dict(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4,
                use_viewdirs=True)
    full.update(kw)
    m = nerf.models.FlexibleNeRFModel(**full)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.synth_state_dict(11, **full).items()})
    m = m.to(dev)
    x = g["mlp_in"]
    x = x[:, : m.dim_xyz + m.dim_dir]
    with torch.no_grad():
        out = m(G(x, dev))
    assert rel_err(C(out), g[tag]) < 1e-5


@pytest.mark.parametrize("name", list(CASES))
def test_run_network_on_golden_points(golden, dev, name):
    import nerf
    g = golden(name)
    mkw, wfn, rkw = CASES[name]
    mc, mf = make_models(mkw, *wfn(), dev)
    ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
    rd = torch.from_numpy(g["rd"])
    vd = (rd / rd.norm(p=2, dim=-1).unsqueeze(-1)).to(dev)
    rays = torch.cat([G(g["ro"], dev), G(g["rd"], dev), torch.zeros(len(rd), 2, device=dev), vd], -1)
    with torch.no_grad():
        rf_c = nerf.run_network(mc, G(g["pts_coarse"], dev), rays, 4096, ex, ed)
        rf_f = nerf.run_network(mf, G(g["pts_fine"], dev), rays, 4096, ex, ed)
    assert rel_err(C(rf_c), g["rf_coarse"]) < TOL
    assert rel_err(C(rf_f), g["rf_fine"]) < TOL


@pytest.mark.parametrize("name", list(CASES))
def test_render_goldens_end_to_end(golden, dev, name, monkeypatch):
    """predict_and_render_radiance (one dn_render_rays call) on the golden rays/weights with the reference's
    recorded RNG draws injected through torch.rand / torch.randn."""
    import nerf
    g = golden(name)
    mkw, wfn, rkw = CASES[name]
    mc, mf = make_models(mkw, *wfn(), dev)
    cfg = make_cfg(rkw)
    draws = draws_of(g)
    if draws is not None:
        q_rand = [G(draws["t_rand"], dev), G(draws["u"], dev)]
        q_randn = [G(draws["noise_c"], dev), G(draws["noise_f"], dev)]
        monkeypatch.setattr(torch, "rand", lambda *a, **k: q_rand.pop(0))
        monkeypatch.setattr(torch, "randn", lambda *a, **k: q_randn.pop(0))
    ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
    with torch.no_grad():
        out = nerf.run_one_iter_of_nerf(1, len(g["ro"]), 1.0, mc, mf, G(g["ro"], dev)[None], G(g["rd"], dev)[None], cfg,
                                        mode="validation", encode_position_fn=ex, encode_direction_fn=ed,
                                        m_thres_cand=np.arange(5, 105, 5))
    assert len(out) == 6 + len(M_THRES)
    names = ["rgb_coarse", "depth_coarse", "acc_coarse", "rgb_fine", "depth_fine", "acc_fine"]
    for n, o in zip(names, out[:6]):
        ref = g["out_" + n]
        assert o.shape[:2] == (1, len(g["ro"]))
        assert rel_err(C(o).reshape(ref.shape), ref) < TOL, n
    dex = np.stack([C(o).reshape(-1) for o in out[6:]])
    # Dex depth = z_fine[first sigma > m]: same value unless an ulp in sigma flips the argmax (SURVEY.md hard part 4)
    assert (np.abs(dex - g["out_dex_fine"]) <= TOL * np.abs(g["out_dex_fine"]).max()).mean() > 0.995
    if draws is None:
        with torch.no_grad():
            out6 = nerf.run_one_iter_of_nerf(1, 8, 1.0, mc, mf, G(g["ro"][:8], dev)[None], G(g["rd"][:8], dev)[None], cfg,
                                             mode="validation", encode_position_fn=ex, encode_direction_fn=ed)
        assert len(out6) == 6  # m_thres_cand=None: exactly six outputs (eval_nerf.py:175-187)


@pytest.mark.parametrize("precision,psnr_min", [("bf16", 35.0), ("fp16", 48.0)])
def test_16bit_modes_psnr(golden, dev, precision, psnr_min):
    """Throughput modes: bf16 / fp16 MFMA with fp32 accumulation, judged on PSNR against the fp32 reference outputs."""
    import nerf
    name = "render_lego_val"
    g = golden(name)
    mkw, wfn, rkw = CASES[name]
    mc, mf = make_models(mkw, *wfn(), dev)
    ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
    nerf.set_precision(precision)
    try:
        with torch.no_grad():
            out = nerf.run_one_iter_of_nerf(1, len(g["ro"]), 1.0, mc, mf, G(g["ro"], dev)[None], G(g["rd"], dev)[None],
                                            make_cfg(rkw), mode="validation", encode_position_fn=ex,
                                            encode_direction_fn=ed, m_thres_cand=list(M_THRES))
    finally:
        nerf.set_precision("fp32")
    mse = float(np.mean((C(out[3]).reshape(-1, 3) - g["out_rgb_fine"]) ** 2))
    psnr = -10.0 * np.log10(max(mse, 1e-12))
    print(f"{precision} vs reference rgb_fine PSNR {psnr:.1f} dB")
    assert psnr > psnr_min


def test_large_render_properties_and_oracle(dev):
    """C2-shaped workload (D8/W256, 64+128) at a size the oracle still finishes in seconds, plus
    size-independent properties: determinism, chunk invariance, sorted merged depths, acc == sum(weights)."""
    import nerf
    from nerf import _ops, synthetic as syn
    from oracle import nerf_oracle as oc
    kw = CASES["render_d8w256_val"][0]
    sd_c, sd_f = CASES["render_d8w256_val"][1]()
    mc, mf = make_models(kw, sd_c, sd_f, dev)
    h = w = 400
    e_mat, k_mat = torch.from_numpy(syn.scene_pose(11)), torch.from_numpy(syn.intrinsic(h, w))
    ro, rd = nerf.get_ray_bundle(h, w, float(k_mat[0, 0]), e_mat.to(dev), k_mat.to(dev))
    sel = torch.from_numpy(syn.select_rays(h, w, 1536, seed=9)).to(dev)
    ro, rd = ro.reshape(-1, 3)[sel].contiguous(), rd.reshape(-1, 3)[sel].contiguous()
    rkw = dict(num_coarse=64, num_fine=128, near=2.0, far=6.0)
    ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
    with torch.no_grad():
        a = nerf.run_one_iter_of_nerf(h, w, 1.0, mc, mf, ro, rd, make_cfg(rkw, 4096), mode="train", encode_position_fn=ex,
                                      encode_direction_fn=ed, m_thres_cand=list(M_THRES))
        b = nerf.run_one_iter_of_nerf(h, w, 1.0, mc, mf, ro, rd, make_cfg(rkw, 500), mode="train", encode_position_fn=ex,
                                      encode_direction_fn=ed, m_thres_cand=list(M_THRES))
    for x, y in zip(a, b):
        assert torch.equal(x, y)  # chunking (500-ray chunks, ragged tail) does not change a single bit
    cfg_o = oc.RenderCfg(chunksize=4096, m_thres=M_THRES, **rkw)
    mcfg = oc.ModelCfg(**kw)
    with torch.no_grad():
        ref = oc.run_one_iter(ro.cpu(), rd.cpu(), oc.to_torch_sd(sd_c), oc.to_torch_sd(sd_f), mcfg, mcfg, cfg_o)
    for i in range(6):
        assert rel_err(C(a[i]), ref[i].numpy()) < TOL, i
    dex = np.stack([C(o) for o in a[6:]])
    dex_ref = np.stack([o.numpy() for o in ref[6:]])
    assert (np.abs(dex - dex_ref) <= TOL * np.abs(dex_ref).max()).mean() > 0.995
    # stage properties
    rays = oc.pack_rays(ro.cpu(), rd.cpu(), cfg_o).to(dev)
    z_c = _ops.coarse_depths(rays, 64, False, None)
    rf = _ops.run_network_rays(mc.packed(), rays, z_c)
    rgb, disp, acc, wts, depth, _ = _ops.volume_render_fwd(rf, z_c, rays[:, 3:6], None, 0.0, False, [])
    assert rel_err(C(wts.sum(-1)), C(acc)) < 1e-5
    assert float(acc.min()) >= 0.0 and float(acc.max()) <= 1.0 + 1e-5
    z_f = _ops.fine_depths(z_c, wts, 128, None)
    assert bool((z_f[:, 1:] >= z_f[:, :-1]).all())
    merged = torch.sort(torch.cat([z_c, _ops.sample_pdf(0.5 * (z_c[:, 1:] + z_c[:, :-1]), wts[:, 1:-1], 128)], -1), -1)[0]
    assert torch.equal(merged, z_f)


def test_host_tensors_and_bad_configs_fail_loudly(dev):
    import nerf
    with pytest.raises(RuntimeError):
        nerf.volume_render_radiance_field(torch.zeros(2, 4, 4), torch.zeros(2, 4), torch.zeros(2, 3))
    with pytest.raises(RuntimeError):
        nerf.sample_pdf(torch.zeros(2, 16), torch.zeros(2, 15), 8, det=True)
    from nerf import _ops
    with pytest.raises(RuntimeError):
        _ops.PackedMLP(dict(num_layers=8, hidden_size=192, skip_connect_every=4, num_encoding_fn_xyz=10,
                            num_encoding_fn_dir=4, include_input_xyz=1, include_input_dir=1, use_viewdirs=1,
                            log_sampling_xyz=1, log_sampling_dir=1), dev)
    with pytest.raises(RuntimeError):
        _ops.sample_pdf(torch.zeros(2, 4, device=dev), torch.zeros(2, 3, device=dev), 8)  # row too short


@pytest.mark.parametrize("name", ["train_lego", "train_d8w256"])
def test_train_step_matches_reference(golden, dev, name, monkeypatch):
    """One training iteration (perturbed sampling + density noise, the reference's recorded draws injected):
    loss, parameter gradients of both nets and - for the lego nets - the parameters after one Adam step."""
    import nerf
    from conftest import load_golden
    g = golden(name)
    mkw, wfn, rkw = CASES[name]
    mc, mf = make_models(mkw, *wfn(), dev)
    cfg = make_cfg(rkw)
    draws = draws_of(g)
    q_rand = [G(draws["t_rand"], dev), G(draws["u"], dev)]
    q_randn = [G(draws["noise_c"], dev), G(draws["noise_f"], dev)]
    monkeypatch.setattr(torch, "rand", lambda *a, **k: q_rand.pop(0))
    monkeypatch.setattr(torch, "randn", lambda *a, **k: q_randn.pop(0))
    ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
    params = list(mc.parameters()) + list(mf.parameters())
    opt = torch.optim.Adam(params, lr=5e-3)
    out = nerf.run_one_iter_of_nerf(1, len(g["ro"]), 1.0, mc, mf, G(g["ro"], dev), G(g["rd"], dev), cfg, mode="train",
                                    encode_position_fn=ex, encode_direction_fn=ed, m_thres_cand=list(M_THRES))
    assert not q_rand and not q_randn  # all four draws consumed, in the reference's order
    assert out[0].requires_grad and out[3].requires_grad and not out[6].requires_grad
    target = G(g["target"], dev)
    loss = nerf.img2mse(out[0][..., :3], target) + nerf.img2mse(out[3][..., :3], target)
    assert abs(loss.item() - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    loss.backward()
    for pref, m in (("gc_", mc), ("gf_", mf)):
        for k, p in m.named_parameters():
            gr = C(p.grad)
            if pref + k in g:
                assert rel_err(gr, g[pref + k]) < 1e-3, k
            else:
                assert rel_err(gr.reshape(-1)[::97], g[pref + k + ".sub"]) < 1e-3, k
                nrm = float(g[pref + k + ".norm"])
                assert abs(np.linalg.norm(gr.astype(np.float64)) - nrm) < 1e-3 * nrm, k
    if name == "train_lego":
        opt.step()
        post = load_golden("train_lego_post_adam")
        # Adam's first step moves every element by ~lr*g/(|g|+1e-8): for the few elements whose gradient is at the
        # 1e-8 level an ulp in g flips the whole step, so compare by the fraction of elements that agree
        n_close = n_all = 0
        for pref, m in (("pc_", mc), ("pf_", mf)):
            for k, p in m.state_dict().items():
                ref = post[pref + k]
                close = np.abs(C(p) - ref) <= 1e-4 * max(np.abs(ref).max(), 1e-30)
                n_close += int(close.sum())
                n_all += close.size
                assert close.mean() > 0.85, (k, close.mean())
                assert np.abs(C(p) - ref).max() < 2.5 * 5e-3, k  # never more than ~2 lr apart
        assert n_close / n_all > 0.99


def test_fused_training_kernels(golden, dev, monkeypatch):
    """Kernel-level checks of the fused training path (training forward with saved activations + backward-data
    chain on the transposed weight stream; the end-to-end train-step tests above already run through it):
      * the route is really taken (FusedNetFn) and the training forward equals the inference kernel bit for bit;
      * fp32 mode: parameter gradients of run_network for a random upstream gradient equal PyTorch autograd over the
        same nn.Linear composition to 1e-4 (exact-fp32 MFMA chains both ways);
      * bf16 mode: the same gradients agree with the fp32 ones to cosine > 0.95 per tensor (measured 0.96-1.00) (same points, so only
        kernel arithmetic differs - end to end the coarse pass would also move the fine samples)."""
    import nerf
    from nerf import _ops, _train
    calls = []
    orig = _train.FusedNetFn.apply
    monkeypatch.setattr(_train.FusedNetFn, "apply", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    for name in ("train_d8w256", "train_lego"):
        g = golden(name)
        mkw, wfn, _ = CASES[name]
        pts = G(g["pts_fine"], dev)
        rd = G(g["rd"], dev)
        vd = torch.nn.functional.normalize(rd, dim=-1)
        rays = torch.cat([torch.zeros(len(rd), 8, device=dev), vd], -1)
        g_up = G(np.random.default_rng(3).normal(size=pts.shape[:2] + (4,)).astype(np.float32), dev)
        ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
        grads = {}
        for prec in ("fp32", "bf16-s16"):
            nerf.set_precision(prec)
            try:
                _, mf = make_models(mkw, *wfn(), dev)
                out = nerf.run_network(mf, pts, rays, 4096, ex, ed)
                assert out.requires_grad
                (out * g_up).sum().backward()
                grads[prec] = {k: C(p.grad).astype(np.float64) for k, p in mf.named_parameters()}
                if prec == "fp32":
                    with torch.no_grad():
                        inference = _ops.run_network_pts(mf.packed(), pts, vd, pts.shape[1])
                    assert torch.equal(out.detach().reshape(-1, 4), inference)
            finally:
                nerf.set_precision("fp32")
        # autograd over the plain module composition (HIP positional encoding + nn.Linear on the device)
        _, mref = make_models(mkw, *wfn(), dev)
        emb = torch.cat([ex(pts.reshape(-1, 3)), ed(vd[:, None, :].expand(pts.shape).reshape(-1, 3))], -1)
        (mref._forward_modules(emb).reshape(out.shape) * g_up).sum().backward()
        for k, p in mref.named_parameters():
            ref = C(p.grad).astype(np.float64).reshape(-1)

            def cosine(a, b):
                return float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-30))
            # a random upstream gradient on raw sigma (x200 density head) makes these sums cancel heavily: two fp32
            # evaluations with different summation orders agree to ~1e-3; the realistic-gradient parity (1e-3 against
            # the reference's recorded gradients) is pinned by test_train_step_matches_reference
            assert cosine(grads["fp32"][k].reshape(-1), ref) > 0.99999, (name, k)
            assert rel_err(grads["fp32"][k].reshape(-1), ref) < 2e-2, (name, k)
            cos_bf = cosine(grads["bf16-s16"][k].reshape(-1), grads["fp32"][k].reshape(-1))
            print(f"{name} {k}: bf16-vs-fp32 gradient cosine {cos_bf:.4f}")
            assert cos_bf > 0.95, (name, k, cos_bf)  # measured 0.96-1.00: bf16 activations and gradients, fp32 accumulate
    assert len(calls) == 4


@pytest.mark.parametrize("n_rays,nc,nf,lindisp,white,perturb", [
    (1, 64, 128, False, False, False),      # a single ray
    (37, 10, 7, False, True, False),        # minimum coarse count (weights row of 8), odd fine count, ragged ray count
    (37, 33, 70, True, False, True),        # nothing a multiple of 32/64; lindisp; perturbed with injected draws
    (19, 128, 256, False, False, False),    # BASELINE config 5 sampling (128 + 256 = 384 samples: 6 chunks of 64)
    (130, 64, 0, False, True, False),       # coarse only (num_fine = 0): superset behaviour
])
def test_ragged_and_extreme_shapes_against_oracle(dev, n_rays, nc, nf, lindisp, white, perturb, monkeypatch):
    """Edge shapes through the whole path (4x128 lego nets), HIP vs the CPU oracle on the same inputs."""
    import nerf
    from golden_cases import D4, lego_weights
    from oracle import nerf_oracle as oc
    rng = np.random.default_rng(n_rays * 1000 + nc)
    sd_c, sd_f = lego_weights()
    mc, mf = make_models(D4, sd_c, sd_f, dev)
    ro = torch.from_numpy(rng.normal(0, 0.3, size=(n_rays, 3)).astype(np.float32) + np.array([0, 0, 4], np.float32))
    rd = torch.from_numpy(rng.normal(0, 0.2, size=(n_rays, 3)).astype(np.float32) + np.array([0, 0, -1], np.float32))
    rkw = dict(num_coarse=nc, num_fine=nf, near=2.0, far=6.0, lindisp=lindisp, white_background=white, perturb=perturb,
               noise_std=0.1 if perturb else 0.0)
    draws = None
    if perturb:
        draws = dict(t_rand=torch.from_numpy(rng.uniform(size=(n_rays, nc)).astype(np.float32)),
                     noise_c=torch.from_numpy(rng.normal(size=(n_rays, nc)).astype(np.float32)),
                     u=torch.from_numpy(rng.uniform(size=(n_rays, nf)).astype(np.float32)),
                     noise_f=torch.from_numpy(rng.normal(size=(n_rays, nc + nf)).astype(np.float32)))
        q_rand = [draws["t_rand"].to(dev), draws["u"].to(dev)]
        q_randn = [draws["noise_c"].to(dev), draws["noise_f"].to(dev)]
        monkeypatch.setattr(torch, "rand", lambda *a, **k: q_rand.pop(0))
        monkeypatch.setattr(torch, "randn", lambda *a, **k: q_randn.pop(0))
    ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
    thres = [5.0, 50.0, 500.0]
    with torch.no_grad():
        out = nerf.run_one_iter_of_nerf(1, n_rays, 1.0, mc, mf if nf > 0 else None, ro.to(dev), rd.to(dev), make_cfg(rkw),
                                        mode="train", encode_position_fn=ex, encode_direction_fn=ed, m_thres_cand=thres)
    cfg_o = oc.RenderCfg(chunksize=4096, m_thres=thres, **{k: v for k, v in rkw.items()})
    mcfg = oc.ModelCfg(**D4)
    with torch.no_grad():
        ref = oc.run_one_iter(ro, rd, oc.to_torch_sd(sd_c), oc.to_torch_sd(sd_f) if nf > 0 else None, mcfg, mcfg, cfg_o, draws)
    assert len(out) == len(ref) == 6 + len(thres)
    for i in range(6):
        if ref[i] is None:
            assert out[i] is None
            continue
        assert out[i].shape == ref[i].shape
        assert rel_err(C(out[i]), ref[i].numpy()) < TOL, i
    dex = np.stack([C(o) for o in out[6:]])
    dex_ref = np.stack([o.numpy() for o in ref[6:]])
    assert (np.abs(dex - dex_ref) <= TOL * np.abs(dex_ref).max()).mean() > 0.98


def test_empty_inputs(dev):
    """Zero rays / zero points are legal and produce empty outputs, not launches."""
    import nerf
    from nerf import _ops
    from golden_cases import D4, lego_weights
    mc, mf = make_models(D4, *lego_weights(), dev)
    rays = torch.zeros(0, 11, device=dev)
    assert _ops.coarse_depths(rays, 64, False).shape == (0, 64)
    z = torch.zeros(0, 64, device=dev)
    assert _ops.run_network_rays(mc.packed(), rays, z).shape == (0, 64, 4)
    out = _ops.volume_render_fwd(torch.zeros(0, 64, 4, device=dev), z, torch.zeros(0, 3, device=dev), None, 0.0, False, [5.0])
    assert out[0].shape == (0, 3) and out[5].shape == (1, 0)
    assert _ops.fine_depths(z, z, 128).shape == (0, 192)
    r = _ops.render_rays(mc.packed(), mf.packed(), rays, 64, 128, False, 0.0, False, [5.0, 10.0])
    assert r[3].shape == (0, 3) and r[6].shape == (2, 0)
    assert nerf.positional_encoding(torch.zeros(0, 3, device=dev), 10).shape == (0, 63)


def test_training_driver_ir_head_and_dex_config_shapes(dev):
    """BASELINE configs 3 / 5 as shapes: the as-shipped 4x128 nets with the Dex-NeRF sampling (64+64, then 128+256) and
    the IR luminance loss head; short runs must reduce the loss (bf16 kernels)."""
    import nerf
    import train_dexnerf
    try:
        for extra in (["--num-coarse", "64", "--num-fine", "64"], ["--num-coarse", "128", "--num-fine", "256", "--ir"]):
            res = train_dexnerf.main(["--iters", "120", "--size", "24", "--views", "4", "--num-random-rays", "256", "--layers", "4",
                                      "--width", "128", "--validate-every", "0", "--quiet", "--precision", "bf16"] + extra)
            first, last = res["history"][0], res["history"][-1]
            assert np.isfinite(last[1]) and last[1] < 0.5 * first[1], (extra, first, last)
    finally:
        nerf.set_precision("fp32")


@pytest.mark.parametrize("precision", ["fp32", "bf16-s16", "bf16"])
def test_training_driver_learns_a_synthetic_scene(dev, precision):
    """End-to-end: the build-owned driver (reference loop: random view + random rays, MSE_c + MSE_f, Adam with the
    exponential LR, Dex threshold sweep) trains a 4x128 student on a teacher scene through the fused HIP training
    kernels; the training PSNR must rise by > 10 dB in 300 iterations and the held-out view must follow."""
    import nerf
    import train_dexnerf
    try:
        res = train_dexnerf.main(["--iters", "300", "--size", "32", "--views", "6", "--num-random-rays", "512", "--layers", "4",
                                  "--width", "128", "--validate-every", "0", "--quiet", "--precision", precision])
    finally:
        nerf.set_precision("fp32")
    first, last = res["history"][0], res["history"][-1]
    assert last[2] - first[2] > 10.0, (first, last)
    assert res["val_psnr"] > 18.0, res["val_psnr"]


@pytest.mark.parametrize("width,l_xyz,viewdirs", [(128, 6, True), (256, 6, True), (128, 10, False), (128, 6, False)])
def test_fused_network_other_instances(dev, width, l_xyz, viewdirs):
    """Kernel instances the goldens do not exercise (L_xyz = 6 defaults of FlexibleNeRFModel, the fc_out head at W=128
    whose stream is padded mid-phase): fused kernel vs the nn.Linear composition fed by the HIP encoding kernel."""
    import nerf
    torch.manual_seed(5)
    m = nerf.models.FlexibleNeRFModel(num_layers=5, hidden_size=width, skip_connect_every=2, num_encoding_fn_xyz=l_xyz,
                                      num_encoding_fn_dir=4, use_viewdirs=viewdirs).to(dev)
    n, s = 21, 40
    pts = torch.randn(n, s, 3, device=dev) * 1.5
    vd = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=-1)
    rays = torch.cat([torch.zeros(n, 8, device=dev), vd], -1)
    ex, ed = nerf.get_embedding_function(l_xyz), (nerf.get_embedding_function(4) if viewdirs else None)
    with torch.no_grad():
        fused = nerf.run_network(m, pts, rays, 4096, ex, ed)
        emb = ex(pts.reshape(-1, 3))
        if viewdirs:
            emb = torch.cat([emb, ed(vd[:, None, :].expand(n, s, 3).reshape(-1, 3))], -1)
        ref = m._forward_modules(emb).reshape(n, s, 4)
        direct = m(emb).reshape(n, s, 4)   # FlexibleNeRFModel.forward(x) on embedded rows -> dn_mlp_forward_encoded
    assert rel_err(C(fused), C(ref)) < TOL
    assert rel_err(C(direct), C(ref)) < TOL


@pytest.mark.parametrize("depth,width,viewdirs", [(5, 128, False), (5, 128, True), (6, 256, False), (2, 128, True)])
def test_fused_training_other_shapes(dev, depth, width, viewdirs):
    """Fused training path on shapes the goldens do not cover: even/odd trunk depth (ping-pong pairs + tail), a skip
    every 2 layers, the fc_out head (no view directions: custom 4-row output-gradient piece), the smallest trunk.
    Parameter gradients vs PyTorch autograd over the same nn.Linear composition, fp32 and bf16."""
    import nerf
    from nerf import _train
    torch.manual_seed(11)
    base = nerf.models.FlexibleNeRFModel(num_layers=depth, hidden_size=width, skip_connect_every=2, num_encoding_fn_xyz=10,
                                         num_encoding_fn_dir=4, use_viewdirs=viewdirs).to(dev)
    assert _train.train_fused_ok(base)
    n, s = 19, 48
    pts = torch.randn(n, s, 3, device=dev)
    vd = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=-1)
    rays = torch.cat([torch.zeros(n, 8, device=dev), vd], -1)
    g_up = torch.randn(n, s, 4, device=dev)
    ex, ed = nerf.get_embedding_function(10), (nerf.get_embedding_function(4) if viewdirs else None)
    emb = ex(pts.reshape(-1, 3))
    if viewdirs:
        emb = torch.cat([emb, ed(vd[:, None, :].expand(n, s, 3).reshape(-1, 3))], -1)
    (base._forward_modules(emb).reshape(n, s, 4) * g_up).sum().backward()
    ref = {k: C(p.grad).astype(np.float64).reshape(-1) for k, p in base.named_parameters()}
    for prec, cos_min, rel_max in (("fp32", 0.99999, 1e-3), ("bf16-s16", 0.95, None)):
        nerf.set_precision(prec)
        try:
            m = nerf.models.FlexibleNeRFModel(num_layers=depth, hidden_size=width, skip_connect_every=2,
                                              num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=viewdirs).to(dev)
            m.load_state_dict(base.state_dict())
            out = nerf.run_network(m, pts, rays, 4096, ex, ed)
            (out * g_up).sum().backward()
            for k, p in m.named_parameters():
                a = C(p.grad).astype(np.float64).reshape(-1)
                cos = float(a @ ref[k] / max(np.linalg.norm(a) * np.linalg.norm(ref[k]), 1e-30))
                assert cos > cos_min, (prec, k, cos)
                if rel_max is not None:
                    assert rel_err(a, ref[k]) < rel_max, (prec, k)
        finally:
            nerf.set_precision("fp32")


def test_render_is_hipgraph_capturable(golden, dev):
    """The C ABI never allocates or synchronises, so a whole predict_and_render_radiance chunk (6 kernels) can be
    captured into a hipGraph and replayed; the replay must reproduce the eager outputs bit for bit."""
    from nerf import _ops
    from oracle import nerf_oracle as oc
    g = golden("render_lego_val")
    mkw, wfn, rkw = CASES["render_lego_val"]
    mc, mf = make_models(mkw, *wfn(), dev)
    cfg = oc.RenderCfg(**rkw)
    rays = oc.pack_rays(torch.from_numpy(g["ro"]), torch.from_numpy(g["rd"]), cfg).to(dev)
    pc, pf = mc.packed(), mf.packed()
    thres = [5.0, 10.0]
    eager = _ops.render_rays(pc, pf, rays, 64, 64, False, 0.0, True, thres)   # also warms up (function attributes)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        _ops.render_rays(pc, pf, rays, 64, 64, False, 0.0, True, thres)      # workspace for this stream
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            captured = _ops.render_rays(pc, pf, rays, 64, 64, False, 0.0, True, thres)
    torch.cuda.current_stream().wait_stream(side)
    for t in captured:
        t.zero_()
    graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(eager, captured):
        assert torch.equal(a, b)


def test_eval_driver_loads_reference_format_checkpoints(golden, dev, tmp_path):
    """eval_nerf.py counterpart: a checkpoint in the reference's dict format (here: the real lego-lowres 4x128 weights
    re-wrapped, and one written by the training driver) loads with the network shape inferred from the tensors and
    renders the same image as a direct call."""
    import eval_nerf
    import nerf
    import train_dexnerf
    from golden_cases import lego_weights
    sd_c, sd_f = lego_weights()
    ck = tmp_path / "lego.ckpt"
    torch.save({"iter": 199999, "model_coarse_state_dict": {k: torch.from_numpy(v) for k, v in sd_c.items()},
                "model_fine_state_dict": {k: torch.from_numpy(v) for k, v in sd_f.items()},
                "optimizer_state_dict": {}, "loss": 0.0, "psnr": 21.4}, ck)
    try:
        res = eval_nerf.main(["--checkpoint", str(ck), "--size", "24", "--views", "2", "--num-fine", "64", "--white-background",
                              "--precision", "fp32", "--savedir", str(tmp_path / "out"), "--quiet", "--m-thres", "20"])
        assert res["frames"][0][0].shape == (24, 24, 3) and (tmp_path / "out" / "0001.png").exists()
        m = eval_nerf.model_from_state_dict({k: torch.from_numpy(v) for k, v in sd_c.items()}, dev)
        assert (m.num_layers, m.hidden_size, m.num_encoding_fn_xyz, m.num_encoding_fn_dir) == (4, 128, 10, 4)
        # a checkpoint written by the training driver (D8/W256-shaped keys incl. the wide skip layer) round-trips
        ck2 = tmp_path / "student.ckpt"
        train_dexnerf.main(["--iters", "3", "--size", "16", "--views", "2", "--num-random-rays", "64", "--layers", "8",
                            "--width", "256", "--validate-every", "0", "--quiet", "--precision", "fp32", "--save", str(ck2)])
        saved = torch.load(ck2, map_location="cpu")
        assert set(saved) == {"iter", "model_coarse_state_dict", "model_fine_state_dict", "optimizer_state_dict", "loss", "psnr"}
        m2 = eval_nerf.model_from_state_dict(saved["model_fine_state_dict"], dev)
        assert (m2.num_layers, m2.hidden_size, m2.skip_layers) == (8, 256, [4])
        res2 = eval_nerf.main(["--checkpoint", str(ck2), "--size", "16", "--views", "1", "--precision", "fp32", "--quiet"])
        assert torch.isfinite(res2["frames"][0][0]).all()
    finally:
        nerf.set_precision("fp32")


def test_weight_grad_all_equals_per_layer_launches(dev):
    """dn_mlp_weight_grad_all (every layer in one launch, workgroups shared out among the layers) must give the same
    gradients as one dn_mlp_weight_grad launch per layer on the same saved buffers: identical products, only the
    fp32 summation order of the partials differs (1e-5 relative); ragged point count so the padded tail is masked."""
    import nerf
    from nerf import _ops, _train, synthetic as syn
    nerf.set_precision("bf16-s16")
    try:
        for kw in (dict(num_layers=8, hidden_size=256, skip_connect_every=4), dict(num_layers=4, hidden_size=128, skip_connect_every=2),
                   dict(num_layers=3, hidden_size=128, skip_connect_every=4, use_viewdirs=False)):
            kw = dict(dict(num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True), **kw)
            m = nerf.models.FlexibleNeRFModel(**kw)
            m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.synth_state_dict(5, **kw).items()})
            m = m.to(dev)
            n_rays, s = 37, 21   # 777 points: not a multiple of 32
            pts = torch.randn(n_rays * s, 3, device=dev)
            vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1) if kw["use_viewdirs"] else None
            pk = m.packed()
            mods = m.linear_modules()
            _ops.pack_backward(pk, [x.weight for x in mods])
            out, act, masks = _ops.run_network_train(pk, pts, vd, s)
            n = out.shape[0]
            grads = _ops.mlp_backward_data(pk, torch.randn(n, 4, device=dev), masks, n)
            res = _ops.mlp_weight_grad_all(pk, act, grads, n, [tuple(x.weight.shape) for x in mods])
            slots, gslots, kh = _train._slots(m, pk.precision)
            w = m.hidden_size
            per_layer = {}

            def one(mod, g_slot, n_out, x_slot, x_width, pe_kind):
                d_w = torch.zeros_like(mod.weight); d_b = torch.zeros_like(mod.bias)
                _ops.mlp_weight_grad(pk, act, grads, n, g_slot, n_out, x_slot, x_width, pe_kind, d_w, d_b)
                per_layer[mod] = (d_w, d_b)

            one(m.layer1, gslots["layer1"], w, 0, 0, 1)
            x_slot = slots["layer1"]
            for i, layer in enumerate(m.layers_xyz):
                one(layer, gslots["trunk0"] + i * kh, w, x_slot, w, 1 if i in m.skip_layers else 0)
                x_slot = slots["trunk0"] + i * kh
            if m.use_viewdirs:
                one(m.fc_feat, gslots["feat"], w, x_slot, w, 0)
                one(m.fc_alpha, gslots["out"] + 1, 1, x_slot, w, 0)
                one(m.layers_dir[0], gslots["dirout"], w // 2, slots["feat"], w, 2)
                one(m.fc_rgb, gslots["out"], 3, slots["dirout"], w // 2, 0)
            else:
                one(m.fc_out, gslots["out"], 4, x_slot, w, 0)
            for mod, (d_w, d_b) in zip(mods, res):
                ref_w, ref_b = per_layer[mod]
                assert d_w.shape == mod.weight.shape and torch.isfinite(d_w).all()
                assert rel_err(C(d_w), C(ref_w)) < 1e-5 and rel_err(C(d_b), C(ref_b)) < 1e-5
                assert float(ref_w.abs().max()) > 0
    finally:
        nerf.set_precision("fp32")


def test_fused_optimizer_step_reaches_the_kernels(dev):
    """Adam(fused=True) changes the parameters without bumping tensor versions; the next render must run on the new
    weights (the packed stream is rebuilt after any optimizer step)."""
    import nerf
    from nerf import _ops, synthetic as syn
    kw = dict(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True)
    m = nerf.models.FlexibleNeRFModel(**kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.synth_state_dict(9, **kw).items()})
    m = m.to(dev)
    pts = torch.randn(64 * 8, 3, device=dev)
    vd = torch.nn.functional.normalize(torch.randn(64, 3, device=dev), dim=-1)
    ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
    emb = torch.cat([ex(pts), ed(vd[:, None, :].expand(64, 8, 3).reshape(-1, 3))], -1)
    with torch.no_grad():
        before = _ops.run_network_pts(m.packed(), pts, vd, 8)
    opt = torch.optim.Adam(m.parameters(), lr=1e-2, fused=True)
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    with torch.no_grad():
        after = _ops.run_network_pts(m.packed(), pts, vd, 8)
        ref = m._forward_modules(emb)
    assert not torch.allclose(before, after)
    assert rel_err(C(after), C(ref)) < TOL


def test_validation_extras_on_device(golden, dev):
    """N3: the Dex threshold sweep (all candidates, one kernel, one copy) and the error image against the reference's
    recorded compute_err_metric / depth_error_img outputs: counts and colours exact, mean |err| to 1e-6 relative
    (fp64 accumulation here, ATen's fp32 mean there)."""
    import nerf
    g = golden("val_extras")
    gt = G(g["err_gt"], dev)
    pred = G(g["err_pred"], dev)
    best, errs = nerf.dex_error_sweep(gt, [pred[k] for k in range(pred.shape[0])])      # reference ground mask (0, 1.25)
    ref = g["err_out"]
    for k, e in enumerate(errs):
        assert abs(e["depth_abs_err"] - ref[k, 0]) <= 1e-6 * ref[k, 0]
        assert [e["depth_err2"], e["depth_err4"], e["depth_err8"]] == list(ref[k, 1:])
    assert best == int(np.argmin(ref[:, 0])) == 0
    mask = torch.from_numpy(g["err_mask"]).to(dev)
    _, errs_m = nerf.dex_error_sweep(gt, pred, mask=mask)                                 # explicit mask, stacked maps
    assert [e["depth_err4"] for e in errs_m] == list(ref[:, 2])
    single = nerf.compute_err_metric(gt, pred[3], mask)
    assert single["depth_err8"] == ref[3, 3] and abs(single["depth_abs_err"] - ref[3, 0]) <= 1e-6 * ref[3, 0]
    for k in (1, 4):
        img = nerf.depth_error_img(pred[k][None] * 1000, gt[None] * 1000, mask[None])
        assert isinstance(img, np.ndarray)
        np.testing.assert_array_equal(img, g[f"err_img_{k}"])


def test_ray_selection_and_ndc_on_device(golden, dev):
    """N4: one kernel from pixel draws to packed ray rows + target pixels, against the rows the reference's
    run_one_iter_of_nerf handed to predict_and_render_radiance for the same draws; NDC warp against its golden."""
    import nerf
    g = golden("val_extras")
    sel = nerf.RaySelector(30, 40, torch.from_numpy(g["sel_E"]), torch.from_numpy(g["sel_K"]), float(g["sel_near"]),
                           float(g["sel_far"]), device=dev)
    pix = sel.from_reference_choice(g["sel_inds"])
    rays, target = sel.select(pix, G(g["sel_image"], dev))
    np.testing.assert_array_equal(C(target), g["sel_target"][:, :3])
    ref = g["sel_rays"]
    np.testing.assert_array_equal(C(rays)[:, [0, 1, 2, 6, 7]], ref[:, [0, 1, 2, 6, 7]])   # origin (same host inverse), near, far
    assert rel_err(C(rays)[:, 3:6], ref[:, 3:6]) < 1e-6 and rel_err(C(rays)[:, 8:], ref[:, 8:]) < 1e-6
    # the rows feed the render directly
    kw = dict(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True)
    from nerf import synthetic as syn
    m = nerf.models.FlexibleNeRFModel(**kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.synth_state_dict(3, **kw).items()})
    m = m.to(dev)
    cfg = make_cfg(dict(near=2.0, far=6.0, num_coarse=16, num_fine=0, perturb=False, noise_std=0.0, white=False, lindisp=False))
    ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
    with torch.no_grad():
        a = nerf.predict_and_render_radiance(rays, m, None, cfg, mode="validation", encode_position_fn=ex, encode_direction_fn=ed)
        ro, rd = rays[:, :3], rays[:, 3:6]
        b = nerf.run_one_iter_of_nerf(1, rays.shape[0], 1.0, m, None, ro, rd, cfg, mode="train", encode_position_fn=ex,
                                      encode_direction_fn=ed)
    assert rel_err(C(a[0]), C(b[0])) < 1e-5
    # random draws on the device: distinct, in range
    p = sel.random_pixels(200)
    assert p.is_cuda and len(torch.unique(p)) == 200 and int(p.max()) < 1200 and int(p.min()) >= 0
    k = golden("kat")
    o, d = nerf.ndc_rays(378, 504, 407.5, 1.0, G(k["ndc_o"], dev), G(k["ndc_d"], dev))
    np.testing.assert_array_equal(C(o), k["ndc_out_o"])
    np.testing.assert_array_equal(C(d), k["ndc_out_d"])


@pytest.mark.parametrize("precision", ["bf16-s16", "fp32"])
def test_backward_chain_stages_against_matmul(dev, precision):
    """Kernel-level check of the backward-data chain on its own buffers: every trunk stage's stored gradient must equal
    (next stage's stored gradient) @ W masked by the saved ReLU pattern, recomputed here with plain matmuls from the
    unpacked native buffers.  Catches data-path faults that the end-to-end cosine checks only see as a small loss of
    accuracy (e.g. a store racing a register reuse corrupts one dword in a few lanes)."""
    import nerf
    from nerf import _ops, _train, synthetic as syn
    nerf.set_precision(precision)
    try:
        kw = dict(num_layers=8, hidden_size=256, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True)
        m = nerf.models.FlexibleNeRFModel(**kw)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.synth_state_dict(43, sigma_bias=-20.0, **kw).items()})
        m = m.to(dev)
        pk = m.packed()
        _ops.pack_backward(pk, [x.weight for x in m.linear_modules()])
        torch.manual_seed(0)
        n_rays, s = 41, 24          # 984 points: the last workgroup tile is ragged
        pts = torch.randn(n_rays * s, 3, device=dev)
        vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1)
        out, act, masks = _ops.run_network_train(pk, pts, vd, s)
        n = out.shape[0]
        grads = _ops.mlp_backward_data(pk, torch.randn(n, 4, device=dev), masks, n)
        slots, gslots, kh = _train._slots(m, pk.precision)
        w = m.hidden_size

        def rows(which, buf, slot):
            return _ops.mlp_unpack(pk, which, buf, n, slot, w, 0, torch.empty((n, w), dtype=torch.float32, device=dev))

        lowp = (lambda t: t.to(torch.bfloat16).float()) if precision == "bf16-s16" else (lambda t: t)
        tol = 1.5e-2 if precision == "bf16-s16" else 1e-5     # bf16: the stored gradient is rounded to 8 bits
        d_next = rows(1, grads, gslots["trunk0"] + (m.num_layers - 2) * kh)       # d pre-activation of layers_xyz[D-2]
        for i in range(m.num_layers - 2, -1, -1):
            weight = lowp(m.layers_xyz[i].weight.detach()[:, :w])
            d_x = d_next @ weight                                                   # d (input of layers_xyz[i])
            if i > 0:
                x_i = rows(0, act, slots["trunk0"] + (i - 1) * kh)                  # = relu(pre_{i-1}) saved by the forward
                expect = d_x * (x_i > 0)
                got = rows(1, grads, gslots["trunk0"] + (i - 1) * kh)
            else:
                expect = d_x                                                        # layer1 has no activation
                got = rows(1, grads, gslots["layer1"])
            assert rel_err(C(got), C(expect)) < tol, (precision, i, rel_err(C(got), C(expect)))
            d_next = got
    finally:
        nerf.set_precision("fp32")


def test_multiview_selector_and_graph_captured_training(golden, dev):
    """The device-indexed camera selection equals the host-constant one for every view, and the training driver's
    captured HIP graph (whole iteration: selection, render, backward, Adam) learns like the eager loop."""
    import nerf
    import train_dexnerf
    from nerf import synthetic as syn
    g = golden("val_extras")
    h, w = 30, 40
    poses = [torch.from_numpy(syn.scene_pose(i)) for i in range(3)] + [torch.from_numpy(g["sel_E"])]
    kmat = torch.from_numpy(g["sel_K"])
    images = torch.rand(4, h, w, 3)
    multi = nerf.MultiViewRaySelector(h, w, poses, kmat, 2.0, 6.0, images=images, device=dev)
    pix = multi.random_pixels(100)
    for v in range(4):
        single = nerf.RaySelector(h, w, poses[v], kmat, 2.0, 6.0, device=dev)
        r1, t1 = single.select(pix, images[v].to(dev))
        multi.view.fill_(v)
        r2, t2 = multi.select(pix)
        assert torch.equal(r1, r2) and torch.equal(t1, t2)
    try:
        common = ["--iters", "240", "--size", "32", "--views", "6", "--num-random-rays", "512", "--layers", "4", "--width", "128",
                  "--validate-every", "0", "--quiet", "--precision", "bf16"]
        graphed = train_dexnerf.main(common)
        eager = train_dexnerf.main(common + ["--no-hip-graph"])
    finally:
        nerf.set_precision("fp32")
    for res in (graphed, eager):
        assert res["history"][-1][2] - res["history"][0][2] > 8.0, res["history"]
    assert abs(graphed["val_psnr"] - eager["val_psnr"]) < 3.0, (graphed["val_psnr"], eager["val_psnr"])


def test_training_from_a_messytable_directory(dev, tmp_path):
    """On-disk format -> loader -> device selection -> fused training -> Dex validation, end to end: a synthetic scene is
    written in the reference's MessyTable layout (540x960 PNGs, mm depth PNGs, meta.pkl with the 1080x1920 intrinsic),
    read back by nerf.load_messytable_data at the fork's 270x480 / K/4 convention, and trained on by the driver."""
    import importlib.util
    import nerf
    import train_dexnerf
    spec = importlib.util.spec_from_file_location("make_synthetic_messytable", os.path.join(REPO, "scripts", "make_synthetic_messytable.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    root = str(tmp_path / "scene")
    mod.write_scene(root, n_train=5, n_val=1, n_test=1)
    imgs, poses, _, hwf, i_split, intrinsics, depths = nerf.load_messytable_data(root, half_res=True)
    assert imgs.shape == (7, 270, 480, 3) and hwf[:2] == [270, 480] and [len(s) for s in i_split] == [5, 1, 1]
    assert float(intrinsics[0, 0, 0]) == 400.0 and float(intrinsics[0, 0, 2]) == 240.0 and float(intrinsics[0, 1, 2]) == 135.0
    assert 0.25 < float(depths[depths > 0].median()) < 1.4
    try:
        res = train_dexnerf.main(["--messytable", root, "--iters", "400", "--num-random-rays", "1024", "--layers", "4", "--width", "128",
                                  "--num-fine", "64", "--near", "0.3", "--far", "1.3", "--validate-every", "0", "--quiet",
                                  "--precision", "bf16", "--m-thres", "100"])
    finally:
        nerf.set_precision("fp32")
    first, last = res["history"][0], res["history"][-1]
    assert last[2] - first[2] > 8.0, (first, last)
    assert res["val_psnr"] > 15.0 and np.isfinite(res["dex_abs_err_mm"]), res


def test_data_parallel_training_loop_two_ranks_one_gpu(dev, tmp_path):
    """The driver's data-parallel path end to end on real kernels: two ranks (gloo backend - RCCL refuses two ranks on one
    device) share this GPU, start from broadcast weights, train on disjoint views, exchange each network's gradients once per
    step, and must end with bit-identical replicas that have learned.  Run twice: the iteration replayed as HIP graphs around the
    exchange (nerf.GraphedTrainStep: draw .. fine backward | coarse backward | Adam, the all-reduces launched eagerly in between)
    and every kernel launched from Python."""
    import subprocess
    import sys
    ms = {}
    for mode, extra in (("graphs", ""), ("eager", "--no-hip-graph")):
        env = dict(os.environ, DEXNERF_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", CKDIR=str(tmp_path), DP_EXTRA=extra, DP_ITERS="220")
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                              "--master-port", "29533", os.path.join(REPO, "scripts", "dp_rehearsal.py")],
                             env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [ln.split() for ln in out.stdout.splitlines() if ln.startswith("RESULT")]
        assert len(lines) == 2, out.stdout[-2000:]
        for _, rank, first, last, _ms, n_graphs in lines:
            assert float(last) - float(first) > 8.0, lines     # both ranks log their (local) training PSNR
            assert int(n_graphs) == (3 if mode == "graphs" else 0), lines
        ms[mode] = max(float(ln[4]) for ln in lines)
        a = torch.load(os.path.join(str(tmp_path), "dp_rank0.ckpt"), map_location="cpu")
        b = torch.load(os.path.join(str(tmp_path), "dp_rank1.ckpt"), map_location="cpu")
        for key in ("model_coarse_state_dict", "model_fine_state_dict"):
            for name in a[key]:
                assert torch.equal(a[key][name], b[key][name]), (mode, key, name)    # replicas stayed bit-identical
    # (timing: gloo's wait() blocks the HOST until the GPU has produced the segment and the CPU exchange is done, so every iteration
    # drains the launch pipeline and neither loop can run ahead - the figures are recorded, and only a gross regression is gated;
    # over RCCL the wait is a stream dependency and the host stays ahead: unmeasured here, one GPU per box)
    _record_measurement("dp_two_ranks_one_gpu_ms_per_iter", ms)
    assert ms["graphs"] < 1.5 * ms["eager"], ms


def test_fp32_weight_grad_kernel_against_unpacked_gemms(dev):
    """Exact-fp32 weight-gradient kernel (v_mfma_f32_32x32x2_f32 on the native fp32 buffers, one launch per network)
    against dY^T X formed with float64 matmuls on the unpacked rows, every layer incl. the skip layer's positional-encoding
    columns, the view-direction layer and the 3-/1-row heads; ragged point count."""
    import nerf
    from nerf import _ops, _train, synthetic as syn
    nerf.set_precision("fp32")
    torch.manual_seed(17)
    for kw in (dict(num_layers=8, hidden_size=256, skip_connect_every=4), dict(num_layers=5, hidden_size=128, skip_connect_every=2),
               dict(num_layers=3, hidden_size=128, skip_connect_every=4, use_viewdirs=False)):
        kw = dict(dict(num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True), **kw)
        m = nerf.models.FlexibleNeRFModel(**kw)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.synth_state_dict(11, sigma_gain=5.0, sigma_bias=0.0, **kw).items()})
        m = m.to(dev)
        n_rays, s = 45, 23     # 1035 points
        pts = torch.randn(n_rays * s, 3, device=dev)
        vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1) if kw["use_viewdirs"] else None
        pk = m.packed()
        mods = m.linear_modules()
        _ops.pack_backward(pk, [x.weight for x in mods])
        out, act, masks = _ops.run_network_train(pk, pts, vd, s)
        n = out.shape[0]
        g = torch.randn(n, 4, device=dev)
        grads = _ops.mlp_backward_data(pk, g, masks, n)
        res = dict(zip(mods, _ops.mlp_weight_grad_all(pk, act, grads, n, [tuple(x.weight.shape) for x in mods])))
        slots, gslots, kh = _train._slots(m, pk.precision)
        w = m.hidden_size

        def rows(which, buf, slot, width, kind=0):
            return _ops.mlp_unpack(pk, which, buf, n, slot, width, kind, torch.empty((n, width), dtype=torch.float32, device=dev)).double()

        pe_xyz = rows(0, act, slots["xyz"], m.dim_xyz, 1)
        checks = [(m.layer1, rows(1, grads, gslots["layer1"], w), pe_xyz)]
        x_prev = rows(0, act, slots["layer1"], w)
        for i, layer in enumerate(m.layers_xyz):
            x = torch.cat((x_prev, pe_xyz), -1) if i in m.skip_layers else x_prev
            checks.append((layer, rows(1, grads, gslots["trunk0"] + i * kh, w), x))
            x_prev = rows(0, act, slots["trunk0"] + i * kh, w)
        if m.use_viewdirs:
            checks.append((m.fc_feat, rows(1, grads, gslots["feat"], w), x_prev))
            checks.append((m.fc_alpha, g[:, 3:4].double(), x_prev))
            x_dir = torch.cat((rows(0, act, slots["feat"], w), rows(0, act, slots["dir"], m.dim_dir, 2)), -1)
            checks.append((m.layers_dir[0], rows(1, grads, gslots["dirout"], w // 2), x_dir))
            checks.append((m.fc_rgb, g[:, :3].double(), rows(0, act, slots["dirout"], w // 2)))
        else:
            checks.append((m.fc_out, g.double(), x_prev))
        for mod, dy, x in checks:
            d_w, d_b = res[mod]
            assert rel_err(C(d_w), C(dy.t() @ x)) < 2e-5, (kw, mod)
            # (a bias gradient is a sum of ~1000 signed terms, for fc_alpha a single number: the fp32 rounding of the partial sums is
            # relative to the terms, not to a result that may cancel to a hundredth of them)
            db_err = np.abs(C(d_b).astype(np.float64) - C(dy.sum(0))).max()
            assert db_err < 2e-5 * max(np.abs(C(dy.sum(0))).max(), 1e-2 * float(dy.abs().sum(0).max())), (kw, mod, db_err)


@pytest.mark.parametrize("prec16", ["bf16", "fp16"])
def test_48_point_geometry_against_32_point_and_fp32(dev, monkeypatch, prec16):
    """The 48-points-per-wave bf16 / fp16 inference kernel (mlp_fused48.hip, the default) against the 32-point 16-bit
    kernel (DEXNERF_BF16_GEOM=32: same products, different fp32 accumulation grouping and cosine phase form) and against
    the exact-fp32 kernel, on ragged point counts around the 384-point workgroup tile, both input forms, with / without
    view directions, odd / even trunk depth, skip at different layers.  The two bf16 kernels form the same products and
    differ only where a different fp32 summation order flips a bf16 rounding, which later layers can amplify in single
    outputs - so the layout check is the MEAN difference (measured 4e-6 .. 1e-4 of the output range; a wrong column or
    row anywhere gives O(1)): < 3e-4; outliers: max < 8e-2.  Against fp32 both sit at the bf16 level (mean < 2e-2,
    max < 8e-2) and equally so (mean errors within 10 %)."""
    import nerf
    from nerf import _ops, synthetic as syn
    gen = torch.Generator(device="cpu").manual_seed(5)
    for (D, view, skip, width) in [(8, True, 4, 256), (8, False, 4, 256), (5, True, 2, 256), (2, True, 4, 256), (3, False, 100, 256),
                                   (9, True, 3, 256), (4, True, 4, 128), (6, False, 2, 128), (3, True, 2, 128)]:
        kw = dict(num_layers=D, hidden_size=width, skip_connect_every=skip, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=view)
        sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(11 + D, sigma_gain=5.0, sigma_bias=0.0, **kw).items()}
        packed = {}
        for prec in ("fp32", prec16):
            nerf.set_precision(prec)
            m = nerf.models.FlexibleNeRFModel(**kw); m.load_state_dict(sd); m = m.to(dev)
            packed[prec] = (m, m.packed())
        nerf.set_precision("fp32")
        for n_rays, s in [(1, 1), (1, 383), (1, 385), (77, 5), (3, 1000), (129, 192)]:
            pts = torch.randn(n_rays, s, 3, generator=gen).to(dev)
            vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=gen), dim=-1).to(dev)
            rays = torch.cat([torch.randn(n_rays, 3, generator=gen).to(dev), vd * 1.5, torch.zeros(n_rays, 2, device=dev), vd], -1).contiguous()
            z = torch.sort(torch.rand(n_rays, s, generator=gen) * 4 + 2, -1)[0].to(dev).contiguous()
            for form in ("pts", "rays"):
                def run(prec):
                    pk = packed[prec][1]
                    with torch.no_grad():
                        if form == "pts":
                            return _ops.run_network_pts(pk, pts.reshape(-1, 3), vd if view else None, s)
                        return _ops.run_network_rays(pk, rays, z)
                ref = run("fp32")
                monkeypatch.setenv("DEXNERF_BF16_GEOM", "32")
                o32 = run(prec16)
                monkeypatch.delenv("DEXNERF_BF16_GEOM")
                o48 = run(prec16)
                scale = float(ref.abs().max()) + 1e-6
                case = (width, D, view, skip, n_rays, s, form)
                assert torch.isfinite(o48).all(), case
                d = (o48 - o32).abs() / scale
                assert float(d.max()) < 8e-2, (case, float(d.max()))
                e48, e32 = (o48 - ref).abs() / scale, (o32 - ref).abs() / scale
                assert float(e48.max()) < 8e-2 and float(e32.max()) < 8e-2, (case, float(e48.max()), float(e32.max()))
                if ref.numel() >= 1000:   # means only where they are statistics (one 5e-2 outlier in 1500 values is 3e-5)
                    assert float(d.mean()) < 3e-4, (case, float(d.mean()))
                    assert float(e48.mean()) < 2e-2, (case, float(e48.mean()))
                    assert abs(float(e48.mean()) - float(e32.mean())) < 0.1 * float(e32.mean()) + 1e-5, (case, float(e48.mean()), float(e32.mean()))
    # a D the 48-point kernel's LDS budget excludes falls back to the 32-point kernel transparently (D = 12 here)
    kw = dict(num_layers=12, hidden_size=256, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True)
    sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(3, sigma_gain=5.0, sigma_bias=0.0, **kw).items()}
    nerf.set_precision(prec16)
    try:
        m = nerf.models.FlexibleNeRFModel(**kw); m.load_state_dict(sd); m = m.to(dev)
        pts = torch.randn(500, 3, device=dev)
        vd = torch.nn.functional.normalize(torch.randn(500, 3, device=dev), dim=-1)
        with torch.no_grad():
            a = _ops.run_network_pts(m.packed(), pts, vd, 1)
            monkeypatch.setenv("DEXNERF_BF16_GEOM", "32")
            b = _ops.run_network_pts(m.packed(), pts, vd, 1)
        assert torch.equal(a, b)
    finally:
        nerf.set_precision("fp32")


@pytest.mark.parametrize("prec16", ["bf16", "fp16"])
def test_explicit_schedule_instances_equal_the_compiler_scheduled_kernel_bit_for_bit(dev, monkeypatch, prec16):
    """The fixed-shape render instances run the explicit schedule (mlp_stage48.h run_stage48x: MFMAs, conversions and ReLUs as ordered
    asm statements, the previous tile's epilogue in the MFMA gaps, weight DMAs one wave at a time); the run-time-shape kernel
    (DEXNERF_G48_RUNTIME_SHAPE=1) is compiler-scheduled from the round-1 stage loop.  Same pieces, same products, same accumulation
    order: the outputs must be EQUAL - paper net (32-piece barrier period) and as-shipped net (16-piece period, points form: the
    non-overlapped instance), one point .. more tiles than workgroups (every workgroup's first-tile prologue and its steady state,
    the stream wrap, the tail padding), both input forms."""
    import nerf
    from nerf import _ops, synthetic as syn
    gen = torch.Generator(device="cpu").manual_seed(9)
    nerf.set_precision(prec16)
    try:
        for (D, width) in ((8, 256), (4, 128)):
            kw = dict(num_layers=D, hidden_size=width, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True)
            m = nerf.models.FlexibleNeRFModel(**kw)
            m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.synth_state_dict(7 + D, sigma_gain=5.0, sigma_bias=0.0, **kw).items()})
            m = m.to(dev)
            pk = m.packed()
            for n_rays, s in ((1, 1), (1, 385), (77, 5), (700, 192), (2100, 100)):      # 2100 x 100 = 547 tiles of 384 points > 256 workgroups
                pts = torch.randn(n_rays, s, 3, generator=gen).to(dev)
                vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=gen), dim=-1).to(dev)
                rays = torch.cat([torch.randn(n_rays, 3, generator=gen).to(dev), vd * 1.5, torch.zeros(n_rays, 2, device=dev), vd], -1).contiguous()
                z = torch.sort(torch.rand(n_rays, s, generator=gen) * 4 + 2, -1)[0].to(dev).contiguous()
                for form in ("pts", "rays"):
                    def run():
                        with torch.no_grad():
                            if form == "pts":
                                return _ops.run_network_pts(pk, pts.reshape(-1, 3), vd, s)
                            return _ops.run_network_rays(pk, rays, z)
                    monkeypatch.setenv("DEXNERF_G48_NO_OVERLAP", "1")      # (as-shipped net, rays form: the plain fixed instance)
                    fixed = run()
                    monkeypatch.setenv("DEXNERF_G48_RUNTIME_SHAPE", "1")
                    runtime = run()
                    monkeypatch.delenv("DEXNERF_G48_RUNTIME_SHAPE")
                    monkeypatch.delenv("DEXNERF_G48_NO_OVERLAP")
                    assert torch.isfinite(fixed).all() and torch.equal(fixed, runtime), (prec16, width, n_rays, s, form)
    finally:
        nerf.set_precision("fp32")


# ---- the kernel / configurations the bench line times, gated against the reference-recorded golden and the oracle ----
def dex_agreement(dex, dex_ref, tol_scale=TOL):
    """Dex-depth agreement between two (K, N) stacks: fraction of (threshold, ray) entries within tol_scale * max|ref|,
    and the worst absolute miss in metres (a flipped first-crossing moves the readout to another sample's depth)."""
    dex, dex_ref = np.asarray(dex, np.float64), np.asarray(dex_ref, np.float64)
    miss = np.abs(dex - dex_ref)
    return float((miss <= tol_scale * np.abs(dex_ref).max()).mean()), float(miss.max())


def _record_measurement(key, values):
    """Measured figures behind a gate, appended to gpurun_out/test_measurements.jsonl (read back after a GPU run to keep the
    floors just under what the kernels deliver)."""
    import json
    out = os.path.join(REPO, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "test_measurements.jsonl"), "a") as f:
        f.write(json.dumps({"test": key, **{k: float(v) for k, v in values.items()}}) + "\n")


def psnr_db(a, b, peak=1.0):
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return float(10.0 * np.log10(peak * peak / max(mse, 1e-14)))


# floors just under the measured figures (bf16 41.6 dB / 37.0 dB / 0.981, fp16 57.5 / 53.7 / 0.9974): a regression that doubles the
# Dex misses, or costs half a dB, fails
@pytest.mark.parametrize("precision,rgb_floor,depth_floor,dex_floor", [("bf16", 41.0, 36.5, 0.975), ("fp16", 57.0, 53.0, 0.996)])
def test_headline_kernel_16bit_against_reference_golden(golden, dev, precision, rgb_floor, depth_floor, dex_floor):
    """What bench.py times - D8/W256, 64+128, the 48-points-per-wave 16-bit kernel - end to end on the rays of the
    reference-recorded golden `render_d8w256_val` (192 rays; fp32 reference outputs): rgb PSNR, depth PSNR (peak = far - near
    = 4 m) and Dex-depth agreement (reference nerf/volume_rendering_utils.py:51-58: the readout is a thresholded first
    crossing, so one flipped sample moves it by a whole sample spacing or more).  Stage-wise as well: the 16-bit fine
    network on the golden fine points, Dex readout on ITS sigma against the readout on the golden sigma at identical depths -
    isolates the network's arithmetic from the resampling it feeds."""
    import nerf
    from nerf import _ops
    name = "render_d8w256_val"
    g = golden(name)
    mkw, wfn, rkw = CASES[name]
    mc, mf = make_models(mkw, *wfn(), dev)
    ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
    thres = list(M_THRES)
    nerf.set_precision(precision)
    try:
        with torch.no_grad():
            out = nerf.run_one_iter_of_nerf(1, len(g["ro"]), 1.0, mc, mf, G(g["ro"], dev)[None], G(g["rd"], dev)[None],
                                            make_cfg(rkw), mode="validation", encode_position_fn=ex,
                                            encode_direction_fn=ed, m_thres_cand=thres)
            rd = torch.from_numpy(g["rd"])
            vd = (rd / rd.norm(p=2, dim=-1).unsqueeze(-1)).to(dev)
            rays = torch.cat([G(g["ro"], dev), G(g["rd"], dev), torch.zeros(len(rd), 2, device=dev), vd], -1)
            rf16 = nerf.run_network(mf, G(g["pts_fine"], dev), rays, 4096, ex, ed)
    finally:
        nerf.set_precision("fp32")
    rgb_psnr = psnr_db(C(out[3]).reshape(-1, 3), g["out_rgb_fine"])
    depth_psnr = psnr_db(C(out[4]).reshape(-1), g["out_depth_fine"], peak=4.0)
    dex = np.stack([C(o).reshape(-1) for o in out[6:]])
    frac, worst = dex_agreement(dex, g["out_dex_fine"])
    # stage-wise: same depths, only sigma differs
    z = G(g["z_fine"], dev)
    rdv = G(g["rd"], dev)
    with torch.no_grad():
        dex16 = _ops.volume_render_fwd(rf16, z, rdv, None, 0.0, False, thres, want_weights=False)[5]
        dex32 = _ops.volume_render_fwd(G(g["rf_fine"], dev), z, rdv, None, 0.0, False, thres, want_weights=False)[5]
    sfrac, sworst = dex_agreement(C(dex16), C(dex32))
    sig_err = rel_err(C(rf16)[..., 3], g["rf_fine"][..., 3])
    print(f"{precision} D8/W256 64+128 vs reference golden: rgb {rgb_psnr:.1f} dB, depth {depth_psnr:.1f} dB, Dex agreement "
          f"{frac:.4f} (worst miss {worst:.3f} m); fixed-depth Dex agreement {sfrac:.4f} (worst {sworst:.3f} m), sigma rel err {sig_err:.2e}")
    _record_measurement(f"headline_{precision}", dict(rgb_psnr=rgb_psnr, depth_psnr=depth_psnr, dex_agree=frac, dex_worst_m=worst,
                                                     fixed_depth_dex_agree=sfrac, sigma_rel_err=sig_err))
    assert rgb_psnr > rgb_floor and depth_psnr > depth_floor
    assert frac > dex_floor and sfrac > dex_floor
    assert np.array_equal(C(dex32), g["vf_dex"])  # the readout itself is exact on the golden sigma


def test_bf16_mode_renders_in_guarded_fp16_by_default(golden, dev):
    """nerf.set_precision('bf16') + the default render policy: a no-grad render runs the fp16 instance of the kernel (the Dex depth
    sweep reads validation renders: reference train_dexnerf_rgb.py:391-408, nerf/volume_rendering_utils.py:51-58) - it equals the
    'fp16' mode's render bit for bit and differs from the pure-bf16 one; training stays on the bf16 kernels (same loss as with the
    policy off).  The guard: weights that drive a hidden activation beyond fp16's range (65504) make the raw radiance field
    non-finite - the compositing passes count that, the render is repeated in bf16 (finite) with ONE warning, and the policy
    stays off for the rest of the process."""
    import warnings

    import nerf
    from nerf import train_utils
    name = "render_d8w256_val"
    g = golden(name)
    mkw, wfn, rkw = CASES[name]
    mc, mf = make_models(mkw, *wfn(), dev)
    ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
    ro, rd = G(g["ro"], dev)[None], G(g["rd"], dev)[None]

    def render():
        with torch.no_grad():
            return nerf.run_one_iter_of_nerf(1, ro.shape[1], 1.0, mc, mf, ro, rd, make_cfg(rkw), mode="validation",
                                             encode_position_fn=ex, encode_direction_fn=ed, m_thres_cand=list(M_THRES))
    try:
        nerf.set_precision("fp16")
        ref16 = render()
        nerf.set_precision("bf16")
        nerf.set_render_policy("bf16")
        pure = render()
        nerf.set_render_policy("fp16")
        assert nerf.get_render_policy() == "fp16"
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            pol = render()
        for a, b in zip(pol, ref16):
            assert torch.equal(a, b)
        assert not torch.equal(pol[3], pure[3])
        # training is untouched by the policy
        losses = []
        for policy in ("fp16", "bf16"):
            nerf.set_render_policy(policy)
            torch.manual_seed(3)
            out = nerf.run_one_iter_of_nerf(1, ro.shape[1], 1.0, mc, mf, ro[0], rd[0], make_cfg(rkw), mode="train",
                                            encode_position_fn=ex, encode_direction_fn=ed, m_thres_cand=list(M_THRES))
            assert out[3].requires_grad
            losses.append(float(out[3].detach().double().sum()))
        assert losses[0] == losses[1]
        # the guard
        nerf.set_render_policy("fp16")
        with torch.no_grad():
            mf.layers_xyz[2].weight.mul_(3.0e4)   # weights stay inside fp16, the layer's outputs (~1e5) do not; nothing for bf16
        nerf.models.mark_parameters_updated()
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            guarded = render()
            again = render()
        assert sum("fp16 render produced non-finite" in str(w.message) for w in caught) == 1
        assert train_utils._FP16_RENDER_DISABLED[0]
        nerf.set_render_policy("bf16")
        pure_big = render()
        for a, b, c in zip(guarded, pure_big, again):
            assert torch.equal(a, b) and torch.equal(c, b)
        assert bool(torch.isfinite(guarded[3]).all())
    finally:
        train_utils._FP16_RENDER_DISABLED[0] = False
        nerf.set_precision("fp32")


def test_fp32_mode_on_16384_rays_of_the_bench_scene(dev):
    """The exact-fp32 mode at the size bench.py samples (16,384 rays of the 400x400 bench view, D8/W256, 64+128) against the CPU
    oracle, STAGE-WISE (end-to-end index equality is not achievable once the network differs by an ulp, so every
    stage is checked on the oracle's own inputs; reference nerf/train_utils.py:136-190):
      A. coarse network + compositing on the oracle's coarse depths: raw field, rgb / depth / acc <= 1e-4 on EVERY ray;
      B. resampling + merge on the oracle's coarse weights: merged depths bit-exact on every ray (sample_pdf_2 + sort, :163-173);
      C. fine network + compositing on the oracle's merged depths: raw field, rgb / depth / acc <= 1e-4 on EVERY ray;
      D. the Dex readout on the oracle's raw field: every (threshold, ray) entry exact; on this kernel's field >= 0.9999;
      E. end to end: every ray over 1e-4 has at least one merged depth that differs from the oracle's - the excess is resampling
         flips (an ulp in a coarse sigma moving a sample across a density edge), nothing else; p99.9 <= 2e-5."""
    import bench
    import nerf
    from nerf import _ops
    models, cfg, ro, rd, ex, ed = bench.build_scene(dev, 0)
    with torch.no_grad():
        out = nerf.run_one_iter_of_nerf(bench.H, bench.W, 1.0, models[0], models[1], ro, rd, cfg, mode="validation",
                                        encode_position_fn=ex, encode_direction_fn=ed, m_thres_cand=bench.M_THRES)
    cb, sel, ref, aux = bench.cpu_baseline(16384, return_aux=True)
    idx = torch.from_numpy(sel).to(dev)
    n = len(sel)

    def per_ray_err(a, b):
        a = C(a).reshape(n, -1).astype(np.float64)
        b = b.numpy().reshape(n, -1).astype(np.float64)
        return np.abs(a - b).max(-1) / np.abs(b).max()

    rays = aux["rays"].to(dev)
    # (the stages below run on the ORACLE's ray rows; this library's own rows may differ from them by an ulp in the unit view vector -
    # CPU vs device norm - which is part of the end-to-end comparison E, not of the stage checks)
    rows_equal = bool(torch.equal(rays, _ops.pack_ray_rows(ro.reshape(-1, 3)[idx], rd.reshape(-1, 3)[idx], rd.reshape(-1, 3)[idx], 2.0, 6.0)))
    rd_s = rays[:, 3:6].contiguous()
    pc, pf = models[0].packed(precision=_ops._hip.PREC_F32), models[1].packed(precision=_ops._hip.PREC_F32)
    stage = {}
    with torch.no_grad():
        # A: coarse pass on the oracle's depths
        z_c = _ops.coarse_depths(rays, bench.NC, False, None)
        assert torch.equal(z_c.cpu(), aux["z_coarse"])
        rf_c = _ops.run_network_rays(pc, rays, aux["z_coarse"].to(dev))
        rgb_c, _, acc_c, w_c, depth_c, _ = _ops.volume_render_fwd(rf_c, z_c, rd_s, None, 0.0, False, bench.M_THRES)
        stage["rf_coarse"] = per_ray_err(rf_c, aux["rf_coarse"])
        stage["rgb_coarse"] = per_ray_err(rgb_c, ref[0]); stage["depth_coarse"] = per_ray_err(depth_c, ref[1]); stage["acc_coarse"] = per_ray_err(acc_c, ref[2])
        # B: the sampler + merge on the oracle's weights
        z_f_inj = _ops.fine_depths(aux["z_coarse"].to(dev), aux["w_coarse"].to(dev), bench.NF, None)
        flips_b = int((z_f_inj.cpu() != aux["z_fine"]).any(-1).sum())
        # C: fine pass on the oracle's merged depths
        z_f = aux["z_fine"].to(dev)
        rf_f = _ops.run_network_rays(pf, rays, z_f)
        rgb_f, _, acc_f, _, depth_f, dex_f = _ops.volume_render_fwd(rf_f, z_f, rd_s, None, 0.0, False, bench.M_THRES, want_weights=False)
        stage["rf_fine"] = per_ray_err(rf_f, aux["rf_fine"])
        stage["rgb_fine"] = per_ray_err(rgb_f, ref[3]); stage["depth_fine"] = per_ray_err(depth_f, ref[4]); stage["acc_fine"] = per_ray_err(acc_f, ref[5])
        # D: the Dex readout given the oracle's field / this kernel's field
        dex_given = _ops.volume_render_fwd(aux["rf_fine"].to(dev), z_f, rd_s, None, 0.0, False, bench.M_THRES, want_weights=False)[5]
        theirs = torch.stack(list(ref[6:]))
        dex_exact = float((dex_given.cpu() == theirs).double().mean())
        dex_stage = float((dex_f.cpu() == theirs).double().mean())
        # E: this library's own merged depths (the stage composition is the one-call render bit for bit)
        z_f_own = _ops.fine_depths(z_c, w_c, bench.NF, None)
    worst = {k: (int((v > 1e-4).sum()), float(v.max())) for k, v in stage.items()}
    for k, (over, mx) in worst.items():
        assert over == 0, (k, over, mx, worst)
    assert flips_b == 0, f"sampler + merge on the oracle's weights: {flips_b} rays with a differing merged depth"
    assert dex_exact == 1.0, dex_exact
    assert dex_stage >= 0.9999, dex_stage
    flipped = (z_f_own.cpu() != aux["z_fine"]).any(-1).numpy()
    e2e = {}
    for i, nm in ((0, "rgb_coarse"), (3, "rgb_fine"), (4, "depth_fine"), (5, "acc_fine")):
        per_ray = per_ray_err(out[i].reshape(bench.H * bench.W, -1)[idx], ref[i])
        over = per_ray > 1e-4
        e2e[nm] = (int(over.sum()), float(np.quantile(per_ray, 0.999)), float(per_ray.max()))
        assert not (over & ~flipped).any(), (nm, "a ray over 1e-4 whose merged depths equal the oracle's", e2e[nm])
        assert e2e[nm][1] <= 2e-5, (nm, e2e[nm])
    dex = bench.dex_agreement(out, ref, sel, dev)
    _record_measurement("fp32_16384", dict(**{"stage_max_" + k: v[1] for k, v in worst.items()}, rays_with_a_flipped_depth=int(flipped.sum()), ray_rows_equal=rows_equal,
                                            rgb_fine_over=e2e["rgb_fine"][0], rgb_fine_p999=e2e["rgb_fine"][1], rgb_fine_max=e2e["rgb_fine"][2],
                                            dex_given_sigma_exact=dex_exact, dex_stage=dex_stage, dex_agree=dex["agree_frac"],
                                            cpu_rays_per_s=cb["value"]))
    assert dex["agree_frac"] >= 0.9999, dex


def test_config5_training_step_fp32_luminance_head_against_oracle_autograd(dev, monkeypatch):
    """BASELINE.json configs[4] as configured: 128 coarse + 256 fine samples, exact fp32, D8/W256 nets, the IR loss head (MSE on
    the luminance 0.299 r + 0.587 g + 0.114 b of both passes: reference train_nerf_ir.py:260-263), perturbed sampling + density
    noise with the same injected draws on both sides - one training step on the fused kernels against autograd through the CPU
    oracle: loss at 1e-4, every parameter gradient of both networks at 1e-3 (the gradient tolerance of
    test_train_step_matches_reference)."""
    import nerf
    from nerf import synthetic as syn
    from oracle import nerf_oracle as oc
    kw = CASES["render_d8w256_val"][0]
    sd_c, sd_f = CASES["render_d8w256_val"][1]()
    mc, mf = make_models(kw, sd_c, sd_f, dev)
    h = w = 400
    nc, nf, n = 128, 256, 96
    e_mat, k_mat = torch.from_numpy(syn.scene_pose(9)), torch.from_numpy(syn.intrinsic(h, w))
    ro, rd = nerf.get_ray_bundle(h, w, float(k_mat[0, 0]), e_mat.to(dev), k_mat.to(dev))
    sel = torch.from_numpy(syn.select_rays(h, w, n, seed=77)).to(dev)
    ro, rd = ro.reshape(-1, 3)[sel].contiguous(), rd.reshape(-1, 3)[sel].contiguous()
    gen = torch.Generator().manual_seed(123)
    draws = dict(t_rand=torch.rand(n, nc, generator=gen), noise_c=torch.randn(n, nc, generator=gen),
                 u=torch.rand(n, nf, generator=gen), noise_f=torch.randn(n, nc + nf, generator=gen))
    target = torch.rand(n, 3, generator=gen)
    rkw = dict(num_coarse=nc, num_fine=nf, near=2.0, far=6.0, perturb=True, noise_std=0.2)

    def lum(t):
        return 0.299 * t[..., 0] + 0.587 * t[..., 1] + 0.114 * t[..., 2]
    q_rand = [draws["t_rand"].to(dev), draws["u"].to(dev)]
    q_randn = [draws["noise_c"].to(dev), draws["noise_f"].to(dev)]
    monkeypatch.setattr(torch, "rand", lambda *a, **k: q_rand.pop(0))
    monkeypatch.setattr(torch, "randn", lambda *a, **k: q_randn.pop(0))
    ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
    out = nerf.run_one_iter_of_nerf(h, w, 1.0, mc, mf, ro, rd, make_cfg(rkw, 4096), mode="train", encode_position_fn=ex,
                                    encode_direction_fn=ed, m_thres_cand=list(M_THRES))
    monkeypatch.undo()
    assert not q_rand and not q_randn
    tgt = target.to(dev)
    loss = nerf.img2mse(lum(out[0]), lum(tgt)) + nerf.img2mse(lum(out[3]), lum(tgt))
    loss.backward()
    # the oracle, with autograd through its own nn.functional composition
    tsd_c, tsd_f = oc.to_torch_sd(sd_c, requires_grad=True), oc.to_torch_sd(sd_f, requires_grad=True)
    cfg_o = oc.RenderCfg(chunksize=4096, m_thres=M_THRES, num_coarse=nc, num_fine=nf, near=2.0, far=6.0, perturb=True, noise_std=0.2)
    mcfg = oc.ModelCfg(**kw)
    ref = oc.run_one_iter(ro.cpu(), rd.cpu(), tsd_c, tsd_f, mcfg, mcfg, cfg_o, draws=draws)
    mse = torch.nn.functional.mse_loss
    loss_ref = mse(lum(ref[0]), lum(target)) + mse(lum(ref[3]), lum(target))
    loss_ref.backward()
    assert abs(loss.item() - loss_ref.item()) < 1e-4 * abs(loss_ref.item()), (loss.item(), loss_ref.item())
    worst = 0.0
    for model, tsd in ((mc, tsd_c), (mf, tsd_f)):
        for k, p in model.named_parameters():
            err = rel_err(C(p.grad), tsd[k].grad.numpy())
            worst = max(worst, err)
            assert err < 1e-3, (k, err)
    _record_measurement("config5_train_fp32_ir", dict(loss=loss.item(), loss_ref=loss_ref.item(), worst_grad_rel_err=worst))


@pytest.mark.parametrize("nc,nf,tag", [(64, 192, "config 4 sampling"), (128, 256, "config 5 sampling")])
def test_config4_and_config5_sampling_on_d8w256_fp32(dev, nc, nf, tag):
    """BASELINE.json configs[3] (64+192) and configs[4] (128+256, fp32) sample counts on the D8/W256 nets those configs name,
    256 rays of a scene view, exact-fp32 mode against the CPU oracle at 1e-4."""
    import nerf
    from nerf import synthetic as syn
    from oracle import nerf_oracle as oc
    kw = CASES["render_d8w256_val"][0]
    sd_c, sd_f = CASES["render_d8w256_val"][1]()
    mc, mf = make_models(kw, sd_c, sd_f, dev)
    h = w = 800
    e_mat, k_mat = torch.from_numpy(syn.scene_pose(5)), torch.from_numpy(syn.intrinsic(h, w))
    ro, rd = nerf.get_ray_bundle(h, w, float(k_mat[0, 0]), e_mat.to(dev), k_mat.to(dev))
    sel = torch.from_numpy(syn.select_rays(h, w, 256, seed=nc + nf)).to(dev)
    ro, rd = ro.reshape(-1, 3)[sel].contiguous(), rd.reshape(-1, 3)[sel].contiguous()
    rkw = dict(num_coarse=nc, num_fine=nf, near=2.0, far=6.0)
    ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
    with torch.no_grad():
        out = nerf.run_one_iter_of_nerf(h, w, 1.0, mc, mf, ro, rd, make_cfg(rkw, 4096), mode="train", encode_position_fn=ex,
                                        encode_direction_fn=ed, m_thres_cand=list(M_THRES))
    cfg_o = oc.RenderCfg(chunksize=4096, m_thres=M_THRES, **rkw)
    mcfg = oc.ModelCfg(**kw)
    with torch.no_grad():
        ref = oc.run_one_iter(ro.cpu(), rd.cpu(), oc.to_torch_sd(sd_c), oc.to_torch_sd(sd_f), mcfg, mcfg, cfg_o)
    for i in range(6):
        assert rel_err(C(out[i]), ref[i].numpy()) < TOL, (tag, i)
    frac, worst = dex_agreement(np.stack([C(o) for o in out[6:]]), np.stack([o.numpy() for o in ref[6:]]))
    print(f"{tag}: Dex agreement {frac:.4f}, worst miss {worst:.3f} m")
    assert frac > 0.995, (tag, frac)


def test_config4_full_size_render_properties_bf16(dev):
    """BASELINE.json configs[3] at full size on one GPU: 800x800 rays, 64+192 samples, D8/W256, bf16 (the 48-point kernel) -
    too large for the oracle, so size-independent properties: determinism, chunk invariance (one 640,000-ray chunk vs
    ragged 100,003-ray chunks: bit-identical), ascending merged depths, acc == sum(weights), acc in [0, 1], finite maps,
    Dex depths inside [near, far]; and a 512-ray subset against the fp32 oracle on PSNR."""
    import nerf
    from nerf import _ops, synthetic as syn
    from oracle import nerf_oracle as oc
    kw = CASES["render_d8w256_val"][0]
    sd_c, sd_f = CASES["render_d8w256_val"][1]()
    mc, mf = make_models(kw, sd_c, sd_f, dev)
    h = w = 800
    e_mat, k_mat = torch.from_numpy(syn.scene_pose(2)), torch.from_numpy(syn.intrinsic(h, w))
    ro, rd = nerf.get_ray_bundle(h, w, float(k_mat[0, 0]), e_mat.to(dev), k_mat.to(dev))
    rkw = dict(num_coarse=64, num_fine=192, near=2.0, far=6.0)
    ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
    nerf.set_precision("bf16")
    try:
        def run(chunk):
            with torch.no_grad():
                return nerf.run_one_iter_of_nerf(h, w, 1.0, mc, mf, ro, rd, make_cfg(rkw, chunk), mode="validation",
                                                 encode_position_fn=ex, encode_direction_fn=ed, m_thres_cand=list(M_THRES))
        a = run(h * w)
        b = run(h * w)
        c = run(100003)
        for x, y, z_ in zip(a, b, c):
            assert x.shape[:2] == (h, w)
            assert torch.equal(x, y)      # deterministic
            assert torch.equal(x, z_)     # chunking never changes a bit
        for t in a[:6]:
            assert bool(torch.isfinite(t).all())
        acc = a[5]
        assert float(acc.min()) >= 0.0 and float(acc.max()) <= 1.0 + 1e-5
        dex = torch.stack(list(a[6:]))
        assert float(dex.min()) >= 2.0 - 1e-4 and float(dex.max()) <= 6.0 + 1e-4
        # stage properties on a 65,536-ray block of the same image
        rays = oc.pack_rays(ro.reshape(-1, 3)[:65536].cpu(), rd.reshape(-1, 3)[:65536].cpu(), oc.RenderCfg(chunksize=4096, m_thres=M_THRES, **rkw)).to(dev)
        z_c = _ops.coarse_depths(rays, 64, False, None)
        rf = _ops.run_network_rays(mc.packed(), rays, z_c)
        _, _, acc_c, wts, _, _ = _ops.volume_render_fwd(rf, z_c, rays[:, 3:6], None, 0.0, False, [])
        assert rel_err(C(wts.sum(-1)), C(acc_c)) < 1e-5
        z_f = _ops.fine_depths(z_c, wts, 192, None)
        assert z_f.shape == (65536, 256) and bool((z_f[:, 1:] >= z_f[:, :-1]).all())
    finally:
        nerf.set_precision("fp32")
    sel = torch.from_numpy(syn.select_rays(h, w, 512, seed=4)).to(dev)
    cfg_o = oc.RenderCfg(chunksize=4096, m_thres=M_THRES, **rkw)
    mcfg = oc.ModelCfg(**kw)
    with torch.no_grad():
        ref = oc.run_one_iter(ro.reshape(-1, 3)[sel].cpu(), rd.reshape(-1, 3)[sel].cpu(), oc.to_torch_sd(sd_c), oc.to_torch_sd(sd_f), mcfg, mcfg, cfg_o)
    psnr = psnr_db(C(a[3].reshape(-1, 3)[sel]), ref[3].numpy())
    frac, worst = dex_agreement(np.stack([C(o.reshape(-1)[sel]) for o in a[6:]]), np.stack([o.numpy() for o in ref[6:]]))
    print(f"800x800 64+192 bf16: rgb PSNR {psnr:.1f} dB vs the fp32 oracle on 512 rays; Dex agreement {frac:.4f}, worst miss {worst:.3f} m")
    assert psnr > 38.0


@pytest.mark.parametrize("precision", ["fp32", "bf16-s16", "bf16"])
def test_one_call_training_path_equals_stage_composition(golden, dev, precision, monkeypatch):
    """predict_and_render_radiance under autograd as ONE C-ABI call forward
    (dn_render_rays_train) and one backward (dn_render_rays_backward), against the stage-by-stage Python composition of the
    same kernels (coarse depths -> FusedNetFn -> VolumeRenderFn -> fine depths -> FusedNetFn -> VolumeRenderFn) on the
    reference's recorded training rays and RNG draws: the six maps, the Dex depths and every saved-gradient-independent
    quantity are bit-identical; the parameter gradients agree to the reordering noise of the weight-gradient kernel's fp32
    atomics (its partial sums arrive in a different order from launch to launch).  Also with a FlatGradBucket attached:
    the two half-calls (fine, then coarse) accumulate straight into the bucket and autograd hands back no gradient."""
    import nerf
    from nerf import parallel, train_utils
    name = "train_d8w256"
    g = golden(name)
    mkw, wfn, rkw = CASES[name]
    cfg = make_cfg(rkw)
    draws = draws_of(g)
    ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
    target = G(g["target"], dev)

    def run(stagewise, bucket=False):
        mc, mf = make_models(mkw, *wfn(), dev)
        b = parallel.FlatGradBucket([mc, mf]) if bucket else None
        q_rand = [G(draws["t_rand"], dev), G(draws["u"], dev)]
        q_randn = [G(draws["noise_c"], dev), G(draws["noise_f"], dev)]
        monkeypatch.setattr(torch, "rand", lambda *a, **k: q_rand.pop(0))
        monkeypatch.setattr(torch, "randn", lambda *a, **k: q_randn.pop(0))
        monkeypatch.setattr(train_utils, "_STAGEWISE_TRAINING", [stagewise])
        out = nerf.run_one_iter_of_nerf(1, len(g["ro"]), 1.0, mc, mf, G(g["ro"], dev), G(g["rd"], dev), cfg, mode="train",
                                        encode_position_fn=ex, encode_direction_fn=ed, m_thres_cand=list(M_THRES))
        assert not q_rand and not q_randn
        # a loss that also pulls on depth and acc, so all six upstream gradients are exercised
        loss = (nerf.img2mse(out[0], target) + nerf.img2mse(out[3], target) + 0.1 * out[1].mean() + 0.05 * out[2].mean()
                + 0.1 * out[4].mean() + 0.05 * out[5].mean())
        if b is not None:
            b.zero()
        loss.backward()
        grads = {("c" if m is mc else "f") + "." + k: p.grad.detach().clone() for m in (mc, mf) for k, p in m.named_parameters()}
        return [o.detach() for o in out], grads, b

    nerf.set_precision(precision)
    try:
        calls = []
        orig = nerf._train.RenderRaysTrainFn.apply
        monkeypatch.setattr(nerf._train.RenderRaysTrainFn, "apply", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
        out_s, grads_s, _ = run(True)
        assert not calls
        out_1, grads_1, _ = run(False)
        assert len(calls) == 1
        out_b, grads_b, bucket = run(False, bucket=True)
    finally:
        nerf.set_precision("fp32")
    for a, b, c in zip(out_s, out_1, out_b):
        assert torch.equal(a, b) and torch.equal(a, c)
    for k in grads_s:
        scale = float(grads_s[k].abs().max()) + 1e-30
        assert float((grads_s[k] - grads_1[k]).abs().max()) <= 2e-5 * scale, k
        assert float((grads_s[k] - grads_b[k]).abs().max()) <= 2e-5 * scale, k
    lo, hi = bucket.flat.data_ptr(), bucket.flat.data_ptr() + 4 * bucket.flat.numel()
    assert all(lo <= p.grad.data_ptr() < hi for p in bucket.params)


def test_inputs_that_require_grad_take_the_differentiable_path(golden, dev):
    """Pose / ray optimisation: points or rays that require grad must GET a gradient.  The fused training kernels
    differentiate w.r.t. the parameters only, so such calls run the torch encoding + nn.Linear composition (and FusedNetFn
    refuses them loudly rather than returning None); values agree with the fused path, and d(out)/d(pts) matches a central
    finite difference of the fused forward."""
    import nerf
    from nerf import _train
    name = "render_lego_val"
    g = golden(name)
    mkw, wfn, _ = CASES[name]
    mc, _ = make_models(mkw, *wfn(), dev)
    ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
    pts = G(g["pts_coarse"][:8, :16], dev).clone().requires_grad_(True)
    rd = G(g["rd"][:8], dev)
    rays = torch.cat([torch.zeros(8, 8, device=dev), torch.nn.functional.normalize(rd, dim=-1)], -1)
    out = nerf.run_network(mc, pts, rays, 4096, ex, ed)
    with torch.no_grad():
        fused = nerf.run_network(mc, pts.detach(), rays, 4096, ex, ed)
    assert rel_err(C(out), C(fused)) < 1e-4
    w = torch.linspace(0.5, 1.5, out.numel(), device=dev).reshape(out.shape)
    (out * w).sum().backward()
    assert pts.grad is not None and bool(torch.isfinite(pts.grad).all()) and float(pts.grad.abs().max()) > 0
    assert all(p.grad is not None for p in mc.parameters())
    # central difference along one coordinate of one point, through the fused (exact-fp32) forward
    eps = 1e-3
    with torch.no_grad():
        d = torch.zeros_like(pts); d[3, 5, 1] = eps
        fd = ((nerf.run_network(mc, pts.detach() + d, rays, 4096, ex, ed) - nerf.run_network(mc, pts.detach() - d, rays, 4096, ex, ed)) * w).sum() / (2 * eps)
    assert abs(float(fd) - float(pts.grad[3, 5, 1])) < 2e-2 * max(abs(float(fd)), 1.0)
    with pytest.raises(RuntimeError):
        _train.FusedNetFn.apply(mc, pts.reshape(-1, 3), rays[:, -3:], 16, True, True, *[p for m in mc.linear_modules() for p in (m.weight, m.bias)])


def test_second_device_gets_its_own_launch_attributes():
    """hipFuncAttributeMaxDynamicSharedMemorySize is per device: a process that launches the > 64 KiB-LDS kernels on
    a second GPU must set it there too.  Needs two visible GPUs (the gpurun boxes have one: skipped there)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    import nerf
    from nerf import _ops, synthetic as syn
    kw = CASES["render_d8w256_val"][0]
    sd = {k: torch.from_numpy(v) for k, v in syn.synth_state_dict(42, **kw).items()}
    outs = []
    nerf.set_precision("bf16")
    try:
        for idx in (0, 1):
            d = torch.device("cuda", idx)
            with torch.cuda.device(d):
                m = nerf.models.FlexibleNeRFModel(**kw); m.load_state_dict(sd); m = m.to(d)
                pts = torch.linspace(-1, 1, 3 * 1000, device=d).reshape(1000, 3)
                vd = torch.nn.functional.normalize(torch.ones(1000, 3, device=d), dim=-1)
                with torch.no_grad():
                    outs.append(_ops.run_network_pts(m.packed(), pts, vd, 1).cpu())
    finally:
        nerf.set_precision("fp32")
    assert torch.equal(outs[0], outs[1])


def test_bench_multi_gpu_leg_rehearsal_two_ranks_one_gpu(dev):
    """`bench.py --gpus 2` under torch.distributed.run, as the driver launches it, with two gloo ranks sharing this GPU (RCCL
    refuses two ranks on one device): the render timing, the data-parallel training step (`train_dp`: broadcast, per-network
    all-reduce through the FlatGradBucket, fused Adam) and the one JSON line on rank 0.  Numbers are meaningless here (one GPU,
    gloo); the path must run and the line must carry the fields the N > 1 contract asks for."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, DEXNERF_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["unit"] == "rays/s" and "cpu_baseline" not in line
    assert abs(line["value"] - 2 * 160000 / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-6
    dp = line["train_dp"]
    assert dp["n_gpus"] == 2 and dp["rccl_ranks"] == 2 and dp["allreduce_bytes"] == 2 * 595844 * 4
    assert dp["rays_per_s"] > 0 and dp["allreduce_ms"] > 0 and dp["roofline"]["bound"] == "hbm"
    assert dp["hip_graphs_per_step"] == 3 and dp["graph_fallback"] is None       # graphs around the exchange (nerf.GraphedTrainStep)
    rs = dp["render_sharded"]                                                    # ONE 800x800 image, its rows split over the ranks
    assert rs["n_gpus"] == 2 and rs["render_sharded_ms"] > rs["all_gather_ms"] > 0 and rs["gathered_bytes_per_rank"] == 800 * 800 * 4 * (10 + 20)


def test_bench_started_plainly_with_two_gpus_launches_its_own_ranks(dev):
    """`python bench.py --gpus 2` WITHOUT torch.distributed.run (how the driver starts the N = 1 case): bench.py starts a child
    torch.distributed.run itself before touching the GPU, relays the one JSON line and exits with the child's code (two gloo ranks on
    this GPU here; --no-train keeps the rehearsal short)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TORCHELASTIC_RUN_ID")}
    env.update(DEXNERF_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-train"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["dtype"] == line["config_dtype"] == "bf16"
    assert abs(line["value"] - 2 * 160000 / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-6


def test_training_psnr_at_matched_iterations_against_the_cpu_oracle(dev):
    """SURVEY section 8(d) metric (b) as a gate (reference definition train_dexnerf_rgb.py:264-279): 300 iterations of this library's
    training iteration (nerf.FusedTrainStep + nerf.FlatAdam, device-side draws, bf16 kernels) against 300 iterations of autograd
    through the CPU oracle from the same initial weights on the same teacher images - the training PSNR at the final mark within 1 dB,
    every mark within 2 dB (each side makes its own draws: the comparison is statistical), and both well above the starting level."""
    import bench
    res = bench.train_psnr_vs_oracle(dev, iters=300, precision="bf16")
    rows = res["marks"]
    assert [r["iteration"] for r in rows] == [100, 200, 300]
    assert abs(res["final_delta_db"]) <= 1.0, rows
    # (the curve leaves its first plateau between iterations 100 and 200 - a few iterations earlier or later per draw sequence - so
    # the middle marks are loose)
    assert all(abs(r["hip_psnr_db"] - r["oracle_psnr_db"]) <= 4.0 for r in rows), rows
    assert rows[-1]["hip_psnr_db"] > rows[0]["hip_psnr_db"] + 5.0 and rows[-1]["oracle_psnr_db"] > rows[0]["oracle_psnr_db"] + 5.0, rows
    _record_measurement("train_psnr_vs_oracle", dict(final_hip=rows[-1]["hip_psnr_db"], final_oracle=rows[-1]["oracle_psnr_db"],
                                                      delta=res["final_delta_db"], oracle_cpu_s=res["oracle_cpu_s"]))


def test_as_shipped_render_instance_with_overlapped_encoding_is_bit_identical(dev):
    """The as-shipped 4 x 128 nets on rays + depths run the forward instance that encodes tile t + 1 inside tile t's trunk stages
    (mlp_forward48_kernel<128, F, 4, 0, 1, 0, 1>: csrc/mlp_fused48.hip).  Same arithmetic, another schedule: its output must equal
    the plain fixed-shape instance's (DEXNERF_G48_NO_OVERLAP=1) bit for bit - one point, ragged tails, one tile, more tiles than
    workgroups (every workgroup then runs its first-tile prologue and the steady state), bf16 and fp16."""
    import nerf
    from nerf import _hip, _ops, synthetic as syn
    kw = dict(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True)
    nerf.set_precision("bf16")
    try:
        m = nerf.models.FlexibleNeRFModel(**kw)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.synth_state_dict(43, sigma_bias=-2.0, **kw).items()})
        m = m.to(dev)
        for prec in (_hip.PREC_BF16, _hip.PREC_F16):
            pk = m.packed(precision=prec)
            for n, s in ((1, 1), (7, 55), (1000, 64), (3000, 128), (70001, 64)):
                g = torch.Generator(device=dev).manual_seed(n + s)
                rd = torch.nn.functional.normalize(torch.randn(n, 3, device=dev, generator=g), dim=-1)
                rays = torch.cat([torch.randn(n, 3, device=dev, generator=g), rd, torch.full((n, 1), 0.3, device=dev),
                                  torch.full((n, 1), 4.0, device=dev), rd], -1).contiguous()
                z = torch.sort(torch.rand(n, s, device=dev, generator=g) * 3.7 + 0.3, -1)[0].contiguous()
                out = _ops.run_network_rays(pk, rays, z)
                os.environ["DEXNERF_G48_NO_OVERLAP"] = "1"
                try:
                    plain = _ops.run_network_rays(pk, rays, z)
                finally:
                    os.environ.pop("DEXNERF_G48_NO_OVERLAP", None)
                assert bool(torch.isfinite(out).all()) and torch.equal(out, plain), (prec, n, s)
    finally:
        nerf.set_precision("fp32")


def test_ray_rows_packed_by_one_kernel_equal_the_torch_composition(dev):
    """dn_pack_ray_rows (what run_one_iter_of_nerf uses for device rays) against the reference's composition - norm, divide,
    ones_like x2, cat (nerf/train_utils.py:220-250) - bit for bit, with and without view directions, and through the NDC branch
    (view directions from the unwarped rays)."""
    import nerf
    from nerf import _ops
    g = torch.Generator(device=dev).manual_seed(4)
    n = 5000
    ro = torch.randn(n, 3, device=dev, generator=g)
    rd = torch.randn(n, 3, device=dev, generator=g) * 3.0
    near, far = 0.3, 4.0
    viewdirs = rd / rd.norm(p=2, dim=-1).unsqueeze(-1)
    expect = torch.cat([ro, rd, near * torch.ones_like(rd[..., :1]), far * torch.ones_like(rd[..., :1]), viewdirs], dim=-1).float()
    assert torch.equal(_ops.pack_ray_rows(ro, rd, rd, near, far), expect)
    assert torch.equal(_ops.pack_ray_rows(ro, rd, None, near, far), expect[:, :8].contiguous())
    other = torch.randn(n, 3, device=dev, generator=g)
    v2 = other / other.norm(p=2, dim=-1).unsqueeze(-1)
    assert torch.equal(_ops.pack_ray_rows(ro, rd, other, near, far)[:, 8:], v2)


def test_in_kernel_compositing_is_bit_identical_to_the_two_kernel_render(dev):
    """DEXNERF_FUSED_COMPOSITE=1: the fixed-shape 48-point instances composite the rays of a tile themselves (raw rows staged in LDS,
    one wave per finished ray runs composite_body.h - the body composite_fwd_kernel runs) instead of writing the radiance field to
    HBM.  Off by default (it is slower: DESIGN 4.7f); where it applies every map must equal the two-kernel render bit for bit -
    both widths, fp16 and bf16, whole rays per 384-point tile (64, 128, 192 samples; 32: twelve rays per tile, more than waves),
    a sample count that does not divide the tile (falls back), ragged ray counts, white background, coarse-only."""
    import nerf
    from nerf import _hip, _ops, synthetic as syn
    thres = [float(m) for m in range(5, 105, 5)]
    nerf.set_precision("bf16")
    try:
        for width, layers, biases in ((256, 8, (-150.0, -20.0)), (128, 4, (-15.0, -2.0))):
            kw = dict(num_layers=layers, hidden_size=width, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True)
            nets = []
            for seed, b in zip((42, 43), biases):
                m = nerf.models.FlexibleNeRFModel(**kw)
                m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.synth_state_dict(seed, sigma_bias=b, **kw).items()})
                nets.append(m.to(dev))
            for prec in (_hip.PREC_F16, _hip.PREC_BF16):
                pc, pf = nets[0].packed(precision=prec), nets[1].packed(precision=prec)
                for n, nc, nf, white in ((1, 64, 128, False), (777, 64, 128, True), (3000, 64, 64, False), (500, 64, 192, False), (300, 32, 0, False)):
                    g = torch.Generator(device=dev).manual_seed(n + nc)
                    rd = torch.nn.functional.normalize(torch.tensor([0.0, 0.0, -1.0], device=dev) + 0.3 * torch.randn(n, 3, device=dev, generator=g), dim=-1) * 1.3
                    ro = torch.tensor([0.0, 0.0, 4.0], device=dev) + 0.05 * torch.randn(n, 3, device=dev, generator=g)
                    rays = torch.cat([ro, rd, torch.full((n, 1), 2.0, device=dev), torch.full((n, 1), 6.0, device=dev),
                                      torch.nn.functional.normalize(rd, dim=-1)], -1).contiguous()
                    plain = _ops.render_rays(pc, pf if nf else None, rays, nc, nf, False, 0.0, white, thres)
                    os.environ["DEXNERF_FUSED_COMPOSITE"] = "1"
                    try:
                        fused = _ops.render_rays(pc, pf if nf else None, rays, nc, nf, False, 0.0, white, thres)
                    finally:
                        os.environ.pop("DEXNERF_FUSED_COMPOSITE", None)
                    for a, b in zip(fused, plain):
                        assert (a is None) == (b is None) and (a is None or torch.equal(a, b)), (width, prec, n, nc, nf)
    finally:
        nerf.set_precision("fp32")


def test_training_driver_resumes_from_its_checkpoint(dev, tmp_path):
    """train_dexnerf.py: 90 iterations + checkpoint + 60 more from the checkpoint against 150 uninterrupted ones (fused step, flat
    Adam, device-side draws).  (A run is bit-reproducible since round 4 - test_training_run_is_a_pure_function_of_its_seed - but a resumed
    run re-captures its graph and re-packs from the loaded weights; the check here is on what a resume must restore: the
    checkpoint is in the reference's format with torch.optim.Adam's state layout; the resumed run starts at iteration 90 at the
    checkpoint's loss level (a lost optimizer state or re-initialised weights would not) and ends where the uninterrupted run ends."""
    import sys
    sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd"))
    import nerf
    import train_dexnerf
    base = ["--size", "32", "--views", "6", "--num-random-rays", "512", "--layers", "4", "--width", "128", "--validate-every", "0", "--quiet",
            "--precision", "bf16-s8", "--lr-decay", "1"]
    ck_a, ck_b = (os.path.join(str(tmp_path), n) for n in ("a.ckpt", "b.ckpt"))
    try:
        first = train_dexnerf.main(base + ["--iters", "90", "--save", ck_a])
        resumed = train_dexnerf.main(base + ["--iters", "150", "--load-checkpoint", ck_a, "--save", ck_b])
        full = train_dexnerf.main(base + ["--iters", "150"])
    finally:
        nerf.set_precision("fp32")
    a, b = torch.load(ck_a, map_location="cpu"), torch.load(ck_b, map_location="cpu")
    assert a["iter"] == 90 and b["iter"] == 150 and set(a) >= {"model_coarse_state_dict", "model_fine_state_dict", "optimizer_state_dict", "loss", "psnr"}
    st = a["optimizer_state_dict"]["state"]
    assert float(st[0]["step"]) == 90.0 and set(st[0]) == {"step", "exp_avg", "exp_avg_sq"}
    assert float(b["optimizer_state_dict"]["state"][0]["step"]) == 150.0            # the step count went on from 90
    assert isinstance(a["optimizer_state_dict"]["param_groups"][0]["lr"], float)
    assert resumed["history"][0][0] == 100                                          # first logged iteration of the resumed run (90 .. 149)
    assert abs(resumed["history"][0][2] - first["history"][-1][2]) < 3.0, (resumed["history"][0], first["history"][-1])   # dB
    assert abs(resumed["history"][-1][2] - full["history"][-1][2]) < 2.5, (resumed["history"][-1], full["history"][-1])
    assert resumed["history"][-1][2] > first["history"][0][2] + 5.0                  # and it kept learning


def test_training_run_is_a_pure_function_of_its_seed(dev, tmp_path):
    """Two runs of train_dexnerf.py with one seed end in bit-identical weights and optimizer moments (the reference on the CPU is
    reproducible; round 3's weight-gradient kernel summed its workgroups' partials with fp32 atomics and was not): the draws are a
    function of (seed, iteration), the loss head and Adam sum in a fixed order, and the weight gradients are reduced in workgroup order
    (dn_render_rays_backward_ws).  Both bf16 training modes: 'bf16' (8-bit saves, fp8 MFMA) and 'bf16-s16'; and the atomics form,
    switched on explicitly, gives the same gradients to rounding."""
    import sys
    sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd"))
    import nerf
    import train_dexnerf
    from nerf import _ops
    try:
        for prec in ("bf16", "bf16-s16"):
            base = ["--size", "32", "--views", "6", "--num-random-rays", "512", "--layers", "4", "--width", "128", "--validate-every", "0",
                    "--quiet", "--precision", prec, "--iters", "80"]
            paths = [os.path.join(str(tmp_path), f"{prec}_{k}.ckpt") for k in range(2)]
            for path in paths:
                train_dexnerf.main(base + ["--save", path])
            a, b = (torch.load(p, map_location="cpu") for p in paths)
            for key in ("model_coarse_state_dict", "model_fine_state_dict"):
                for name in a[key]:
                    assert torch.equal(a[key][name], b[key][name]), (prec, key, name)
            for idx, st in a["optimizer_state_dict"]["state"].items():
                assert torch.equal(st["exp_avg"], b["optimizer_state_dict"]["state"][idx]["exp_avg"]), (prec, idx)
        # one gradient evaluation: fixed-order reduction vs atomics - the same sums to fp32 reordering noise
        nerf.set_precision("bf16")
        mkw = dict(num_layers=8, hidden_size=256, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True)
        torch.manual_seed(4)
        m = nerf.models.FlexibleNeRFModel(**mkw).to(dev)
        ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
        pts = torch.randn(300, 40, 3, device=dev)
        vd = torch.nn.functional.normalize(torch.randn(300, 3, device=dev), dim=-1)
        rays = torch.cat([torch.zeros(300, 8, device=dev), vd], -1)
        g_up = torch.randn(300, 40, 4, device=dev)
        grads = {}
        for mode in (True, True, False):
            _ops.set_deterministic_weight_gradients(mode)
            m.zero_grad(set_to_none=True)
            (nerf.run_network(m, pts, rays, 1 << 20, ex, ed) * g_up).sum().backward()
            grads.setdefault(mode, []).append(torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone())
        assert torch.equal(grads[True][0], grads[True][1])
        scale = float(grads[True][0].abs().max())
        assert float((grads[True][0] - grads[False][0]).abs().max()) <= 1e-5 * scale
    finally:
        _ops.set_deterministic_weight_gradients(True)
        nerf.set_precision("fp32")


def test_flat_adam_against_torch_adam(dev):
    """nerf.FlatAdam (dn_adam_step: one launch over flat parameter / gradient / moment buffers, the step count and the reference's
    learning-rate schedule inside the kernel) against torch.optim.Adam in float64 on the same gradients for 25 steps
    (reference: train_dexnerf_rgb.py:146-148, 280-289): parameters and both moments within 2e-6 relative; the gradients are
    cleared by the step; torch.optim.Adam's state_dict format round-trips in both directions."""
    import nerf
    from nerf import parallel
    torch.manual_seed(3)
    mkw = dict(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True)
    nets = [nerf.models.FlexibleNeRFModel(**mkw).to(dev) for _ in range(2)]
    ref_params = [p.detach().double().cpu().clone().requires_grad_(True) for m in nets for p in m.parameters()]
    bucket = parallel.FlatGradBucket(nets)
    lr0, factor, steps_decay = 5e-3, 0.1, 40.0
    opt = nerf.FlatAdam(bucket, lr=lr0, lr_decay_factor=factor, lr_decay_steps=steps_decay, zero_grads=True)
    for p, q in zip(bucket.params, ref_params):
        assert p.data_ptr() >= opt.flat_params.data_ptr() and torch.equal(p.detach().cpu().double(), q.detach())   # flattened in place
    ref = torch.optim.Adam(ref_params, lr=lr0)
    t32_params = [q.detach().float().to(dev).requires_grad_(True) for q in ref_params]       # torch's own fp32 fused Adam: the yardstick
    t32 = torch.optim.Adam(t32_params, lr=lr0, fused=True)
    gen = torch.Generator().manual_seed(5)
    for it in range(25):
        for o in (ref, t32):
            for group in o.param_groups:
                group["lr"] = lr0 * factor ** (it / steps_decay)
        for p, q, t in zip(bucket.params, ref_params, t32_params):
            g = torch.randn(q.shape, generator=gen, dtype=torch.float64) * (10.0 ** float(torch.randint(-6, 1, (1,), generator=gen)))
            q.grad = g.clone()
            p.grad.copy_(g.float())
            t.grad = g.float().to(dev)
        opt.step()
        ref.step()
        t32.step()
        assert float(bucket.flat.abs().max()) == 0.0                       # zero_grads: the bucket is clean for the next iteration
    assert float(opt.step_state[0]) == 25.0 and abs(opt.last_lr() - lr0 * factor ** (24 / steps_decay)) < 1e-9

    def rel(a, b):
        return float((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30))
    worst = [0.0, 0.0]
    for p, q, t in zip(bucket.params, ref_params, t32_params):
        st, rt, tt = opt.state[p], ref.state[q], t32.state[t]
        for mine, theirs, torch32 in ((p.detach(), q.detach(), t.detach()), (st["exp_avg"], rt["exp_avg"], tt["exp_avg"]),
                                      (st["exp_avg_sq"], rt["exp_avg_sq"], tt["exp_avg_sq"])):
            e_mine, e_t32 = rel(mine, theirs), rel(torch32, theirs)
            worst = [max(worst[0], e_mine), max(worst[1], e_t32)]
            assert e_mine < max(1.5 * e_t32, 1e-6), (e_mine, e_t32)      # as close to float64 Adam as torch's fp32 kernel is
    _record_measurement("flat_adam_rel_err_vs_f64", dict(mine=worst[0], torch_fp32_fused=worst[1]))
    # checkpoints move between the two optimizers - through a FILE, as a resumed run gets them (torch.save / torch.load keep storage
    # sharing: a state dict whose parameters share one step tensor would advance it once per parameter per torch step)
    import io
    sd = opt.state_dict()
    assert sd["param_groups"][0]["lr"] == pytest.approx(lr0 * factor ** (24 / steps_decay), rel=1e-12)   # the decayed value, as the reference saves it (train_dexnerf_rgb.py:284-289, :449)
    buf = io.BytesIO()
    torch.save(sd, buf)
    buf.seek(0)
    sd = torch.load(buf, map_location=dev)
    t_params = [p.detach().clone().requires_grad_(True) for p in bucket.params]
    t_opt = torch.optim.Adam(t_params, lr=lr0)
    t_opt.load_state_dict(sd)
    assert float(t_opt.state[t_params[0]]["step"]) == 25.0
    for t in t_params:
        t.grad = torch.ones_like(t)
    t_opt.step()
    assert all(float(t_opt.state[t]["step"]) == 26.0 for t in t_params)          # ONE step for every parameter
    assert float(opt.step_state[0]) == 25.0                                       # and the live optimizer's state was not aliased
    # torch.optim.Adam with a per-iteration decayed lr (the reference's loop) -> FlatAdam with the in-kernel schedule: the stored lr is
    # already decayed and must not be decayed again
    k = 7
    t2_params = [p.detach().clone().requires_grad_(True) for p in bucket.params]
    t2 = torch.optim.Adam(t2_params, lr=lr0)
    for it in range(k):
        for t in t2_params:
            t.grad = torch.full_like(t, 1e-3)
        t2.step()
        for group in t2.param_groups:
            group["lr"] = lr0 * factor ** (it / steps_decay)
    nets2 = [nerf.models.FlexibleNeRFModel(**mkw).to(dev) for _ in range(2)]
    b2 = parallel.FlatGradBucket(nets2)
    opt2 = nerf.FlatAdam(b2, lr=lr0, lr_decay_factor=factor, lr_decay_steps=steps_decay)
    opt2.load_state_dict(t2.state_dict())
    assert float(opt2.step_state[0]) == float(k)
    for p in b2.params:
        p.grad.fill_(1e-3)
    opt2.step()
    assert opt2.last_lr() == pytest.approx(lr0 * factor ** (k / steps_decay), rel=1e-6)
    # moments survive the torch -> flat direction
    opt3 = nerf.FlatAdam(parallel.FlatGradBucket([nerf.models.FlexibleNeRFModel(**mkw).to(dev) for _ in range(2)]), lr=lr0)
    opt3.load_state_dict(opt.state_dict())      # (a fresh one: torch's load_state_dict adopted `sd`'s tensors and t_opt.step() advanced them)
    assert float(opt3.step_state[0]) == 25.0 and torch.equal(opt3.exp_avg, opt.exp_avg) and torch.equal(opt3.exp_avg_sq, opt.exp_avg_sq)


# ---- the draws of a training iteration made on the device (csrc/dn_rng.h, nerf.FusedTrainStep) ---------------------------------
def test_device_pixel_draw_is_a_permutation_with_uniform_marginals(dev):
    """dn_select_rays_draw: the pixels of an iteration are a keyed permutation of the H W pixels, so ANY prefix is a draw without
    replacement (reference: np.random.choice(H W, n, replace=False), train_dexnerf_rgb.py:229-236).  Checked: a full-length draw
    hits every pixel exactly once (several image sizes, incl. non-powers of two); consecutive iterations give different draws, the
    same (seed, iteration) the same one; over 4,000 iterations of 256-ray draws on a 37 x 29 image every pixel is chosen about
    equally often (chi-square over the 1,073 pixels within 5 sigma of its mean) and no draw repeats a pixel; the ray rows and
    targets equal dn_select_rays_indirect's on the drawn pixels."""
    from nerf import _ops
    cams = torch.zeros(2, 16, device=dev)
    cams[:, 0] = cams[:, 4] = cams[:, 8] = 1.0
    cams[:, 12], cams[:, 13], cams[:, 14] = 50.0, 3.0, 2.0
    view = torch.zeros((), dtype=torch.int32, device=dev)
    for (h, w) in ((4, 4), (37, 29), (64, 64), (100, 75)):
        st = _ops.new_rng_state(11, dev)
        _, _, pix = _ops.select_rays_draw(h, w, cams, view, 2.0, 6.0, st, h * w, want_pixels=True)
        assert torch.equal(torch.sort(pix)[0], torch.arange(h * w, device=dev))
        assert st.tolist()[2] == 0 and st.tolist()[3] == 0          # the draw publishes nxt as cur; only the loss kernel advances nxt
        again = _ops.select_rays_draw(h, w, cams, view, 2.0, 6.0, _ops.new_rng_state(11, dev), h * w, want_pixels=True)[2]
        assert torch.equal(pix, again)
        other = _ops.select_rays_draw(h, w, cams, view, 2.0, 6.0, _ops.new_rng_state(11, dev, first_iteration=1), h * w, want_pixels=True)[2]
        assert h * w < 32 or not torch.equal(pix, other)
    h, w, n = 37, 29, 256
    images = torch.rand(2, h, w, 3, device=dev)
    counts = torch.zeros(h * w, dtype=torch.int64, device=dev)
    iters = 4000
    st = _ops.new_rng_state(5, dev)
    loss3 = torch.zeros(3, device=dev)
    dummy = torch.zeros(n, 3, device=dev)
    for it in range(iters):
        rays, target, pix = _ops.select_rays_draw(h, w, cams, view, 2.0, 6.0, st, n, images, want_pixels=True)
        if it < 3:
            assert len(torch.unique(pix)) == n
            ref_rays, ref_target = _ops.select_rays_indirect(h, w, cams, view, 2.0, 6.0, pix, images)
            assert torch.equal(rays, ref_rays) and torch.equal(target, ref_target)
        counts += torch.bincount(pix, minlength=h * w)
        _ops.mse2_loss(dummy, dummy, dummy, rng_state=st)          # what advances the iteration counter in a training loop
    assert st.tolist()[3] == iters and int(counts.sum()) == iters * n
    expect = iters * n / (h * w)
    # sampling without replacement: variance of a pixel's count = iters * p (1 - p), p = n / (H W)
    p = n / (h * w)
    chi2 = float(((counts.double() - expect) ** 2).sum() / (iters * p * (1 - p)))
    dof = h * w - 1
    assert abs(chi2 - dof) < 5.0 * (2 * dof) ** 0.5, (chi2, dof)
    assert int(counts.min()) > 0.8 * expect and int(counts.max()) < 1.2 * expect
    # view = None: the training view is drawn in the kernel as well (np.random.choice(i_train), train_dexnerf_rgb.py:223): one
    # view per iteration (all rays of a draw come from it), every view about equally often, rows equal to the explicit-view call
    n_views = 5
    cams5 = cams[:1].repeat(n_views, 1).contiguous()
    cams5[:, 9] = torch.arange(n_views, device=dev, dtype=torch.float32)          # the camera position tells the views apart
    imgs5 = torch.rand(n_views, h, w, 3, device=dev)
    st = _ops.new_rng_state(9, dev)
    seen = torch.zeros(n_views, dtype=torch.int64)
    for it in range(1000):
        rays, target, pix = _ops.select_rays_draw(h, w, cams5, None, 2.0, 6.0, st, 64, imgs5, want_pixels=True)
        v = rays[:, 0]
        assert float(v.min()) == float(v.max())
        vi = int(v[0])
        seen[vi] += 1
        if it < 5:
            ref_rays, ref_target = _ops.select_rays_indirect(h, w, cams5, torch.tensor(vi, dtype=torch.int32, device=dev), 2.0, 6.0, pix, imgs5)
            assert torch.equal(rays, ref_rays) and torch.equal(target, ref_target)
        _ops.mse2_loss(dummy[:64], dummy[:64], dummy[:64], rng_state=st)
    assert int(seen.min()) > 150 and int(seen.max()) < 250, seen      # 200 expected, sigma 12.6


def test_in_kernel_uniforms_and_normals(dev):
    """The jitter / resampling uniforms and the density-noise normals the training kernels draw themselves (Philox-4x32-10 + Box-
    Muller, csrc/dn_rng.h; reference: torch.rand / torch.randn, nerf/train_utils.py:126-133, volume_rendering_utils.py:32-38):
    moments, range, independence across streams and iterations, reproducibility."""
    from nerf import _ops
    st = _ops.new_rng_state(123, dev)
    n = 1 << 20
    u = _ops.rng_fill(st, 0, (n,)).double()
    assert float(u.min()) >= 0.0 and float(u.max()) < 1.0
    assert abs(float(u.mean()) - 0.5) < 4 * (1 / 12 / n) ** 0.5 and abs(float(u.var()) - 1 / 12) < 1e-3
    hist = torch.histc(u.float(), bins=64, min=0.0, max=1.0).double()
    chi2 = float(((hist - n / 64) ** 2 / (n / 64)).sum())
    assert abs(chi2 - 63) < 5 * (2 * 63) ** 0.5, chi2
    z = _ops.rng_fill(st, 1, (n,), normal=True).double()
    assert bool(torch.isfinite(z).all())
    assert abs(float(z.mean())) < 4 / n ** 0.5 and abs(float(z.var()) - 1.0) < 5e-3
    assert abs(float((z ** 3).mean())) < 2e-2 and abs(float((z ** 4).mean()) - 3.0) < 5e-2
    assert abs(float((z.abs() > 3).double().mean()) - 0.0027) < 5e-4
    u2 = _ops.rng_fill(st, 2, (n,)).double()
    assert abs(float(((u - 0.5) * (u2 - 0.5)).mean())) < 4 / 12 / n ** 0.5          # streams are uncorrelated
    assert abs(float(((u[:-1] - 0.5) * (u[1:] - 0.5)).mean())) < 4 / 12 / n ** 0.5   # and so are neighbours
    assert torch.equal(_ops.rng_fill(_ops.new_rng_state(123, dev), 0, (n,)).double(), u)
    u_next = _ops.rng_fill(_ops.new_rng_state(123, dev, first_iteration=1), 0, (n,)).double()
    assert not torch.equal(u_next, u) and abs(float(((u - 0.5) * (u_next - 0.5)).mean())) < 4 / 12 / n ** 0.5


@pytest.mark.parametrize("precision,luminance", [("fp32", False), ("bf16-s8", False), ("fp32", True)])
def test_fused_train_step_equals_the_autograd_path_on_the_same_draws(dev, precision, luminance, monkeypatch):
    """nerf.FusedTrainStep (device pixel draw, in-kernel jitter / u / density noise, dn_mse2_loss, no autograd graph) against
    predict_and_render_radiance under autograd + torch's mse_loss fed the SAME pixels and the SAME draws (read back with dn_rng_fill):
    the six maps bit for bit, the loss and both MSEs to fp32 rounding, the parameter gradients of both networks to the reordering
    noise of the weight-gradient kernel's atomics; and the iteration counter advances exactly once per step."""
    import nerf
    from nerf import _ops, parallel, synthetic as syn
    h, w, n = 40, 52, 512
    mkw = dict(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True)
    cfg = make_cfg(dict(num_coarse=64, num_fine=64, near=2.0, far=6.0, perturb=True, noise_std=0.2, white_background=True), chunksize=4096)
    ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
    poses = [torch.from_numpy(syn.scene_pose(k)) for k in (3, 9)]
    kmat = torch.from_numpy(syn.intrinsic(h, w))
    images = torch.rand(2, h, w, 3, device=dev)
    nerf.set_precision(precision)
    try:
        def models():
            return make_models(mkw, syn.synth_state_dict(21, sigma_bias=-1.0, **mkw), syn.synth_state_dict(22, sigma_bias=-1.0, **mkw), dev)
        sel = nerf.MultiViewRaySelector(h, w, poses, [kmat, kmat], 2.0, 6.0, images=images, device=dev)
        sel.view.fill_(1)
        mc, mf = models()
        bucket = parallel.FlatGradBucket([mc, mf])
        step = nerf.FusedTrainStep(mc, mf, sel, cfg, bucket, ex, ed, n, seed=77, luminance=luminance, first_iteration=5)
        # what the kernels will draw in iteration 5
        peek = _ops.new_rng_state(77, dev, 5)
        rays_ref, target_ref, pix = _ops.select_rays_draw(h, w, sel.cams, sel.view, 2.0, 6.0, peek, n, images, want_pixels=True)
        draws = [_ops.rng_fill(peek, 0, (n, 64)), _ops.rng_fill(peek, 1, (n, 64), normal=True), _ops.rng_fill(peek, 2, (n, 64)),
                 _ops.rng_fill(peek, 3, (n, 128), normal=True)]
        loss3 = step.forward_backward()
        assert step.rng_state.tolist()[2:] == [5, 6]
        grads_fused = [p.grad.detach().clone() for p in bucket.params]
        maps_fused = step._keep[2]
        # the autograd path on the same rays and draws
        mc2, mf2 = models()
        q_rand, q_randn = [draws[0], draws[2]], [draws[1], draws[3]]
        monkeypatch.setattr(torch, "rand", lambda *a, **k: q_rand.pop(0))
        monkeypatch.setattr(torch, "randn", lambda *a, **k: q_randn.pop(0))
        out = nerf.predict_and_render_radiance(rays_ref, mc2, mf2, cfg, mode="train", encode_position_fn=ex, encode_direction_fn=ed, m_thres_cand=None)
        monkeypatch.undo()
        assert not q_rand and not q_randn
        for a, b in zip(maps_fused[:6], out[:6]):
            assert torch.equal(a, b.detach())

        def head(t):
            return (0.299 * t[..., 0] + 0.587 * t[..., 1] + 0.114 * t[..., 2]) if luminance else t
        mse_c, mse_f = nerf.img2mse(head(out[0]), head(target_ref)), nerf.img2mse(head(out[3]), head(target_ref))
        (mse_c + mse_f).backward()
        got = loss3.tolist()
        assert abs(got[1] - mse_c.item()) < 1e-5 * abs(mse_c.item()) and abs(got[2] - mse_f.item()) < 1e-5 * abs(mse_f.item())
        assert abs(got[0] - (mse_c + mse_f).item()) < 1e-5 * abs((mse_c + mse_f).item())
        tol = 1e-4 if precision == "fp32" else 2e-3
        for g, p in zip(grads_fused, list(mc2.parameters()) + list(mf2.parameters())):
            assert rel_err(C(g), C(p.grad)) < tol, (tuple(g.shape), rel_err(C(g), C(p.grad)))
        # a second step: new pixels, new draws, counter 6 -> 7
        step.forward_backward()
        assert step.rng_state.tolist()[2:] == [6, 7]
    finally:
        nerf.set_precision("fp32")


# ---- 8-bit saved tensors (DN_PREC_BF16_S8, nerf.set_precision("bf16-s8")) -----------------------------------------------------
@pytest.mark.parametrize("depth,width,viewdirs,skip", [(8, 256, True, 4), (4, 128, True, 4), (8, 256, False, 3), (5, 128, False, 2), (3, 256, True, 2)])
def test_s8_training_kernels_in_the_48_point_geometry(dev, depth, width, viewdirs, skip):
    """DN_PREC_BF16_S8 = the training forward and the backward-data chain on v_mfma_f32_16x16x32_bf16, 48 points per wave (fixed-shape
    instances: D8/W256 and 4x128 with view directions; run-time-shape ones: the rest), saved tensors at 8 bits in the s8-48 layout.
      * the training forward's radiance field equals the bf16 48-point INFERENCE kernel's bit for bit;
      * every saved activation (unpacked with dn_mlp_unpack) is the bf16 32-point training kernel's saved activation rounded to
        e4m3 - same products, another fp32 summation grouping, so compared to 3 mantissa bits: |a8 - a16| <= 0.07 |a16| + 2^-8
        on >= 99.9 % of the elements - and both encodings likewise;
      * every saved layer gradient is the 32-point bf16 backward chain's (same upstream gradient; the masks of each geometry's
        own forward) to e5m2's 2 mantissa bits: |g8 - g16| <= 0.14 |g16| + 1e-9 on >= 99 %, cosine >= 0.995 per stage;
      * the fp8-MFMA weight-gradient kernel on the 8-bit buffers against the bf16 kernel on the bf16 ones: cosine > 0.99 per
        tensor.  Ragged point count (37 x 53 points: partial 384-point tile, partial 32-point record, partial 16-point group)."""
    import nerf
    from nerf import _hip, _ops, _train
    nerf.set_precision("bf16-s16")
    try:
        torch.manual_seed(3)
        m = nerf.models.FlexibleNeRFModel(num_layers=depth, hidden_size=width, skip_connect_every=skip, num_encoding_fn_xyz=10,
                                          num_encoding_fn_dir=4, use_viewdirs=viewdirs).to(dev)
        pk = m.packed()
        weights = [x.weight for x in m.linear_modules()]
        _ops.pack_backward(pk, weights, _hip.PREC_BF16)
        _ops.pack_backward(pk, weights, _hip.PREC_BF16_S8)
        assert _ops.s8_supported(pk)
        n_rays, s = 37, 53
        n = n_rays * s
        pts = torch.rand(n, 3, device=dev) * 2 - 1
        vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1) if viewdirs else None
        g_out = torch.randn(n, 4, device=dev) * 1e-4
        out, act, masks = _ops.run_network_train(pk, pts, vd, s, prec=_hip.PREC_BF16)
        out8, act8, masks8 = _ops.run_network_train(pk, pts, vd, s, prec=_hip.PREC_BF16_S8)
        with torch.no_grad():
            inference = _ops.run_network_pts(pk, pts, vd, s)
        assert torch.equal(out8, inference)
        assert act8.numel() < 0.62 * act.numel()
        grads = _ops.mlp_backward_data(pk, g_out, masks, n, prec=_hip.PREC_BF16)
        grads8 = _ops.mlp_backward_data(pk, g_out, masks8, n, prec=_hip.PREC_BF16_S8)
        assert grads8.numel() < 0.62 * grads.numel()
        # stage by stage through the unpackers of the two layouts
        slots, gslots, kh = _train._slots(m, _hip.PREC_BF16)            # 32-point layout: slots in 16-deep pieces
        d, w = depth, width
        khu = w // 64                                                    # s8-48 layout: slots in 64-feature units per 16-point group
        s8 = {"xyz": 0, "dir": 1, "layer1": 1 + (1 if viewdirs else 0)}
        s8["trunk0"] = s8["layer1"] + khu
        s8["feat"] = s8["trunk0"] + (d - 1) * khu
        s8["dirout"] = s8["feat"] + (khu if viewdirs else 0)
        g8 = {"dirout": 0, "feat": (w // 128) if viewdirs else 0}
        g8["trunk0"] = g8["feat"] + (khu if viewdirs else 0)
        g8["layer1"] = g8["trunk0"] + (d - 1) * khu
        g8["out"] = g8["layer1"] + khu

        def rows16(which, buf, slot, width_, kind=0):
            return C(_ops.mlp_unpack(pk, which, buf, n, slot, width_, kind, torch.zeros((n, width_), dtype=torch.float32, device=dev))).astype(np.float64)

        def rows8(which, buf, slot, width_, kind=0, cols=None):
            o = torch.zeros((n, cols or width_), dtype=torch.float32, device=dev)
            return C(_ops.mlp_unpack(pk, which, buf, n, slot, width_, kind, o, prec=_hip.PREC_BF16_S8)).astype(np.float64)

        def close_act(a8, a16, what):
            ok = np.abs(a8 - a16) <= 0.07 * np.abs(a16) + 2.0 ** -8
            assert ok.mean() >= 0.999, (what, ok.mean(), np.abs(a8 - a16).max())

        def close_grad(a8, a16, what):
            ok = np.abs(a8 - a16) <= 0.14 * np.abs(a16) + 1e-9
            cos = float((a8 * a16).sum() / max(np.linalg.norm(a8) * np.linalg.norm(a16), 1e-300))
            assert ok.mean() >= 0.99 and cos >= 0.995, (what, ok.mean(), cos)

        close_act(rows8(0, act8, s8["xyz"], m.dim_xyz, 1), rows16(0, act, slots["xyz"], m.dim_xyz, 1), "xyz encoding")
        if viewdirs:
            close_act(rows8(0, act8, s8["dir"], m.dim_dir, 2), rows16(0, act, slots["dir"], m.dim_dir, 2), "view-direction encoding")
        close_act(rows8(0, act8, s8["layer1"], w), rows16(0, act, slots["layer1"], w), "layer1")
        close_grad(rows8(1, grads8, g8["layer1"], w), rows16(1, grads, gslots["layer1"], w), "d layer1")
        for i in range(d - 1):
            close_act(rows8(0, act8, s8["trunk0"] + i * khu, w), rows16(0, act, slots["trunk0"] + i * kh, w), f"layers_xyz[{i}]")
            close_grad(rows8(1, grads8, g8["trunk0"] + i * khu, w), rows16(1, grads, gslots["trunk0"] + i * kh, w), f"d layers_xyz[{i}]")
        if viewdirs:
            close_act(rows8(0, act8, s8["feat"], w), rows16(0, act, slots["feat"], w), "fc_feat")
            close_grad(rows8(1, grads8, g8["feat"], w), rows16(1, grads, gslots["feat"], w), "d fc_feat")
            # (the 64-feature units of a W = 128 net's 64-wide layers_dir.0 output: one unit)
            close_act(rows8(0, act8, s8["dirout"], max(w // 2, 64))[:, : w // 2], rows16(0, act, slots["dirout"], w // 2), "layers_dir.0")
            close_grad(rows8(1, grads8, g8["dirout"], max(w // 2, 64))[:, : w // 2], rows16(1, grads, gslots["dirout"], w // 2), "d layers_dir.0")
        custom = rows8(1, grads8, g8["out"], 8, 3, cols=8)
        want = C(g_out).astype(np.float64)
        if viewdirs:
            close_grad(custom[:, [0, 1, 2, 4]], want, "custom output-gradient unit")
        else:
            close_grad(custom[:, :4], want, "custom output-gradient unit")
        # the fp8-MFMA weight-gradient kernel on the 8-bit buffers against the bf16 kernel on the bf16 buffers
        shapes = [tuple(x.weight.shape) for x in m.linear_modules()]
        ref = _ops.mlp_weight_grad_all(pk, act, grads, n, shapes, prec=_hip.PREC_BF16)
        got = _ops.mlp_weight_grad_all(pk, act8, grads8, n, shapes, prec=_hip.PREC_BF16_S8)
        for (w16, b16), (w8, b8), mod in zip(ref, got, m.linear_modules()):
            for a, b in ((w8, w16), (b8, b16)):
                a, b = C(a).astype(np.float64).reshape(-1), C(b).astype(np.float64).reshape(-1)
                cos = float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))
                assert cos > 0.99, (tuple(mod.weight.shape), cos)
    finally:
        nerf.set_precision("fp32")


@pytest.mark.parametrize("depth,width", [(8, 256), (4, 128)])
def test_s8_training_kernels_on_two_point_groups_per_wave_equal_the_three_group_ones(dev, depth, width, monkeypatch):
    """Small training launches of the two fixed shapes run 256-point workgroup tiles (two 16-point groups per wave) instead of 384-point
    ones (csrc/mlp_geo48.h g48_train_groups: fewer idle compute units, no short last round).  Same per-point arithmetic, the same saved
    units at the same addresses (the layout is by group), other mask-word addresses (per wave tile) that forward and backward agree on:
    radiance field, saved activations, saved gradients and weight gradients must be IDENTICAL to the three-group instances' - forced
    either way with DEXNERF_G48_TRAIN_GROUPS - on ragged point counts (the default rule's own choice runs in every other 8-bit test:
    small launches take two groups, the 4096-ray steps three)."""
    import nerf
    from nerf import _hip, _ops
    nerf.set_precision("bf16")
    try:
        torch.manual_seed(11)
        m = nerf.models.FlexibleNeRFModel(num_layers=depth, hidden_size=width, skip_connect_every=4, num_encoding_fn_xyz=10,
                                          num_encoding_fn_dir=4, use_viewdirs=True).to(dev)
        pk = m.packed()
        S8 = _hip.PREC_BF16_S8
        _ops.pack_backward(pk, [x.weight for x in m.linear_modules()], S8)
        shapes = [tuple(x.weight.shape) for x in m.linear_modules()]
        for n_rays, s in ((37, 53), (1024, 64), (700, 129)):
            n = n_rays * s
            g = torch.Generator(device=dev).manual_seed(n)
            pts = torch.rand(n, 3, device=dev, generator=g) * 2 - 1
            vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev, generator=g), dim=-1)
            g_out = torch.randn(n, 4, device=dev, generator=g) * 1e-4
            res = {}
            for groups in ("2", "3"):
                monkeypatch.setenv("DEXNERF_G48_TRAIN_GROUPS", groups)
                out, act, masks = _ops.run_network_train(pk, pts, vd, s, prec=S8)
                grads = _ops.mlp_backward_data(pk, g_out, masks, n, prec=S8)
                wg = _ops.mlp_weight_grad_all(pk, act, grads, n, shapes, prec=S8)
                rec = 2 * 1024 * ((n + 31) // 32)    # bytes of one unit slot over the records that hold real points
                res[groups] = (out, act, grads, wg, rec)
            monkeypatch.delenv("DEXNERF_G48_TRAIN_GROUPS")
            (o2, a2, g2, w2, _), (o3, a3, g3, w3, _) = res["2"], res["3"]
            assert torch.equal(o2, o3), (n_rays, s)
            # saved units: identical wherever a real point lives (the tilings pad different tails with copies of the last point)
            for slot in range(4):
                rows2 = _ops.mlp_unpack(pk, 0, a2, n, slot, 64, 0, torch.zeros((n, 64), device=dev), prec=S8)
                rows3 = _ops.mlp_unpack(pk, 0, a3, n, slot, 64, 0, torch.zeros((n, 64), device=dev), prec=S8)
                assert torch.equal(rows2, rows3), ("activation unit", slot, n_rays, s)
                rows2 = _ops.mlp_unpack(pk, 1, g2, n, slot, 64, 0, torch.zeros((n, 64), device=dev), prec=S8)
                rows3 = _ops.mlp_unpack(pk, 1, g3, n, slot, 64, 0, torch.zeros((n, 64), device=dev), prec=S8)
                assert torch.equal(rows2, rows3), ("gradient unit", slot, n_rays, s)
            for (dw2, db2), (dw3, db3), shp in zip(w2, w3, shapes):
                assert torch.equal(dw2, dw3) and torch.equal(db2, db3), (shp, n_rays, s)
    finally:
        nerf.set_precision("fp32")


def test_s8_saturates_instead_of_overflowing(dev):
    """The 8-bit conversions do not saturate in the default mode (an e4m3 overflow converts to NaN, an e5m2 one to infinity: measured,
    scripts/micro/cvt_scale_probe.hip); the training kernels run them with MODE.FP16_OVFL set, under which they do
    (scripts/micro/cvt_sat_probe.hip).  A network driven to activations beyond 448 and gradients beyond 57344 / scale must still give
    finite weight gradients."""
    import nerf
    from nerf import _hip, _ops
    nerf.set_precision("bf16-s16")
    try:
        torch.manual_seed(5)
        m = nerf.models.FlexibleNeRFModel(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10,
                                          num_encoding_fn_dir=4, use_viewdirs=True).to(dev)
        with torch.no_grad():
            m.layers_xyz[0].weight.mul_(3.0e3)        # activations of the order of 1e3 - 1e4 (> 448)
        pk = m.packed()
        _ops.pack_backward(pk, [x.weight for x in m.linear_modules()], _hip.PREC_BF16_S8)
        n_rays, s = 24, 40
        n = n_rays * s
        pts = torch.rand(n, 3, device=dev) * 2 - 1
        vd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1)
        out8, act8, masks8 = _ops.run_network_train(pk, pts, vd, s, prec=_hip.PREC_BF16_S8)
        big = _ops.mlp_unpack(pk, 0, act8, n, 2 + 128 // 64, 128, 0, torch.zeros((n, 128), dtype=torch.float32, device=dev), prec=_hip.PREC_BF16_S8)
        assert bool(torch.isfinite(big).all()) and float(big.max()) == 448.0      # layers_xyz[0]'s output: saturated, not NaN
        g_out = torch.randn(n, 4, device=dev) * 10.0                               # x 65536 >> 57344
        grads8 = _ops.mlp_backward_data(pk, g_out, masks8, n, prec=_hip.PREC_BF16_S8)
        shapes = [tuple(x.weight.shape) for x in m.linear_modules()]
        for d_w, d_b in _ops.mlp_weight_grad_all(pk, act8, grads8, n, shapes, prec=_hip.PREC_BF16_S8):
            assert bool(torch.isfinite(d_w).all()) and bool(torch.isfinite(d_b).all())
    finally:
        nerf.set_precision("fp32")


def test_s8_training_step_gradients(golden, dev):
    """One whole training step (both D8/W256 nets, 64 + 128 samples, perturbed) in the three training modes: the 'bf16-s8' mode's
    forward agrees with the bf16 mode's to the accumulation order (loss within 2e-3), and its parameter gradients are as close to the fp32 mode's as the bf16
    mode's are (cosine per tensor; the gate: within 5e-3 of the bf16 mode's cosine against fp32 and >= 0.93 - layer1 of a bf16
    step sits at 0.955 on these rays - and >= 0.998 against the bf16 mode's own gradient)."""
    import nerf
    from nerf import synthetic as syn
    mkw = dict(num_layers=8, hidden_size=256, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True)
    sds = [{k: torch.from_numpy(v) for k, v in syn.synth_state_dict(seed, sigma_bias=-1.0, **mkw).items()} for seed in (41, 43)]
    cfg = make_cfg(dict(num_coarse=64, num_fine=128, near=2.0, far=6.0, perturb=True, noise_std=0.0))
    ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
    n = 1536
    gen = torch.Generator().manual_seed(2)
    ro = torch.tensor([0.0, 0.0, 4.0]).expand(n, 3) + 0.01 * torch.randn(n, 3, generator=gen)
    rd = torch.nn.functional.normalize(torch.tensor([0.0, 0.0, -1.0]) + 0.3 * torch.randn(n, 3, generator=gen), dim=-1)
    rays = torch.cat([ro, rd, torch.full((n, 1), 2.0), torch.full((n, 1), 6.0), rd], -1).to(dev)
    target = torch.rand(n, 3, generator=gen).to(dev)
    res = {}
    try:
        for prec in ("fp32", "bf16-s16", "bf16-s8"):
            nerf.set_precision(prec)
            assert nerf.get_precision() == {"bf16-s8": "bf16"}.get(prec, prec)      # ('bf16-s8' is the older name of 'bf16')
            models = []
            for sd in sds:
                m = nerf.models.FlexibleNeRFModel(**mkw); m.load_state_dict(sd); models.append(m.to(dev))
            torch.manual_seed(9)
            out = nerf.predict_and_render_radiance(rays, models[0], models[1], cfg, mode="train", encode_position_fn=ex,
                                                   encode_direction_fn=ed, m_thres_cand=M_THRES)
            loss = nerf.img2mse(out[0], target) + nerf.img2mse(out[3], target)
            loss.backward()
            res[prec] = (float(loss.detach()), [C(p.grad).astype(np.float64).reshape(-1) for m in models for p in m.parameters()])
    finally:
        nerf.set_precision("fp32")
    # (the 8-bit mode's forward is the 48-point kernel, the bf16 mode's the 32-point one: same products, another fp32 grouping)
    assert abs(res["bf16-s8"][0] - res["bf16-s16"][0]) < 2e-3 * abs(res["bf16-s16"][0])
    assert abs(res["bf16-s16"][0] - res["fp32"][0]) < 5e-3 * abs(res["fp32"][0])

    def cos(a, b):
        return float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))
    for g8, g16, g32 in zip(res["bf16-s8"][1], res["bf16-s16"][1], res["fp32"][1]):
        assert cos(g8, g32) > 0.93 and cos(g8, g32) > cos(g16, g32) - 5e-3, (cos(g8, g32), cos(g16, g32))
        assert cos(g8, g16) > 0.998, cos(g8, g16)
    whole = [np.concatenate(res[p][1]) for p in ("fp32", "bf16-s16", "bf16-s8")]
    assert cos(whole[2], whole[0]) > 0.99 and cos(whole[2], whole[0]) > cos(whole[1], whole[0]) - 1e-3, (cos(whole[2], whole[0]), cos(whole[1], whole[0]))


def test_s8_statistics_and_per_launch_scale_under_a_sum_reduced_loss(dev):
    """e5m2's range under the fixed scale 2^16 is made for a mean-reduced MSE.  With the SUM-reduced loss (per-point gradients x the
    number of ray channels) the saved layer gradients clip at 57344 / 2^16: the statistics the weight-gradient kernel counts must
    say so (this is what train_dexnerf.py warns on), and nerf.set_s8_grad_scale(0) - every backward-data launch takes its scale
    from its largest upstream gradient - must bring the whole-gradient cosine against the bf16 mode back to >= 0.999."""
    import nerf
    from nerf import synthetic as syn
    mkw = dict(num_layers=8, hidden_size=256, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True)
    sds = [{k: torch.from_numpy(v) for k, v in syn.synth_state_dict(seed, sigma_bias=-1.0, **mkw).items()} for seed in (41, 43)]
    cfg = make_cfg(dict(num_coarse=64, num_fine=128, near=2.0, far=6.0, perturb=True, noise_std=0.0))
    ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
    n = 1536
    gen = torch.Generator().manual_seed(2)
    ro = torch.tensor([0.0, 0.0, 4.0]).expand(n, 3) + 0.01 * torch.randn(n, 3, generator=gen)
    rd = torch.nn.functional.normalize(torch.tensor([0.0, 0.0, -1.0]) + 0.3 * torch.randn(n, 3, generator=gen), dim=-1)
    rays = torch.cat([ro, rd, torch.full((n, 1), 2.0), torch.full((n, 1), 6.0), rd], -1).to(dev)
    target = torch.rand(n, 3, generator=gen).to(dev)

    def step(prec, reduction, scale=None):
        nerf.set_precision(prec)
        if scale is not None:
            nerf.set_s8_grad_scale(scale)
        models = []
        for sd in sds:
            m = nerf.models.FlexibleNeRFModel(**mkw); m.load_state_dict(sd); models.append(m.to(dev))
        torch.manual_seed(9)
        out = nerf.predict_and_render_radiance(rays, models[0], models[1], cfg, mode="train", encode_position_fn=ex,
                                               encode_direction_fn=ed, m_thres_cand=M_THRES)
        mse = torch.nn.functional.mse_loss
        (mse(out[0], target, reduction=reduction) + mse(out[3], target, reduction=reduction)).backward()
        grad = np.concatenate([C(p.grad).astype(np.float64).reshape(-1) for m in models for p in m.parameters()])
        return grad, (nerf.s8_grad_stats() if prec in ("bf16", "bf16-s8") else None)

    def cos(a, b):
        return float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))
    try:
        ref_sum, _ = step("bf16-s16", "sum")
        # the DEFAULT ('bf16', no scale set): 8-bit saved tensors with the scale chosen per launch - nothing clips under a sum-reduced loss
        g_default, st_default = step("bf16", "sum")
        assert st_default["saturated"] == 0.0 and all(0.0 < v < 65536.0 for v in st_default["scale"]), st_default
        assert np.isfinite(g_default).all() and cos(g_default, ref_sum) >= 0.999, cos(g_default, ref_sum)
        _, st_mean = step("bf16-s8", "mean", 65536.0)
        assert st_mean["sampled"] > 10_000 and st_mean["saturated"] == 0.0 and st_mean["scale"] == [65536.0], st_mean
        g_fixed, st_fixed = step("bf16-s8", "sum", 65536.0)
        assert st_fixed["saturated"] > 1e-4 and st_fixed["scale"] == [65536.0], st_fixed     # the counter fires ...
        assert cos(g_fixed, ref_sum) < 0.999, cos(g_fixed, ref_sum)                        # ... and the clipping is real
        g_auto, st_auto = step("bf16-s8", "sum", 0.0)
        assert st_auto["saturated"] == 0.0 and all(0.0 < v < 65536.0 and np.log2(v) == round(np.log2(v)) for v in st_auto["scale"]), st_auto
        assert np.isfinite(g_auto).all() and cos(g_auto, ref_sum) >= 0.999, cos(g_auto, ref_sum)
        _record_measurement("s8_sum_loss", dict(saturated_fixed=st_fixed["saturated"], cos_fixed=cos(g_fixed, ref_sum),
                                                cos_auto=cos(g_auto, ref_sum), scale_auto_max=max(st_auto["scale"]), floor_auto=st_auto["floor"]))
    finally:
        nerf.set_s8_grad_scale(0.0)       # the default: per launch
        nerf.set_precision("fp32")


def test_gradient_scale_from_the_compositing_backward_equals_the_separate_reduction(dev, tmp_path, monkeypatch):
    """'bf16' training takes the e5m2 scale of every backward launch from its largest upstream gradient.  Inside dn_render_rays_backward
    that maximum comes from the compositing backward (one word per workgroup, reduced by the network backward: csrc/composite.hip,
    mlp_train48.hip) instead of a reduction launch of its own (absmax_kernel, what a stand-alone dn_mlp_backward_data runs;
    DEXNERF_S8_ABSMAX_KERNEL=1 forces it here too).  A maximum has no order: both ways must pick the same power of two, so two runs of
    the training driver - ragged ray count: the last workgroup of the compositing backward has idle waves - end in identical weights."""
    import sys
    sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd"))
    import nerf
    import train_dexnerf
    try:
        base = ["--size", "32", "--views", "6", "--num-random-rays", "509", "--layers", "4", "--width", "128", "--validate-every", "0",
                "--quiet", "--precision", "bf16", "--iters", "60", "--no-hip-graph"]
        paths = [os.path.join(str(tmp_path), f"scale_{k}.ckpt") for k in range(2)]
        scales = []
        for k, path in enumerate(paths):
            if k == 1:
                monkeypatch.setenv("DEXNERF_S8_ABSMAX_KERNEL", "1")
            train_dexnerf.main(base + ["--save", path])
            st = nerf.s8_grad_stats()
            assert st is not None and st["saturated"] == 0.0 and all(0.0 < v and np.log2(v) == round(np.log2(v)) for v in st["scale"]), st
            scales.append(st["scale"])
        monkeypatch.delenv("DEXNERF_S8_ABSMAX_KERNEL")
        assert scales[0] == scales[1], scales
        a, b = (torch.load(p, map_location="cpu") for p in paths)
        for key in ("model_coarse_state_dict", "model_fine_state_dict"):
            for name in a[key]:
                assert torch.equal(a[key][name], b[key][name]), (key, name)
    finally:
        nerf.set_precision("fp32")


def test_llff_capture_renders_through_the_ndc_branch(dev, tmp_path):
    """SURVEY 8f N2 + S2b end to end: a forward-facing capture on disk in the LLFF layout -> nerf.load_llff_data -> 4-argument
    get_ray_bundle (camera-to-world convention) -> run_one_iter_of_nerf with dataset.no_ndc = False (near 0, far 1: the LLFF
    configs of train_nerf_rgb.py) against the same chunk rendered from rows warped with the reference's elementwise formula
    (nerf_helpers.py:172-199 in torch ops; view directions from the UNwarped rays, train_utils.py:220-238)."""
    import nerf
    from PIL import Image
    from nerf import synthetic as syn
    root = str(tmp_path)
    os.makedirs(os.path.join(root, "images"))
    rng = np.random.default_rng(1)
    h, w, f = 20, 28, 30.0
    rows = []
    for k, (x, y) in enumerate([(x, y) for y in (-0.3, 0.0, 0.3) for x in (-0.4, 0.0, 0.4)]):
        # LLFF block columns: (down, right, back, position, hwf) for a camera at (x, y, 0) looking down -z
        block = np.stack([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0], [x, y, 0.0], [h, w, f]], axis=1)
        rows.append(np.concatenate([block.reshape(-1), [2.0, 8.0]]))
        Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)).save(os.path.join(root, "images", f"{k:02d}.png"))
    np.save(os.path.join(root, "poses_bounds.npy"), np.stack(rows))
    images, poses, bds, render_poses, i_test = nerf.load_llff_data(root, factor=1)
    assert images.shape == (9, h, w, 3) and i_test == 4 and render_poses.shape == (120, 3, 5)
    hh, ww, focal = (int(poses[0, 0, 4]), int(poses[0, 1, 4]), float(poses[0, 2, 4]))
    assert (hh, ww, focal) == (h, w, f)
    pose = torch.from_numpy(np.vstack([render_poses[7, :3, :4], [[0, 0, 0, 1]]]).astype(np.float32)).to(dev)
    ro, rd = nerf.get_ray_bundle(hh, ww, focal, pose)
    mkw = dict(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True)
    models = make_models(mkw, syn.synth_state_dict(5, sigma_bias=1.0, **mkw), syn.synth_state_dict(6, sigma_bias=1.0, **mkw), dev)
    cfg = make_cfg(dict(num_coarse=32, num_fine=48, near=0.0, far=1.0))
    cfg.dataset.no_ndc = False
    ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
    with torch.no_grad():
        out = nerf.run_one_iter_of_nerf(hh, ww, focal, models[0], models[1], ro, rd, cfg, mode="validation", encode_position_fn=ex,
                                        encode_direction_fn=ed, m_thres_cand=M_THRES)
        o, d = ro.reshape(-1, 3), rd.reshape(-1, 3)
        t = -(1.0 + o[:, 2]) / d[:, 2]
        o = o + t[:, None] * d
        sx, sy = -1.0 / (ww / (2.0 * focal)), -1.0 / (hh / (2.0 * focal))
        wo = torch.stack([sx * o[:, 0] / o[:, 2], sy * o[:, 1] / o[:, 2], 1.0 + 2.0 / o[:, 2]], -1)
        wd = torch.stack([sx * (d[:, 0] / d[:, 2] - o[:, 0] / o[:, 2]), sy * (d[:, 1] / d[:, 2] - o[:, 1] / o[:, 2]), -2.0 / o[:, 2]], -1)
        vd = torch.nn.functional.normalize(rd.reshape(-1, 3), dim=-1)
        rows = torch.cat([wo, wd, torch.zeros_like(wo[:, :1]), torch.ones_like(wo[:, :1]), vd], -1)
        ref = nerf.predict_and_render_radiance(rows, models[0], models[1], cfg, mode="validation", encode_position_fn=ex,
                                               encode_direction_fn=ed, m_thres_cand=M_THRES)
    assert out[3].shape == (hh, ww, 3) and out[4].shape == (hh, ww)
    for a, b in zip(out, ref):
        assert rel_err(C(a).reshape(-1), C(b).reshape(-1)) < TOL
    depth = C(out[4])
    assert depth.min() >= 0.0 and depth.max() <= 1.0 + 1e-6          # NDC depths live in [0, 1]
    assert 0.0 <= float(C(out[5]).min()) and float(C(out[5]).max()) <= 1.0 + 1e-5


def test_training_from_an_llff_capture(dev, tmp_path):
    """N2 + S2b + N1 end to end: a forward-facing capture of the built-in teacher scene written in the LLFF layout (poses_bounds.npy in
    LLFF's (down, right, back) axes + images/) -> nerf.load_llff_data -> train_dexnerf.py --llff: NDC rays through dn_ndc_rays inside
    the (HIP-graph-captured) iteration, validation through run_one_iter_of_nerf's NDC branch.  The student must learn (+6 dB)."""
    import nerf
    import train_dexnerf
    from PIL import Image
    from nerf import synthetic as syn
    root = str(tmp_path)
    os.makedirs(os.path.join(root, "images"))
    h = w = 48
    f = 60.0
    mkw = dict(num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True)
    teacher = make_models(mkw, syn.synth_state_dict(42, sigma_bias=-150.0, **mkw), syn.synth_state_dict(43, sigma_bias=-20.0, **mkw), dev)
    cfg = make_cfg(dict(num_coarse=64, num_fine=64, near=2.0, far=6.0))
    ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
    rows = []
    k = 0
    for y in (-0.5, 0.0, 0.5):
        for x in (-0.6, -0.2, 0.2, 0.6):
            c2w = np.eye(4, dtype=np.float32)
            c2w[:3, 3] = [x, y, 4.0]                                   # looking down -z at the scene around the origin
            ro, rd = nerf.get_ray_bundle(h, w, f, torch.from_numpy(c2w).to(dev))
            with torch.no_grad():
                out = nerf.run_one_iter_of_nerf(h, w, f, teacher[0], teacher[1], ro, rd, cfg, mode="validation", encode_position_fn=ex,
                                                encode_direction_fn=ed, m_thres_cand=M_THRES)
            img = (C(out[3]).clip(0, 1) * 255 + 0.5).astype(np.uint8)
            Image.fromarray(img).save(os.path.join(root, "images", f"{k:02d}.png"))
            block = np.stack([-c2w[:3, 1], c2w[:3, 0], c2w[:3, 2], c2w[:3, 3], np.array([h, w, f], dtype=np.float32)], axis=1)
            rows.append(np.concatenate([block.reshape(-1), [2.0, 6.0]]))
            k += 1
    np.save(os.path.join(root, "poses_bounds.npy"), np.stack(rows).astype(np.float64))
    try:
        res = train_dexnerf.main(["--llff", root, "--llff-factor", "1", "--llffhold", "6", "--iters", "300", "--num-random-rays", "512",
                                  "--layers", "4", "--width", "128", "--num-fine", "64", "--validate-every", "0", "--quiet",
                                  "--precision", "bf16"])
    finally:
        nerf.set_precision("fp32")
    first, last = res["history"][0], res["history"][-1]
    assert np.isfinite(last[1]) and last[2] - first[2] > 6.0, (first, last)
    assert res["val_psnr"] > 12.0 and "dex_best_threshold" not in res


def test_training_leaves_the_inference_stream_stale_and_a_render_refreshes_it(dev):
    """dn_mlp_pack_parts: the training entry points re-pack only the core stream after an optimizer step (they never run the
    48-point inference kernel); the first render afterwards must bring that kernel's own stream up to date - a bf16 render after
    training steps equals the render of a fresh model holding the same parameters, bit for bit."""
    import nerf
    nerf.set_precision("bf16")
    try:
        torch.manual_seed(21)
        mkw = dict(num_layers=8, hidden_size=256, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4, use_viewdirs=True)
        m = nerf.models.FlexibleNeRFModel(**mkw).to(dev)
        ex, ed = nerf.get_embedding_function(10), nerf.get_embedding_function(4)
        n, s = 64, 48
        pts = torch.randn(n, s, 3, device=dev)
        vd = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=-1)
        rays = torch.cat([torch.zeros(n, 8, device=dev), vd], -1)
        with torch.no_grad():
            before = nerf.run_network(m, pts, rays, 1 << 20, ex, ed)          # packs both streams
        pk = m.packed()
        assert pk.key48 == pk.key
        opt = torch.optim.Adam(m.parameters(), lr=1e-2, fused=True)
        for _ in range(2):
            out = nerf.run_network(m, pts, rays, 1 << 20, ex, ed)             # training path: core stream only
            opt.zero_grad(set_to_none=True)
            (out ** 2).mean().backward()
            opt.step()
        out = nerf.run_network(m, pts, rays, 1 << 20, ex, ed)                 # (a training forward on the updated parameters)
        pk = m.packed(train=True)
        assert pk.key48 != pk.key                                             # the inference stream is two optimizer steps old
        # a caller that kept the packed object and renders with it directly would get the OLD weights: refused, not served
        from nerf import _ops
        with pytest.raises(RuntimeError, match="48-point inference stream is older"):
            _ops.run_network_pts(pk, pts.reshape(-1, 3), vd, s)
        with torch.no_grad():
            after = nerf.run_network(m, pts, rays, 1 << 20, ex, ed)           # render: refreshes it
        assert m.packed().key48 == m.packed().key
        fresh = nerf.models.FlexibleNeRFModel(**mkw).to(dev)
        fresh.load_state_dict(m.state_dict())
        with torch.no_grad():
            ref = nerf.run_network(fresh, pts, rays, 1 << 20, ex, ed)
        assert torch.equal(after, ref)
        assert not torch.equal(after, before)
    finally:
        nerf.set_precision("fp32")
