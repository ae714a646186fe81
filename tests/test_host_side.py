"""CPU-only checks of the host side: the C-ABI library loads and exports what include/dexnerf_hip.h declares,
argument validation happens before any GPU work, and the drop-in package mirrors the reference's names,
state_dict keys and tolerant call surface.  No compute calls into the library are made here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import REPO
from golden_cases import lego_weights


@pytest.fixture(scope="module")
def hiplib():
    from nerf import _hip
    if not _hip.available():
        import __graft_entry__ as ge
        ge.build()
    return _hip.lib()


def test_library_exports_every_declared_symbol(hiplib):
    from nerf import _hip
    header = open(os.path.join(REPO, "include", "dexnerf_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(dn_[a-z_0-9]+)\s*\(", header)))
    assert set(declared) == set(_hip.EXPORTS)
    raw = ctypes.CDLL(_hip.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert hiplib.dn_abi_version() == 2


def test_argument_validation_needs_no_gpu(hiplib):
    from nerf import _hip
    d = _hip.MlpDesc(num_layers=8, hidden_size=256, skip_connect_every=4, num_encoding_fn_xyz=10,
                     num_encoding_fn_dir=4, include_input_xyz=1, include_input_dir=1, use_viewdirs=1,
                     log_sampling_xyz=1, log_sampling_dir=1)
    # D8/W256: bias tiles (8 + 7*8 + 9 + 4 + 1) = 78 -> 10 KiB; pieces 1184 (bf16) / 2368 (fp32) KiB
    # bf16 / fp16 nets carry a second stream for the 48-points-per-wave inference kernel: 2464 bias rows -> 10 KiB,
    # 1.5 KiB of encoding tables, 1184 pieces of 16 x 32
    null = None
    assert hiplib.dn_mlp_weight_grad_pair(ctypes.byref(d), _hip.PREC_BF16, null, null, 4, null, null, null, null, 4, null, null, null) == -1000
    assert hiplib.dn_mlp_packed_bytes(ctypes.byref(d), _hip.PREC_BF16) == (10 + 1184) * 1024 + (10 + 1184) * 1024 + 1536
    assert hiplib.dn_mlp_packed_bytes(ctypes.byref(d), _hip.PREC_F16) == hiplib.dn_mlp_packed_bytes(ctypes.byref(d), _hip.PREC_BF16)
    assert hiplib.dn_mlp_packed_bytes(ctypes.byref(d), _hip.PREC_F32) == 10 * 1024 + 2368 * 1024
    d.hidden_size = 192
    assert hiplib.dn_mlp_packed_bytes(ctypes.byref(d), _hip.PREC_BF16) == 0
    assert b"outside the fused HIP kernel" in hiplib.dn_last_error()
    assert hiplib.dn_coarse_depths(None, 11, 4, 64, 0, None, None, None) == -1000
    assert hiplib.dn_sample_pdf(None, None, None, 1, 64, 8, None, None, None) == -1000
    assert hiplib.dn_render_workspace_bytes(1000, 64, 128) >= 1000 * (64 * 6 + 192 * 5) * 4
    # the training pair: its workspace adds the gradient scratch of one network; NULL / misaligned
    # arguments and an unknown `nets` mask are refused before anything is enqueued
    ws_train = hiplib.dn_render_train_workspace_bytes(1000, 64, 128)
    assert ws_train >= hiplib.dn_render_workspace_bytes(1000, 64, 128) + 1000 * 192 * 4 * 4
    assert hiplib.dn_render_train_workspace_bytes(-1, 64, 128) == 0
    null = None
    assert hiplib.dn_render_rays_train(ctypes.byref(d), null, ctypes.byref(d), null, 1, null, 11, 4, 64, 128, 0, 0.0, 0, None, 0,
                                       null, null, null, null, null, null, null, null, null, null, null, null, null, null,
                                       null, null, null, 0, null) == -1000
    assert b"dn_render_rays_train" in hiplib.dn_last_error()
    assert hiplib.dn_render_rays_backward(ctypes.byref(d), null, ctypes.byref(d), null, 1, null, 11, 4, 64, 128, 0.0, 0, null, null,
                                          null, null, null, null, null, null, null, null, null, null, null, null, null,
                                          None, None, None, None, 3, null, null) == -1000
    assert hiplib.dn_render_rays_train(ctypes.byref(d), null, ctypes.byref(d), null, 1, null, 11, 0, 64, 128, 0, 0.0, 0, None, 0,
                                       null, null, null, null, null, null, null, null, null, null, null, null, null, null,
                                       null, null, null, 0, null) == 0     # zero rays: nothing to do
    # the device-side draws of a training iteration: argument checks (no launch without a GPU)
    assert hiplib.dn_select_rays_draw(4, 4, null, null, 0, 2.0, 6.0, null, 4, null, 0, null, null, null, null) == -1000
    assert hiplib.dn_mse2_loss(null, null, null, 4, 0, null, null, null, null, null) == -1000
    assert hiplib.dn_rng_fill(null, 0, 4, 0, null, null) == -1000 and hiplib.dn_rng_fill(null, 0, 0, 0, null, null) == 0
    assert hiplib.dn_pack_ray_rows(null, null, null, 2.0, 6.0, 4, null, null) == -1000 and hiplib.dn_pack_ray_rows(null, null, null, 2.0, 6.0, 0, null, null) == 0
    assert hiplib.dn_adam_step(null, null, null, null, 4, null, null, 1e-3, 1.0, 0.9, 0.999, 1e-8, 0, null) == -1000
    # 8-bit saved tensors (DN_PREC_BF16_S8): the same tiles at one 1 KiB unit per PAIR of bf16 pieces; bf16 arithmetic only
    d.hidden_size = 256
    sizes = {}
    for prec in (_hip.PREC_BF16, _hip.PREC_BF16_S8):
        a, m, g = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
        assert hiplib.dn_mlp_train_sizes(ctypes.byref(d), prec, 1000, ctypes.byref(a), ctypes.byref(m), ctypes.byref(g)) == 0
        sizes[prec] = (a.value, m.value, g.value)
    a16, m16, g16 = sizes[_hip.PREC_BF16]
    a8, m8, g8 = sizes[_hip.PREC_BF16_S8]
    # the 8-bit mode runs the 48-point geometry (mlp_geo48.h): 1 KiB units of 64 features per 16-point group, two groups per
    # 32-point record, whole workgroup tiles of whichever tiling pads further - 384 points (1000 points = 3 tiles = 36 records) or 256
    # (4 tiles = 32 records); two 1 KiB mask words per wave tile and masked stage, wave tiles of the 256-point tiling (4 x 8)
    dd, ww, v = d.num_layers, d.hidden_size, int(d.use_viewdirs)
    khu = ww // 64
    act_units = 1 + v + khu * (1 + (dd - 1) + v) + (ww // 128) * v
    grad_units = (ww // 128) * v + khu * v + (dd - 1) * khu + khu + 1
    # (+ the 256-byte scale / statistics record behind the gradient units)
    assert a8 == 36 * 2 * act_units * 1024 and g8 == 36 * 2 * grad_units * 1024 + 256 and m8 == 4 * 8 * (dd - 1 + 2 * v) * 2 * 1024
    assert a8 < 0.6 * a16 and g8 < 0.6 * g16   # (half the bytes per point; 36 records against the 32-point kernels' 32 tiles)
    # networks outside the 48-point kernels are refused (and train in plain bf16)
    d_deep = _hip.MlpDesc(**{k: getattr(d, k) for k, _ in d._fields_})
    d_deep.num_layers = 12
    assert hiplib.dn_mlp_train_sizes(ctypes.byref(d_deep), _hip.PREC_BF16_S8, 1000, ctypes.byref(a), ctypes.byref(m), ctypes.byref(g)) == -1001
    assert hiplib.dn_mlp_backward_packed_bytes(ctypes.byref(d_deep), _hip.PREC_BF16_S8) == 0 and hiplib.dn_mlp_backward_packed_bytes(ctypes.byref(d), _hip.PREC_BF16_S8) > 0
    assert hiplib.dn_fp16_range_guard(ctypes.byref(d)) in (0, 1)
    assert hiplib.dn_set_s8_grad_scale(3.0) == -1000 and b"power of two" in hiplib.dn_last_error()
    assert hiplib.dn_set_s8_grad_scale(65536.0) == 0
    with pytest.raises(RuntimeError):
        _hip.check(-1000, "probe")


def test_model_mirrors_reference_state_dict():
    import nerf
    sd_c, _ = lego_weights()
    m = nerf.models.FlexibleNeRFModel(num_encoding_fn_xyz=10, num_encoding_fn_dir=4)  # as-shipped defaults: 4 x 128
    assert list(m.state_dict().keys()) == list(sd_c.keys())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_c.items()})
    assert sum(p.numel() for p in m.parameters()) == 84548
    big = nerf.models.FlexibleNeRFModel(num_layers=8, hidden_size=256, skip_connect_every=4, num_encoding_fn_xyz=10)
    assert sum(p.numel() for p in big.parameters()) == 595844
    assert big.layers_xyz[4].weight.shape == (256, 319) and big.skip_layers == [4]
    assert tuple(big.layers_dir[0].weight.shape) == (128, 283)
    for name in ("VeryTinyNeRFModel", "MultiHeadNeRFModel", "ReplicateNeRFModel", "PaperNeRFModel", "FlexibleNeRFModel"):
        assert hasattr(nerf.models, name)


def test_model_host_forward_matches_golden(golden):
    """nn.Module semantics on host tensors (layer1 without activation, skip order cat(x, xyz), alpha from the trunk)."""
    import nerf
    from nerf import synthetic as syn
    g = golden("kat")
    full = dict(num_layers=8, hidden_size=256, skip_connect_every=4, num_encoding_fn_xyz=10, num_encoding_fn_dir=4)
    m = nerf.models.FlexibleNeRFModel(**full)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.synth_state_dict(11, **full).items()})
    with torch.no_grad():
        out = m(torch.from_numpy(g["mlp_in"])).numpy()
    assert np.abs(out - g["mlp_d8w256"]).max() <= 1e-6 * np.abs(g["mlp_d8w256"]).max()


def test_tiny_nerf_plumbing_helpers_on_host(golden):
    """BASELINE config 1: tiny_nerf.py imports these four names and runs them on CPU tensors."""
    from nerf import cumprod_exclusive, get_minibatches, get_ray_bundle, positional_encoding
    g = golden("kat")
    ro, rd = get_ray_bundle(20, 30, 1.0, torch.from_numpy(g["rb1_E"]), torch.from_numpy(g["rb1_K"]))
    np.testing.assert_array_equal(ro.numpy(), g["rb1_ro"])
    np.testing.assert_array_equal(rd.numpy(), g["rb1_rd"])
    np.testing.assert_array_equal(positional_encoding(torch.from_numpy(g["pe_x"]), 10).numpy(), g["pe_l10"])
    np.testing.assert_array_equal(cumprod_exclusive(torch.from_numpy(g["cpe_in"])).numpy(), g["cpe_out"])
    assert [b.shape[0] for b in get_minibatches(torch.zeros(10, 3), 4)] == [4, 4, 2]
    # stale 4-argument callers (tiny_nerf.py:127, eval_nerf.py:174): upstream camera-to-world convention
    c2w = torch.eye(4)
    c2w[:3, 3] = torch.tensor([1.0, 2.0, 3.0])
    ro, rd = get_ray_bundle(4, 6, 10.0, c2w)
    assert ro.shape == (4, 6, 3) and torch.allclose(ro[0, 0], torch.tensor([1.0, 2.0, 3.0]))
    assert torch.allclose(rd[2, 3], torch.tensor([0.0, 0.0, -1.0]))
    assert torch.allclose(rd[0, 0], torch.tensor([-0.3, 0.2, -1.0]))


def test_misc_helpers(golden):
    import nerf
    g = golden("kat")
    o, d = nerf.ndc_rays(378, 504, 407.5, 1.0, torch.from_numpy(g["ndc_o"]), torch.from_numpy(g["ndc_d"]))
    np.testing.assert_array_equal(o.numpy(), g["ndc_out_o"])
    np.testing.assert_array_equal(d.numpy(), g["ndc_out_d"])
    mse = nerf.img2mse(torch.from_numpy(g["mse_a"]), torch.from_numpy(g["mse_b"])).item()
    assert nerf.mse2psnr(mse) == float(g["psnr"]) and nerf.mse2psnr(0) == float(g["psnr0"])
    e = nerf.get_embedding_function(num_encoding_functions=10, include_input=True, log_sampling=False)
    assert (e.num_encoding_functions, e.include_input, e.log_sampling) == (10, True, False)
    cfg = nerf.CfgNode({"nerf": {"train": {"num_coarse": 64}}, "dataset": {"near": 2}})
    assert cfg.nerf.train.num_coarse == 64 and getattr(cfg.nerf, "train").num_coarse == 64 and cfg.dataset.near == 2
    assert "num_coarse: 64" in cfg.dump()
    err = nerf.compute_err_metric(torch.tensor([[1.0, 2.0]]), torch.tensor([[1.001, 2.01]]), torch.tensor([[True, True]]))
    assert abs(err["depth_abs_err"] - 5.5) < 1e-3 and err["depth_err2"] == 0.5 and err["depth_err8"] == 0.5
    img = nerf.depth_error_img(torch.zeros(1, 32, 240), torch.ones(1, 32, 240) * 0.5, torch.ones(1, 32, 240, dtype=torch.bool))
    assert img.shape == (32, 240, 3)


def test_hot_path_refuses_host_tensors():
    """No CPU fallback: host tensors are rejected loudly rather than silently computed elsewhere."""
    import nerf
    with pytest.raises(RuntimeError, match="ROCm device"):
        nerf.volume_render_radiance_field(torch.zeros(2, 4, 4), torch.zeros(2, 4), torch.zeros(2, 3))
    with pytest.raises(RuntimeError, match="ROCm device"):
        nerf.sample_pdf(torch.zeros(2, 16), torch.zeros(2, 15), 8, det=True)
    m = nerf.models.FlexibleNeRFModel()
    with pytest.raises(RuntimeError, match="ROCm device"):
        nerf.run_network(m, torch.zeros(2, 4, 3), torch.zeros(2, 11), 16, nerf.get_embedding_function(6),
                         nerf.get_embedding_function(4))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "dex-nerf_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(root, f)).read()
                assert "oracle" not in text.replace("no oracle in the fork", ""), os.path.join(root, f)


def test_param_key_tracks_fused_optimizer_steps():
    """Adam(fused=True) updates parameters without bumping tensor versions; the packed-weights cache key must still
    change, or the HIP kernels would keep running on stale weights."""
    import nerf
    m = nerf.models.FlexibleNeRFModel()
    k0 = m.param_key()
    assert m.param_key() == k0
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, fused=True)
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    versions = [p._version for p in m.parameters()]
    opt.step()
    assert m.param_key() != k0
    k1 = m.param_key()
    with torch.no_grad():
        m.layer1.weight.mul_(2.0)   # ordinary in-place op: version bump
    assert m.param_key() != k1
    del versions


def test_validation_helpers_host_path(golden):
    """compute_err_metric / depth_error_img on host tensors (the reference's own composition) against the recorded
    reference outputs; RaySelector's index conversion from the reference's column-major pixel draws."""
    import nerf
    g = golden("val_extras")
    gt, mask = torch.from_numpy(g["err_gt"]), torch.from_numpy(g["err_mask"])
    for k in range(g["err_pred"].shape[0]):
        e = nerf.compute_err_metric(gt, torch.from_numpy(g["err_pred"][k]), mask)
        assert [e["depth_abs_err"], e["depth_err2"], e["depth_err4"], e["depth_err8"]] == list(g["err_out"][k])
    img = nerf.depth_error_img(torch.from_numpy(g["err_pred"][1])[None] * 1000, gt[None] * 1000, mask[None])
    np.testing.assert_array_equal(img, g["err_img_1"])
    sel = nerf.RaySelector(30, 40, torch.from_numpy(g["sel_E"]), torch.from_numpy(g["sel_K"]), 2.0, 6.0, device="cpu")
    pix = sel.from_reference_choice(g["sel_inds"])
    np.testing.assert_array_equal(torch.from_numpy(g["sel_image"]).reshape(-1, 4)[pix].numpy(), g["sel_target"])


def test_committed_bench_line_follows_the_contract():
    """The bench line committed under profiles/ (written by bench.py on an MI355X) carries every field the driver's
    contract names, with the roofline and cpu_baseline objects."""
    import json
    path = os.path.join(REPO, "profiles", "r04_bench.json")
    line = json.loads(open(path).read().strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["unit"] == "rays/s" and line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert "workload" in line["config"] and "model" not in line["config"]
    roof = line["roofline"]
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert roof["traffic"] is None or roof["traffic"] > 0
    # round 2 additions: measured-traffic fields, Dex agreement next to the PSNR, the exact-fp32 mode, the training roofline
    assert roof["algorithmic_bytes"] == 160000 * 192 * 20 + 160000 * 44
    assert 0.0 < line["dex_vs_oracle"]["agree_frac"] <= 1.0 and line["dex_vs_oracle"]["worst_miss_m"] >= 0.0
    assert line["fp32_mode"]["frac"] == pytest.approx(line["fp32_mode"]["tflops"] / 157.3)
    tr = line["train"]["roofline"]
    assert tr["bound"] == "hbm" and tr["unit"] == "GB/s" and abs(tr["frac"] - tr["achieved"] / tr["peak"]) < 1e-9
    # round 4: the headline is the config's dtype, the library's default render policy is a leg; every BASELINE config has a leg;
    # the measured traffic comes from a PMC record that names the timed kernel instance
    assert line["dtype"] == line["config_dtype"] == line["render_dtype"] == "bf16" and line["default_policy_fp16_render"]["dtype"] == "fp16"
    assert roof["traffic_record"] == "profiles/r04_pmc_fine_net_bf16.json" and roof["traffic"] > roof["algorithmic_bytes"]
    for leg in ("c3_render", "c3_render_128_192", "c4_render", "c5_render"):
        assert line[leg]["value"] > 0 and 0.0 < line[leg]["frac"] < 1.0, leg
    assert line["c5_render"]["dtype"] == "fp32" and line["c5_train"]["rays_per_s"] > 0 and line["train_s16_mode"]["ms_per_step"] > line["train"]["ms_per_step"]
    marks = line["train_psnr_vs_oracle"]["marks"]
    assert marks[-1]["iteration"] == 300 and abs(line["train_psnr_vs_oracle"]["final_delta_db"]) <= 1.0
    cb = line["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    # value = rays of all ranks / time: consistent with ms_per_step
    rays = line["config"]["rays_per_step_per_gpu"] * line["n_gpus"]
    assert abs(line["value"] - rays / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-6


def test_bench_refuses_a_pmc_record_of_another_kernel_or_other_sources(tmp_path, monkeypatch):
    """bench.py copies `roofline.traffic` from a committed rocprofv3 --pmc record: only from one that names the kernel instance the run
    times and was collected on the kernel sources of this tree (scripts/pmc_collect.py stamps both)."""
    import json
    import sys
    sys.path.insert(0, REPO)
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    csrc = tmp_path / "dex-nerf_amd" / "csrc"
    csrc.mkdir(parents=True)
    for name in bench.KERNEL_SOURCES:
        (csrc / name).write_text("// " + name)
    sha = bench.kernel_source_sha16()
    good = {"kernel": "void dn::mlp_forward48_kernel<256, 1, 8, 16u, 1, 0, 2, 0>(dn::FwdParams, dn::G48Params)", "source_sha16": sha,
            "derived": {"hbm_bytes_per_launch": 6.4e8, "matrix_pipe_busy_frac": 0.8}}
    (prof / "r04_pmc_fine_net_bf16.json").write_text(json.dumps(good))
    rec = bench.pmc_record("bf16")
    assert rec["hbm_bytes_per_launch"] == 6.4e8 and rec["record"] == "profiles/r04_pmc_fine_net_bf16.json"
    (prof / "r04_pmc_fine_net_bf16.json").write_text(json.dumps(dict(good, kernel="void dn::mlp_forward48_kernel<256, 2, 8, 16u, 1, 0, 2, 0>(...)")))
    rec = bench.pmc_record("bf16")
    assert "hbm_bytes_per_launch" not in rec and "refused" in rec["record"]
    (prof / "r04_pmc_fine_net_bf16.json").write_text(json.dumps(good))
    (csrc / bench.KERNEL_SOURCES[0]).write_text("// edited")
    rec = bench.pmc_record("bf16")
    assert "hbm_bytes_per_launch" not in rec and "refused" in rec["record"]
    assert bench.pmc_record("fp16")["record"].startswith("no committed")
    # a record that carries the kernel BINARY's identity (scripts/codeobj.py kernel_isa_sha16: the instruction stream of the instance in
    # the built library) is tied by that - it survives edited sources (comments, refactors), not another instruction stream
    isa = bench.kernel_isa_sha16("bf16")
    if isa is not None:   # (the LLVM tools of the ROCm image)
        (prof / "r04_pmc_fine_net_bf16.json").write_text(json.dumps(dict(good, isa_sha16=isa)))
        rec = bench.pmc_record("bf16")   # (the sources were edited above: the source hash no longer matches)
        assert rec.get("hbm_bytes_per_launch") == 6.4e8 and rec["record_tied_by"] == "isa_sha16", rec
        (prof / "r04_pmc_fine_net_bf16.json").write_text(json.dumps(dict(good, isa_sha16="0" * 16)))
        rec = bench.pmc_record("bf16")
        assert "hbm_bytes_per_launch" not in rec and "kernel binary" in rec["record"], rec


def test_run_reference_launcher_resolves_the_builds_package(tmp_path):
    """The drop-in launcher (dex-nerf_amd/run_reference.py): a script that sits next to a decoy `nerf/` directory - the
    situation of every reference script (train_dexnerf_rgb.py:15-19 beside nerf-pytorch/nerf/) - must import THIS build's
    package, keep its sibling modules importable and run with its own directory as the working directory.  The plain
    `PYTHONPATH=... python script.py` launch of the same script resolves to the decoy (sys.path[0] = script directory):
    that is the pitfall the launcher exists for."""
    import subprocess
    import sys
    scripts = tmp_path / "nerf-pytorch"
    (scripts / "nerf").mkdir(parents=True)
    (scripts / "nerf" / "__init__.py").write_text("WHO = 'decoy'\n")
    (scripts / "sibling_helper.py").write_text("VALUE = 41\n")
    (scripts / "train_dummy.py").write_text(
        "import os, sys\nimport nerf\nimport sibling_helper\n"
        "from nerf import CfgNode, get_embedding_function, get_ray_bundle, models, run_one_iter_of_nerf\n"
        "print('NERF_FILE=' + os.path.abspath(nerf.__file__))\nprint('CWD=' + os.getcwd())\n"
        "print('ARGV=' + ' '.join(sys.argv[1:]))\nprint('SIB=%d' % (sibling_helper.VALUE + 1))\n"
        "print('MAIN=' + __name__)\n")
    pkg = os.path.join(REPO, "dex-nerf_amd")
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, os.path.join(pkg, "run_reference.py"), str(scripts / "train_dummy.py"), "--config", "x.yml"],
                       capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = dict(line.split("=", 1) for line in r.stdout.strip().splitlines() if "=" in line)
    assert out["NERF_FILE"] == os.path.join(pkg, "nerf", "__init__.py")
    assert os.path.realpath(out["CWD"]) == os.path.realpath(str(scripts))
    assert out["ARGV"] == "--config x.yml" and out["SIB"] == "42" and out["MAIN"] == "__main__"
    # the PYTHONPATH recipe does NOT do this: the script directory wins
    env["PYTHONPATH"] = pkg
    (scripts / "which.py").write_text("import nerf\nprint(getattr(nerf, 'WHO', 'build'))\n")
    r2 = subprocess.run([sys.executable, str(scripts / "which.py")], capture_output=True, text=True, env=env, timeout=300)
    assert r2.stdout.strip() == "decoy"
