"""CPU oracle for the nerf-pytorch / Dex-NeRF ray-marching hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (dex-nerf_amd/) may import this module;
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the checker
or as the timed CPU baseline.  It is a from-scratch restatement (fp32, PyTorch CPU ops in the same
order as the reference so that ATen's CPU arithmetic - double-accumulated cumprod/cumsum, the sum
kernel's association order, FMA-form linspace - is reproduced) of the reference functions cited in
each docstring (paths relative to /root/reference/nerf-pytorch/).

Parity pin: every function here is checked against golden vectors captured from the imported
reference (tests/golden/*.npz, written by tests/golden/make_golden.py) in tests/test_oracle_golden.py.
`exact_sampler.py` next to this file restates the sampler with explicit numpy arithmetic (no torch
kernels) for the bit-exact index check.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import torch


# ----------------------------------------------------------------------------------------------
# configuration (the attribute bag the reference reads at train_utils.py:111-199,222-252)
# ----------------------------------------------------------------------------------------------
@dataclass
class RenderCfg:
    num_coarse: int = 64
    num_fine: int = 128
    near: float = 2.0
    far: float = 6.0
    lindisp: bool = False
    perturb: bool = False
    noise_std: float = 0.0
    white_background: bool = False
    chunksize: int = 16384
    use_viewdirs: bool = True
    m_thres: Sequence[float] = field(default_factory=lambda: tuple(float(x) for x in range(5, 105, 5)))


@dataclass
class ModelCfg:
    num_layers: int = 8
    hidden_size: int = 256
    skip_connect_every: int = 4
    num_encoding_fn_xyz: int = 10
    num_encoding_fn_dir: int = 4
    include_input_xyz: bool = True
    include_input_dir: bool = True
    use_viewdirs: bool = True
    log_sampling_xyz: bool = True
    log_sampling_dir: bool = True

    @property
    def dim_xyz(self):
        return (3 if self.include_input_xyz else 0) + 6 * self.num_encoding_fn_xyz

    @property
    def dim_dir(self):
        return ((3 if self.include_input_dir else 0) + 6 * self.num_encoding_fn_dir) if self.use_viewdirs else 0

    def skip_layers(self):
        """Trunk layers whose input is cat(x, xyz): where models.py:210 builds the wide Linear."""
        return [i for i in range(self.num_layers - 1)
                if i % self.skip_connect_every == 0 and i > 0 and i != self.num_layers - 1]


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))


# ----------------------------------------------------------------------------------------------
# S1  get_ray_bundle + meshgrid_xy      (nerf/nerf_helpers.py:28-40, 67-112)
# ----------------------------------------------------------------------------------------------
def get_ray_bundle(height, width, extrinsic, intrinsic):
    """Pixel grid -> (ray_origins, ray_directions), both (H, W, 3).

    ii[h, w] = w, jj[h, w] = h; dir = [(ii - cx)/fx, (jj - cy)/fx, 1] (fx divides y as well,
    nerf_helpers.py:100-101); rd = sum_k dir_k * inv(E[:3,:3])[:, k]; ro = inv(E)[:3, 3].
    """
    E, K = _t(extrinsic), _t(intrinsic)
    cols = torch.arange(width, dtype=E.dtype)
    rows = torch.arange(height, dtype=E.dtype)
    ii = cols[None, :].expand(height, width)
    jj = rows[:, None].expand(height, width)
    d = torch.stack(((ii - K[0, 2]) / K[0, 0], (jj - K[1, 2]) / K[0, 0], torch.ones_like(ii)), dim=-1)
    rinv = torch.inverse(E[:3, :3])
    rd = (d[..., None, :] * rinv).sum(dim=-1)
    ro = torch.inverse(E)[:3, -1].expand(rd.shape)
    return ro, rd


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """Forward-facing NDC warp (nerf/nerf_helpers.py:172-199)."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    o = rays_o + t[..., None] * rays_d
    sx = -1.0 / (W / (2.0 * focal))
    sy = -1.0 / (H / (2.0 * focal))
    o0 = sx * o[..., 0] / o[..., 2]
    o1 = sy * o[..., 1] / o[..., 2]
    o2 = 1.0 + 2.0 * near / o[..., 2]
    d0 = sx * (rays_d[..., 0] / rays_d[..., 2] - o[..., 0] / o[..., 2])
    d1 = sy * (rays_d[..., 1] / rays_d[..., 2] - o[..., 1] / o[..., 2])
    d2 = -2.0 * near / o[..., 2]
    return torch.stack((o0, o1, o2), -1), torch.stack((d0, d1, d2), -1)


# ----------------------------------------------------------------------------------------------
# S5  positional_encoding               (nerf/nerf_helpers.py:115-159)
# ----------------------------------------------------------------------------------------------
def frequency_bands(num_fns, log_sampling=True, dtype=torch.float32):
    if num_fns == 0:
        return torch.zeros(0, dtype=dtype)
    if log_sampling:
        return 2.0 ** torch.linspace(0.0, num_fns - 1, num_fns, dtype=dtype)
    return torch.linspace(1.0, 2.0 ** (num_fns - 1), num_fns, dtype=dtype)


def positional_encoding(x, num_fns=6, include_input=True, log_sampling=True):
    """[x, sin(f0 x), cos(f0 x), sin(f1 x), cos(f1 x), ...]; per frequency [sin(3), cos(3)]."""
    x = _t(x)
    parts = [x] if include_input else []
    for f in frequency_bands(num_fns, log_sampling, x.dtype):
        parts.append(torch.sin(x * f))
        parts.append(torch.cos(x * f))
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=-1)


# ----------------------------------------------------------------------------------------------
# S6m FlexibleNeRFModel.forward         (nerf/models.py:185-256), functional over a state dict
# ----------------------------------------------------------------------------------------------
def flexible_mlp(sd, x, mc: ModelCfg):
    """sd: {'layer1.weight': (W, dim_xyz), ...} tensors; x: (B, dim_xyz + dim_dir) -> (B, 4).

    layer1 has NO activation (models.py:238); ReLU follows each layers_xyz[i] (:246); the skip
    concatenates (x, xyz) in that order (:245); alpha comes from the trunk output, not from feat
    (:249); feat = relu(fc_feat(x)) (:248).
    """
    lin = torch.nn.functional.linear
    xyz = x[..., : mc.dim_xyz]
    h = lin(xyz, sd["layer1.weight"], sd["layer1.bias"])
    skips = set(mc.skip_layers())
    for i in range(mc.num_layers - 1):
        if i in skips:
            h = torch.cat((h, xyz), dim=-1)
        h = torch.relu(lin(h, sd[f"layers_xyz.{i}.weight"], sd[f"layers_xyz.{i}.bias"]))
    if not mc.use_viewdirs:
        return lin(h, sd["fc_out.weight"], sd["fc_out.bias"])
    view = x[..., mc.dim_xyz:]
    feat = torch.relu(lin(h, sd["fc_feat.weight"], sd["fc_feat.bias"]))
    alpha = lin(h, sd["fc_alpha.weight"], sd["fc_alpha.bias"])
    g = torch.relu(lin(torch.cat((feat, view), dim=-1), sd["layers_dir.0.weight"], sd["layers_dir.0.bias"]))
    rgb = lin(g, sd["fc_rgb.weight"], sd["fc_rgb.bias"])
    return torch.cat((rgb, alpha), dim=-1)


def to_torch_sd(sd_np, requires_grad=False):
    out = {}
    for k, v in sd_np.items():
        t = _t(v).clone().float()
        if requires_grad:
            t.requires_grad_(True)
        out[k] = t
    return out


# ----------------------------------------------------------------------------------------------
# S4  run_network                       (nerf/train_utils.py:72-89)
# ----------------------------------------------------------------------------------------------
def run_network(sd, pts, viewdirs, mc: ModelCfg, chunksize=16384):
    """pts (N, S, 3), viewdirs (N, 3) or None -> raw radiance field (N, S, 4) = [r, g, b, sigma]."""
    flat = pts.reshape(-1, 3)
    emb = positional_encoding(flat, mc.num_encoding_fn_xyz, mc.include_input_xyz, mc.log_sampling_xyz)
    if mc.use_viewdirs:
        vd = viewdirs[:, None, :].expand(pts.shape).reshape(-1, 3)
        emb = torch.cat((emb, positional_encoding(vd, mc.num_encoding_fn_dir, mc.include_input_dir,
                                                  mc.log_sampling_dir)), dim=-1)
    outs = [flexible_mlp(sd, emb[i:i + chunksize], mc) for i in range(0, emb.shape[0], chunksize)]
    return torch.cat(outs, dim=0).reshape(*pts.shape[:-1], 4)


# ----------------------------------------------------------------------------------------------
# S6  volume_render_radiance_field      (nerf/volume_rendering_utils.py:6-70)
#     cumprod_exclusive                 (nerf/nerf_helpers.py:43-64)
# ----------------------------------------------------------------------------------------------
def cumprod_exclusive(t):
    c = torch.cumprod(t, dim=-1)
    return torch.cat((torch.ones_like(c[..., :1]), c[..., :-1]), dim=-1)


def volume_render(rf, z, rd, noise=None, noise_std=0.0, white_background=False, m_thres=()):
    """Returns dict(rgb, disp, acc, weights, depth, dex=[K tensors], sigma).

    dists = [z[i+1]-z[i], 1e10] * ||rd||; sigma = relu(raw + noise*std); alpha = 1-exp(-sigma*dists);
    w = alpha * excl_cumprod(1 - alpha + 1e-10); Dex depth for threshold m = z[first i with sigma_i > m]
    (index 0 if none: torch.argmax of an all-zero int row returns 0), volume_rendering_utils.py:51-58.
    `noise` is the raw N(0,1) draw (reference: torch.randn(...) * std, :32-38).
    """
    big = torch.full_like(z[..., :1], 1e10)
    dists = torch.cat((z[..., 1:] - z[..., :-1], big), dim=-1)
    dists = dists * rd[..., None, :].norm(p=2, dim=-1)
    rgb = torch.sigmoid(rf[..., :3])
    raw_sigma = rf[..., 3]
    if noise_std > 0.0:
        raw_sigma = raw_sigma + noise * noise_std
    sigma = torch.relu(raw_sigma)
    alpha = 1.0 - torch.exp(-sigma * dists)
    weights = alpha * cumprod_exclusive(1.0 - alpha + 1e-10)
    rgb_map = (weights[..., None] * rgb).sum(dim=-2)
    depth_map = (weights * z).sum(dim=-1)
    acc_map = weights.sum(dim=-1)
    dex = []
    rows = torch.arange(z.shape[0])
    for m in m_thres:
        first = torch.argmax((sigma > float(m)).to(torch.int32), dim=-1)
        dex.append(z[rows, first])
    disp_map = 1.0 / torch.max(1e-10 * torch.ones_like(depth_map), depth_map / acc_map)
    if white_background:
        rgb_map = rgb_map + (1.0 - acc_map[..., None])
    return dict(rgb=rgb_map, disp=disp_map, acc=acc_map, weights=weights, depth=depth_map, dex=dex, sigma=sigma)


# ----------------------------------------------------------------------------------------------
# S7  sample_pdf_2                      (nerf/nerf_helpers.py:262-304) + torchsearchsorted
#     (third-party, unpinned; numpy side="right" semantics: count of cdf <= u)
# ----------------------------------------------------------------------------------------------
def sample_pdf(bins, weights, num_samples, det=True, u=None, return_aux=False):
    w = weights + 1e-5
    pdf = w / torch.sum(w, dim=-1, keepdim=True)
    cdf = torch.cumsum(pdf, dim=-1)
    cdf = torch.cat((torch.zeros_like(cdf[..., :1]), cdf), dim=-1).contiguous()
    if u is None:
        if det:
            u = torch.linspace(0.0, 1.0, steps=num_samples, dtype=w.dtype).expand(*cdf.shape[:-1], num_samples)
        else:
            u = torch.rand(*cdf.shape[:-1], num_samples, dtype=w.dtype)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = (inds - 1).clamp(min=0)
    above = inds.clamp(max=cdf.shape[-1] - 1)
    cdf_b, cdf_a = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    bin_b, bin_a = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_b) / denom
    samples = bin_b + t * (bin_a - bin_b)
    if return_aux:
        return samples, dict(cdf=cdf, u=u, inds=inds)
    return samples


# ----------------------------------------------------------------------------------------------
# S3  predict_and_render_radiance       (nerf/train_utils.py:92-202)
# ----------------------------------------------------------------------------------------------
def coarse_depths(near, far, num_coarse, lindisp=False, t_rand=None):
    """near/far (N,1).  t=linspace(0,1,Nc); z = near(1-t)+far t, or the lindisp form (:120-122);
    stratified jitter z = lower + (upper-lower) * t_rand (:126-133)."""
    t = torch.linspace(0.0, 1.0, num_coarse, dtype=near.dtype)
    if not lindisp:
        z = near * (1.0 - t) + far * t
    else:
        z = 1.0 / (1.0 / near * (1.0 - t) + 1.0 / far * t)
    z = z.expand(near.shape[0], num_coarse)
    if t_rand is not None:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat((mids, z[..., -1:]), dim=-1)
        lower = torch.cat((z[..., :1], mids), dim=-1)
        z = lower + (upper - lower) * t_rand
    return z


def predict_and_render(rays, sd_c, sd_f, mc_c: ModelCfg, mc_f: ModelCfg, cfg: RenderCfg, draws=None,
                       return_aux=False):
    """rays (N, 8 or 11) = [ro3, rd3, near, far, (viewdir3)].

    draws: optional dict(t_rand (N,Nc), noise_c (N,Nc), u (N,Nf), noise_f (N,Nc+Nf)); when absent and the
    mode needs them they are drawn with torch.rand/randn in the reference's order.
    Returns the reference's tuple order (:201): rgb_c, depth_c, acc_c, rgb_f, depth_f, acc_f, *dex_f.
    """
    draws = dict(draws or {})
    n = rays.shape[0]
    ro, rd = rays[..., :3], rays[..., 3:6]
    near, far = rays[..., 6:7], rays[..., 7:8]
    vd = rays[..., 8:11] if cfg.use_viewdirs else None
    t_rand = None
    if cfg.perturb:
        t_rand = draws.get("t_rand")
        if t_rand is None:
            t_rand = torch.rand(n, cfg.num_coarse, dtype=rays.dtype)
    z_c = coarse_depths(near, far, cfg.num_coarse, cfg.lindisp, t_rand)
    pts_c = ro[..., None, :] + rd[..., None, :] * z_c[..., :, None]
    rf_c = run_network(sd_c, pts_c, vd, mc_c, cfg.chunksize)
    noise_c = draws.get("noise_c")
    if cfg.noise_std > 0.0 and noise_c is None:
        noise_c = torch.randn(n, cfg.num_coarse, dtype=rays.dtype)
    vc = volume_render(rf_c, z_c, rd, noise_c, cfg.noise_std, cfg.white_background, cfg.m_thres)
    aux = dict(z_coarse=z_c, pts_coarse=pts_c, rf_coarse=rf_c, vc=vc)
    if cfg.num_fine <= 0 or sd_f is None:
        # superset behaviour (the fork raises NameError here, SURVEY.md section 2 Breakages)
        out = (vc["rgb"], vc["depth"], vc["acc"], None, None, None) + tuple(vc["dex"])
        return (out, aux) if return_aux else out
    z_mid = 0.5 * (z_c[..., 1:] + z_c[..., :-1])
    u = draws.get("u")
    zs, sp_aux = sample_pdf(z_mid, vc["weights"][..., 1:-1], cfg.num_fine, det=(not cfg.perturb), u=u,
                            return_aux=True)
    zs = zs.detach()
    z_f, _ = torch.sort(torch.cat((z_c, zs), dim=-1), dim=-1)
    pts_f = ro[..., None, :] + rd[..., None, :] * z_f[..., :, None]
    rf_f = run_network(sd_f, pts_f, vd, mc_f, cfg.chunksize)
    noise_f = draws.get("noise_f")
    if cfg.noise_std > 0.0 and noise_f is None:
        noise_f = torch.randn(n, cfg.num_coarse + cfg.num_fine, dtype=rays.dtype)
    vf = volume_render(rf_f, z_f, rd, noise_f, cfg.noise_std, cfg.white_background, cfg.m_thres)
    out = (vc["rgb"], vc["depth"], vc["acc"], vf["rgb"], vf["depth"], vf["acc"]) + tuple(vf["dex"])
    if return_aux:
        aux.update(z_samples=zs, sp=sp_aux, z_fine=z_f, pts_fine=pts_f, rf_fine=rf_f, vf=vf)
        return out, aux
    return out


# ----------------------------------------------------------------------------------------------
# S2  run_one_iter_of_nerf              (nerf/train_utils.py:205-288)
# ----------------------------------------------------------------------------------------------
def pack_rays(ro, rd, cfg: RenderCfg):
    ro, rd = ro.reshape(-1, 3), rd.reshape(-1, 3)
    parts = [ro, rd, cfg.near * torch.ones_like(rd[..., :1]), cfg.far * torch.ones_like(rd[..., :1])]
    if cfg.use_viewdirs:
        parts.append(rd / rd.norm(p=2, dim=-1).unsqueeze(-1))
    return torch.cat(parts, dim=-1)


def run_one_iter(ro, rd, sd_c, sd_f, mc_c, mc_f, cfg: RenderCfg, draws=None):
    """Flat outputs (N,3)/(N,) in the reference's order; ray chunking as train_utils.py:252-271."""
    rays = pack_rays(ro, rd, cfg)
    chunks = []
    for i in range(0, rays.shape[0], cfg.chunksize):
        sl = slice(i, i + cfg.chunksize)
        d = None if draws is None else {k: v[sl] for k, v in draws.items()}
        chunks.append(predict_and_render(rays[sl], sd_c, sd_f, mc_c, mc_f, cfg, d))
    cols = list(zip(*chunks))
    return tuple(torch.cat(c, dim=0) if c[0] is not None else None for c in cols)


def nerf_loss(outputs, target):
    """loss = mse(rgb_coarse, tgt) + mse(rgb_fine, tgt) (train_dexnerf_rgb.py:264-277)."""
    mse = torch.nn.functional.mse_loss
    loss = mse(outputs[0][..., :3], target[..., :3])
    if outputs[3] is not None:
        loss = loss + mse(outputs[3][..., :3], target[..., :3])
    return loss


def mse2psnr(mse):
    """-10 log10(mse), 0 -> 1e-5 guard (nerf/nerf_helpers.py:13-17)."""
    import math
    return -10.0 * math.log10(1e-5 if mse == 0 else mse)


# ----------------------------------------------------------------------------------------------
# validation extras and training-ray selection
# ----------------------------------------------------------------------------------------------
def compute_err_metric(depth_gt, depth_pred, mask):
    """nerf/train_utils.py:9-30: mean |pred*1000 - gt*1000| over the mask and the fractions of masked pixels whose
    |gt - pred| exceeds 2e-3 / 4e-3 / 8e-3 (all fp32 tensor arithmetic)."""
    gt, pred = _t(depth_gt), _t(depth_pred)
    mask = torch.as_tensor(np.asarray(mask), dtype=torch.bool)
    g, p = gt[mask], pred[mask]
    diff = torch.abs(g - p)
    n = diff.numel()
    return dict(depth_abs_err=torch.mean(torch.abs(p * 1000 - g * 1000)).item(),
                depth_err2=int((diff > 2e-3).sum()) / n, depth_err4=int((diff > 4e-3).sum()) / n,
                depth_err8=int((diff > 8e-3).sum()) / n)


def error_colormap():
    """nerf/train_utils.py:31-45: rows [lo, hi, r, g, b] (colours / 255), float32."""
    edges = [0.0, 0.00001] + [2000.0 / 2 ** e for e in range(10, 1, -1)] + [np.inf]
    rgb = [(0, 0, 0), (49, 54, 149), (69, 117, 180), (116, 173, 209), (171, 217, 233), (224, 243, 248), (254, 224, 144),
           (253, 174, 97), (244, 109, 67), (215, 48, 39), (165, 0, 38)]
    cols = np.array([[edges[i], edges[i + 1], *rgb[i]] for i in range(11)], dtype=np.float32)
    cols[:, 2:5] /= 255.0
    return cols


def depth_error_img(est, gt, mask, abs_thres=1.0):
    """nerf/train_utils.py:46-70 for one (H,W) map: colour-coded |gt - est| / abs_thres, masked-out pixels black, the
    colour legend (20-pixel swatches) over the top ten rows."""
    est, gt = np.asarray(est, np.float32), np.asarray(gt, np.float32)
    mask = np.asarray(mask, bool)
    err = np.abs(gt - est)
    err[~mask] = 0
    err[mask] = err[mask] / np.float32(abs_thres)
    cols = error_colormap()
    img = np.zeros(gt.shape + (3,), np.float32)
    for row in cols:
        img[(err >= row[0]) & (err < row[1])] = row[2:]
    img[~mask] = 0.0
    for i, row in enumerate(cols):
        img[:10, i * 20:(i + 1) * 20, :] = row[2:]
    return img


def select_training_rays(height, width, extrinsic, intrinsic, select_inds, image, near, far):
    """train_dexnerf_rgb.py:229-242 + nerf/train_utils.py:225-250: the reference enumerates pixels column-major
    (its coordinate grid is meshgrid_xy(arange(H), arange(W)) flattened, so draw f means pixel (f % H, f // H)),
    gathers origin / direction / target there, and run_one_iter_of_nerf packs [ro, rd, near, far, rd / ||rd||]."""
    ro, rd = get_ray_bundle(height, width, extrinsic, intrinsic)
    f = torch.as_tensor(np.asarray(select_inds), dtype=torch.int64)
    h, w = f % height, f // height
    ro_s, rd_s = ro[h, w, :], rd[h, w, :]
    target = _t(image)[h, w]
    viewdirs = rd_s / rd_s.norm(p=2, dim=-1).unsqueeze(-1)
    near_t = near * torch.ones_like(rd_s[..., :1])
    far_t = far * torch.ones_like(rd_s[..., :1])
    return torch.cat((ro_s, rd_s, near_t, far_t, viewdirs), dim=-1), target
