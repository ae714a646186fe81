"""Bit-exact restatement of the hierarchical sampler's index arithmetic, in explicit numpy fp32/fp64.

TEST INFRASTRUCTURE ONLY (see nerf_oracle.py header).  No torch kernels are called here: every
float operation is spelled out so the result does not depend on the host's ATen dispatch.

Restates sample_pdf_2 (nerf/nerf_helpers.py:262-304) as executed by PyTorch-CPU in the reference
(SURVEY.md section 8a row S7):
  * `torch.sum(w, -1)` of a contiguous fp32 row = ATen's AVX2 cascade-sum inner kernel:
    8-lane vectors, 4 ILP accumulators over the first 4*(nvec//4) vectors, remaining whole vectors
    into accumulator 0, accumulators folded 0+=1,+=2,+=3, then a scalar chain
    acc=0; acc+=tail elements (in order); acc+=lane partials 0..7 (in order);
  * `torch.cumsum` = sequential fp64 accumulator, rounded to fp32 per element;
  * searchsorted(side="right") = count of cdf entries <= u (numpy semantics; the third-party
    torchsearchsorted op, un-vendored and unpinned - requirements.txt:9).
Pinned against the reference's recorded cdf/inds in tests/test_oracle_golden.py.
"""
import numpy as np

F = np.float32


def _ceil_log2(x):
    return 0 if x <= 1 else int(x - 1).bit_length()


def _multi_row_sum(vecs):
    """ATen multi_row_sum<acc, nrows=4>: vecs (R, G, 4, 8) -> (R, 4, 8) with cascade levels."""
    r, g = vecs.shape[0], vecs.shape[1]
    levels = 4
    level_power = max(4, _ceil_log2(g) // levels)
    step = 1 << level_power
    mask = step - 1
    acc = np.zeros((levels, r, 4, 8), F)
    i = 0
    while i + step <= g:
        for _ in range(step):
            acc[0] = acc[0] + vecs[:, i]
            i += 1
        for j in range(1, levels):
            acc[j] = acc[j] + acc[j - 1]
            acc[j - 1] = 0
            if (i & (mask << (j * level_power))) != 0:
                break
    while i < g:
        acc[0] = acc[0] + vecs[:, i]
        i += 1
    for j in range(1, levels):
        acc[0] = acc[0] + acc[j]
    return acc[0]


def aten_sum_lastdim(x):
    """fp32 row sums of a contiguous (R, L) array in ATen-CPU's association order."""
    x = np.ascontiguousarray(x, dtype=F)
    r, length = x.shape
    nvec = length // 8
    g = nvec // 4
    if g > 0:
        part = _multi_row_sum(x[:, : g * 32].reshape(r, g, 4, 8))
    else:
        part = np.zeros((r, 4, 8), F)
    for v in range(g * 4, nvec):
        part[:, 0] = part[:, 0] + x[:, v * 8: v * 8 + 8]
    p0 = part[:, 0]
    for k in range(1, 4):
        p0 = p0 + part[:, k]
    acc = np.zeros((r,), F)
    for k in range(nvec * 8, length):
        acc = acc + x[:, k]
    for k in range(8):
        acc = acc + p0[:, k]
    return acc.astype(F)


def cumsum_f64_rounded(x):
    """Sequential fp64 running sum, each prefix rounded to fp32 (ATen-CPU cumsum on float)."""
    return np.cumsum(x.astype(np.float64), axis=-1).astype(F)


def linspace_f32(start, end, steps):
    """torch.linspace on CPU, fp32: i < steps/2 ? start + step*i : end - step*(steps-1-i), with
    step = (end-start)/(steps-1) in fp32 and each element formed by one fused multiply-add."""
    if steps == 1:
        return np.array([start], F)
    step = F((F(end) - F(start)) / F(steps - 1))
    i = np.arange(steps)
    lo = (np.float64(start) + np.float64(step) * i).astype(F)            # fma: exact product, one rounding
    hi = (np.float64(end) - np.float64(step) * (steps - 1 - i)).astype(F)
    return np.where(i < steps // 2, lo, hi).astype(F)


def sample_pdf_exact(bins, weights, num_samples, u=None):
    """bins (R, B), weights (R, B-1) fp32 -> dict(cdf (R,B), u, inds int64 (R,Nf), samples (R,Nf))."""
    bins = np.ascontiguousarray(bins, dtype=F)
    w = (np.ascontiguousarray(weights, dtype=F) + F(1e-5)).astype(F)
    s = aten_sum_lastdim(w)
    pdf = (w / s[:, None]).astype(F)
    cdf = np.concatenate([np.zeros((w.shape[0], 1), F), cumsum_f64_rounded(pdf)], axis=-1)
    if u is None:
        u = np.broadcast_to(linspace_f32(0.0, 1.0, num_samples), (w.shape[0], num_samples))
    u = np.ascontiguousarray(u, dtype=F)
    inds = (cdf[:, None, :] <= u[:, :, None]).sum(-1).astype(np.int64)
    below = np.maximum(inds - 1, 0)
    above = np.minimum(inds, cdf.shape[-1] - 1)
    cdf_b = np.take_along_axis(cdf, below, -1)
    cdf_a = np.take_along_axis(cdf, above, -1)
    bin_b = np.take_along_axis(bins, below, -1)
    bin_a = np.take_along_axis(bins, above, -1)
    denom = (cdf_a - cdf_b).astype(F)
    denom = np.where(denom < F(1e-5), F(1.0), denom).astype(F)
    t = ((u - cdf_b).astype(F) / denom).astype(F)
    samples = (bin_b + (t * (bin_a - bin_b).astype(F)).astype(F)).astype(F)
    return dict(cdf=cdf, u=u, inds=inds, samples=samples)
