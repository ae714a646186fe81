"""Developer probe: HIP-event times of the non-network kernels of one C2 render (compositing with / without the Dex readout,
fine-depth sampler, coarse depths) on 160,000 rays."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "dex-nerf_amd"))
from nerf import _ops

dev = torch.device("cuda:0")
n = 160000
thres = [float(m) for m in range(5, 105, 5)]


def timed(f, reps=10):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for s in (64, 192):
    rf = torch.randn(n, s, 4, device=dev) * 3
    z = torch.sort(torch.rand(n, s, device=dev) * 4 + 2, -1)[0].contiguous()
    rd = torch.randn(n, 3, device=dev)
    for k, name in ((thres, "20 Dex thresholds"), ([], "no Dex readout")):
        for ww in (True, False):
            t = timed(lambda: _ops.volume_render_fwd(rf, z, rd, None, 0.0, False, k, want_weights=ww))
            gb = n * s * 20 + (n * s * 4 if ww else 0)
            print(f"composite S={s} {name}, weights {'out' if ww else 'not written'}: {t:.0f} us ({gb / t / 1e6:.2f} TB/s)")
zc = torch.sort(torch.rand(n, 64, device=dev) * 4 + 2, -1)[0].contiguous()
w = torch.rand(n, 64, device=dev)
t = timed(lambda: _ops.fine_depths(zc, w, 128, None))
print(f"fine_depths 64+128 det: {t:.0f} us")
u = torch.rand(n, 128, device=dev)
t = timed(lambda: _ops.fine_depths(zc, w, 128, u))
print(f"fine_depths 64+128 random u: {t:.0f} us")
