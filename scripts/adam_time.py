"""Developer probe: dn_adam_step alone on flat buffers of two D8/W256 networks (1,191,688 parameters), HIP-event timing."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "dex-nerf_amd")]
from nerf import _ops
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1191688
p, g, m, v = (torch.randn(n, device=dev) for _ in range(4))
v.abs_()
state = torch.zeros(12, device=dev)[:10]
state[4:10].view(torch.float64).copy_(torch.tensor([1.0, 1.0, 1.0], dtype=torch.float64))
for _ in range(5):
    _ops.adam_step(p, g, m, v, state, 5e-4, 1.0, (0.9, 0.999), 1e-8, True)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); a.record()
for _ in range(200):
    _ops.adam_step(p, g, m, v, state, 5e-4, 1.0, (0.9, 0.999), 1e-8, True)
b.record(); torch.cuda.synchronize()
print(f"n = {n}: {a.elapsed_time(b) / 200 * 1e3:.2f} us per step ({n * 32 / (a.elapsed_time(b) / 200 * 1e-3) / 1e12:.2f} TB/s of 32 B per parameter)")
