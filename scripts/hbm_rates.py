"""Developer probe: HBM write-only / read-only / copy rates with torch's own kernels (yardsticks for the training kernels)."""
import torch
dev = torch.device("cuda:0")
n = 1 << 30   # 4 GiB fp32
x = torch.empty(n, dtype=torch.float32, device=dev); y = torch.empty_like(x)
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps): f()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e-3
dt = t(lambda: x.fill_(1.0)); print(f"fill (write only): {n * 4 / dt / 1e12:.2f} TB/s")
dt = t(lambda: x.sum()); print(f"sum (read only): {n * 4 / dt / 1e12:.2f} TB/s")
dt = t(lambda: y.copy_(x)); print(f"copy: {2 * n * 4 / dt / 1e12:.2f} TB/s (read + write)")
dt = t(lambda: torch.add(x, 1.0, out=y)); print(f"add: {2 * n * 4 / dt / 1e12:.2f} TB/s (read + write)")
