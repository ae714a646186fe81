"""Developer probe: what the vendor GEMM library reaches on this box (bf16, fp32 accumulate) - a practical yardstick next
to the 2.5 PFLOP/s datasheet peak used by bench.py's roofline: a large square GEMM, and the MLP's own layer shape
(points x 256) @ (256 x 256) as a standalone library call (activations through HBM)."""
import torch, time
dev = torch.device("cuda:0")
def rate(m, n, k, reps=20):
    a = torch.randn(m, k, device=dev, dtype=torch.bfloat16); b = torch.randn(k, n, device=dev, dtype=torch.bfloat16)
    for _ in range(3): c = a @ b
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps): c = a @ b
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / reps
    return 2.0 * m * n * k / ms / 1e9, ms
for (m, n, k) in ((8192, 8192, 8192), (16384, 16384, 8192), (786432, 256, 256), (6291456, 256, 256), (30720000, 256, 256)):
    tf, ms = rate(m, n, k, reps=10 if m > 1e6 else 20)
    print(f"bf16 GEMM {m} x {n} x {k}: {ms:.3f} ms  {tf:.0f} TFLOP/s", flush=True)
