"""Developer probe: milliseconds per rendered image (whole path, host glue included) for BASELINE configs 2 and 3."""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "dex-nerf_amd")]
import nerf, bench
dev = torch.device("cuda:0")
nerf.set_precision("bf16")
for tag, args in (("C2 400x400 64+128 D8/W256", (bench.H, bench.W, 64, 128, None, 2.0, 6.0)),
                  ("C3 270x480 64+64 4x128", (270, 480, 64, 64, dict(bench.MODEL_KW, num_layers=4, hidden_size=128), 0.3, 4.0))):
    models, cfg, ro, rd, ex, ed = bench.build_scene(dev, 0, *args)
    for _ in range(2):
        bench.render(models, cfg, ro, rd, ex, ed)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        bench.render(models, cfg, ro, rd, ex, ed)
    torch.cuda.synchronize()
    print(f"{tag}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per image", flush=True)
