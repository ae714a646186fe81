import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "dex-nerf_amd")]
from nerf import _ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(4)
n = 5000
ro = torch.randn(n, 3, device=dev, generator=g); rd = torch.randn(n, 3, device=dev, generator=g) * 3.0
nrm_t = rd.norm(p=2, dim=-1)
x, y, z = rd[:, 0], rd[:, 1], rd[:, 2]
cands = {"(xx+yy)+zz": torch.sqrt((x * x + y * y) + z * z), "xx+(yy+zz)": torch.sqrt(x * x + (y * y + z * z)), "(xx+zz)+yy": torch.sqrt((x * x + z * z) + y * y),
         "f64": torch.sqrt((rd.double() ** 2).sum(-1)).float(), "fma": torch.sqrt(torch.addcmul(torch.addcmul(x * x, y, y), z, z)),
         "fma2": torch.sqrt(torch.addcmul(torch.addcmul(z * z, y, y), x, x))}
for k, v in cands.items():
    print(k, int((v != nrm_t).sum()))
rows = _ops.pack_ray_rows(ro, rd, rd, 0.3, 4.0)
vt = rd / nrm_t.unsqueeze(-1)
print("cols differing:", [(c, int((rows[:, 8 + c] != vt[:, c]).sum())) for c in range(3)], "other cols", int((rows[:, :6] != torch.cat([ro, rd], -1)).sum()))
v_mine_nrm = rd / cands["(xx+yy)+zz"].unsqueeze(-1)
print("division with my norm vs torch viewdirs:", int((v_mine_nrm != vt).sum()), " rows vs my-norm division:", int((rows[:, 8:] != v_mine_nrm).sum()))
