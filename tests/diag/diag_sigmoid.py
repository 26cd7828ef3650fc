"""Device sigmoid of the compositing kernels against torch over the whole useful range (one-sample rays: comp_rgb = sigmoid(raw) * w)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aon_amd import ops
dev = torch.device("cuda:0")
x = torch.cat([torch.linspace(-100, 100, 200001), torch.tensor([float("inf"), -float("inf"), float("nan"), -87.5, -88.9, 89.0])])
n = x.numel()
raw = torch.zeros(n, 1, 4)
raw[:, 0, 0] = x; raw[:, 0, 1] = -x; raw[:, 0, 2] = 0.5 * x
raw[:, 0, 3] = 1.0   # sigma 1 * delta 1e10 -> alpha = 1, w = 1
t = torch.full((n, 1), 3.0)
d = torch.tensor([[0.0, 0.0, 1.0]]).expand(n, 3).contiguous()
cr, acc, w, dep = ops.composite_raw(raw.to(dev), t.to(dev), d.to(dev), False, ops.ACT_VANILLA)
ref = torch.sigmoid(raw[:, 0, :3].double())
err = (cr.cpu().double() - ref).abs()
fin = torch.isfinite(x)
print("max abs err (finite x):", err[fin].max().item(), "at x =", x[fin][err[fin].max(1).values.argmax()].item())
print("specials (inf, -inf, nan, -87.5, -88.9, 89):", cr[-6:].cpu().tolist())
