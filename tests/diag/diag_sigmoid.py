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

# softplus of the articulated activation (sigma = softplus(raw - 1)): two-sample rays with t = (3, 4), unit direction, rgb raw 0:
# alpha_0 = 1 - exp(-sigma_0), so sigma_0 = -log(1 - w_0) is readable from the first weight where it is not saturated
xs = torch.linspace(-12, 3, 60001)
n = xs.numel()
raw = torch.zeros(n, 2, 4)
raw[:, 0, 3] = xs + 1.0
raw[:, 1, 3] = -50.0
t = torch.tensor([[3.0, 4.0]]).expand(n, 2).contiguous()
d = torch.tensor([[0.0, 0.0, 1.0]]).expand(n, 3).contiguous()
_, _, w, _ = ops.composite_raw(raw.to(dev), t.to(dev), d.to(dev), False, ops.ACT_ARTICULATED)
sig_true = torch.nn.functional.softplus(xs.double())
w_true = 1 - torch.exp(-sig_true)
print("articulated first weight, max abs err vs fp64:", (w[:, 0].cpu().double() - w_true).abs().max().item())
big = torch.tensor([25.0, 100.0, float("inf"), -float("inf"), float("nan"), -100.0])
raw = torch.zeros(6, 2, 4); raw[:, 0, 3] = big + 1.0; raw[:, 1, 3] = -50.0
t = torch.tensor([[3.0, 3.0 + 1e-3]]).expand(6, 2).contiguous()
_, _, w, _ = ops.composite_raw(raw.to(dev), t.to(dev), d[:6].contiguous().to(dev), False, ops.ACT_ARTICULATED)
print("softplus specials through alpha (x = 25, 100, inf, -inf, nan, -100; delta 1e-3):", w[:, 0].cpu().tolist(),
      "expected", (1 - torch.exp(-torch.nn.functional.softplus(big.double()) * (torch.tensor(3.0 + 1e-3).float().double() - 3.0))).tolist())
