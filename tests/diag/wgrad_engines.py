"""Weight gradients of the articulated level with the fp32 and the bf16x3 training engine on the SAME planes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import aon_amd.synthetic as syn
from aon_amd import ops

dev = torch.device("cuda:0")
n, S = 40, 193
sd = syn.make_art_state_dict(seed=2, density_scale=10.0)
params = {k[len("fine_mlp."):]: v.to(dev) for k, v in sd.items() if k.startswith("fine_mlp.")}
g = torch.Generator().manual_seed(0)
lat = {"density": torch.randn(1, 128, generator=g).to(dev) * 0.1, "color": torch.randn(1, 128, generator=g).to(dev) * 0.1,
       "articulation": torch.randn(1, 32, generator=g).to(dev) * 0.1}
packed, packed_bwd, small = ops.pack_art_mlp(params), ops.pack_art_mlp_bwd(params), ops.art_prepare(params, lat)
rays = syn.random_rays(n, seed=21)
t = torch.sort(torch.rand(n, S, generator=g) * 4 + 2, dim=-1).values
o, d, v, tt = (x.to(dev) for x in (rays["rays_o"], rays["rays_d"], rays["viewdirs"], t))
raw, planes, masks = ops.art_mlp_fwd_train(packed, small, o, d, v, tt)
rgb = ops.composite_raw(raw, tt, d, True, ops.ACT_ARTICULATED)[0]
g_rgb = 2.0 * (rgb - 0.5) / (n * 3)
d_raw = ops.composite_bwd(raw, tt, d, g_rgb, None, None, True, ops.ACT_ARTICULATED, planes.shape[1])
dplanes, dxp = ops.art_bwd_chain(packed_bwd, small, d_raw, masks, planes)
res = {}
for eng in ("fp32", "bf16x3"):
    ops.set_train_engine(eng)
    grads, g_lat = ops.art_wgrad(planes, dplanes, d_raw, dxp, params, lat)
    res[eng] = {k: x.double().cpu() for k, x in grads.items()}
ops.set_train_engine("fp32")
# fp64 reference for a few layers straight from the planes
P, D = planes.double().cpu(), dplanes.double().cpu()
for k in sorted(res["fp32"]):
    a, b = res["fp32"][k], res["bf16x3"][k]
    print(f"{k:<32} rel diff bf16x3 vs fp32 engine: {((a - b).norm() / (a.norm() + 1e-300)).item():.2e}   |grad| {a.norm().item():.3e}")
