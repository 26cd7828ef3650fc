import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aon_amd.synthetic as syn
from aon_amd import ops
from oracle import nerf_oracle as orc
dev = torch.device("cuda:0")
sd = syn.make_art_state_dict(seed=2, density_scale=10.0)
prefix = "fine_mlp."
params = {k[len(prefix):]: v.to(dev) for k, v in sd.items() if k.startswith(prefix)}
lib = syn.make_code_library_state(seed=0, n_max_objs=2)
lat_cpu = orc.code_library(lib, torch.tensor([1]), torch.tensor([6]))
lat = {k: v.to(dev) for k, v in lat_cpu.items()}
packed, small = ops.pack_art_mlp(params), ops.art_prepare(params, lat)
n, S = 24, 193
rays = syn.random_rays(n, seed=21)
gen = torch.Generator().manual_seed(21)
t = torch.sort(torch.rand(n, S, generator=gen) * 4 + 2, dim=-1).values
target = torch.rand(n, 3, generator=gen)
o, d, v, tt = (x.to(dev) for x in (rays["rays_o"], rays["rays_d"], rays["viewdirs"], t))
raw = ops.art_mlp_fwd(packed, small, o, d, v, tt)
rgb = ops.composite_raw(raw, tt, d, True, ops.ACT_ARTICULATED)[0]
g_rgb = 2.0 * (rgb - target.to(dev)) / (n * 3)
d_raw = ops.composite_bwd(raw, tt, d, g_rgb, None, None, True, ops.ACT_ARTICULATED, ops.padded_samples(n * S))[: n * S].reshape(n, S, 4).cpu()
def ref(dtype):
    r = raw.cpu().to(dtype).requires_grad_(True)
    c = torch.sigmoid(r[..., :3]) * 1.002 - 0.001
    sg = torch.nn.functional.softplus(r[..., 3:] - 1.0)
    comp = orc.volumetric_rendering(c, sg, t.to(dtype), rays["rays_d"].to(dtype), True)[0]
    (comp * g_rgb.cpu().to(dtype)).sum().backward()
    return r.grad
g64, g32 = ref(torch.float64), ref(torch.float32)
def rel(a, b): return (torch.linalg.norm((a.double() - b.double())) / torch.linalg.norm(b.double())).item()
print("sigma grad: hip vs f64", rel(d_raw[..., 3], g64[..., 3]), " f32-autograd vs f64", rel(g32[..., 3], g64[..., 3]))
print("rgb grad:   hip vs f64", rel(d_raw[..., :3], g64[..., :3]), " f32-autograd vs f64", rel(g32[..., :3], g64[..., :3]))
err = (d_raw[..., 3].double() - g64[..., 3]).abs()
i = err.argmax(); print("worst", divmod(i.item(), S), err.max().item(), g64[..., 3].abs().max().item())
r_, s_ = divmod(i.item(), S)
print("around worst: hip", d_raw[r_, max(0,s_-2):s_+3, 3].tolist(), "f64", g64[r_, max(0,s_-2):s_+3, 3].tolist())
print("raw sigma there", raw.cpu()[r_, max(0,s_-2):s_+3, 3].tolist())
