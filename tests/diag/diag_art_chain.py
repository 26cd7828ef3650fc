import sys, os, torch
import torch.nn.functional as F
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
packed, packed_bwd, small = ops.pack_art_mlp(params), ops.pack_art_mlp_bwd(params), ops.art_prepare(params, lat)
n, S = 24, 193
rays = syn.random_rays(n, seed=21)
gen = torch.Generator().manual_seed(21)
t = torch.sort(torch.rand(n, S, generator=gen) * 4 + 2, dim=-1).values
target = torch.rand(n, 3, generator=gen)
o, d, v, tt = (x.to(dev) for x in (rays["rays_o"], rays["rays_d"], rays["viewdirs"], t))
raw, planes, masks = ops.art_mlp_fwd_train(packed, small, o, d, v, tt)
rgb = ops.composite_raw(raw, tt, d, True, ops.ACT_ARTICULATED)[0]
g_rgb = 2.0 * (rgb - target.to(dev)) / (n * 3)
d_raw = ops.composite_bwd(raw, tt, d, g_rgb, None, None, True, ops.ACT_ARTICULATED, ops.plane_samples(planes))
dplanes, dxp = ops.art_bwd_chain(packed_bwd, small, d_raw, masks, planes)
N = n * S
dt = torch.float64
W = {k[len(prefix):]: v.to(dt).requires_grad_(True) for k, v in sd.items() if k.startswith(prefix)}
L = {k: v.to(dt) for k, v in lat_cpu.items()}
pos = orc.cast_rays(t.to(dt), rays["rays_o"].to(dt), rays["rays_d"].to(dt)).reshape(-1, 3)
shape, app, art = L["density"].expand(N, -1), L["color"].expand(N, -1), L["articulation"].expand(N, -1)
zs = {}
def lin(name, x, tag, relu=True):
    z = F.linear(x, W[name + ".weight"], W[name + ".bias"]); z.retain_grad(); zs[tag] = z
    return F.relu(z) if relu else z
x = torch.cat([pos, shape, art], -1)
for i in range(4): x = lin(f"deformations_linear.{i}", x, f"d{i}")
xd = F.linear(x, W["deformation_layer.weight"], W["deformation_layer.bias"]) + pos; xd.retain_grad(); zs["xd"] = xd
e = orc.pos_enc(xd, 0, 10); e.retain_grad(); zs["enc"] = e
x = torch.cat([e, shape], -1); inputs = x
for i in range(8):
    x = lin(f"pts_linears.{i}", x, f"h{i}")
    if i == 4: x = torch.cat([x, inputs], -1)
sig = F.linear(x, W["density_layer.weight"], W["density_layer.bias"])
bott = lin("bottleneck_layer", x, "bot", relu=False)
venc = orc.pos_enc(rays["viewdirs"].to(dt), 0, 4)[:, None, :].expand(n, S, 27).reshape(-1, 27)
x = torch.cat([bott, venc, app], -1)
for i in range(4): x = lin(f"views_linear.{i}", x, f"v{i}")
rgbraw = F.linear(x, W["rgb_layer.weight"], W["rgb_layer.bias"])
rawo = torch.cat([rgbraw, sig], -1)
rawo.backward(d_raw[:N].cpu().to(dt))   # inject the HIP d_raw so only the chain is compared
def rel(a, b): return (torch.linalg.norm(a.double() - b.double()) / torch.linalg.norm(b.double()).clamp_min(1e-300)).item()
dp = ops.plane_rows_view(dplanes).cpu()
rows = {"v3": 2944 + 384, "v2": 2944 + 256, "v1": 2944 + 128, "v0": 2944, "bot": 2656}
for i in range(8): rows[f"h{i}"] = 608 + 256 * i
for i in range(4): rows[f"d{i}"] = 32 + 128 * i
for tag in ["v3", "v2", "v1", "v0", "bot", "h7", "h6", "h5", "h4", "h3", "h2", "h1", "h0", "d3", "d2", "d1", "d0"]:
    width = zs[tag].shape[1]
    print(tag, f"{rel(dp[rows[tag]: rows[tag] + width, :N].T, zs[tag].grad):.2e}")
print("dxp", f"{rel(dxp[:N, :3].cpu(), zs['xd'].grad):.2e}")
# forward check too
pl = ops.plane_rows_view(planes).cpu()
print("fwd h6", rel(pl[608 + 256 * 6: 608 + 256 * 7, :N].T, F.relu(zs["h6"]).detach()), "fwd xd", rel(pl[3:6, :N].T, zs["xd"].detach()))
