import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aon_amd.synthetic as syn
from aon_amd import ops
from oracle import nerf_oracle as orc
dev = torch.device("cuda:0")
sd = syn.make_art_state_dict(seed=0, density_scale=30.0)
prefix = "fine_mlp."
params = {k[len(prefix):]: v.to(dev) for k, v in sd.items() if k.startswith(prefix)}
lat_cpu = orc.code_library(syn.make_code_library_state(0, 2), torch.tensor([1]), torch.tensor([3]))
lat = {k: v.to(dev) for k, v in lat_cpu.items()}
packed, small = ops.pack_art_mlp(params), ops.art_prepare(params, lat)
for n, S in ((6, 65), (1, 65), (2, 64), (40, 65), (300, 193)):
    rays = syn.random_rays(n, seed=n)
    t = torch.sort(torch.rand(n, S, generator=torch.Generator().manual_seed(n)) * 4 + 2, dim=-1).values
    pos = orc.cast_rays(t, rays["rays_o"], rays["rays_d"]); venc = orc.pos_enc(rays["viewdirs"], 0, 4)
    rgb_o, sig_o = orc.art_mlp(sd, prefix, pos, venc, lat_cpu)
    for rep in range(2):
        a = ops.art_mlp_fwd(packed, small, rays["rays_o"].to(dev), rays["rays_d"].to(dev), rays["viewdirs"].to(dev), t.to(dev)).cpu()
        b = ops.art_mlp_fwd_pos(packed, small, pos.to(dev), venc.to(dev)).cpu()
        print(n, S, rep, "fused err", (a[..., :3] - rgb_o).abs().max().item(), "pos err", (b[..., :3] - rgb_o).abs().max().item())
