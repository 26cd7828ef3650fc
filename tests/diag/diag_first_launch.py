import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aon_amd.synthetic as syn
from aon_amd import ops
from oracle import nerf_oracle as orc
dev = torch.device("cuda:0")
sd = syn.make_nerf_state_dict(seed=0, density_scale=30.0)
params = {k[len("fine_mlp."):]: v.to(dev) for k, v in sd.items() if k.startswith("fine_mlp.")}
packed = ops.pack_vanilla_mlp(params)
for n, S in ((1, 65), (1, 65), (40, 65), (1, 65), (2000, 65), (1, 65)):
    rays = syn.random_rays(n, seed=7)
    t = torch.sort(torch.rand(n, S, generator=torch.Generator().manual_seed(7)) * 4 + 2, dim=-1).values
    enc = orc.pos_enc(orc.cast_rays(t, rays["rays_o"], rays["rays_d"]), 0, 10); venc = orc.pos_enc(rays["viewdirs"], 0, 4)
    rgb_o, sig_o = orc.nerf_mlp(sd, "fine_mlp.", enc, venc)
    raw = ops.mlp_fwd(packed, rays["rays_o"].to(dev), rays["rays_d"].to(dev), rays["viewdirs"].to(dev), t.to(dev)).cpu()
    err = (raw[..., :3] - rgb_o).abs()
    print(n, S, "max err", err.max().item(), "per-128-sample-pass max:", [round(x, 5) for x in err.reshape(-1, 3).max(1).values.split(128)[0:0]] , "first/last sample err", err.reshape(-1,3)[0].max().item(), err.reshape(-1,3)[-1].max().item())
print("---- enc variant")
for n, S in ((1, 65), (1, 65), (8, 65), (40, 65), (2000, 65), (1, 65), (8, 65)):
    rays = syn.random_rays(n, seed=7)
    t = torch.sort(torch.rand(n, S, generator=torch.Generator().manual_seed(7)) * 4 + 2, dim=-1).values
    enc = orc.pos_enc(orc.cast_rays(t, rays["rays_o"], rays["rays_d"]), 0, 10); venc = orc.pos_enc(rays["viewdirs"], 0, 4)
    rgb_o, sig_o = orc.nerf_mlp(sd, "fine_mlp.", enc, venc)
    raw = ops.mlp_fwd_enc(packed, enc.to(dev), venc.to(dev)).cpu()
    err = (raw[..., :3] - rgb_o).abs().reshape(-1, 3).max(1).values
    bad = (err > 1e-4).nonzero().flatten()
    print(n, S, "max err", err.max().item(), "n bad samples", bad.numel(), "of", err.numel(), "first bad idx", bad[:8].tolist(), "bad mod 32:", sorted(set((bad % 32).tolist()))[:40])
