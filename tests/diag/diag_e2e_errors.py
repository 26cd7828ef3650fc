"""How far is the HIP end-to-end render from the reference's own outputs?  Smooth fixture (G15, every ray) and the sharp
x30 fixtures (G8 vanilla, G11 articulated; all rays and far-plane-robust rays)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import aon_amd.synthetic as syn  # noqa: E402
from aon_amd.models.vanilla_nerf.model import NeRF  # noqa: E402
from aon_amd.models.vanilla_nerf.model_autodecoder import NeRF_AE_Art  # noqa: E402
from oracle import nerf_oracle as orc  # noqa: E402

dev = torch.device("cuda:0")


def G(name):
    z = np.load(f"tests/golden/{name}.npz")
    return {k: (torch.from_numpy(z[k]) if z[k].ndim else z[k].item()) for k in z.files}


def rep(tag, out, g, prefix, ok=None):
    for lvl, name in ((0, "coarse"), (1, "fine")):
        parts = []
        for j, q in enumerate(("rgb", "acc", "depth")):
            d = (out[lvl][j].cpu() - g[f"{prefix}_{name}_{q}"]).abs()
            if ok is not None:
                d = d[ok]
            parts.append(f"{q} {float(d.max()):.2e}")
        print(f"{tag:<34} {name:<6} " + "  ".join(parts))


g = G("g15_smooth")
m = NeRF().to(dev)
m.load_state_dict(syn.make_smooth_nerf_state_dict())
rays = {k: g[k].to(dev) for k in ("rays_o", "rays_d", "viewdirs")}
with torch.no_grad():
    rep("smooth vanilla det", m(rays, False, True, 2.0, 6.0), g, "van_det")
    rep("smooth vanilla rnd", m(rays, True, False, 2.0, 6.0, t_rand=g["t_rand"].to(dev), u=g["u"].to(dev)), g, "van_rnd")
a = NeRF_AE_Art().to(dev)
a.load_state_dict(syn.make_art_state_dict(seed=5, density_scale=2.0))
arays = {k: g["art_" + k].to(dev) for k in ("rays_o", "rays_d", "viewdirs")}
lat = {k: g["art_lat_" + k].to(dev) for k in ("density", "color", "articulation")}
with torch.no_grad():
    rep("smooth articulated det", a(arays, False, True, 2.0, 6.0, lat), g, "art_det")

g8 = G("g8_nerf_forward")
sd = syn.make_nerf_state_dict(seed=0, density_scale=30.0)
m.load_state_dict(sd)
rays_cpu = {k: g8[k] for k in ("rays_o", "rays_d", "viewdirs")}
rays = {k: v.to(dev) for k, v in rays_cpu.items()}
_, aux = orc.nerf_forward(sd, rays_cpu, False, True, 2.0, 6.0, return_aux=True)
ok = torch.ones(rays_cpu["rays_o"].shape[0], dtype=torch.bool)
for x in aux:
    ok &= x["raw_sigma"][:, -1, 0].abs() > 2e-2
with torch.no_grad():
    out = m(rays, False, True, 2.0, 6.0)
rep("sharp x30 vanilla det, all rays", out, g8, "det")
rep(f"sharp x30 vanilla det, robust {int(ok.sum())}/{ok.numel()}", out, g8, "det", ok)
