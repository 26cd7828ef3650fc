"""Measured distances behind the tolerances of the GPU tests that were loose in round 1: articulated end-to-end (G11) and the
gradients against the reference's own autograd (G9)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import aon_amd.synthetic as syn  # noqa: E402
from aon_amd.models.vanilla_nerf.model import NeRF  # noqa: E402
from aon_amd.models.vanilla_nerf.model_autodecoder import NeRF_AE_Art  # noqa: E402

dev = torch.device("cuda:0")


def G(name):
    z = np.load(f"tests/golden/{name}.npz")
    return {k: (torch.from_numpy(z[k]) if z[k].ndim else z[k].item()) for k in z.files}


g = G("g11_nerf_ae_art")
art_sd = syn.make_art_state_dict(seed=0, density_scale=30.0)
model = NeRF_AE_Art().to(dev)
model.load_state_dict(art_sd)
rays = {k: g[k].to(dev) for k in ("rays_o", "rays_d", "viewdirs")}
lat = lambda tag: {k: g[f"lat_{tag}_{k}"].to(dev) for k in ("density", "color", "articulation")}
for tag, lt, kw, draws in (("det", "train", (False, True), {}), ("tst_nowb", "test", (False, False), {}),
                           ("rnd", "train", (True, True), dict(t_rand=g["t_rand"].to(dev), u=g["u"].to(dev)))):
    with torch.no_grad():
        out = model(rays, kw[0], kw[1], g["near"], g["far"], lat(lt), **draws)
    for lvl, name in ((0, "coarse"), (1, "fine")):
        parts = []
        for j, q in enumerate(("rgb", "acc", "depth")):
            key = f"{tag}_{name}_{q}"
            if key in g:
                d = (out[lvl][j].cpu() - g[key]).abs()
                parts.append(f"{q} max {float(d.max()):.2e} p99 {float(d.flatten().kthvalue(max(1, int(0.99 * d.numel()))).values):.2e}")
        print(f"G11 {tag:<9} {name:<6} " + "  ".join(parts))

g9 = G("g9_backward")
rays9 = {k: g9[k].to(dev) for k in ("rays_o", "rays_d", "viewdirs")}
target = g9["target"].to(dev)


def grads(prefix, named):
    worst = {}
    for name, p in named:
        gr = p.grad.detach().reshape(-1).cpu()
        ref_norm = g9[f"{prefix}|{name}|norm"]
        rms = ref_norm / gr.numel() ** 0.5
        err_norm = abs(gr.double().norm().item() - ref_norm) / max(ref_norm, 1e-12)
        val = g9[f"{prefix}|{name}|val"].double()
        err_val = (gr[g9[f"{prefix}|{name}|idx"]].double() - val).norm().item() / max(val.norm().item(), 48 ** 0.5 * rms, 1e-12)
        grp = name.split(".")[0] + ("" if prefix == "vanilla" else "." + name.split(".")[1].rstrip("0123456789"))
        w = worst.setdefault(grp, [0.0, 0.0])
        w[0], w[1] = max(w[0], err_norm), max(w[1], err_val)
    for k, v in worst.items():
        print(f"G9 {prefix:<8} {k:<40} norm err {v[0]:.2e}   sampled-entry rel L2 {v[1]:.2e}")


m = NeRF().to(dev)
m.load_state_dict(syn.make_nerf_state_dict(seed=0, density_scale=30.0))
out = m(rays9, False, True, g9["near"], g9["far"])
loss = torch.mean((out[0][0] - target) ** 2) + torch.mean((out[1][0] - target) ** 2)
loss.backward()
print("vanilla loss diff", abs(loss.item() - g9["vanilla_loss"]))
grads("vanilla", m.named_parameters())
am = NeRF_AE_Art().to(dev)
am.load_state_dict(art_sd)
l9 = {k: g[f"lat_train_{k}"].to(dev).requires_grad_(True) for k in ("density", "color", "articulation")}
out = am(rays9, False, True, g9["near"], g9["far"], l9)
loss = torch.mean((out[0][0] - target) ** 2) + torch.mean((out[1][0] - target) ** 2)
loss.backward()
print("art loss diff", abs(loss.item() - g9["art_loss"]))
grads("art", am.named_parameters())
for k, v in l9.items():
    ref = g9[f"art_latgrad_{k}"]
    print(f"G9 latent {k:<14} max err / max ref {float((v.grad.cpu() - ref).abs().max() / ref.abs().max()):.2e}")
