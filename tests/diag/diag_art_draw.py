"""Articulated network, coarse / fine level: HIP compositing backward (act = articulated) vs fp64 on the oracle's fp32 raw values."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import nerf_oracle as orc  # noqa: E402
from conftest import load_golden  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def main():
    import aon_amd.synthetic as syn
    from aon_amd import ops

    dev = torch.device("cuda:0")
    g = load_golden("g18_art_degrees")
    m = 96
    rays = {k: g[k][:m] for k in ("rays_o", "rays_d", "viewdirs")}
    lat = {k: g["lat_" + k] for k in ("density", "color", "articulation")}
    target = syn.seeded_uniform(1899, m, 3)
    tr, u = syn.seeded_uniform(1977, m, 65), syn.seeded_uniform(1978, m, 128)
    gk = dict(min_deg_point=0, max_deg_point=6, deg_view=2)
    sd = syn.make_art_state_dict(seed=18, density_scale=2.0, **gk)
    for rnd in (True, False):
        out, aux = orc.nerf_ae_art_forward(sd, rays, rnd, True, 2.0, 6.0, lat, t_rand=tr, u=u, return_aux=True, **gk)
        for lvl in (0, 1):
            a = aux[lvl]
            t, rr0, rs0 = a["t_vals"].detach(), a["raw_rgb"].detach(), a["raw_sigma"].detach()
            S = t.shape[1]

            def grads(dtype):
                rr, rs = rr0.to(dtype).requires_grad_(True), rs0.to(dtype).requires_grad_(True)
                rgb = torch.sigmoid(rr) * (1 + 2 * 0.001) - 0.001
                comp, acc, w, depth = orc.volumetric_rendering(rgb, F.softplus(rs - 1.0), t.to(dtype), rays["rays_d"].to(dtype), True)
                ((comp - target.to(dtype)) ** 2).mean().backward()
                return comp.detach(), rr.grad, rs.grad

            c64, grr64, grs64 = grads(torch.float64)
            c32, grr32, grs32 = grads(torch.float32)
            g_rgb = (2.0 * (c32 - target) / (3 * m)).float()
            raw4 = torch.cat([rr0, rs0], -1).reshape(m * S, 4).contiguous()
            Np = ops.padded_samples(m * S)
            d_raw = ops.composite_bwd(raw4.to(dev), t.to(dev), rays["rays_d"].to(dev), g_rgb.to(dev), None, None, True, ops.ACT_ARTICULATED, Np)[: m * S].cpu().reshape(m, S, 4)
            for name, gs, gr in (("torch fp32", grs32, grr32), ("HIP", d_raw[..., 3:], d_raw[..., :3])):
                print(f"rnd={rnd} level {lvl} {name:10s}: d_sigma rel L2 {(gs.double() - grs64).norm() / grs64.norm():.2e}  d_rgb rel L2 {(gr.double() - grr64).norm() / grr64.norm():.2e}", flush=True)


if __name__ == "__main__":
    main()
