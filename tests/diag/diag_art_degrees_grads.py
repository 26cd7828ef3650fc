"""Gradient accuracy of the articulated network at several encoding degrees (fp64-truth yardstick), incl. the default (0, 10, 4) on the
same weights seed: is a ratio above 5x specific to other degrees?   python tests/diag/diag_art_degrees_grads.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import nerf_oracle as orc  # noqa: E402
from _gradcheck import rel_l2  # noqa: E402
from conftest import load_golden  # noqa: E402


def main():
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model_autodecoder import NeRF_AE_Art

    dev = torch.device("cuda:0")
    g = load_golden("g18_art_degrees")
    rays_cpu = {k: g[k][:96] for k in ("rays_o", "rays_d", "viewdirs")}
    lat_cpu = {k: g["lat_" + k] for k in ("density", "color", "articulation")}
    m = 96
    target = syn.seeded_uniform(1899, m, 3)
    tr, u = syn.seeded_uniform(1977, m, 65), syn.seeded_uniform(1978, m, 128)
    for gk in (dict(min_deg_point=0, max_deg_point=10, deg_view=4), dict(min_deg_point=0, max_deg_point=6, deg_view=2),
               dict(min_deg_point=-1, max_deg_point=9, deg_view=4), dict(min_deg_point=2, max_deg_point=5, deg_view=0)):
        for rnd in (True, False):
            sd = syn.make_art_state_dict(seed=18, density_scale=2.0, **gk)
            model = NeRF_AE_Art(**gk).to(dev)
            model.load_state_dict(sd)

            def oracle_grads(dtype):
                sd_o = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in sd.items()}
                lo = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in lat_cpu.items()}
                out = orc.nerf_ae_art_forward(sd_o, {k: v.to(dtype) for k, v in rays_cpu.items()}, rnd, True, 2.0, 6.0, lo, t_rand=tr.to(dtype), u=u.to(dtype), **gk)
                (orc.img2mse(out[0][0], target.to(dtype)) + orc.img2mse(out[1][0], target.to(dtype))).backward()
                gr = {k: v.grad for k, v in sd_o.items()}
                gr.update({f"latent[{k}]": v.grad for k, v in lo.items()})
                return gr

            truth, ref32 = oracle_grads(torch.float64), oracle_grads(torch.float32)
            lg = {k: v.to(dev).clone().requires_grad_(True) for k, v in lat_cpu.items()}
            out = model({k: v.to(dev) for k, v in rays_cpu.items()}, rnd, True, 2.0, 6.0, lg, t_rand=tr.to(dev), u=u.to(dev))
            (((out[0][0] - target.to(dev)) ** 2).mean() + ((out[1][0] - target.to(dev)) ** 2).mean()).backward()
            hip = {name: p.grad.cpu() for name, p in model.named_parameters()}
            hip.update({f"latent[{k}]": v.grad.cpu() for k, v in lg.items()})
            rows = sorted(((rel_l2(h, truth[n]) / max(rel_l2(ref32[n], truth[n]), 1e-30), rel_l2(h, truth[n]), rel_l2(ref32[n], truth[n]), n) for n, h in hip.items()
                           if rel_l2(h, truth[n]) > 1e-4), reverse=True)
            print(f"{tuple(gk.values())} randomized={rnd}: above 1e-4: {len(rows)}; top:", [(round(r, 1), f"{a:.1e}", f"{b:.1e}", n) for r, a, b, n in rows[:4]], flush=True)


if __name__ == "__main__":
    main()
