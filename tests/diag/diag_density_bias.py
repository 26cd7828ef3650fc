"""Where does the error of a density-bias gradient (sum over all samples of dL/d raw_sigma) come from?  Seed of the training fuzz
(tests/test_hip_fuzz.py), coarse level: the compositing backward of the HIP path against an fp64 evaluation of
volumetric_rendering's backward ON THE SAME fp32 INPUTS, beside torch's own fp32 autograd of the same function.

    python tests/diag/diag_density_bias.py 112
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import nerf_oracle as orc  # noqa: E402
import test_hip_fuzz as tf  # noqa: E402


def main():
    import aon_amd.synthetic as syn
    from aon_amd import ops

    dev = torch.device("cuda:0")
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 112
    rng = np.random.Generator(np.random.PCG64(5000 + seed))
    n = int(rng.integers(8, 160))
    nc, nf = int(rng.integers(2, 121)), int(rng.integers(1, 301))
    lindisp, white = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    noise_std = float(rng.choice([0.0, 0.4]))
    assert seed % 4 != 3, "vanilla seeds only"
    rays_cpu = tf._rays(n, rng, True)
    g = torch.Generator().manual_seed(seed)
    target = torch.rand(n, 3, generator=g)
    draws = dict(t_rand=torch.rand(n, nc + 1, generator=g), u=torch.rand(n, nf, generator=g))
    noise = [torch.rand(n, nc + 1, generator=g), torch.rand(n, nc + 1 + nf, generator=g)]
    kw = dict(num_coarse_samples=nc, num_fine_samples=nf, lindisp=lindisp, noise_std=noise_std)
    gk = [dict(), dict(min_deg_point=0, max_deg_point=7, deg_view=3), dict(min_deg_point=1, max_deg_point=12, deg_view=5)][seed % 3]
    sd = syn.make_general_nerf_state_dict(6000 + seed, **gk)
    print(f"seed {seed}: n {n}, {kw}, {gk}, white {white}")
    out, aux = orc.nerf_forward(sd, rays_cpu, True, white, 2.0, 6.0, return_aux=True, noise=noise, **draws, **kw, **gk)
    for lvl in (0, 1):
        a = aux[lvl]
        t, raw_rgb, raw_sig = a["t_vals"].detach(), a["raw_rgb"].detach(), a["raw_sigma"].detach()   # raw_sigma: after the noise
        S = t.shape[1]

        def grads(dtype):
            rr, rs = raw_rgb.to(dtype).requires_grad_(True), raw_sig.to(dtype).requires_grad_(True)
            comp, acc, w, depth = orc.volumetric_rendering(torch.sigmoid(rr), torch.relu(rs), t.to(dtype), rays_cpu["rays_d"].to(dtype), white)
            loss = ((comp - target.to(dtype)) ** 2).mean()
            loss.backward()
            return comp.detach(), rr.grad, rs.grad

        comp64, grr64, grs64 = grads(torch.float64)
        comp32, grr32, grs32 = grads(torch.float32)
        g_rgb = (2.0 * (comp32 - target) / (3 * n)).float()
        raw4 = torch.cat([raw_rgb, raw_sig], -1).reshape(n * S, 4).contiguous()
        Np = ops.padded_samples(n * S)
        d_raw = ops.composite_bwd(raw4.to(dev), t.to(dev), rays_cpu["rays_d"].to(dev), g_rgb.to(dev), None, None, white, ops.ACT_VANILLA, Np)
        d_raw = d_raw[: n * S].cpu().reshape(n, S, 4)
        hs, hr = d_raw[..., 3:], d_raw[..., :3]
        tot = grs64.sum().item()
        print(f" level {lvl}: S {S}; sum d_sigma truth {tot:.6e}; sum |d_sigma| {grs64.abs().sum().item():.3e} (cancellation x{grs64.abs().sum().item() / abs(tot):.0f})")
        for name, gs, gr in (("torch fp32 autograd", grs32, grr32), ("HIP composite_bwd", hs, hr)):
            es = (gs.double() - grs64)
            print(f"   {name:20s}: per-sample d_sigma rel L2 {es.norm() / grs64.norm():.2e}, max abs {es.abs().max():.2e};  bias (sum) rel err "
                  f"{abs(gs.double().sum().item() - tot) / abs(tot):.2e};  d_rgb rel L2 {(gr.double() - grr64).norm() / grr64.norm():.2e}")
        # where along the ray the HIP error sits
        es = (hs.double() - grs64).abs().squeeze(-1)
        et = (grs32.double() - grs64).abs().squeeze(-1)
        idx = torch.argsort(es.sum(0), descending=True)[:5]
        print("   sample indices with the largest summed |err| (HIP):", idx.tolist(), [f"{es.sum(0)[i]:.1e}" for i in idx], " torch there:", [f"{et.sum(0)[i]:.1e}" for i in idx])
        rays_bad = torch.argsort(es.sum(1), descending=True)[:3]
        for r in rays_bad.tolist():
            j = int(es[r].argmax())
            print(f"   ray {r}: worst sample {j}: truth {grs64[r, j, 0]:.6e} hip {hs[r, j, 0]:.6e} torch {grs32[r, j, 0]:.6e}; sigma {raw_sig[r, j, 0]:.4f} "
                  f"delta {(t[r, min(j + 1, S - 1)] - t[r, j]):.4e}; sum signed err over the ray: hip {(hs.double() - grs64)[r].sum():.2e} torch {(grs32.double() - grs64)[r].sum():.2e}")


if __name__ == "__main__":
    main()
