"""Is a fine-level difference of a few 1e-4 between the HIP render and the fp32 oracle conditioning or a bug?  Compare
both with the oracle evaluated in fp64 on the failing sweep cases (tests/test_hip_fuzz.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import aon_amd.synthetic as syn
from aon_amd.models.vanilla_nerf.model import NeRF
from oracle import nerf_oracle as orc
from test_hip_fuzz import _rays

dev = torch.device("cuda:0")
for seed, n in ((10, 129), (18, 513)):
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    white, unit = bool(seed & 2), seed % 5 != 4
    near, far = ((2.0, 6.0), (1.5, 7.0), (2.5, 5.5))[seed % 3]
    scale = (30.0, 5.0, 60.0)[(seed // 3) % 3]
    rays_cpu = _rays(n, rng, unit)
    sd = syn.make_nerf_state_dict(seed=seed, density_scale=scale)
    model = NeRF().to(dev); model.load_state_dict(sd)
    with torch.no_grad():
        out = model({k: v.to(dev) for k, v in rays_cpu.items()}, False, white, near, far)
    o32 = orc.nerf_forward(sd, rays_cpu, False, white, near, far)
    o64 = orc.nerf_forward({k: v.double() for k, v in sd.items()}, {k: v.double() for k, v in rays_cpu.items()}, False, white, near, far)
    for lvl in (0, 1):
        h = out[lvl][0].cpu().double(); a = o32[lvl][0].double(); b = o64[lvl][0]
        e_h32, e_h64, e_3264 = (h - a).abs().max(-1).values, (h - b).abs().max(-1).values, (a - b).abs().max(-1).values
        i = int(e_h32.argmax())
        print(f"seed {seed} level {lvl}: max|hip-o32| {e_h32.max():.2e} (ray {i}: hip-o64 {e_h64[i]:.2e}, o32-o64 {e_3264[i]:.2e}); "
              f"max|hip-o64| {e_h64.max():.2e}; max|o32-o64| {e_3264.max():.2e}")
