"""BASELINE config 1 literally: Sapien single-scene vanilla NeRF, 320x240, 64 coarse samples only (num_levels=1).
The reference runs it on the CPU; here: the HIP path on cuda:0 and, beside it, the CPU restatement (oracle) on a bounded
sample of the same frame.  Lives under tests/ because its CPU leg imports the oracle."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import aon_amd.synthetic as syn
    from aon_amd import ops
    from aon_amd.models.vanilla_nerf.model import NeRF

    dev = torch.device("cuda:0")
    H, W = 240, 320
    sd = syn.make_nerf_state_dict(seed=0, density_scale=30.0)
    model = NeRF(num_levels=1).to(dev)
    model.load_state_dict(sd)
    ro, vd = ops.raygen(syn.look_at_pose(), H, W, syn.focal_from_fovy(H), device=dev)
    rays = {"rays_o": ro, "rays_d": vd, "viewdirs": vd}
    with torch.no_grad():
        model(rays, False, True, syn.NEAR, syn.FAR)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            out = model(rays, False, True, syn.NEAR, syn.FAR)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
    res = {"workload": "config 1: 320x240, 65 coarse evals/ray, num_levels=1", "hip_rays_per_s": H * W / dt, "hip_ms_per_frame": dt * 1e3}
    if "--no-cpu" not in sys.argv:
        from oracle import nerf_oracle as orc   # CPU leg only

        torch.set_num_threads(min(32, os.cpu_count() or 1))
        n = 3840 * 4
        rc = {k: v[:n].cpu() for k, v in rays.items()}
        with torch.no_grad():
            orc.nerf_forward(sd, {k: v[:3840] for k, v in rc.items()}, False, True, syn.NEAR, syn.FAR, num_levels=1)
            t0 = time.perf_counter()
            ref = [orc.nerf_forward(sd, {k: v[i: i + 3840] for k, v in rc.items()}, False, True, syn.NEAR, syn.FAR, num_levels=1)[0][0]
                   for i in range(0, n, 3840)]
            dtc = time.perf_counter() - t0
        ref = torch.cat(ref)
        mse = torch.mean((out[0][0][:n].cpu() - ref) ** 2).item()
        res.update({"cpu_rays_per_s": n / dtc, "cpu_threads": torch.get_num_threads(), "cpu_sample": f"{n} rays in 3840-ray chunks, {dtc:.1f} s",
                    "psnr_vs_oracle_db": -10.0 * torch.log10(torch.tensor(max(mse, 1e-20))).item(), "speedup": H * W / dt / (n / dtc)})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
