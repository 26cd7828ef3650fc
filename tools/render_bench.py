"""Render throughput of the articulated path (BASELINE config 4: sapien_multi / vanilla_autodecoder, 320x240, 1 GPU)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import aon_amd.synthetic as syn
    from aon_amd import ops
    from aon_amd.models.vanilla_nerf.model_autodecoder import NeRF_AE_Art
    import types

    from aon_amd.models.code_library import CodeLibraryArticulated

    dev = torch.device("cuda:0")
    H, W = (240, 320) if "--full" not in sys.argv else (480, 640)
    model = NeRF_AE_Art().to(dev)
    model.load_state_dict(syn.make_art_state_dict(seed=0, density_scale=30.0))
    lib = CodeLibraryArticulated(types.SimpleNamespace(N_max_objs=1, N_obj_code_length=128)).to(dev)
    lib.load_state_dict(syn.make_code_library_state(0, 1))
    with torch.no_grad():
        lat = lib({"instance_id": torch.tensor([0], device=dev), "articulation_id": torch.tensor([3], device=dev)})
    ro, vd = ops.raygen(syn.look_at_pose(), H, W, syn.focal_from_fovy(H), device=dev)
    rays = {"rays_o": ro, "rays_d": vd, "viewdirs": vd}
    with torch.no_grad():
        model(rays, False, True, 2.0, 6.0, lat)
        torch.cuda.synchronize()
        ops.profile_begin()
        t0 = time.perf_counter()
        steps = 3
        for _ in range(steps):
            model(rays, False, True, 2.0, 6.0, lat)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ms, launches, samples = ops.profile_end()
    flop_lit = 1_589_760  # reference-literal FLOP per sample, latent columns included
    print(json.dumps({"workload": f"articulated render {W}x{H}", "rays_per_s": H * W / dt, "ms_per_frame": dt * 1e3,
                      "mlp_kernel_tflops_reference_literal": samples * flop_lit / (ms * 1e-3) / 1e12,
                      "mlp_kernel_frac_of_fp32_matrix_peak": samples * flop_lit / (ms * 1e-3) / 1e12 / 157.3}))


if __name__ == "__main__":
    main()
