"""Quick A/B of the headline region (BASELINE config 2: vanilla 640x480, 65 + 193 evaluations per ray) without bench.py's other legs:
    [AON_HIP_LIB=...] python tools/headline_bench.py [--steps 3] [--literal] [--tag name]
One JSON line: rays/s, ms per frame, the fused MLP kernel's HIP-event time and its executed / reference-literal fraction of the fp32
matrix peak.  AON_HIP_LIB selects an experiment build of the library (tools/exp_tu.sh)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--literal", action="store_true")
    ap.add_argument("--no-view-bias", action="store_true", help="the view-encoding chunk per sample (round 5's first form) instead of the per-ray view bias")
    ap.add_argument("--tag", default=os.environ.get("AON_HIP_LIB", "default"))
    args = ap.parse_args()
    import aon_amd.synthetic as syn
    from aon_amd import ops
    from aon_amd.models.vanilla_nerf.model import NeRF

    dev = torch.device("cuda:0")
    ops.set_bottleneck_fold(not args.literal)
    ops.set_view_bias(not args.no_view_bias)
    H, W = 480, 640
    model = NeRF().to(dev)
    model.load_state_dict(syn.make_nerf_state_dict(seed=0, density_scale=30.0))
    ro, vd = ops.raygen(syn.look_at_pose(), H, W, syn.focal_from_fovy(H), device=dev)
    rays = {"rays_o": ro, "rays_d": vd, "viewdirs": vd}
    with torch.no_grad():
        out = model(rays, False, True, 2.0, 6.0)
        torch.cuda.synchronize()
        ops.profile_begin()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            model(rays, False, True, 2.0, 6.0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        ms, launches, samples = ops.profile_end()
        del out
    vb = not args.literal and not args.no_view_bias
    mac = 593_408 - (0 if args.literal else 65_536) - (128 * 28 if vb else 0)   # (the chunk runs 14 two-deep steps: 28 columns incl. one of padding)
    ex = samples * mac * 2 / (ms * 1e-3) / 1e12
    lit = samples * 593_408 * 2 / (ms * 1e-3) / 1e12
    print(json.dumps({"tag": args.tag, "fold": not args.literal, "view_bias": vb, "rays_per_s": H * W / dt, "ms_per_frame": dt * 1e3,
                      "mlp_ms_per_launch": ms / launches, "frac_executed": ex / 157.3, "frac_reference_literal": lit / 157.3}))


if __name__ == "__main__":
    main()
