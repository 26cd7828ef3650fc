"""The layer-wise engine (csrc/aon_gmlp.hip) measured: NeRFMLP.forward, whole-path render and a training step for a few constructor
geometries, HIP-event timing on torch's stream (the engine launches there), one JSON line each.  On the default geometry the
fused kernels are timed beside it -- the price of generality.

    python tools/general_engine_bench.py [--rays 4096] [--reps 5]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK = 157.3   # TFLOP/s, fp32 matrix (MI355X_MICROARCH.md)


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]


def macs_per_sample(geom):
    W, Wc, P, V = geom.netwidth, geom.netwidth_condition, geom.pos_size, geom.view_pos_size
    m = 0
    for l in range(geom.netdepth):
        m += W * (P if l == 0 else (W + P if geom.cat_before(l) else W))
    m += W * W + W * geom.num_density_channels + Wc * (W + V) + (geom.netdepth_condition - 1) * Wc * Wc + geom.num_rgb_channels * Wc
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import aon_amd.synthetic as syn
    from aon_amd import ops
    from aon_amd.models.vanilla_nerf.model import NeRF

    dev = torch.device("cuda:0")
    n = args.rays
    frame = syn.make_rays(64, n // 64, syn.look_at_pose(), syn.focal_from_fovy(64))
    rays = {k: v[:n].to(dev) for k, v in frame.items()}
    target = torch.rand(n, 3, device=dev)
    geoms = {"default (0,10,4) 8x256 + 1x128": dict(),
             "degrees (0,6,2)": dict(min_deg_point=0, max_deg_point=6, deg_view=2),
             "degrees (0,16,4)": dict(min_deg_point=0, max_deg_point=16, deg_view=4)}
    for name, gk in geoms.items():
        geom = ops.MlpGeometry(**gk)
        mac = macs_per_sample(geom)
        sd = syn.make_general_nerf_state_dict(7, **gk)
        pc = {k[11:]: v.to(dev) for k, v in sd.items() if k.startswith("coarse_mlp.")}
        pf = {k[9:]: v.to(dev) for k, v in sd.items() if k.startswith("fine_mlp.")}
        # stage level: NeRFMLP.forward on 193 samples per ray
        S = 193
        x = torch.rand(n, S, geom.pos_size, device=dev) * 2 - 1
        v = torch.rand(n, geom.view_pos_size, device=dev) * 2 - 1
        ms = timeit(lambda: ops.gmlp_fwd(geom, pf, x, v), args.reps)
        fl = 2.0 * mac * n * S
        print(json.dumps({"what": "NeRFMLP.forward, layer-wise engine", "geometry": name, "samples": n * S, "ms": round(ms, 3),
                          "tflops": round(fl / ms / 1e9, 2), "frac_fp32_matrix_peak": round(fl / ms / 1e9 / PEAK, 4)}), flush=True)
        if geom.is_default:
            packed = ops.pack_vanilla_mlp(pf)
            ms_f = timeit(lambda: ops.mlp_fwd_enc(packed, x, v), args.reps)
            print(json.dumps({"what": "NeRFMLP.forward, fused kernel (caller-encoded inputs)", "geometry": name, "samples": n * S, "ms": round(ms_f, 3),
                              "tflops": round(fl / ms_f / 1e9, 2), "frac_fp32_matrix_peak": round(fl / ms_f / 1e9 / PEAK, 4)}), flush=True)
        # whole path + training step through the module
        model = NeRF(**gk).to(dev)
        model.load_state_dict(sd)
        if geom.is_default:
            model._general = True      # force the layer-wise engine on the default geometry (measurement only)
        fused_flag = getattr(model, "_fused_inference", False)   # what the constructor chose
        fused_inf = fused_flag and not geom.is_default
        model._fused_inference = False
        with torch.no_grad():
            ms = timeit(lambda: model(rays, False, True, 2.0, 6.0), args.reps)
        fl = 2.0 * mac * n * 258
        print(json.dumps({"what": "NeRF.forward (65 + 193), layer-wise engine", "geometry": name, "rays": n, "ms": round(ms, 3), "rays_per_s": round(n / ms * 1e3),
                          "frac_fp32_matrix_peak": round(fl / ms / 1e9 / PEAK, 4)}), flush=True)
        if fused_inf:   # up to 10 / 4 frequency levels: inference also runs on the fused kernels (padded encodings)
            model._fused_inference = True
            with torch.no_grad():
                ms = timeit(lambda: model(rays, False, True, 2.0, 6.0), args.reps)
            print(json.dumps({"what": "NeRF.forward (65 + 193), fused kernels on padded encodings", "geometry": name, "rays": n, "ms": round(ms, 3),
                              "rays_per_s": round(n / ms * 1e3), "frac_fp32_matrix_peak": round(fl / ms / 1e9 / PEAK, 4)}), flush=True)
        opt = torch.optim.Adam(model.parameters(), lr=5e-4)
        if fused_inf:
            def fstep():
                opt.zero_grad(set_to_none=True)
                out = model(rays, True, True, 2.0, 6.0)
                (((out[0][0] - target) ** 2).mean() + ((out[1][0] - target) ** 2).mean()).backward()
                opt.step()
            ms = timeit(fstep, args.reps)
            print(json.dumps({"what": "training step (fwd + bwd + Adam), fused kernels on padded encodings", "geometry": name, "rays": n, "ms": round(ms, 3),
                              "rays_per_s": round(n / ms * 1e3), "frac_fp32_matrix_peak": round(3 * fl / ms / 1e9 / PEAK, 4)}), flush=True)
            model._fused_inference = False

        def step():
            opt.zero_grad(set_to_none=True)
            out = model(rays, True, True, 2.0, 6.0)
            loss = ((out[0][0] - target) ** 2).mean() + ((out[1][0] - target) ** 2).mean()
            loss.backward()
            opt.step()

        ms = timeit(step, args.reps)
        print(json.dumps({"what": "training step (fwd + bwd + Adam), layer-wise engine", "geometry": name, "rays": n, "ms": round(ms, 3),
                          "rays_per_s": round(n / ms * 1e3), "frac_fp32_matrix_peak": round(3 * fl / ms / 1e9 / PEAK, 4)}), flush=True)
        if geom.is_default:
            # back to what the constructor chose: BOTH switches (round 3 left _fused_inference off here, so its "fused kernels"
            # rows of the default geometry were the layer-wise engine again: VERDICT r3)
            model._general = False
            model._fused_inference = fused_flag
            assert fused_flag, "the default geometry runs on the fused kernels"
            with torch.no_grad():
                ms = timeit(lambda: model(rays, False, True, 2.0, 6.0), args.reps)
            print(json.dumps({"what": "NeRF.forward (65 + 193), fused kernels", "geometry": name, "rays": n, "ms": round(ms, 3), "rays_per_s": round(n / ms * 1e3),
                              "frac_fp32_matrix_peak": round(fl / ms / 1e9 / PEAK, 4)}), flush=True)
            ms = timeit(step, args.reps)
            print(json.dumps({"what": "training step (fwd + bwd + Adam), fused kernels", "geometry": name, "rays": n, "ms": round(ms, 3),
                              "rays_per_s": round(n / ms * 1e3), "frac_fp32_matrix_peak": round(3 * fl / ms / 1e9 / PEAK, 4)}), flush=True)


if __name__ == "__main__":
    main()
