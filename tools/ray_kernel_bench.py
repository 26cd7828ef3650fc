"""Per-launch timing of the per-ray (non-GEMM) kernels with HIP events, against their algorithmic HBM bytes:

    python tools/ray_kernel_bench.py [--rays 61440,307200] [--reps 20]

One JSON line per (kernel, size): us per launch, TB/s on the algorithmic bytes, fraction of the 8 TB/s HBM peak.  Inputs are
the real intermediate buffers of a vanilla render of the synthetic scene (raw records out of the MLP kernel), so the data-dependent
paths (guarded scan, rank merge) run as they do in the product."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK = 8.0e12
# algorithmic bytes per ray
B_COMP_C = 65 * (16 + 4) + 12 + 20 + 65 * 4        # records + t + dir in; outputs + weights out
B_COMP_C_NOW = 65 * (16 + 4) + 12 + 20             # ... without the weights
B_COMP_F = 193 * (16 + 4) + 12 + 20
B_PDF = 65 * 4 + 63 * 4 + 193 * 4                  # t_coarse + weights in, t_fine out (u is one shared 512-byte row)
B_FUSED = 65 * (16 + 4) + 12 + 20 + 193 * 4        # records + t + dir in; outputs + t_fine out


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    torch.cuda._sleep(40_000_000)   # ~20 ms of GPU spin: the host enqueues all reps behind it, so no launch gap sits between events
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2] * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", default="61440,307200")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--tag", default=os.environ.get("AON_HIP_LIB", "product"))
    ap.add_argument("--only", default="", help="substring of the kernel label to run ('=label' for an exact match)")
    args = ap.parse_args()
    import aon_amd.synthetic as syn
    from aon_amd import ops

    dev = torch.device("cuda:0")
    H, W = 480, 640
    ro, vd = ops.raygen(syn.look_at_pose(), H, W, syn.focal_from_fovy(H), device=dev)
    sd = {k: v.to(dev) for k, v in syn.make_nerf_state_dict(seed=0, density_scale=30.0).items()}
    pc = ops.pack_vanilla_mlp({k[len("coarse_mlp."):]: v for k, v in sd.items() if k.startswith("coarse_mlp.")})
    pf = ops.pack_vanilla_mlp({k[len("fine_mlp."):]: v for k, v in sd.items() if k.startswith("fine_mlp.")})
    for n in [int(x) for x in args.rays.split(",")]:
        o, d = ro[:n].contiguous(), vd[:n].contiguous()
        t_c, _ = ops.sample_along_rays(o, d, 64, 2.0, 6.0, want_coords=False)
        raw_c = ops.mlp_fwd(pc, o, d, d, t_c)
        _, _, w_c, _ = ops.composite_raw(raw_c, t_c, d, True, ops.ACT_VANILLA)
        t_f = ops.sample_pdf_t(t_c, w_c)
        raw_f = ops.mlp_fwd(pf, o, d, d, t_f)
        u_rand = torch.rand(n, 128, device=dev)

        def emit(name, fn, bytes_per_ray):
            if args.only and (args.only[1:] != name if args.only.startswith("=") else args.only not in name):
                return
            us = timeit(fn, args.reps)
            tbs = n * bytes_per_ray / (us * 1e-6) / 1e12
            print(json.dumps({"tag": args.tag, "kernel": name, "rays": n, "us": round(us, 2), "MB": round(n * bytes_per_ray / 1e6, 1),
                              "TBps": round(tbs, 3), "frac_hbm": round(tbs * 1e12 / HBM_PEAK, 3)}), flush=True)

        emit("sample_along_rays S=65", lambda: ops.sample_along_rays(o, d, 64, 2.0, 6.0, want_coords=False), 65 * 4)
        emit("composite S=65 (+weights)", lambda: ops.composite_raw(raw_c, t_c, d, True, ops.ACT_VANILLA), B_COMP_C)
        emit("composite S=65 (no weights)", lambda: ops.composite_raw(raw_c, t_c, d, True, ops.ACT_VANILLA, want_weights=False), B_COMP_C_NOW)
        emit("composite S=193", lambda: ops.composite_raw(raw_f, t_f, d, True, ops.ACT_VANILLA, want_weights=False), B_COMP_F)
        emit("composite S=193 articulated act", lambda: ops.composite_raw(raw_f, t_f, d, True, ops.ACT_ARTICULATED, want_weights=False), B_COMP_F)
        emit("sample_pdf (shared u)", lambda: ops.sample_pdf_t(t_c, w_c), B_PDF)
        emit("sample_pdf (per-ray u)", lambda: ops.sample_pdf_t(t_c, w_c, u_rand), B_PDF + 512)
        emit("composite_pdf fused (shared u)", lambda: ops.composite_pdf(raw_c, t_c, d, True, ops.ACT_VANILLA), B_FUSED)
        emit("composite_pdf fused (per-ray u)", lambda: ops.composite_pdf(raw_c, t_c, d, True, ops.ACT_VANILLA, u_rand), B_FUSED + 512)
        del raw_c, raw_f, t_f, w_c
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
