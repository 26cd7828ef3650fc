"""Per-kernel timing of the training / render kernels with HIP events on the launch stream (no profiler needed):

    python tools/kernel_bench.py [--rays 4096] [--reps 5] [--only fwd_train,art_fwd_train,...]
    AON_HIP_LIB=articulated-object-nerf_amd/libaon_hip_x.so python tools/kernel_bench.py     # an experiment build

One JSON line per kernel: ms per launch, executed and reference-literal TFLOP/s.  Sizes default to one level pair of a
4096-ray training step (65 and 193 samples per ray)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import os as _os

_FOLD = 0 if _os.environ.get("AON_BOTTLENECK_FOLD", "") == "0" else 65_536   # round 5: bottleneck_layer folded into views_linear[0] (default)
VAN_MAC_LIT = 593_408             # reference-literal MACs / sample, vanilla
VAN_MAC = VAN_MAC_LIT - _FOLD     # executed
ART_MAC = 794_880                 # articulated, latent columns included
ART_MAC_EXEC = 794_880 - 102_400 - _FOLD  # latent columns folded into biases: 128*(128+32) + 2*256*128 + 128*128; bottleneck fold


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


def wgrad_kinds(args):
    """Each job kind of the grouped weight-gradient kernel alone, `nl` identical layers filling the chip: executed TFLOP/s per kind
    (the costs of wg_cost() in csrc/aon_wgrad.h are these rates' reciprocals)."""
    from aon_amd import _lib, ops

    lib = _lib.lib
    dev = torch.device("cuda:0")
    rows = 3456
    for S in (65, 193):
        Np = ops.padded_samples(args.rays * S)
        planes = torch.randn((Np // 32, rows // 4, 32, 4), device=dev)
        dplanes = torch.randn((Np // 32, rows // 4, 32, 4), device=dev) * 1e-3
        ws = torch.empty(int(lib.aon_wgrad_workspace_bytes()), dtype=torch.uint8, device=dev)
        for kind, name, M, K, nl in ((0, "256x256", 256, 256, 8), (1, "128x128", 128, 128, 16), (2, "256x64", 256, 64, 16), (3, "128x256", 128, 256, 16), (4, "128x32", 128, 32, 16)):
            def run():
                rc = lib.aon_wgrad_kind_bench(kind, nl, planes.data_ptr(), dplanes.data_ptr(), rows, Np, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
                assert rc == 0, lib.aon_last_error()
            ms = timeit(run, args.reps)
            flops = 2.0 * M * K * Np * nl
            print(json.dumps({"tag": args.tag, "kernel": f"wgrad kind {name} x {nl} layers", "S": S, "ms": round(ms, 4), "tflops": round(flops / ms / 1e9, 2),
                              "frac": round(flops / ms / 1e9 / 157.3, 4), "operand_GBs": round((M + K) * 4 * Np * nl / ms / 1e6, 1)}), flush=True)


def wgrad_probe(args):
    """Where the grouped weight-gradient launch spends its time: per job (layer) the workgroups' entry / exit clocks, from the probe
    of aon_set_wgrad_probe, for the articulated network at both levels.  One JSON line per job: kind, workgroups, steps per
    workgroup, mean / max duration, idle tail behind its last workgroup until the launch ends."""
    import ctypes as C

    import aon_amd.synthetic as syn
    from aon_amd import _lib, ops

    lib = _lib.lib
    dev = torch.device("cuda:0")
    n = args.rays
    asd = {k: v.to(dev) for k, v in syn.make_art_state_dict(seed=0, density_scale=30.0).items()}
    art = {k[len("fine_mlp."):]: v for k, v in asd.items() if k.startswith("fine_mlp.")}
    lat = {"density": torch.randn(1, 128, device=dev) * 0.1, "color": torch.randn(1, 128, device=dev) * 0.1,
           "articulation": torch.randn(1, 32, device=dev) * 0.1}
    pa, pab, small = ops.pack_art_mlp(art), ops.pack_art_mlp_bwd(art), ops.art_prepare(art, lat)
    H, W = 480, 640
    ro, vd = ops.raygen(syn.look_at_pose(), H, W, syn.focal_from_fovy(H), device=dev)
    idx = torch.randint(0, H * W, (n,), device=dev)
    o, d = ro[idx].contiguous(), vd[idx].contiguous()
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    names = {0: "256x256", 1: "128x128", 2: "256x64", 3: "128x256", 4: "128x32"}
    for S in (65, 193):
        t, _ = ops.sample_along_rays(o, d, S - 1, 2.0, 6.0, want_coords=False)
        raw, planes, masks = ops.art_mlp_fwd_train(pa, small, o, d, d, t.contiguous())
        d_raw = torch.randn(ops.plane_samples(planes), 4, device=dev) * 1e-3
        dpl, dxp = ops.art_bwd_chain(pab, small, d_raw, masks, planes)
        probe = torch.zeros(2 * 320, dtype=torch.int64, device=dev)
        ops.art_wgrad(planes, dpl, d_raw, dxp, art, lat, packed_bwd=pab)
        torch.cuda.synchronize()
        lib.aon_set_wgrad_probe(probe.data_ptr())
        try:
            ops.art_wgrad(planes, dpl, d_raw, dxp, art, lat, packed_bwd=pab)
            torch.cuda.synchronize()
        finally:
            lib.aon_set_wgrad_probe(None)
        jobs = (C.c_int32 * (6 * 24))()
        nj = lib.aon_wgrad_plan(1, ops.plane_samples(planes), cus, jobs, 24, None)
        assert nj > 0, lib.aon_last_error()
        pr = probe.cpu().reshape(-1, 2).double() / 100.0     # microseconds
        total = int(max(jobs[6 * j + 1] + jobs[6 * j + 2] for j in range(nj)))   # workgroups of the launch (a workgroup may serve several jobs)
        t0, t1 = pr[:total, 0].min().item(), pr[:total, 1].max().item()
        print(json.dumps({"tag": args.tag, "S": S, "launch_us": round(t1 - t0, 1), "workgroups": total, "jobs": nj}), flush=True)
        for j in range(nj):
            kind, b, c, steps = jobs[6 * j], jobs[6 * j + 1], jobs[6 * j + 2], jobs[6 * j + 3]
            dur = pr[b: b + c, 1] - pr[b: b + c, 0]     # whole duration of the workgroups that serve this job (they may serve its neighbours too)
            print(json.dumps({"S": S, "job": j, "kind": names[kind], "wgs": c, "steps": steps, "wg_dur_us_mean": round(dur.mean().item(), 1),
                              "wg_dur_us_max": round(dur.max().item(), 1), "idle_tail_us": round(t1 - pr[b: b + c, 1].max().item(), 1)}), flush=True)
        del raw, planes, masks, dpl, dxp
        torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--tag", default=os.environ.get("AON_HIP_LIB", "product"))
    ap.add_argument("--wgrad-kinds", action="store_true", help="isolated rate of each weight-gradient job kind (aon_wgrad_kind_bench)")
    ap.add_argument("--wgrad-probe", action="store_true", help="per-job workgroup durations inside the grouped weight-gradient launch (aon_set_wgrad_probe)")
    args = ap.parse_args()
    if args.wgrad_kinds:
        return wgrad_kinds(args)
    if args.wgrad_probe:
        return wgrad_probe(args)
    only = set(filter(None, args.only.split(",")))
    import aon_amd.synthetic as syn
    from aon_amd import ops

    dev = torch.device("cuda:0")
    n = args.rays
    H, W = 480, 640
    ro, vd = ops.raygen(syn.look_at_pose(), H, W, syn.focal_from_fovy(H), device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    idx = torch.randint(0, H * W, (n,), device=dev, generator=g)
    o, d = ro[idx].contiguous(), vd[idx].contiguous()
    sd = {k: v.to(dev) for k, v in syn.make_nerf_state_dict(seed=0, density_scale=30.0).items()}
    van = {k[len("fine_mlp."):]: v for k, v in sd.items() if k.startswith("fine_mlp.")}
    asd = {k: v.to(dev) for k, v in syn.make_art_state_dict(seed=0, density_scale=30.0).items()}
    art = {k[len("fine_mlp."):]: v for k, v in asd.items() if k.startswith("fine_mlp.")}
    lat = {"density": torch.randn(1, 128, device=dev) * 0.1, "color": torch.randn(1, 128, device=dev) * 0.1,
           "articulation": torch.randn(1, 32, device=dev) * 0.1}
    pv, pvb = ops.pack_vanilla_mlp(van), ops.pack_vanilla_mlp_bwd(van)
    pa, pab, small = ops.pack_art_mlp(art), ops.pack_art_mlp_bwd(art), ops.art_prepare(art, lat)

    def emit(name, S, ms, mac_exec, mac_lit, extra=None):
        smp = n * S
        rec = {"tag": args.tag, "kernel": name, "rays": n, "S": S, "ms": round(ms, 4),
               "tflops_executed": round(smp * mac_exec * 2 / (ms * 1e-3) / 1e12, 2),
               "tflops_literal": round(smp * mac_lit * 2 / (ms * 1e-3) / 1e12, 2)}
        rec["frac_executed"] = round(rec["tflops_executed"] / 157.3, 4)
        if extra:
            rec.update(extra)
        print(json.dumps(rec), flush=True)

    def want(k):
        return not only or k in only

    for S in (65, 193):
        t, _ = ops.sample_along_rays(o, d, S - 1, 2.0, 6.0, want_coords=False)
        t = t.contiguous()
        if want("fwd"):
            emit("mlp_fwd", S, timeit(lambda: ops.mlp_fwd(pv, o, d, d, t), args.reps), VAN_MAC, VAN_MAC_LIT)
        if want("art_fwd"):
            emit("art_mlp_fwd", S, timeit(lambda: ops.art_mlp_fwd(pa, small, o, d, d, t), args.reps), ART_MAC_EXEC, ART_MAC)
        if want("fwd_train") or want("bwd_chain") or want("wgrad"):
            raw, planes, masks = ops.mlp_fwd_train(pv, o, d, d, t)
            if want("fwd_train"):
                emit("mlp_fwd_train", S, timeit(lambda: ops.mlp_fwd_train(pv, o, d, d, t), args.reps), VAN_MAC, VAN_MAC_LIT,
                     {"plane_GB": round(planes.numel() * 4 / 1e9, 3)})
            d_raw = torch.randn(ops.plane_samples(planes), 4, device=dev) * 1e-3
            if want("bwd_chain"):
                bw_mac = VAN_MAC - 256 * 63 * 2 - 128 * 27 - 256 - 3 * 128   # no data gradient into the encodings / heads on the VALU
                emit("mlp_bwd_chain", S, timeit(lambda: ops.mlp_bwd_chain(pvb, pv, d_raw, masks, planes.shape), args.reps), bw_mac, VAN_MAC_LIT)
            if want("wgrad"):
                dpl = ops.mlp_bwd_chain(pvb, pv, d_raw, masks, planes.shape)
                emit("vanilla_wgrad(all layers)", S, timeit(lambda: ops.vanilla_wgrad(planes, dpl, d_raw, pvb), args.reps), VAN_MAC, VAN_MAC_LIT)
                del dpl
            del raw, planes, masks
        if want("art_fwd_train") or want("art_bwd_chain") or want("art_wgrad"):
            raw, planes, masks = ops.art_mlp_fwd_train(pa, small, o, d, d, t)
            if want("art_fwd_train"):
                emit("art_mlp_fwd_train", S, timeit(lambda: ops.art_mlp_fwd_train(pa, small, o, d, d, t), args.reps), ART_MAC_EXEC, ART_MAC,
                     {"plane_GB": round(planes.numel() * 4 / 1e9, 3)})
            d_raw = torch.randn(ops.plane_samples(planes), 4, device=dev) * 1e-3
            if want("art_bwd_chain"):
                bw_mac = ART_MAC_EXEC - 128 * 3 - 128 * 27 - 256 - 3 * 128 + 0
                emit("art_bwd_chain", S, timeit(lambda: ops.art_bwd_chain(pab, small, d_raw, masks, planes), args.reps), bw_mac, ART_MAC)
            if want("art_wgrad"):
                dpl, dxp = ops.art_bwd_chain(pab, small, d_raw, masks, planes)
                emit("art_wgrad(all layers)", S, timeit(lambda: ops.art_wgrad(planes, dpl, d_raw, dxp, art, lat, packed_bwd=pab), args.reps), ART_MAC_EXEC, ART_MAC)
                del dpl, dxp
            del raw, planes, masks
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
