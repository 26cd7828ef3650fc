"""Training-step timing of the HIP path (vanilla NeRF, reference training_step: model.py:256-282 + Adam :386-389).

    python tools/train_bench.py --rays 2048 --steps 5
"""
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
    ap.add_argument("--rays", type=int, default=2048)   # the reference's hard-coded per-GPU batch (model.py:426)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-overlap", action="store_true", help="backward of both levels on the caller's stream (A/B of the two-stream backward)")
    ap.add_argument("--no-fwd-overlap", action="store_true", help="training forward on the caller's stream only (A/B of the two ray halves on two streams)")
    ap.add_argument("--no-fwd-merge", action="store_true", help="training forward without the merged coarse(A) | fine(A)+coarse(B) | fine(B) launches (round 4 default)")
    ap.add_argument("--no-bwd-merge", action="store_true", help="one backward-chain launch per level (round 3) instead of the merged two-segment launch")
    ap.add_argument("--no-view-bias", action="store_true", help="vanilla: the view-encoding chunk per sample instead of the per-ray view bias")
    ap.add_argument("--torch-loss", action="store_true", help="the loss lines as torch ops (~47 launches) instead of helper.train_loss")
    ap.add_argument("--late-heads", action="store_true", help="every head reduction behind the chain (round 4) instead of the chain-independent ones beside it")
    ap.add_argument("--articulated", action="store_true", help="NeRF_AE_Art + CodeLibraryArticulated (BASELINE config 5 per GPU)")
    ap.add_argument("--foreach-adam", action="store_true", help="torch.optim.Adam's default foreach form instead of fused=True (the harness's choice on a GPU in round 5)")
    ap.add_argument("--classes", action="store_true", help="a second timed loop with the library's per-kernel-class HIP-event timers on: ms per step of forward / chain / weight gradients")
    ap.add_argument("--exchange", action="store_true", help="run parallel.allreduce_gradients(force=True) every step under a world-size-1 RCCL group: the fixed cost of the "
                                                             "data-parallel exchange (in place on the arena; with --torch-adam: the round-5 bucket with its 2 x 83-tensor copies)")
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam (fused unless --foreach-adam) instead of the parameter arena + ArenaAdam the harness builds since round 6")
    args = ap.parse_args()
    import aon_amd.synthetic as syn
    from aon_amd import ops
    from aon_amd.models.vanilla_nerf.helper import train_loss
    from aon_amd.models.vanilla_nerf.model import NeRF

    dev = torch.device("cuda:0")
    ops.set_bwd_overlap(2 if os.environ.get("AON_BWD_AUX") else (not args.no_overlap))
    ops.set_fwd_merge(not args.no_fwd_merge)
    ops.set_bwd_merge(not args.no_bwd_merge)
    ops.set_bwd_early_heads(not args.late_heads)
    ops.set_view_bias(not args.no_view_bias)
    ops.set_fwd_overlap(not args.no_fwd_overlap and not args.no_overlap)
    if os.environ.get("AON_FWD_PARTS"):
        from aon_amd import _lib
        _lib.lib.aon_set_fwd_overlap(int(os.environ["AON_FWD_PARTS"]))
    lib = None
    if args.articulated:
        import types

        from aon_amd.models.code_library import CodeLibraryArticulated
        from aon_amd.models.vanilla_nerf.model_autodecoder import NeRF_AE_Art

        model = NeRF_AE_Art().to(dev)
        model.load_state_dict(syn.make_art_state_dict(seed=0, density_scale=30.0))
        lib = CodeLibraryArticulated(types.SimpleNamespace(N_max_objs=1, N_obj_code_length=128)).to(dev)
        lib.load_state_dict(syn.make_code_library_state(seed=0, n_max_objs=1))
        batch = {"instance_id": torch.tensor([0], device=dev), "articulation_id": torch.tensor([3], device=dev)}
        if args.torch_adam or args.foreach_adam:
            opt = torch.optim.Adam(list(model.parameters()) + list(lib.parameters()), lr=5e-4, betas=(0.9, 0.999), fused=not args.foreach_adam)
        else:
            from aon_amd.models.vanilla_nerf.model import build_adam
            opt = build_adam([model, lib], 5e-4)
    else:
        model = NeRF().to(dev)
        model.load_state_dict(syn.make_nerf_state_dict(seed=0, density_scale=30.0))
        if args.torch_adam or args.foreach_adam:
            opt = torch.optim.Adam(model.parameters(), lr=5e-4, betas=(0.9, 0.999), fused=not args.foreach_adam)
        else:
            from aon_amd.models.vanilla_nerf.model import build_adam
            opt = build_adam([model], 5e-4)
    H, W = 480, 640
    ro, vd = ops.raygen(syn.look_at_pose(), H, W, syn.focal_from_fovy(H), device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    idx = torch.randint(0, H * W, (args.rays,), device=dev, generator=g)
    rays = {"rays_o": ro[idx].contiguous(), "rays_d": vd[idx].contiguous(), "viewdirs": vd[idx].contiguous()}
    target = torch.rand(args.rays, 3, device=dev, generator=g)

    exchange = None
    if args.exchange:
        import socket

        import torch.distributed as dist
        from aon_amd.parallel import allreduce_gradients

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        both = torch.nn.ModuleList([model] + ([lib] if lib is not None else []))
        exchange = lambda: allreduce_gradients(both, force=True)   # noqa: E731

    def step():
        opt.zero_grad(set_to_none=True)
        if lib is not None:
            latents = lib(batch)
            out = model(rays, True, True, syn.NEAR, syn.FAR, latents)
            codes = (latents["density"], latents["color"], latents["articulation"])
        else:
            out = model(rays, True, True, syn.NEAR, syn.FAR)
            codes = ()
        if args.torch_loss:
            reg = sum(torch.mean(torch.norm(c, dim=0)) for c in codes) if codes else 0.0
            loss = torch.mean((out[0][0] - target) ** 2) + torch.mean((out[1][0] - target) ** 2) + 1e-4 * reg
        else:
            loss, _ = train_loss(out, target, codes, 1e-4)   # the harness's loss lines in two launches (helper.train_loss)
        loss.backward()
        if exchange is not None:
            exchange()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    st0 = torch.cuda.memory_stats()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    host_ms = []
    marks[0].record()
    t0 = time.perf_counter()
    for i in range(args.steps):
        th = time.perf_counter()
        loss = step()
        host_ms.append((time.perf_counter() - th) * 1e3)
        marks[i + 1].record()
    enqueue_ms = (time.perf_counter() - t0) / args.steps * 1e3   # host time to ENQUEUE a step (the device runs behind): must stay well below ms_per_step
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    classes = None
    if args.classes:
        ops.profile_begin()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t1) / args.steps
        ops.profile_end()
        classes = {k: round(v[0] / args.steps, 3) for k, v in ops.profile_classes().items() if v[1]}
        classes["step_ms_with_timers"] = round(dt2 * 1e3, 3)
    st1 = torch.cuda.memory_stats()
    alloc = {"device_mallocs_in_timed_loop": st1.get("num_device_alloc", 0) - st0.get("num_device_alloc", 0),
             "device_frees_in_timed_loop": st1.get("num_device_free", 0) - st0.get("num_device_free", 0),
             "reserved_GB": round(st1.get("reserved_bytes.all.current", 0) / 2 ** 30, 2), "reserved_GB_before": round(st0.get("reserved_bytes.all.current", 0) / 2 ** 30, 2)}
    per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]   # device time between the ends of consecutive steps
    worst = max(range(args.steps), key=lambda i: per_step[i])
    flop = args.rays * 258 * (1_589_760 if args.articulated else 1_186_816) * 3  # fwd + 2x bwd, reference-literal
    print(json.dumps({"model": "articulated" if args.articulated else "vanilla", "rays_per_step": args.rays, "ms_per_step": dt * 1e3, "host_enqueue_ms_per_step": enqueue_ms, "exchange": bool(args.exchange), "optimizer": type(opt).__name__ + ("" if type(opt).__name__ == "ArenaAdam" else (" foreach" if args.foreach_adam else " fused")), "rays_per_s": args.rays / dt,
                      "train_tflops_3x_fwd": flop / dt / 1e12, "loss": loss.item(), "allocator": alloc,
                      "step_ms_device": {"median": sorted(per_step)[len(per_step) // 2], "max": per_step[worst], "argmax": worst, "host_ms_of_that_step": host_ms[worst],
                                         "host_ms_max": max(host_ms), "host_argmax": max(range(args.steps), key=lambda i: host_ms[i])}, **({"classes_ms_per_step": classes} if classes else {})}))


if __name__ == "__main__":
    main()
