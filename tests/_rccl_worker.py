"""Worker of tests/test_rccl_multi_gpu.py: one process per GPU under `python -m torch.distributed.run`, RCCL over xGMI (torch backend
"nccl").  Every rank checks its own view and the verdicts are combined with an all-reduce, so ANY rank's failure fails the launch:

  (a) BASELINE config 3: `render_frame_sharded` -- contiguous ray ranges, ONE all_gather_into_tensor of 20 B/ray -- gives every rank the
      frame rank 0 renders alone, bit for bit (models/interface.py:31-51's job without its pixel interleave), and every rank took part;
  (b) BASELINE config 5's exchange: after `allreduce_gradients` every rank holds the mean of the per-rank single-GPU gradients
      (gathered independently with dist.all_gather and averaged in fp64), 1e-6 relative;
  (c) DDP's contract (run.py:151, find_unused_parameters=False): a rank that produced no gradient for a parameter makes EVERY rank raise
      UnevenGradientsError at the next exchange -- none of them enters the collective alone;
  (d) round 6: the exchange in place on a parameter arena (aon_amd/arena.py) -- same mean, gradients never leave their slots -- and the
      one-launch Adam behind it leaves every rank with bit-identical parameters.

Writes one JSON line to the path in argv[1] (rank 0)."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    out_path = sys.argv[1]
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    import aon_amd.synthetic as syn
    from aon_amd import parallel as par
    from aon_amd.datasets.ray_utils import get_frame_rays
    from aon_amd.models.vanilla_nerf.model import NeRF

    force = world == 1   # a one-GPU box still drives the real collectives
    res = {"world": world}
    ok = True

    def agree(flag: bool) -> bool:
        t = torch.tensor([1.0 if flag else 0.0], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    # ---- (a) sharded frame == rank 0's unsharded frame ----
    H, W = 240, 320
    model = NeRF().to(dev)
    model.load_state_dict(syn.make_nerf_state_dict(seed=0, density_scale=30.0))
    focal, c2w = syn.focal_from_fovy(H), syn.look_at_pose(4.0, 30.0, 30.0)
    raygen = lambda h, w, f, c, b, e: get_frame_rays(h, w, f, c, b, e, device=dev)   # noqa: E731
    with torch.no_grad():
        rgb, acc, depth = par.render_frame_sharded(model, H, W, focal, c2w, syn.NEAR, syn.FAR, True, raygen, force=force)
        ro, vd = get_frame_rays(H, W, focal, c2w, device=dev)
        ref = [x.clone() for x in model({"rays_o": ro, "rays_d": vd, "viewdirs": vd}, False, True, syn.NEAR, syn.FAR)[1]]
    for x in ref:
        dist.broadcast(x, src=0)     # rank 0's own render is the reference on every rank
    same = rgb.shape == (H * W, 3) and torch.equal(rgb, ref[0]) and torch.equal(acc, ref[1]) and torch.equal(depth, ref[2])
    seen = torch.ones(1, device=dev)
    dist.all_reduce(seen)
    res["ranks_seen"] = int(seen.item())
    res["sharded_frame_bit_equal"] = agree(same)
    ok &= res["sharded_frame_bit_equal"] and res["ranks_seen"] == world

    # ---- (b) gradient exchange == mean of the per-rank gradients ----
    n = 256
    rays = {k: v.to(dev) for k, v in syn.random_rays(n, seed=40 + rank).items()}
    target = syn.seeded_uniform(90 + rank, n, 3).to(dev)
    tr, u = syn.seeded_uniform(70 + rank, n, 65).to(dev), syn.seeded_uniform(80 + rank, n, 128).to(dev)
    par.broadcast_parameters(model, force=force)
    out = model(rays, True, True, syn.NEAR, syn.FAR, t_rand=tr, u=u)
    (((out[0][0] - target) ** 2).mean() + ((out[1][0] - target) ** 2).mean()).backward()
    params = [p for p in model.parameters() if p.requires_grad]
    local_flat = torch.cat([p.grad.reshape(-1) for p in params])
    gathered = [torch.empty_like(local_flat) for _ in range(world)]
    dist.all_gather(gathered, local_flat)
    mean = (sum(g.double() for g in gathered) / world)
    par.allreduce_gradients(model, force=force)
    got = torch.cat([p.grad.reshape(-1) for p in params])
    err = ((got.double() - mean).norm() / mean.norm().clamp_min(1e-30)).item()
    distinct = world == 1 or (gathered[0] - gathered[-1]).abs().max().item() > 0   # ranks really saw different data
    res["grad_exchange_rel_err"] = err
    ok &= agree(err <= 1e-6 and distinct)
    par.check_gradient_exchange()   # the even exchange above: no complaint

    # ---- (d) round 6: the same exchange IN PLACE on a parameter arena, then the one-launch Adam: every rank ends with the same parameters ----
    from aon_amd.arena import ArenaAdam, ParamArena

    model.zero_grad(set_to_none=True)
    arena = ParamArena(model)
    opt = ArenaAdam(arena, lr=5e-4)
    out = model(rays, True, True, syn.NEAR, syn.FAR, t_rand=tr, u=u)
    (((out[0][0] - target) ** 2).mean() + ((out[1][0] - target) ** 2).mean()).backward()
    in_place = all(arena.grad_in_place(i) for i in range(len(arena.params)))
    local2 = torch.cat([p.grad.reshape(-1) for p in params])
    same_as_plain = torch.equal(local2, local_flat)                  # the arena only re-homes storage: the local gradients keep their bits
    gathered2 = [torch.empty_like(local2) for _ in range(world)]
    dist.all_gather(gathered2, local2)
    mean2 = sum(g.double() for g in gathered2) / world
    ptr = arena.grad.data_ptr()
    par.allreduce_gradients(model, force=force)
    got2 = torch.cat([p.grad.reshape(-1) for p in params])
    err2 = ((got2.double() - mean2).norm() / mean2.norm().clamp_min(1e-30)).item()
    in_place &= arena.grad.data_ptr() == ptr and all(arena.grad_in_place(i) for i in range(len(arena.params)))
    opt.step()
    hi, lo = arena.flat[: arena.total].clone(), arena.flat[: arena.total].clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    res["arena_exchange_rel_err"] = err2
    res["arena_in_place"] = agree(bool(in_place and same_as_plain))
    res["arena_same_parameters_after_adam"] = agree(torch.equal(hi, lo) and opt.last_launches == 1)
    ok &= agree(err2 <= 1e-6) and res["arena_in_place"] and res["arena_same_parameters_after_adam"]
    par.check_gradient_exchange()

    # ---- (c) uneven gradient sets raise late, on every rank ----
    if world > 1:
        if rank == 1:
            params[-1].grad = None
        par.allreduce_gradients(model)
        raised = False
        try:
            par.allreduce_gradients(model)
        except par.UnevenGradientsError:
            raised = True
        res["uneven_raised_everywhere"] = agree(raised)
        ok &= res["uneven_raised_everywhere"]
    res["ok"] = agree(ok)
    if rank == 0:
        with open(out_path, "w") as f:
            f.write(json.dumps(res) + "\n")
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if res["ok"] else 1)


if __name__ == "__main__":
    main()
