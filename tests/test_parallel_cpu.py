"""CPU, world sizes 2, 3 and 8 over gloo: the multi-GPU form of the path (contiguous ray-range shards + one all-gather of
the rendered pixels) assembles a frame identical to the single-process result, including uneven shard sizes."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class FakeRenderer:
    """Stands in for the drop-in NeRF module: a deterministic per-ray function, so sharding/gather order is checkable
    without a GPU (the real module is exercised by the -m gpu tests)."""

    def __call__(self, rays, randomized, white_bkgd, near, far):
        o, d = rays["rays_o"], rays["viewdirs"]
        rgb = torch.stack([d[:, 0] * 2 + o[:, 0], d[:, 1] - o[:, 1], d[:, 2] * d[:, 0]], -1)
        acc = d.abs().sum(-1)
        depth = near + (far - near) * d[:, 2].abs()
        coarse = (rgb * 0.5, acc * 0.5, depth)
        return [coarse, (rgb, acc, depth)]


def _cpu_raygen(H, W, focal, c2w, begin, end):
    import aon_amd.synthetic as syn

    r = syn.make_rays(H, W, c2w, focal)
    return r["rays_o"][begin:end].contiguous(), r["viewdirs"][begin:end].contiguous()


def _worker(rank, world, port, H, W, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import aon_amd.synthetic as syn
        from aon_amd import parallel as par

        c2w, focal = syn.look_at_pose(), syn.focal_from_fovy(H)
        rgb, acc, depth = par.render_frame_sharded(FakeRenderer(), H, W, focal, c2w, 2.0, 6.0, True, _cpu_raygen)
        # uneven explicit gather: rank r contributes r+3 rays; every rank states the layout (no count exchange, no host sync)
        n = rank + 3
        lvl = (torch.full((n, 3), float(rank)), torch.arange(n, dtype=torch.float32) + 100 * rank, torch.full((n,), -1.0 * rank))
        g_rgb, g_acc, g_depth = par.all_gather_pixels(lvl, counts=[r + 3 for r in range(world)])
        q.put((rank, *[x.numpy().copy() for x in (rgb, acc, depth, g_rgb, g_acc, g_depth)]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _grad_worker(rank, world, port, q, shard_align, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from aon_amd import parallel as par

        torch.manual_seed(100 + rank)  # ranks start from different parameters and see different data
        net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
        # a parameter NO rank produces a gradient for (num_levels=1 leaves fine_mlp alone): 68 values + 5 flags = 73 -- with
        # shard_align=1 the bucket is padded to the next multiple of the world size only (74 at world 2, 75 at 3, 80 at 8) and the
        # shards cut through parameters and through the flags
        net.register_parameter("unused", torch.nn.Parameter(torch.ones(2)))
        arena = None
        if mode.endswith("+arena"):   # round 6: the parameters in ONE flat arena -- the exchange reduces its gradient buffer in place
            from aon_amd.arena import ParamArena

            mode = mode[: -len("+arena")]
            arena = ParamArena(net)
        par.broadcast_parameters(net)
        x = torch.randn(11, 5)
        net(x).square().mean().backward()
        if mode == "permissive" and rank == 1:
            net[2].bias.grad = None   # a rank that produced no gradient for a parameter still takes part with zeros
        local = [torch.zeros_like(p) if p.grad is None else p.grad.clone() for p in net.parameters()]
        late_error = None
        if mode == "permissive":
            par.allreduce_gradients(net, shard_align=shard_align, find_unused_parameters=True)
        elif mode == "contract":
            # DDP's contract (run.py:151, find_unused_parameters=False): same parameter set on every rank -> no host read at all
            par.allreduce_gradients(net, shard_align=shard_align)
            par.check_gradient_exchange()
        else:   # "violation": rank 1 breaks the contract; like DDP, the NEXT call reports it -- on every rank
            if rank == 1:
                net[2].bias.grad = None
            par.allreduce_gradients(net, shard_align=shard_align)
            try:
                par.allreduce_gradients(net, shard_align=shard_align)
            except par.UnevenGradientsError as e:
                late_error = str(e)
            net[2].bias.grad = torch.zeros(3)
        assert net.unused.grad is None   # as under torch DDP: globally unused parameters keep grad None (Adam skips them)
        if arena is not None:
            # torch's autograd delivered these gradients OUTSIDE the arena: the exchange adopted them into their slots and ran on the flat
            # buffer itself; the slot of the parameter nobody touched holds the zeros this rank contributed, gaps and tail stay zero
            assert arena.intact() and all(arena.grad_in_place(i) for i, p in enumerate(arena.params) if p.grad is not None)
            mask = torch.ones(arena.capacity, dtype=torch.bool)
            for p_, o_ in zip(arena.params, arena.offsets):
                mask[o_: o_ + p_.numel()] = False
            mask[arena.total: arena.total + len(arena.params)] = False     # the per-parameter counts of the exchange
            assert not arena.grad[mask].any() and not arena.grad_view([i for i, p_ in enumerate(arena.params) if p_ is net.unused][0]).any()
        used = [p for n_, p in net.named_parameters() if n_ != "unused"]
        local = [g for g, (n_, _) in zip(local, net.named_parameters()) if n_ != "unused"]
        # plain numpy payloads: torch tensors travel through shared-memory handles that die with the worker
        q.put((rank, [p.detach().numpy().copy() for p in used], [g.numpy().copy() for g in local],
               [p.grad.numpy().copy() for p in used], late_error))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run_grad_workers(world, shard_align, mode):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q, shard_align, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return [(r[0], *[[torch.from_numpy(a) for a in part] for part in r[1:4]], r[4]) for r in res]


@pytest.mark.parametrize("world,shard_align,mode", [(2, 64, "permissive"), (2, 1, "contract"), (3, 1, "permissive"), (3, 64, "contract"),
                                                     (8, 1, "permissive"), (8, 64, "contract"), (2, 64, "contract+arena"), (3, 1, "permissive+arena"),
                                                     (8, 64, "contract+arena")])
def test_data_parallel_gradient_exchange(world, shard_align, mode):
    """broadcast_parameters + allreduce_gradients (the DDP duties of run.py:151) over gloo at world sizes 2, 3 and 8,
    buckets whose shards cut through parameters and flags (shard_align=1: 73 elements over 2 / 3 / 8 ranks), in both modes:
    DDP's contract (no host read) and the permissive one (a rank without a gradient adopts the others' mean)."""
    res = _run_grad_workers(world, shard_align, mode)
    for _, params, _, _, _ in res[1:]:
        for a, b in zip(res[0][1], params):
            assert torch.equal(a, b)                      # same parameters everywhere after the broadcast
    nparam = len(res[0][1])
    for i in range(nparam):
        mean = sum(r[2][i].double() for r in res) / world
        for r in res:
            assert torch.equal(r[3][i], res[0][3][i])     # same averaged gradient everywhere, bit for bit
        torch.testing.assert_close(res[0][3][i].double(), mean, rtol=1e-6, atol=1e-7)


def test_gradient_contract_violation_is_reported_late_on_every_rank():
    """find_unused_parameters=False and ranks that disagree on the parameter set: the exchange itself never reads the device, so the
    violation surfaces at the NEXT call (torch DDP: 'Expected to have finished reduction in the prior iteration'), on every rank."""
    res = _run_grad_workers(3, 1, "violation")
    assert all(r[4] is not None and "find_unused_parameters=True" in r[4] for r in res)


def test_shard_range_is_a_partition():
    from aon_amd.parallel import shard_range

    for n in (0, 1, 7, 307_200, 76_801):
        for world in (1, 2, 3, 8):
            pieces = [shard_range(n, r, world) for r in range(world)]
            assert pieces[0][0] == 0 and pieces[-1][1] == n
            assert all(pieces[i][1] == pieces[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in pieces]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("H,W,world", [(9, 7, 2), (9, 7, 3), (5, 13, 8), (1, 5, 8), (1, 2, 3)])
def test_sharded_frame_equals_single_process(H, W, world):
    """63 rays over 2 (32 / 31) and 3 ranks, 65 over 8 (one rank with 9, seven with 8), and frames with FEWER rays than ranks
    (5 over 8, 2 over 3: `shard_range` gives the trailing ranks empty ranges; they render nothing and still take part in the
    all-gather with an empty message padded to the longest)."""
    import aon_amd.synthetic as syn

    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, H, W, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ro, vd = _cpu_raygen(H, W, syn.focal_from_fovy(H), syn.look_at_pose(), 0, H * W)
    full = FakeRenderer()({"rays_o": ro, "rays_d": vd, "viewdirs": vd}, False, True, 2.0, 6.0)[1]
    want_acc = torch.cat([torch.arange(r + 3, dtype=torch.float32) + 100 * r for r in range(world)])
    want_depth = torch.cat([torch.full((r + 3,), -1.0 * r) for r in range(world)])
    assert len(results) == world
    for rank, *arrs in results:
        rgb, acc, depth, g_rgb, g_acc, g_depth = (torch.from_numpy(a) for a in arrs)
        assert rgb.shape == (H * W, 3)
        assert torch.equal(rgb, full[0]) and torch.equal(acc, full[1]) and torch.equal(depth, full[2])
        assert g_rgb.shape == (want_acc.numel(), 3)
        assert torch.equal(g_acc, want_acc) and torch.equal(g_depth, want_depth)


def test_interface_gather_is_rank_major():
    """LitModel.alter_gather_cat keeps each rank's images contiguous (the reference interleaves per pixel when
    world > 1); single-process behaviour equals the reference's."""
    from aon_amd.models.interface import LitModel

    m = LitModel()
    outs = [{"rgb": torch.arange(24, dtype=torch.float32).reshape(8, 3)}, {"rgb": torch.arange(24, 48, dtype=torch.float32).reshape(8, 3)}]
    imgs = m.alter_gather_cat(outs, "rgb", [(2, 4), (2, 4)])
    assert len(imgs) == 2 and imgs[0].shape == (2, 4, 3) and torch.equal(imgs[1].reshape(-1), torch.arange(24, 48, dtype=torch.float32))
    masks = m.alter_gather_cat([{"m": torch.ones(8, dtype=torch.bool)}], "m", [(2, 4)])
    assert masks[0].shape == (2, 4)
