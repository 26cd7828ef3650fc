"""GPU: the end of a training step on one parameter arena (round 6, aon_amd/arena.py + csrc/aon_optim.hip; VERDICT r5 #1a).

  * the arena only re-homes storage: gradients of the HIP backward are the SAME BITS with and without it, and with it every one lands in
    its slot (``.grad`` is a view of the gradient arena, no copy);
  * ``ArenaAdam`` = ``torch.optim.Adam(lr, betas=(0.9, 0.999))`` (model.py:386-389) as ONE launch: against torch's own Adam on the same
    gradients over several steps of the reference's learning-rate rule, and its ``state_dict`` is interchangeable with torch's;
  * the code library's HIP lookup (``aon_code_library_fwd`` / ``_bwd``) = nn.Embedding forward / dense backward, bit for bit;
  * two graphs over the same parameters (one loss from two renders; two backwards before one step) still accumulate correctly;
  * the data-parallel mean reduces the arena in place (RCCL at world size 1; world 2 / 3 semantics: tests/test_parallel_cpu.py)."""
import os
import socket
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _art_setup(dev, n=192, seed=3):
    import aon_amd.synthetic as syn
    from aon_amd.models.code_library import CodeLibraryArticulated
    from aon_amd.models.vanilla_nerf.model_autodecoder import NeRF_AE_Art

    model = NeRF_AE_Art().to(dev)
    model.load_state_dict(syn.make_art_state_dict(seed=5, density_scale=2.0))
    lib = CodeLibraryArticulated(types.SimpleNamespace(N_max_objs=2, N_obj_code_length=128)).to(dev)
    lib.load_state_dict(syn.make_code_library_state(seed=3, n_max_objs=2))
    rays = {k: v.to(dev) for k, v in syn.random_rays(n, seed=seed).items()}
    target = syn.seeded_uniform(77, n, 3).to(dev)
    draws = (syn.seeded_uniform(78, n, 65).to(dev), syn.seeded_uniform(79, n, 128).to(dev))
    ids = {"instance_id": torch.tensor([1], device=dev), "articulation_id": torch.tensor([6], device=dev)}
    return model, lib, rays, target, draws, ids


def _art_step(model, lib, rays, target, draws, ids, scale=1.0):
    from aon_amd.models.vanilla_nerf.helper import train_loss

    latents = lib(ids)
    out = model(rays, True, True, 2.0, 6.0, latents, t_rand=draws[0], u=draws[1])
    loss, _ = train_loss(out, target, (latents["density"], latents["color"], latents["articulation"]), 1e-4)
    (loss * scale).backward()
    return loss.detach()


def _named_grads(model, lib):
    g = {"model." + k: p.grad for k, p in model.named_parameters()}
    g.update({"lib." + k: p.grad for k, p in lib.named_parameters()})
    return g


def test_gradients_land_in_the_arena_bit_equal(dev):
    from aon_amd.arena import ParamArena

    model, lib, rays, target, draws, ids = _art_setup(dev)
    loss_a = _art_step(model, lib, rays, target, draws, ids)
    plain = {k: v.clone() for k, v in _named_grads(model, lib).items()}
    sd_before = {k: v.clone() for k, v in model.state_dict().items()}
    for p in list(model.parameters()) + list(lib.parameters()):
        p.grad = None
    arena = ParamArena([model, lib])
    assert arena.intact() and all(torch.equal(v, sd_before[k]) for k, v in model.state_dict().items())     # values and names survive the move
    loss_b = _art_step(model, lib, rays, target, draws, ids)
    assert torch.equal(loss_a, loss_b)
    got = _named_grads(model, lib)
    assert set(got) == set(plain)
    for k in plain:
        assert torch.equal(got[k], plain[k]), k
    assert all(arena.grad_in_place(i) for i in range(len(arena.params))), [i for i in range(len(arena.params)) if not arena.grad_in_place(i)]
    # the gaps between the slots and the tail stay zero (the optimiser and the exchange sweep them)
    mask = torch.ones(arena.capacity, dtype=torch.bool, device=dev)
    for p, o in zip(arena.params, arena.offsets):
        mask[o: o + p.numel()] = False
    assert not arena.grad[mask].any() and not arena.flat[mask].any()


def test_vanilla_gradients_land_in_the_arena_bit_equal(dev, nerf_sd):
    import aon_amd.synthetic as syn
    from aon_amd.arena import ParamArena
    from aon_amd.models.vanilla_nerf.model import NeRF

    model = NeRF().to(dev)
    model.load_state_dict(nerf_sd)
    rays = {k: v.to(dev) for k, v in syn.random_rays(160, seed=4).items()}
    target = syn.seeded_uniform(5, 160, 3).to(dev)

    def run():
        out = model(rays, False, True, 2.0, 6.0)
        (((out[0][0] - target) ** 2).mean() + ((out[1][0] - target) ** 2).mean()).backward()
        return {k: p.grad for k, p in model.named_parameters()}

    plain = {k: v.clone() for k, v in run().items()}
    model.zero_grad(set_to_none=True)
    arena = ParamArena(model)
    got = run()
    for k in plain:
        assert torch.equal(got[k], plain[k]), k
    assert all(arena.grad_in_place(i) for i in range(len(arena.params)))


def test_two_graphs_over_the_same_parameters_accumulate(dev):
    """One loss built from two renders, and two backward calls before one optimiser step: only one of two live graphs may write the
    slots directly (ParamArena.claim), a second backward onto existing gradients must ADD."""
    from aon_amd.arena import ParamArena

    model, lib, rays, target, draws, ids = _art_setup(dev, n=128)
    rays2 = {k: v.flip(0).contiguous() for k, v in rays.items()}

    def both(one_backward):
        from aon_amd.models.vanilla_nerf.helper import train_loss

        for p in list(model.parameters()) + list(lib.parameters()):
            p.grad = None
        losses = []
        for r in (rays, rays2):
            latents = lib(ids)
            out = model(r, True, True, 2.0, 6.0, latents, t_rand=draws[0][:128], u=draws[1][:128])
            loss, _ = train_loss(out, target[:128], (latents["density"], latents["color"], latents["articulation"]), 1e-4)
            if one_backward:
                losses.append(loss)
            else:
                loss.backward()
        if one_backward:
            (losses[0] + losses[1]).backward()
        return {k: v.clone() for k, v in _named_grads(model, lib).items()}

    want_one, want_two = both(True), both(False)
    ParamArena([model, lib])
    got_one, got_two = both(True), both(False)
    for k in want_one:
        # (autograd adds the two graphs' gradients in either order: a + b == b + a bit for bit)
        assert torch.equal(got_one[k], want_one[k]), k
        assert torch.equal(got_two[k], want_two[k]), k


def test_arena_adam_is_torch_adam_in_one_launch(dev):
    """Five steps of the reference's optimizer + learning-rate rule on identical gradients: ArenaAdam (one launch over the arena) against
    torch.optim.Adam (its single-tensor form, whose operations aon_adam_step restates).  fp32, one rounding per operation on both sides; the
    only freedom is fused multiply-adds inside torch's kernels: 2e-7 relative to the parameter scale per step."""
    from aon_amd.arena import ArenaAdam, ParamArena
    from aon_amd.models.vanilla_nerf.model import LitNeRF

    model, lib, *_ = _art_setup(dev)
    twin_m, twin_l, *_ = _art_setup(dev)
    arena = ParamArena([model, lib])
    opt = ArenaAdam(arena, lr=5e-4)
    ref_params = list(twin_m.parameters()) + list(twin_l.parameters())
    ref = torch.optim.Adam(ref_params, lr=5e-4, betas=(0.9, 0.999), foreach=False, fused=False)
    rule = LitNeRF({"run_max_steps": 40}, lr_delay_steps=10)
    g = torch.Generator(device=dev).manual_seed(11)
    for step in range(5):
        lr = rule.lr_at_step(step)
        for idx, (pa, pr) in enumerate(zip(arena.params, ref_params)):
            grad = torch.randn(pa.shape, device=dev, generator=g) * (10.0 ** float((step * 5 + idx) % 7 - 6))
            pa.grad = arena.grad_view(idx).copy_(grad) if step % 2 == 0 else grad.clone()   # in place / arriving elsewhere
            pr.grad = grad.clone()
        for o in (opt, ref):
            for pg in o.param_groups:
                pg["lr"] = lr
        opt.step()
        ref.step()
        assert opt.last_launches == 1
        for pa, pr in zip(arena.params, ref_params):
            scale = pr.abs().max().item() + 1e-12
            assert (pa - pr).abs().max().item() <= 2e-7 * scale * (step + 1), (step, tuple(pa.shape), (pa - pr).abs().max().item(), scale)
    # state_dict: torch.optim.Adam's layout, both directions
    sd = opt.state_dict()
    ref2 = torch.optim.Adam(ref_params, lr=1.0, betas=(0.9, 0.999), foreach=False, fused=False)
    ref2.load_state_dict(sd)
    assert ref2.param_groups[0]["lr"] == sd["param_groups"][0]["lr"]
    for i, p in enumerate(ref_params):
        assert torch.equal(ref2.state[p]["exp_avg"], opt.state[arena.params[i]]["exp_avg"]) and float(ref2.state[p]["step"]) == 5.0
    opt2 = ArenaAdam(arena, lr=1.0)
    opt2.load_state_dict(ref.state_dict())
    assert all(s == 5 for s in opt2._steps)
    for i, p in enumerate(ref_params):
        assert torch.equal(opt2.state[arena.params[i]]["exp_avg_sq"], ref.state[p]["exp_avg_sq"])
        assert opt2.state[arena.params[i]]["exp_avg"].data_ptr() == opt2.exp_avg.data_ptr() + 4 * arena.offsets[i]     # adopted INTO the arena


def test_arena_adam_skips_parameters_without_gradient(dev, nerf_sd):
    """num_levels=1 leaves fine_mlp without gradients: torch's Adam skips those parameters and their state does not advance; ArenaAdam too
    (one launch per run of adjacent parameters that do have one)."""
    import aon_amd.synthetic as syn
    from aon_amd.arena import ArenaAdam, ParamArena
    from aon_amd.models.vanilla_nerf.model import NeRF

    model = NeRF(num_levels=1).to(dev)
    model.load_state_dict(nerf_sd)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    arena = ParamArena(model)
    opt = ArenaAdam(arena)
    rays = {k: v.to(dev) for k, v in syn.random_rays(64, seed=9).items()}
    out = model(rays, False, True, 2.0, 6.0)
    out[0][0].square().mean().backward()
    opt.step()
    assert opt.last_launches == 1     # coarse_mlp's 24 tensors are adjacent
    moved = {k: not torch.equal(v, before[k]) for k, v in model.state_dict().items()}
    assert all(moved[k] for k in moved if k.startswith("coarse_mlp.")) and not any(moved[k] for k in moved if k.startswith("fine_mlp."))
    assert all((s == 1) == n.startswith("coarse_mlp.") for s, (n, _) in zip(opt._steps, model.named_parameters()))


def test_code_library_lookup_is_nn_embedding(dev):
    import aon_amd.synthetic as syn
    from aon_amd.models.code_library import CodeLibraryArticulated

    lib = CodeLibraryArticulated(types.SimpleNamespace(N_max_objs=3, N_obj_code_length=128)).to(dev)
    lib.load_state_dict(syn.make_code_library_state(seed=1, n_max_objs=3))
    for inst, art in ((0, 0), (2, 9), (1, 4)):
        ids = {"instance_id": torch.tensor([inst], device=dev), "articulation_id": torch.tensor([art], device=dev)}
        lat = lib(ids)
        want = {"density": lib.embedding_instance_shape(ids["instance_id"]), "color": lib.embedding_instance_appearance(ids["instance_id"]),
                "articulation": lib.embedding_instance_articulation(ids["articulation_id"])}
        ws = {k: torch.randn_like(v) for k, v in want.items()}
        for p in lib.parameters():
            p.grad = None
        sum((lat[k] * ws[k]).sum() for k in lat).backward()
        got = {n: p.grad.clone() for n, p in lib.named_parameters()}
        for p in lib.parameters():
            p.grad = None
        sum((want[k] * ws[k]).sum() for k in want).backward()
        for k in want:
            assert lat[k].shape == want[k].shape and torch.equal(lat[k], want[k]), k
        for n, p in lib.named_parameters():
            assert torch.equal(got[n], p.grad), n
    # batched ids and the test-time table stay on torch's path
    ids = {"instance_id": torch.tensor([0, 2], device=dev), "articulation_id": torch.tensor([3, 3], device=dev)}
    assert lib(ids)["density"].shape == (2, 128)
    # an out-of-range id cannot raise from a kernel: the row is NaN (nn.Embedding raises)
    bad = lib({"instance_id": torch.tensor([7], device=dev), "articulation_id": torch.tensor([1], device=dev)})
    assert torch.isnan(bad["density"]).all() and torch.isfinite(bad["articulation"]).all()


def test_fit_step_runs_on_the_arena(dev, golden):
    """The harness's optimizer is the arena form on a GPU: after a fit_step every gradient sits in its slot, the update took one launch,
    and a checkpoint written with it loads into torch.optim.Adam (and back)."""
    import aon_amd.synthetic as syn
    from aon_amd.arena import ArenaAdam
    from aon_amd.models.vanilla_nerf.model_autodecoder import LitNeRF_AutoDecoder

    lit = LitNeRF_AutoDecoder({"run_max_steps": 40, "N_max_objs": 2, "N_obj_code_length": 128}, lr_delay_steps=10).to(dev)
    lit.model.load_state_dict(syn.make_art_state_dict(seed=5, density_scale=2.0))
    opt = lit.configure_optimizers()
    assert isinstance(opt, ArenaAdam) and len(opt.arena.params) == 83
    rays = syn.random_rays(128, seed=2)
    for i in range(3):
        batch = {k: v.unsqueeze(0).to(dev) for k, v in rays.items()}
        batch["target"] = syn.seeded_uniform(40 + i, 128, 3).unsqueeze(0).to(dev)
        batch["instance_id"] = torch.tensor([i % 2], device=dev)
        batch["articulation_id"] = torch.tensor([i], device=dev)
        loss = lit.fit_step(batch, i, opt)
        assert torch.isfinite(loss)
        assert all(opt.arena.grad_in_place(j) for j in range(83)) and opt.last_launches == 1
    assert opt._steps == [3] * 83
    # configure_optimizers() again (a resumed run builds its optimizer anew): the parameters stay where they are, the arena is reused
    opt2 = lit.configure_optimizers()
    assert opt2.arena is opt.arena and opt2 is not opt
    opt2.load_state_dict(opt.state_dict())
    assert opt2._steps == [3] * 83


def test_allreduce_in_place_over_rccl(dev):
    """parallel.allreduce_gradients on an arena: the exchange runs on the gradient arena itself (reduce-scatter + all-gather through RCCL
    at world size 1), gradients keep their bits and stay in their slots; broadcast_parameters sends the flat buffer."""
    import torch.distributed as dist

    from aon_amd.arena import ParamArena
    from aon_amd.parallel import allreduce_gradients, broadcast_parameters, check_gradient_exchange

    own_group = not dist.is_initialized()
    if own_group:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        model, lib, rays, target, draws, ids = _art_setup(dev, n=128)
        both = torch.nn.ModuleList([model, lib])
        arena = ParamArena([model, lib])
        _art_step(model, lib, rays, target[:128], (draws[0][:128], draws[1][:128]), ids)
        before = {k: v.clone() for k, v in _named_grads(model, lib).items()}
        ptr = arena.grad.data_ptr()
        allreduce_gradients(both, force=True)
        check_gradient_exchange()
        after = _named_grads(model, lib)
        for k in before:
            assert torch.equal(after[k], before[k]), k
        assert arena.grad.data_ptr() == ptr and all(arena.grad_in_place(i) for i in range(len(arena.params)))
        vals = arena.flat.clone()
        broadcast_parameters(both, force=True)
        assert torch.equal(arena.flat, vals)
    finally:
        if own_group:
            dist.destroy_process_group()


@pytest.mark.parametrize("seed", range(8))
def test_arena_sweep_constructor_arguments(dev, seed):
    """The arena path at random constructor arguments (sample counts, lindisp, density noise, the three encoding routes of the vanilla
    network -- the slot-layout temporaries + remap kernels of other degrees write THROUGH the gradient pointers --, num_levels = 1,
    articulated activation scalars): gradients with the arena are the bits without it and sit in their slots; an optimiser step of
    ArenaAdam on them stays with torch.optim.Adam's."""
    import numpy as np

    import aon_amd.synthetic as syn
    from aon_amd.arena import ArenaAdam, ParamArena
    from aon_amd.models.vanilla_nerf.model import NeRF
    from aon_amd.models.vanilla_nerf.model_autodecoder import NeRF_AE_Art
    from oracle import nerf_oracle as orc   # (code-library rows for the latents only: no oracle arithmetic is compared here)

    rng = np.random.Generator(np.random.PCG64(9100 + seed))
    n = int(rng.integers(8, 200))
    nc, nf = int(rng.integers(2, 100)), int(rng.integers(1, 220))
    kw = dict(num_coarse_samples=nc, num_fine_samples=nf, lindisp=bool(rng.integers(0, 2)), noise_std=float(rng.choice([0.0, 0.4])))
    white = bool(rng.integers(0, 2))
    art = seed % 4 == 3
    levels = 1 if seed == 5 else 2
    g = torch.Generator().manual_seed(seed)
    rays = {k: v.to(dev) for k, v in syn.random_rays(n, seed=300 + seed).items()}
    target = torch.rand(n, 3, generator=g).to(dev)
    args = dict(t_rand=torch.rand(n, nc + 1, generator=g).to(dev), u=torch.rand(n, nf, generator=g).to(dev),
                noise=[torch.rand(n, nc + 1, generator=g).to(dev), torch.rand(n, nc + 1 + nf, generator=g).to(dev)][:levels])

    def build():
        if art:
            m = NeRF_AE_Art(rgb_padding=0.02, density_bias=0.3, **kw).to(dev)
            m.load_state_dict(syn.make_art_state_dict(seed=seed, density_scale=2.0))
            lat = {k: v.to(dev).clone().requires_grad_(True) for k, v in
                   orc.code_library(syn.make_code_library_state(seed=seed, n_max_objs=2), torch.tensor([seed % 2]), torch.tensor([seed % 10])).items()}
            return m, lat
        gk = [dict(), dict(min_deg_point=0, max_deg_point=7, deg_view=3), dict(min_deg_point=1, max_deg_point=9, deg_view=2)][seed % 3]
        m = NeRF(num_levels=levels, **kw, **gk).to(dev)
        m.load_state_dict(syn.make_general_nerf_state_dict(6000 + seed, **gk))
        return m, None

    def backward(m, lat):
        out = m(rays, True, white, 2.0, 6.0, lat, **args) if art else m(rays, True, white, 2.0, 6.0, **args)
        sum(((o[0] - target) ** 2).mean() for o in out).backward()
        return {k: p.grad for k, p in m.named_parameters() if p.grad is not None}

    plain_m, plain_lat = build()
    plain = backward(plain_m, plain_lat)
    m, lat = build()
    arena = ParamArena(m)
    got = backward(m, lat)
    assert set(got) == set(plain) and len(got) == (len(arena.params) if levels == 2 else len(arena.params) // 2)
    for k in plain:
        assert torch.equal(got[k], plain[k]), (k, kw)
    with_grad = [i for i, p in enumerate(arena.params) if p.grad is not None]
    assert all(arena.grad_in_place(i) for i in with_grad)
    if lat is not None:
        for k in lat:
            assert torch.equal(lat[k].grad, plain_lat[k].grad), k
    # one optimiser step on these (bit-identical) gradients: the arena's one-launch Adam against torch's on the plain twin (several steps on
    # identical gradients: test_arena_adam_is_torch_adam_in_one_launch; a second step HERE would compare two backward passes of
    # parameters that already differ in their last bits)
    opt, ref = ArenaAdam(arena, lr=5e-4), torch.optim.Adam(plain_m.parameters(), lr=5e-4, betas=(0.9, 0.999), foreach=False, fused=False)
    opt.step()
    ref.step()
    assert opt.last_launches == 1
    for (k, pa), pr in zip(m.named_parameters(), plain_m.parameters()):
        assert (pa - pr).abs().max().item() <= 4e-7 * (pr.abs().max().item() + 1e-12), k


def test_training_buffers_are_pooled(dev):
    """ops._TRAIN_POOL (round 6): the forward -> backward workspace and the backward's scratch are not handed back to torch's caching allocator
    between steps -- the reserved memory of a training loop is flat after its first step (through the allocator, a full-size request that found
    its block carved up went to hipMalloc inside a step: 80-125 ms, `profiles/r06_slowmode.txt`) --, two live graphs never share a workspace,
    and release_workspaces() drops the pool."""
    from aon_amd import ops

    model, lib, rays, target, draws, ids = _art_setup(dev, n=512)
    r2 = {k: v.repeat(1, 1) for k, v in rays.items()}
    d2 = (draws[0].repeat(3, 1)[:512], draws[1].repeat(3, 1)[:512])
    tg = target.repeat(3, 1)[:512]
    ops.release_workspaces()
    assert not ops._TRAIN_POOL

    def step():
        for p in list(model.parameters()) + list(lib.parameters()):
            p.grad = None
        _art_step(model, lib, r2, tg, d2, ids)

    step()
    step()
    torch.cuda.synchronize()
    keys = {k: len(v) for k, v in ops._TRAIN_POOL.items()}
    assert len(keys) == 2 and all(n == 1 for n in keys.values()), keys          # one workspace, one scratch, both back in the pool
    ptrs = sorted(t.data_ptr() for v in ops._TRAIN_POOL.values() for t, _ in v)
    reserved = torch.cuda.memory_stats()["reserved_bytes.all.current"]
    mallocs = torch.cuda.memory_stats()["num_device_alloc"]
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    assert sorted(t.data_ptr() for v in ops._TRAIN_POOL.values() for t, _ in v) == ptrs     # the same two buffers, step after step
    grown = torch.cuda.memory_stats()["reserved_bytes.all.current"] - reserved
    assert grown <= 64 << 20, (grown, torch.cuda.memory_stats()["num_device_alloc"] - mallocs)
    # two live graphs: the second forward must not get the first one's workspace
    for p in list(model.parameters()) + list(lib.parameters()):
        p.grad = None
    lat = lib(ids)
    out_a = model(r2, True, True, 2.0, 6.0, lat, t_rand=d2[0], u=d2[1])
    out_b = model(r2, True, True, 2.0, 6.0, lat, t_rand=d2[0], u=d2[1])
    assert torch.equal(out_a[1][0], out_b[1][0])
    (out_a[1][0].sum() + out_b[1][0].sum()).backward()
    torch.cuda.synchronize()
    assert all(len(v) <= 2 for v in ops._TRAIN_POOL.values())
    ga = {k: p.grad.clone() for k, p in model.named_parameters()}
    for p in list(model.parameters()) + list(lib.parameters()):
        p.grad = None
    out_c = model(r2, True, True, 2.0, 6.0, lib(ids), t_rand=d2[0], u=d2[1])
    (2.0 * out_c[1][0].sum()).backward()
    for k, p in model.named_parameters():
        assert torch.equal(p.grad, ga[k]), k     # (a + a against 2a: exact in fp32, and the backward is exactly linear in the upstream gradient)
    ops.release_workspaces()
    assert not ops._TRAIN_POOL
    # a loop whose batch size changes every step must not pile buffers up: at most _TRAIN_POOL_SIZES distinct sizes are kept
    for n in (64, 96, 128, 160, 192, 224, 256, 288):
        for p in list(model.parameters()) + list(lib.parameters()):
            p.grad = None
        o = model({k: v[:n] for k, v in r2.items()}, True, True, 2.0, 6.0, lib(ids), t_rand=d2[0][:n], u=d2[1][:n])
        o[1][0].sum().backward()
    assert len(ops._TRAIN_POOL) <= ops._TRAIN_POOL_SIZES
    ops.release_workspaces()


def test_gradient_bits_are_pinned(dev):
    """Every gradient tensor of the articulated config-5 step and of the vanilla step on 4096 seeded rays (131 tensors + the two losses),
    hashed (tools/grad_hash.py) and held to tests/golden/g24_gradient_hashes.json: the kernels are deterministic, so ANY change that moves a
    bit of a gradient shows here -- schedule changes (side streams, launch merges, load batching) must not, arithmetic changes update the file
    on purpose (`python tools/grad_hash.py --write`)."""
    import json
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import grad_hash

    want = json.load(open(os.path.join(root, "tests", "golden", "g24_gradient_hashes.json")))
    got = dict(grad_hash.hashes(int(want["n_rays"])))
    assert set(got) == set(want["hashes"])
    moved = [k for k in got if got[k] != want["hashes"][k]]
    assert not moved, f"{len(moved)} of {len(got)} tensors changed bits: {moved[:6]}"


def test_gradient_bits_do_not_depend_on_the_schedule_switches(dev):
    """Round 6's launch merges are schedule changes only: with every one switched off in the environment (the prologue as six pack calls,
    a second stage per level, the early reductions joined behind the chain, one pack launch per network) a fresh process produces the
    pinned gradient hashes too.  (The switches are read once per process, hence the subprocess.)"""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    want = json.load(open(os.path.join(root, "tests", "golden", "g24_gradient_hashes.json")))["hashes"]
    env = dict(os.environ, AON_PACK_STEP="0", AON_PACK_MERGE="0", AON_POST_MERGE="0", AON_EARLY_JOIN="1", AON_ART_AUX_HEADS="0")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "grad_hash.py")], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    got = {}
    for line in out.stdout.splitlines():
        parts = line.split()
        if len(parts) >= 2 and len(parts[-1]) == 16:
            got[" ".join(parts[:-1])] = parts[-1]
    assert set(got) == set(want), (sorted(set(want) - set(got))[:4], sorted(set(got) - set(want))[:4])
    moved = [k for k in got if got[k] != want[k]]
    assert not moved, f"{len(moved)} of {len(got)} tensors changed bits with the merges switched off: {moved[:6]}"
