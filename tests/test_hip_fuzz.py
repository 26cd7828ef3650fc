"""Randomised sweep of the HIP path against the oracle: ragged ray counts, both networks, both sampling modes, both
backgrounds, unusual near/far, non-unit directions, weights of several density scales -- forward renders and training
gradients.  Tolerances are those of the parity tests (PSNR >= 70 dB and max error stated per case); the point of this
file is breadth (sizes and flag combinations the hand-picked cases do not visit), seeds are fixed."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nerf_oracle as orc  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def psnr(a, b):
    return -10.0 * torch.log10(torch.clamp(torch.mean((a - b) ** 2), min=1e-20)).item()


def rel_l2(a, b):
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _rays(n, rng, unit):
    o = torch.from_numpy(rng.normal(size=(n, 3)).astype(np.float32))
    o = 4.0 * o / o.norm(dim=-1, keepdim=True) * torch.from_numpy(rng.uniform(0.9, 1.1, size=(n, 1)).astype(np.float32))
    tgt = torch.from_numpy(rng.uniform(-0.5, 0.5, size=(n, 3)).astype(np.float32))
    d = tgt - o
    v = d / d.norm(dim=-1, keepdim=True)
    return {"rays_o": o.contiguous(), "rays_d": (v if unit else d * 0.3).contiguous(), "viewdirs": v.contiguous()}


CASES = [(seed, n) for seed, n in zip(range(24), (1, 3, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 255, 257, 300, 383, 500, 511, 513, 640, 777, 96,
                                                  160, 224))]


@pytest.mark.parametrize("seed,n", CASES)
def test_forward_sweep(dev, seed, n):
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model import NeRF
    from aon_amd.models.vanilla_nerf.model_autodecoder import NeRF_AE_Art

    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    art = seed % 3 == 2
    randomized, white, unit = bool(seed & 1), bool(seed & 2), seed % 5 != 4
    near, far = ((2.0, 6.0), (1.5, 7.0), (2.5, 5.5))[seed % 3]
    scale = (30.0, 5.0, 60.0)[(seed // 3) % 3]
    rays_cpu = _rays(n, rng, unit)
    rays = {k: v.to(dev) for k, v in rays_cpu.items()}
    g = torch.Generator().manual_seed(seed)
    draws = dict(t_rand=torch.rand(n, 65, generator=g), u=torch.rand(n, 128, generator=g)) if randomized else {}
    draws_dev = {k: v.to(dev) for k, v in draws.items()}
    robust = torch.ones(n, dtype=torch.bool)
    if art:
        sd = syn.make_art_state_dict(seed=seed, density_scale=scale)
        lat = orc.code_library(syn.make_code_library_state(seed=seed, n_max_objs=2), torch.tensor([seed % 2]), torch.tensor([seed % 10]))
        model = NeRF_AE_Art().to(dev)
        model.load_state_dict(sd)
        with torch.no_grad():
            out = model(rays, randomized, white, near, far, {k: v.to(dev) for k, v in lat.items()}, **draws_dev)
        ref = orc.nerf_ae_art_forward(sd, rays_cpu, randomized, white, near, far, lat, **draws)
        ref64 = orc.nerf_ae_art_forward({k: v.double() for k, v in sd.items()}, {k: v.double() for k, v in rays_cpu.items()}, randomized,
                                        white, near, far, {k: v.double() for k, v in lat.items()}, **{k: v.double() for k, v in draws.items()})
    else:
        sd = syn.make_nerf_state_dict(seed=seed, density_scale=scale)
        model = NeRF().to(dev)
        model.load_state_dict(sd)
        with torch.no_grad():
            out = model(rays, randomized, white, near, far, **draws_dev)
        ref, aux = orc.nerf_forward(sd, rays_cpu, randomized, white, near, far, return_aux=True, **draws)
        ref64 = orc.nerf_forward({k: v.double() for k, v in sd.items()}, {k: v.double() for k, v in rays_cpu.items()}, randomized, white,
                                 near, far, **{k: v.double() for k, v in draws.items()})
        # the 1e10-long last interval makes alpha_last a step function of sign(raw sigma_last) in the vanilla model (a
        # discontinuity of the reference's own math): rays whose far-plane density is not robustly signed are
        # compared at the stated end-to-end tolerance only
        for a in aux:
            robust &= a["raw_sigma"][:, -1, 0].abs() > 2e-2 * scale / 30.0
    for lvl in (0, 1):
        rgb, acc, depth = (x.cpu() for x in out[lvl])
        assert rgb.shape == (n, 3) and acc.shape == (n,) and depth.shape == (n,)
        assert torch.isfinite(rgb).all() and torch.isfinite(acc).all()
        err = (rgb - ref[lvl][0]).abs().max(dim=-1).values
        assert (err <= 1e-3).double().mean().item() >= (0.99 if n >= 200 else 1.0 - 2.0 / max(n, 2)), (lvl, err.max().item())
        if robust.any():
            # The fine level is ill-conditioned on sharp random fields: 1-ulp differences of the coarse weights move
            # inverse-CDF draws across thin shells, and the ORACLE ITSELF differs between fp32 and fp64 by up to 2e-2 on such
            # rays while the coarse level agrees to 3e-7 (tests/diag/diag_fuzz_case.py).  So each ray is held to the base
            # tolerance plus three times the oracle's own fp32-vs-fp64 spread on that ray.
            spread = (ref[lvl][0].double() - ref64[lvl][0]).abs().max(dim=-1).values.float()
            spread_acc = (ref[lvl][1].double() - ref64[lvl][1]).abs().float()
            base, base_acc = (1e-3, 2e-3) if art else (2e-4, 2e-4)
            bad = robust & (err > base + 3 * spread)
            assert not bad.any(), (lvl, err[bad].max().item(), spread[bad].max().item())
            bad = robust & ((acc - ref[lvl][1]).abs() > base_acc + 3 * torch.maximum(spread, spread_acc))
            assert not bad.any(), (lvl, (acc - ref[lvl][1]).abs()[bad].max().item())
            if lvl == 0:
                assert psnr(rgb[robust], ref[lvl][0][robust]) >= 70.0


@pytest.mark.parametrize("seed,n", [(0, 1), (1, 5), (2, 33), (3, 100), (4, 130), (5, 200)])
def test_training_gradient_sweep(dev, seed, n):
    """Coarse-level gradients (identical sample positions on both sides) are compared tightly; fine-level ones inherit the
    inverse-CDF input sensitivity and are bounded at the level the dedicated tests state."""
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model import NeRF
    from aon_amd.models.vanilla_nerf.model_autodecoder import NeRF_AE_Art

    rng = np.random.Generator(np.random.PCG64(2000 + seed))
    art = seed % 2 == 1
    white = bool(seed & 2)
    rays_cpu = _rays(n, rng, True)
    g = torch.Generator().manual_seed(100 + seed)
    target = torch.rand(n, 3, generator=g)
    t_rand, u = torch.rand(n, 65, generator=g), torch.rand(n, 128, generator=g)
    rays = {k: v.to(dev) for k, v in rays_cpu.items()}
    if art:
        sd = syn.make_art_state_dict(seed=seed, density_scale=10.0)
        lat_cpu = orc.code_library(syn.make_code_library_state(seed=seed, n_max_objs=1), torch.tensor([0]), torch.tensor([seed % 10]))
        model = NeRF_AE_Art().to(dev)
        model.load_state_dict(sd)
        lat = {k: v.to(dev).requires_grad_(True) for k, v in lat_cpu.items()}
        out = model(rays, True, white, 2.0, 6.0, lat, t_rand=t_rand.to(dev), u=u.to(dev))
        sd_o = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        out_o = orc.nerf_ae_art_forward(sd_o, rays_cpu, True, white, 2.0, 6.0, lat_cpu, t_rand=t_rand, u=u)
    else:
        sd = syn.make_nerf_state_dict(seed=seed, density_scale=10.0)
        model = NeRF().to(dev)
        model.load_state_dict(sd)
        out = model(rays, True, white, 2.0, 6.0, t_rand=t_rand.to(dev), u=u.to(dev))
        sd_o = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        out_o, aux = orc.nerf_forward(sd_o, rays_cpu, True, white, 2.0, 6.0, t_rand=t_rand, u=u, return_aux=True)
        robust = all((a["raw_sigma"][:, -1, 0].abs() > 2e-2 * 10.0 / 30.0).all().item() for a in aux)
    if art:
        robust = True   # softplus density: no far-plane sign discontinuity
    loss = torch.mean((out[0][0] - target.to(dev)) ** 2) + torch.mean((out[1][0] - target.to(dev)) ** 2)
    loss.backward()
    loss_o = orc.img2mse(out_o[0][0], target) + orc.img2mse(out_o[1][0], target)
    loss_o.backward()
    assert abs(loss.item() - loss_o.item()) <= 1e-4 * max(1.0, abs(loss_o.item()))
    for name, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
        ref = sd_o[name].grad
        if ref.norm().item() < 1e-12:
            assert p.grad.abs().max().item() <= 1e-9, name
            continue
        tol = 5e-2 if art else (1e-2 if name.startswith("fine_mlp") else 2e-3)
        if not robust:   # a ray whose far-plane sigma is within rounding of zero flips a whole alpha_last = {0,1} term
            tol = 5e-2
        assert rel_l2(p.grad.cpu(), ref) <= tol, (name, rel_l2(p.grad.cpu(), ref))


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("AON_FUZZ_SEEDS", "16"))))
def test_forward_sweep_constructor_arguments(dev, seed):
    """Random constructor arguments (sample counts 2..100 / 1..200, lindisp, density noise, encoding degrees on all three routes
    -- fused, fused with zero-weight slots, layer-wise engine -- articulated rgb_padding / density_bias), ragged ray counts,
    against the oracle built with the same arguments.  Smooth fields; each ray is held to 5e-5 plus three times the oracle's own
    fp32-vs-fp64 spread on it."""
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model import NeRF
    from aon_amd.models.vanilla_nerf.model_autodecoder import NeRF_AE_Art

    rng = np.random.Generator(np.random.PCG64(3000 + seed))
    n = int(rng.integers(1, 400))
    nc, nf = int(rng.integers(2, 101)), int(rng.integers(1, 201))
    lindisp, randomized, white = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    noise_std = float(rng.choice([0.0, 0.5, 1.0]))
    near, far = ((2.0, 6.0), (0.7, 7.0), (2.5, 5.5))[seed % 3]
    art = seed % 4 == 3
    rays_cpu = _rays(n, rng, True)
    rays = {k: v.to(dev) for k, v in rays_cpu.items()}
    g = torch.Generator().manual_seed(seed)
    draws = dict(t_rand=torch.rand(n, nc + 1, generator=g), u=torch.rand(n, nf, generator=g)) if randomized else {}
    noise = [torch.rand(n, nc + 1, generator=g), torch.rand(n, nc + 1 + nf, generator=g)]
    kw = dict(num_coarse_samples=nc, num_fine_samples=nf, lindisp=lindisp, noise_std=noise_std)
    dd = lambda d: {k: (v.double() if torch.is_tensor(v) else v) for k, v in d.items()}   # noqa: E731
    if art:
        kw.update(rgb_padding=float(rng.choice([0.001, 0.02])), density_bias=float(rng.choice([-1.0, 0.3])))
        sd = syn.make_art_state_dict(seed=seed, density_scale=2.0)
        lat = orc.code_library(syn.make_code_library_state(seed=seed, n_max_objs=2), torch.tensor([seed % 2]), torch.tensor([seed % 10]))
        model = NeRF_AE_Art(**kw).to(dev)
        model.load_state_dict(sd)
        with torch.no_grad():
            out = model(rays, randomized, white, near, far, {k: v.to(dev) for k, v in lat.items()}, noise=[z.to(dev) for z in noise],
                        **{k: v.to(dev) for k, v in draws.items()})
        ref = orc.nerf_ae_art_forward(sd, rays_cpu, randomized, white, near, far, lat, noise=noise, **draws, **kw)
        ref64 = orc.nerf_ae_art_forward(dd(sd), dd(rays_cpu), randomized, white, near, far, dd(lat), noise=[z.double() for z in noise], **dd(draws), **kw)
        robust = torch.ones(n, dtype=torch.bool)
    else:
        gk = [dict(), dict(min_deg_point=0, max_deg_point=7, deg_view=3), dict(min_deg_point=1, max_deg_point=12, deg_view=5),
              dict(min_deg_point=-1, max_deg_point=5, deg_view=0)][(seed // 4) % 4]
        sd = syn.make_general_nerf_state_dict(4000 + seed, **gk)
        model = NeRF(**kw, **gk).to(dev)
        model.load_state_dict(sd)
        with torch.no_grad():
            out = model(rays, randomized, white, near, far, noise=[z.to(dev) for z in noise], **{k: v.to(dev) for k, v in draws.items()})
        ref, aux = orc.nerf_forward(sd, rays_cpu, randomized, white, near, far, return_aux=True, noise=noise, **draws, **kw, **gk)
        ref64 = orc.nerf_forward(dd(sd), dd(rays_cpu), randomized, white, near, far, noise=[z.double() for z in noise], **dd(draws), **kw, **gk)
        robust = torch.ones(n, dtype=torch.bool)
        for lvl, a in enumerate(aux):   # far-plane sign margin of the vanilla relu, with the noise this pass adds
            last = a["raw_sigma"][:, -1, 0] + (noise[lvl][:, -1] * noise_std if (noise_std > 0 and randomized) else 0.0)
            robust &= last.abs() > 2e-2
    for lvl in (0, 1):
        rgb, acc, depth = (x.cpu() for x in out[lvl])
        assert rgb.shape == (n, 3) and torch.isfinite(rgb).all() and torch.isfinite(acc).all()
        if robust.any():
            err = (rgb - ref[lvl][0]).abs().max(dim=-1).values
            spread = (ref[lvl][0].double() - ref64[lvl][0]).abs().max(dim=-1).values.float()
            bad = robust & (err > 5e-5 + 3 * spread)
            assert not bad.any(), (lvl, err[bad].max().item(), spread[bad].max().item(), kw)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("AON_FUZZ_TRAIN_SEEDS", "8"))))
def test_training_sweep_constructor_arguments(dev, seed):
    """loss.backward() at random constructor arguments (sample counts up to 120 + 300, lindisp, density noise, the three encoding
    routes, articulated activation scalars): every gradient as close to the oracle's fp64 autograd as the oracle's own fp32 is
    (tests/_gradcheck.py), smooth fields."""
    import aon_amd.synthetic as syn
    from _gradcheck import assert_as_close_as_fp32
    from aon_amd.models.vanilla_nerf.model import NeRF
    from aon_amd.models.vanilla_nerf.model_autodecoder import NeRF_AE_Art

    rng = np.random.Generator(np.random.PCG64(5000 + seed))
    n = int(rng.integers(8, 160))
    nc, nf = int(rng.integers(2, 121)), int(rng.integers(1, 301))
    lindisp, white = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    noise_std = float(rng.choice([0.0, 0.4]))
    art = seed % 4 == 3
    rays_cpu = _rays(n, rng, True)
    g = torch.Generator().manual_seed(seed)
    target = torch.rand(n, 3, generator=g)
    draws = dict(t_rand=torch.rand(n, nc + 1, generator=g), u=torch.rand(n, nf, generator=g))
    noise = [torch.rand(n, nc + 1, generator=g), torch.rand(n, nc + 1 + nf, generator=g)]
    kw = dict(num_coarse_samples=nc, num_fine_samples=nf, lindisp=lindisp, noise_std=noise_std)
    lat0 = None
    if art:
        kw.update(rgb_padding=float(rng.choice([0.001, 0.02])), density_bias=float(rng.choice([-1.0, 0.3])))
        sd = syn.make_art_state_dict(seed=seed, density_scale=2.0)
        lat0 = orc.code_library(syn.make_code_library_state(seed=seed, n_max_objs=2), torch.tensor([seed % 2]), torch.tensor([seed % 10]))
        model = NeRF_AE_Art(**kw).to(dev)
        gk = {}
    else:
        gk = [dict(), dict(min_deg_point=0, max_deg_point=7, deg_view=3), dict(min_deg_point=1, max_deg_point=12, deg_view=5)][seed % 3]
        sd = syn.make_general_nerf_state_dict(6000 + seed, **gk)
        model = NeRF(**kw, **gk).to(dev)
    model.load_state_dict(sd)

    def oracle_grads(dtype):
        sd_o = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in sd.items()}
        r = {k: v.to(dtype) for k, v in rays_cpu.items()}
        common = dict(noise=[z.to(dtype) for z in noise], **{k: v.to(dtype) for k, v in draws.items()}, **kw)
        if art:
            lat = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in lat0.items()}
            out = orc.nerf_ae_art_forward(sd_o, r, True, white, 2.0, 6.0, lat, **common)
        else:
            lat = {}
            out = orc.nerf_forward(sd_o, r, True, white, 2.0, 6.0, **common, **gk)
        loss = orc.img2mse(out[0][0], target.to(dtype)) + orc.img2mse(out[1][0], target.to(dtype))
        loss.backward()
        gr = {k: v.grad for k, v in sd_o.items()}
        gr.update({f"latent[{k}]": v.grad for k, v in lat.items()})
        return loss.item(), gr

    (_, truth), (loss32, ref32) = oracle_grads(torch.float64), oracle_grads(torch.float32)
    rays = {k: v.to(dev) for k, v in rays_cpu.items()}
    args = dict(noise=[z.to(dev) for z in noise], **{k: v.to(dev) for k, v in draws.items()})
    if art:
        lat = {k: v.to(dev).clone().requires_grad_(True) for k, v in lat0.items()}
        out = model(rays, True, white, 2.0, 6.0, lat, **args)
    else:
        out = model(rays, True, white, 2.0, 6.0, **args)
    loss = ((out[0][0] - target.to(dev)) ** 2).mean() + ((out[1][0] - target.to(dev)) ** 2).mean()
    loss.backward()
    assert abs(loss.item() - loss32) < 5e-6, (loss.item(), loss32, kw, gk)
    hip = {name: p.grad.cpu() for name, p in model.named_parameters()}
    if art:
        hip.update({f"latent[{k}]": lat[k].grad.cpu() for k in lat})
    # a sweep, not a pin: the ratio to the reference-fp32's own error has a distribution over random geometries -- measured over
    # 150 seeds on MI355X (round 4, profiles/r04_fuzz150_training.txt): median of the per-seed worst ratio 1.1x, 90th percentile 4.6x,
    # 14 seeds above 5x, 8 above 10x (12x on trunk layers at the 4e-3 level behind coarse inverse CDFs, 11x on the articulated
    # deformation head), 2 above 20x (38x / 43x, both a density bias at 2-3e-5 ABSOLUTE relative error, see below); a wrong kernel is
    # off by O(1).  The dedicated tests (test_hip_smooth.py, test_hip_training.py G9 inputs, test_hip_ctor_options.py) hold 5x.
    # The one- and three-element head biases (signed sums over every sample: cancellation) are under the SAME factor since round 4 (round
    # 3: their own call with a 1e-2 floor).  Their sums and the whole compositing backward are fp64 on the device now (aon_wgrad.h,
    # aon_train.hip), so what is left is the fp32 noise of the forward's raw sigma values times the cancellation of the sum (x20 - x6,000
    # over these geometries) -- the same for any fp32 forward; where the reference's own fp32 lands is luck (4.8e-7 ... 8.8e-5 on the seeds
    # below).  Measured over 150 seeds (round 4): worst density bias 3.2e-5 (seeds 75, 95), everything else under 2e-5 or inside 5 x the
    # reference -- hence the sweep's floor of 5e-5 for <= 4-element parameters (the dedicated tests keep _gradcheck's 2e-5).
    assert_as_close_as_fp32(hip, truth, ref32, f"seed {seed}: {kw} {gk}", factor=25.0, floor=1e-3, small_floor=5e-5)
