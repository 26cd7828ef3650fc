"""GPU: constructor arguments beyond the reference's defaults (VERDICT r2 missing #3) against G16 -- outputs of the REAL reference
built with other sample counts, ``lindisp=True``, ``noise_std > 0`` and (articulated) other ``rgb_padding`` / ``density_bias`` --
and against the oracle.  Bars as for the default geometry: stratified t and the inverse CDF / merge bit-exact (``torch.equal``);
the whole path on the smooth fields 2e-6 (rgb, acc) / 2e-5 (depth) on every ray; gradients against the oracle's autograd."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nerf_oracle as orc  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def test_lindisp_sampling_bit_exact(dev, golden):
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf import helper

    g = golden("g16_ctor_options")
    o, d = g["lindisp_rays_o"].to(dev), g["lindisp_rays_d"].to(dev)
    for tag in "abc":
        ns, near, far = g[f"lindisp_{tag}_ns"], g[f"lindisp_{tag}_near"], g[f"lindisp_{tag}_far"]
        t, _ = helper.sample_along_rays(o, d, ns, near, far, False, True)
        assert torch.equal(t.cpu(), g[f"lindisp_{tag}_t_det"])
        tr = syn.seeded_uniform(1600 + ns, 48, ns + 1).to(dev)
        t, c = helper.sample_along_rays(o, d, ns, near, far, True, True, t_rand=tr)
        assert torch.equal(t.cpu(), g[f"lindisp_{tag}_t_rnd"])
        torch.testing.assert_close(c.double().sum((0, 1)).cpu(), g[f"lindisp_{tag}_coords_rnd_sum"], rtol=1e-6, atol=0)
        # the t-only kernel of the whole-path calls (four values per thread) gives the same bits
        from aon_amd import ops
        t4, _ = ops.sample_along_rays(o, d, ns, near, far, tr, want_coords=False, lindisp=True)
        assert torch.equal(t4.cpu(), g[f"lindisp_{tag}_t_rnd"])


def test_inverse_cdf_any_size_bit_exact(dev, golden):
    """aon_sample_pdf_n against the reference's draws and sorted unions at nine (bins, draws) geometries: scalar and vector
    paths of ATen's sum, the cascade (K = 599), non-power-of-two unions."""
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf import helper

    g = golden("g16_ctor_options")
    for nb, nf in g["pdf_sizes"].tolist():
        k = f"pdf_{nb}_{nf}"
        t, w = g[f"{k}_t"].to(dev), g[f"{k}_w"].to(dev)
        n = t.shape[0]
        mids = 0.5 * (t[..., 1:] + t[..., :-1])
        u = syn.seeded_uniform(1700 + nb, n, nf).to(dev)
        z = torch.zeros(n, 3, device=dev)
        assert torch.equal(helper.sorted_piecewise_constant_pdf(mids, w, nf, False).cpu(), g[f"{k}_samples_det"]), k
        assert torch.equal(helper.sorted_piecewise_constant_pdf(mids, w, nf, True, u=u).cpu(), g[f"{k}_samples_rnd"]), k
        assert torch.equal(helper.sample_pdf(mids, w, z, z, t, nf, False)[0].cpu(), g[f"{k}_t_fine_det"]), k
        assert torch.equal(helper.sample_pdf(mids, w, z, z, t, nf, True, u=u)[0].cpu(), g[f"{k}_t_fine_rnd"]), k


def test_general_size_kernel_equals_the_specialised_one(dev, golden):
    """At 64 bins / 128 draws both kernels exist: same bits (G6 / G7 inputs, deterministic and random u)."""
    from aon_amd import ops

    g6, g7 = golden("g6_pdf"), golden("g7_sample_pdf")
    t, w, u = g7["t_vals"].to(dev), g7["weights"].to(dev), g7["u"].to(dev)
    for uu in (None, u):
        assert torch.equal(ops.sample_pdf_t_n(t, w, 128, uu), ops.sample_pdf_t(t, w, uu))
    assert torch.equal(ops.sample_pdf_t_n(t, w, 128, None).cpu(), g7["t_fine_det"])
    # adversarial rows: wide dynamic range, many zero weights
    gen = torch.Generator().manual_seed(5)
    n = 4000
    tt = torch.sort(torch.rand((n, 65), generator=gen) * 4 + 2, dim=-1).values.to(dev)
    ww = (torch.rand((n, 63), generator=gen) ** 12 * (torch.rand((n, 63), generator=gen) > 0.3)).to(dev)
    uu = torch.rand((n, 128), generator=gen).to(dev)
    assert torch.equal(ops.sample_pdf_t_n(tt, ww, 128, uu), ops.sample_pdf_t(tt, ww, uu))
    assert torch.equal(ops.sample_pdf_t_n(tt, ww, 128, None), ops.sample_pdf_t(tt, ww, None))


def _check(out, g, tag, drgb=2e-6, ddepth=2e-5):
    for lvl, name in ((0, "coarse"), (1, "fine")):
        rgb, acc, depth = (x.cpu() for x in out[lvl])
        torch.testing.assert_close(rgb, g[f"{tag}_{name}_rgb"], rtol=0, atol=drgb)
        torch.testing.assert_close(acc, g[f"{tag}_{name}_acc"], rtol=0, atol=drgb)
        torch.testing.assert_close(depth, g[f"{tag}_{name}_depth"], rtol=0, atol=ddepth)


def test_vanilla_with_options_end_to_end(dev, golden):
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model import NeRF

    g = golden("g16_ctor_options")
    assert g["van_margin_det"].min() > 0.05 and g["van_margin_rnd"].min() > 0.05   # nothing is masked
    nc, nf, lind = g["van_cfg"].tolist()
    model = NeRF(num_coarse_samples=nc, num_fine_samples=nf, lindisp=bool(lind), noise_std=g["van_noise_std"]).to(dev)
    model.load_state_dict(syn.make_smooth_nerf_state_dict())
    rays = {k: g["van_" + k].to(dev) for k in ("rays_o", "rays_d", "viewdirs")}
    n = rays["rays_o"].shape[0]
    s = g["van_seeds"].tolist()
    tr, u = syn.seeded_uniform(s[0], n, nc + 1).to(dev), syn.seeded_uniform(s[1], n, nf).to(dev)
    nz = [syn.seeded_uniform(s[2], n, nc + 1).to(dev), syn.seeded_uniform(s[3], n, nc + 1 + nf).to(dev)]
    with torch.no_grad():
        _check(model(rays, False, True, 2.0, 6.0), g, "van_det")
        _check(model(rays, True, False, 2.0, 6.0, t_rand=tr, u=u, noise=nz), g, "van_rnd")
        # chunk invariance at this geometry: a sub-range renders to the same bits alone
        full = model(rays, True, False, 2.0, 6.0, t_rand=tr, u=u, noise=nz)
        sub = {k: v[64:160].contiguous() for k, v in rays.items()}
        part = model(sub, True, False, 2.0, 6.0, t_rand=tr[64:160].contiguous(), u=u[64:160].contiguous(), noise=[z[64:160].contiguous() for z in nz])
        for lvl in (0, 1):
            for a, b in zip(full[lvl], part[lvl]):
                assert torch.equal(a[64:160], b)
        # num_levels = 1 at this geometry against the oracle
        m1 = NeRF(num_levels=1, num_coarse_samples=nc, lindisp=True).to(dev)
        m1.load_state_dict(syn.make_smooth_nerf_state_dict())
        o1 = m1(rays, False, True, 2.0, 6.0)
        ref = orc.nerf_forward(syn.make_smooth_nerf_state_dict(), {k: v.cpu() for k, v in rays.items()}, False, True, 2.0, 6.0, num_levels=1,
                               num_coarse_samples=nc, lindisp=True)
        torch.testing.assert_close(o1[0][0].cpu(), ref[0][0], rtol=0, atol=2e-6)


def test_articulated_with_options_end_to_end(dev, golden):
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model_autodecoder import NeRF_AE_Art

    g = golden("g16_ctor_options")
    nc, nf, lind = g["art_cfg"].tolist()
    model = NeRF_AE_Art(num_coarse_samples=nc, num_fine_samples=nf, lindisp=bool(lind), noise_std=g["art_noise_std"],
                        rgb_padding=g["art_rgb_padding"], density_bias=g["art_density_bias"]).to(dev)
    model.load_state_dict(syn.make_art_state_dict(seed=5, density_scale=2.0))
    rays = {k: g["art_" + k].to(dev) for k in ("rays_o", "rays_d", "viewdirs")}
    lat = {k: g["art_lat_" + k].to(dev) for k in ("density", "color", "articulation")}
    n = rays["rays_o"].shape[0]
    s = g["art_seeds"].tolist()
    tr, u = syn.seeded_uniform(s[0], n, nc + 1).to(dev), syn.seeded_uniform(s[1], n, nf).to(dev)
    nz = [syn.seeded_uniform(s[2], n, nc + 1).to(dev), syn.seeded_uniform(s[3], n, nc + 1 + nf).to(dev)]
    with torch.no_grad():
        _check(model(rays, False, True, 2.0, 6.0, lat), g, "art_det")
        _check(model(rays, True, False, 2.0, 6.0, lat, t_rand=tr, u=u, noise=nz), g, "art_rnd")


@pytest.mark.parametrize("net", ["vanilla", "articulated"])
def test_training_step_with_options(dev, net):
    """loss.backward() through the drop-in modules at a non-default geometry (40 + 56 samples, lindisp, noise on the densities,
    randomized) on the smooth fields: every parameter (and latent) gradient as close to the oracle's fp64 autograd as the oracle's
    own fp32 autograd is, up to the factor of tests/test_hip_smooth.py."""
    import aon_amd.synthetic as syn
    from _gradcheck import assert_as_close_as_fp32
    from aon_amd.models.vanilla_nerf.model import NeRF
    from aon_amd.models.vanilla_nerf.model_autodecoder import NeRF_AE_Art

    n, nc, nf = 192, 40, 56
    art = net != "vanilla"
    frame = syn.make_rays(24, 32, syn.look_at_pose(4.0, 60, 20), syn.focal_from_fovy(24))
    rays_cpu = {k: v[::4][:n].contiguous() for k, v in frame.items()}
    rays = {k: v.to(dev) for k, v in rays_cpu.items()}
    target = syn.seeded_uniform(77, n, 3)
    tr, u = syn.seeded_uniform(78, n, nc + 1), syn.seeded_uniform(79, n, nf)
    nz = [syn.seeded_uniform(80, n, nc + 1), syn.seeded_uniform(81, n, nc + 1 + nf)]
    kw = dict(num_coarse_samples=nc, num_fine_samples=nf, lindisp=True, noise_std=0.3)
    lat0 = None
    if not art:
        sd = syn.make_smooth_nerf_state_dict()
        model = NeRF(**kw).to(dev)
    else:
        sd = syn.make_art_state_dict(seed=5, density_scale=2.0)
        lib = syn.make_code_library_state(seed=0, n_max_objs=2)
        lat0 = {"density": lib["embedding_instance_shape.weight"][1:2], "color": lib["embedding_instance_appearance.weight"][1:2],
                "articulation": lib["embedding_instance_articulation.weight"][3:4]}
        kw.update(rgb_padding=0.01, density_bias=-0.5)
        model = NeRF_AE_Art(**kw).to(dev)
    model.load_state_dict(sd)

    def oracle_grads(dtype):
        sd_o = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in sd.items()}
        r = {k: v.to(dtype) for k, v in rays_cpu.items()}
        common = dict(t_rand=tr.to(dtype), u=u.to(dtype), noise=[z.to(dtype) for z in nz], **kw)
        if art:
            lat = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in lat0.items()}
            out = orc.nerf_ae_art_forward(sd_o, r, True, True, 2.0, 6.0, lat, **common)
        else:
            lat = {}
            out = orc.nerf_forward(sd_o, r, True, True, 2.0, 6.0, **common)
        loss = orc.img2mse(out[0][0], target.to(dtype)) + orc.img2mse(out[1][0], target.to(dtype))
        loss.backward()
        gr = {k: v.grad for k, v in sd_o.items()}
        gr.update({f"latent[{k}]": v.grad for k, v in lat.items()})
        return loss.item(), gr

    (_, truth), (loss32, ref32) = oracle_grads(torch.float64), oracle_grads(torch.float32)
    args = dict(t_rand=tr.to(dev), u=u.to(dev), noise=[z.to(dev) for z in nz])
    if art:
        lat = {k: v.to(dev).clone().requires_grad_(True) for k, v in lat0.items()}
        out = model(rays, True, True, 2.0, 6.0, lat, **args)
    else:
        out = model(rays, True, True, 2.0, 6.0, **args)
    loss = ((out[0][0] - target.to(dev)) ** 2).mean() + ((out[1][0] - target.to(dev)) ** 2).mean()
    loss.backward()
    assert abs(loss.item() - loss32) < 2e-6
    hip = {name: p.grad.cpu() for name, p in model.named_parameters()}
    if art:
        hip.update({f"latent[{k}]": lat[k].grad.cpu() for k in lat})
    assert_as_close_as_fp32(hip, truth, ref32, net)


def test_training_above_256_samples_per_ray(dev):
    """More than 256 samples per ray take the eight-block form of the compositing backward (S <= 512): loss and gradients
    against the oracle's autograd at 101 + 250 samples; above 512 the training call refuses."""
    import aon_amd.synthetic as syn
    from _gradcheck import assert_as_close_as_fp32
    from aon_amd.models.vanilla_nerf.model import NeRF

    n, nc, nf = 64, 100, 250
    sd = syn.make_smooth_nerf_state_dict()
    model = NeRF(num_coarse_samples=nc, num_fine_samples=nf).to(dev)
    model.load_state_dict(sd)
    frame = syn.make_rays(16, 16, syn.look_at_pose(4.0, 60, 20), syn.focal_from_fovy(16))
    rays_cpu = {k: v[::4][:n].contiguous() for k, v in frame.items()}
    target = syn.seeded_uniform(91, n, 3)
    out = model({k: v.to(dev) for k, v in rays_cpu.items()}, False, True, 2.0, 6.0)
    loss = ((out[0][0] - target.to(dev)) ** 2).mean() + ((out[1][0] - target.to(dev)) ** 2).mean()
    loss.backward()

    def oracle_grads(dtype):
        sd_o = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in sd.items()}
        o = orc.nerf_forward(sd_o, {k: v.to(dtype) for k, v in rays_cpu.items()}, False, True, 2.0, 6.0, num_coarse_samples=nc, num_fine_samples=nf)
        l = orc.img2mse(o[0][0], target.to(dtype)) + orc.img2mse(o[1][0], target.to(dtype))
        l.backward()
        return l.item(), {k: v.grad for k, v in sd_o.items()}

    (_, truth), (loss32, ref32) = oracle_grads(torch.float64), oracle_grads(torch.float32)
    assert abs(loss.item() - loss32) < 2e-6
    assert_as_close_as_fp32({k: p.grad.cpu() for k, p in model.named_parameters()}, truth, ref32, "351 samples per ray")
    big = NeRF(num_coarse_samples=200, num_fine_samples=400).to(dev)
    with pytest.raises(Exception, match="512 samples"):
        big({k: v.to(dev) for k, v in rays_cpu.items()}, False, True, 2.0, 6.0)
    with torch.no_grad():                       # inference has no such limit
        big({k: v.to(dev) for k, v in rays_cpu.items()}, False, True, 2.0, 6.0)


def test_more_than_two_levels(dev):
    """NeRF(num_levels=3 / 4): every further level resamples from the previous level's t and weights with fine_mlp
    (model.py:162-173); inference through the stage-level calls against the oracle's loop, smooth field, every robust ray."""
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model import NeRF

    sd = syn.make_smooth_nerf_state_dict()
    frame = syn.make_rays(16, 24, syn.look_at_pose(4.0, 60, 20), syn.focal_from_fovy(16))
    rays_cpu = {k: v[::2].contiguous() for k, v in frame.items()}
    rays = {k: v.to(dev) for k, v in rays_cpu.items()}
    n = rays["rays_o"].shape[0]
    for levels, nc, nf in ((3, 64, 128), (4, 24, 40)):
        model = NeRF(num_levels=levels, num_coarse_samples=nc, num_fine_samples=nf).to(dev)
        model.load_state_dict(sd)
        tr, u = syn.seeded_uniform(101, n, nc + 1), syn.seeded_uniform(102, n, nf)
        with torch.no_grad():
            det = model(rays, False, True, 2.0, 6.0)
            rnd = model(rays, True, False, 2.0, 6.0, t_rand=tr.to(dev), u=[u.to(dev)] * (levels - 1))
        ref_d, aux = orc.nerf_forward(sd, rays_cpu, False, True, 2.0, 6.0, num_levels=levels, num_coarse_samples=nc, num_fine_samples=nf, return_aux=True)
        ref_r, aux_r = orc.nerf_forward(sd, rays_cpu, True, False, 2.0, 6.0, num_levels=levels, num_coarse_samples=nc, num_fine_samples=nf, t_rand=tr, u=u,
                                        return_aux=True)
        ok = torch.stack([a["raw_sigma"][:, -1, 0].abs() for a in aux + aux_r]).min(0).values > 0.05
        assert len(det) == levels and ok.float().mean() > 0.8
        for lvl in range(levels):
            assert det[lvl][0].shape == (n, 3)
            for out, ref in ((det, ref_d), (rnd, ref_r)):
                # (every resampling level re-amplifies the last bits of the previous level's weights: measured 5.2e-6 at level 3)
                torch.testing.assert_close(out[lvl][0].cpu()[ok], ref[lvl][0][ok], rtol=0, atol=2e-5)
                torch.testing.assert_close(out[lvl][2].cpu()[ok], ref[lvl][2][ok], rtol=0, atol=2e-4)


def test_training_with_more_than_two_levels(dev):
    """Training at num_levels = 3 and 4 (round 4; round 3 raised): every level an autograd node built from the stage-level training
    calls (autograd.RenderLevelVanilla), fine_mlp's gradients accumulated over the levels that use it (model.py:149-197 under the
    reference's autograd).  Loss = sum of the levels' mse; gradients by the fp64-truth yardstick of tests/_gradcheck.py."""
    import aon_amd.synthetic as syn
    from _gradcheck import assert_as_close_as_fp32
    from aon_amd.models.vanilla_nerf.model import NeRF

    sd = syn.make_smooth_nerf_state_dict()
    frame = syn.make_rays(16, 24, syn.look_at_pose(4.0, 60, 20), syn.focal_from_fovy(16))
    rays_cpu = {k: v[::3].contiguous() for k, v in frame.items()}
    rays = {k: v.to(dev) for k, v in rays_cpu.items()}
    n = rays["rays_o"].shape[0]
    target = syn.seeded_uniform(77, n, 3)
    for levels, nc, nf in ((3, 64, 128), (4, 24, 40)):
        tr, u = syn.seeded_uniform(101, n, nc + 1), syn.seeded_uniform(102, n, nf)
        kw = dict(num_levels=levels, num_coarse_samples=nc, num_fine_samples=nf)

        def oracle_grads(dtype):
            sd_o = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in sd.items()}
            out = orc.nerf_forward(sd_o, {k: v.to(dtype) for k, v in rays_cpu.items()}, True, True, 2.0, 6.0, t_rand=tr.to(dtype), u=u.to(dtype), **kw)
            loss = sum(orc.img2mse(o[0], target.to(dtype)) for o in out)
            loss.backward()
            return loss.item(), {k: v.grad for k, v in sd_o.items()}

        (_, truth), (loss32, ref32) = oracle_grads(torch.float64), oracle_grads(torch.float32)
        model = NeRF(**kw).to(dev)
        model.load_state_dict(sd)
        out = model(rays, True, True, 2.0, 6.0, t_rand=tr.to(dev), u=[u.to(dev)] * (levels - 1))
        assert len(out) == levels and all(o[0].requires_grad for o in out)
        loss = sum(((o[0] - target.to(dev)) ** 2).mean() for o in out)
        loss.backward()
        assert abs(loss.item() - loss32) < 5e-6, (loss.item(), loss32)
        hip = {name: p.grad.cpu() for name, p in model.named_parameters()}
        assert_as_close_as_fp32(hip, truth, ref32, f"num_levels = {levels}", factor=5.0, floor=1e-4)
        # an evaluation call in grad mode on frozen parameters takes the inference route
        for p in model.parameters():
            p.requires_grad_(False)
        assert not model(rays, False, True, 2.0, 6.0)[levels - 1][0].requires_grad


def test_bad_options_are_rejected(dev):
    from aon_amd import ops
    from aon_amd.models.vanilla_nerf.model import NeRF

    with pytest.raises(ValueError):
        NeRF(num_coarse_samples=1)
    with pytest.raises(NotImplementedError):
        NeRF(num_levels=3, max_deg_point=6)          # more than two levels: default network only
    with pytest.raises(ValueError):
        NeRF(num_levels=0)
    m = NeRF(num_coarse_samples=900, num_fine_samples=8000)      # exceeds the per-ray LDS image: the C call says so
    import aon_amd.synthetic as syn
    m = m.to(dev)
    frame = syn.make_rays(4, 4, syn.look_at_pose(), syn.focal_from_fovy(4))
    rays = {k: v.to(dev) for k, v in frame.items()}
    with pytest.raises(Exception, match="too large"):
        with torch.no_grad():
            m(rays, False, True, 2.0, 6.0)
    assert ops.RenderOpts().Sf == 193
