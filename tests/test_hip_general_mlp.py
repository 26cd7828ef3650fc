"""GPU: NeRFMLP / NeRF of non-default constructor geometry on the layer-wise MFMA GEMM engine (csrc/aon_gmlp.hip) against G17 --
outputs of the REAL reference's NeRFMLP(...) / NeRF(min_deg_point, max_deg_point, deg_view, ...) -- against the fused kernels
where both exist (the default geometry), and against the oracle's autograd for the training step."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nerf_oracle as orc  # noqa: E402  (checker only)

GEOM_KEYS = ("min_deg_point", "max_deg_point", "deg_view", "netdepth", "netwidth", "netdepth_condition", "netwidth_condition", "skip_layer",
             "input_ch", "input_ch_view", "num_rgb_channels", "num_density_channels")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def test_gemm_layers_against_fp64(dev):
    """The GEMM kernel alone through aon_gmlp_fwd on a one-layer-deep network with ragged sizes: every output against an fp64
    evaluation (tile edges in M, N and K; unaligned rows; the two-segment view layer with the per-ray condition row)."""
    import aon_amd.synthetic as syn
    from aon_amd import ops

    for kw, n, S in ((dict(max_deg_point=1, deg_view=0, netdepth=1, netwidth=5, netwidth_condition=3), 3, 7),
                     (dict(max_deg_point=5, deg_view=1, netdepth=2, netwidth=131, netwidth_condition=257, netdepth_condition=2), 5, 53),
                     (dict(max_deg_point=10, deg_view=4, netdepth=4, netwidth=256, netwidth_condition=128, skip_layer=2, netdepth_condition=1), 9, 129)):
        geom = ops.MlpGeometry(**kw)
        sd = syn.make_general_nerf_state_dict(11, prefixes=("",), **kw)
        gen = torch.Generator().manual_seed(3)
        x = torch.rand((n, S, geom.pos_size), generator=gen) * 2 - 1
        v = torch.rand((n, geom.view_pos_size), generator=gen) * 2 - 1
        rgb, dens = ops.gmlp_fwd(geom, {k: t.to(dev) for k, t in sd.items()}, x.to(dev), v.to(dev))
        sd64 = {k: t.double() for k, t in sd.items()}
        r64, d64 = orc.nerf_mlp(sd64, "", x.double(), v.double(), skip_layer=geom.skip_layer)
        torch.testing.assert_close(rgb.cpu().double(), r64, rtol=0, atol=5e-6)
        torch.testing.assert_close(dens.cpu().double(), d64, rtol=0, atol=2e-5)


def test_general_mlp_against_the_reference(dev, golden):
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model import NeRFMLP

    g = golden("g17_general_mlp")
    for code in g["mlp_tags"].tolist():
        tag = chr(code)
        kw = dict(zip(GEOM_KEYS, g[f"mlp_{tag}_geom"].tolist()))
        mlp = NeRFMLP(**kw).to(dev)
        mlp.load_state_dict(syn.make_general_nerf_state_dict(1700 + code, prefixes=("",), **kw))
        with torch.no_grad():
            rgb, dens = mlp(g[f"mlp_{tag}_x"].to(dev), g[f"mlp_{tag}_v"].to(dev))
        torch.testing.assert_close(rgb.cpu(), g[f"mlp_{tag}_rgb"], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(dens.cpu(), g[f"mlp_{tag}_density"], rtol=1e-5, atol=2e-5)


def test_general_engine_equals_the_fused_kernels_on_the_default_geometry(dev, golden):
    """Both engines exist for NeRFMLP(0, 10, 4): the layer-wise one must agree with the fused kernel to summation-order level
    (G4: the reference's raw outputs) -- and, whole path, with the fused path's render of a smooth field."""
    import aon_amd.synthetic as syn
    from aon_amd import ops

    g = golden("g4_mlp")
    sd = syn.make_nerf_state_dict(seed=0, density_scale=30.0)
    geom = ops.MlpGeometry()
    for lvl in ("coarse", "fine"):
        params = {k[len(lvl) + 5:]: v.to(dev) for k, v in sd.items() if k.startswith(lvl)}
        rgb, dens = ops.gmlp_fwd(geom, params, g["samples_enc"].to(dev), g["viewdirs_enc"].to(dev))
        torch.testing.assert_close(rgb.cpu(), g[f"raw_rgb_{lvl}"], rtol=0, atol=2e-5)
        torch.testing.assert_close(dens.cpu(), g[f"raw_sigma_{lvl}"], rtol=0, atol=2e-4)
    g15 = golden("g15_smooth")
    sds = syn.make_smooth_nerf_state_dict()
    rays = {k: g15[k][:256].to(dev) for k in ("rays_o", "rays_d", "viewdirs")}
    pc = {k[11:]: v.to(dev) for k, v in sds.items() if k.startswith("coarse_mlp.")}
    pf = {k[9:]: v.to(dev) for k, v in sds.items() if k.startswith("fine_mlp.")}
    outs = ops.grender_fwd(geom, pc, pf, rays["rays_o"], rays["rays_d"], rays["viewdirs"], 2.0, 6.0, True)
    for lvl, name in ((0, "coarse"), (1, "fine")):
        torch.testing.assert_close(outs[lvl][0].cpu(), g15[f"van_det_{name}_rgb"][:256], rtol=0, atol=2e-6)
        torch.testing.assert_close(outs[lvl][2].cpu(), g15[f"van_det_{name}_depth"][:256], rtol=0, atol=1e-5)


def test_nerf_with_other_degrees_end_to_end(dev, golden):
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model import NeRF

    g = golden("g17_general_mlp")
    rays = {k: g["nerf_" + k].to(dev) for k in ("rays_o", "rays_d", "viewdirs")}
    n = rays["rays_o"].shape[0]
    for tag, (drgb, ddepth) in {"p": (2e-6, 2e-5), "q": (2e-5, 5e-4)}.items():   # q: 2^11 frequencies, a far less smooth field
        mn, mx, dv, nc, nf, lind = g[f"nerf_{tag}_cfg"].tolist()
        model = NeRF(min_deg_point=mn, max_deg_point=mx, deg_view=dv, num_coarse_samples=nc, num_fine_samples=nf, lindisp=bool(lind)).to(dev)
        model.load_state_dict(syn.make_general_nerf_state_dict(1750 + ord(tag), min_deg_point=mn, max_deg_point=mx, deg_view=dv))
        ok = g[f"nerf_{tag}_margin"] > 0.05     # far-plane raw sigma robustly signed (helper.py:163), as on the sharp fixtures
        assert ok.float().mean() > 0.7
        tr, u = syn.seeded_uniform(1760 + ord(tag), n, nc + 1).to(dev), syn.seeded_uniform(1770 + ord(tag), n, nf).to(dev)
        with torch.no_grad():
            outs = {"det": model(rays, False, True, 2.0, 6.0), "rnd": model(rays, True, False, 2.0, 6.0, t_rand=tr, u=u)}
        for t2, out in outs.items():
            for lvl, name in ((0, "coarse"), (1, "fine")):
                rgb, acc, depth = (x.cpu() for x in out[lvl])
                torch.testing.assert_close(rgb[ok], g[f"nerf_{tag}_{t2}_{name}_rgb"][ok], rtol=0, atol=drgb)
                torch.testing.assert_close(acc[ok], g[f"nerf_{tag}_{t2}_{name}_acc"][ok], rtol=0, atol=drgb)
                torch.testing.assert_close(depth[ok], g[f"nerf_{tag}_{t2}_{name}_depth"][ok], rtol=0, atol=ddepth)


def test_other_degrees_on_the_fused_inference_kernels(dev, golden):
    """NeRF(min_deg_point, max_deg_point, deg_view) with at most 10 / 4 levels renders on the FUSED kernels (zero-weight slots for
    the missing levels, encodings in the padded layout): against the reference (G17 'p' = (0, 6, 2)), against the oracle at
    (1, 8, 3), (-1, 9, 4) and the degenerate (2, 2, 0), and against the layer-wise engine on the same weights."""
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model import NeRF

    g = golden("g17_general_mlp")
    rays_cpu = {k: g["nerf_" + k] for k in ("rays_o", "rays_d", "viewdirs")}
    rays = {k: v.to(dev) for k, v in rays_cpu.items()}
    n = rays["rays_o"].shape[0]
    for (mn, mx, dv), seed in (((0, 6, 2), 1750 + ord("p")), ((1, 8, 3), 61), ((-1, 9, 4), 62), ((2, 2, 0), 63)):
        kw = dict(min_deg_point=mn, max_deg_point=mx, deg_view=dv)
        sd = syn.make_general_nerf_state_dict(seed, **kw)
        model = NeRF(**kw).to(dev)
        model.load_state_dict(sd)
        assert model._fused_inference and not model._general and not model.coarse_mlp.geometry.is_default
        tr, u = syn.seeded_uniform(64, n, 65).to(dev), syn.seeded_uniform(65, n, 128).to(dev)
        with torch.no_grad():
            fused = [model(rays, False, True, 2.0, 6.0), model(rays, True, False, 2.0, 6.0, t_rand=tr, u=u)]
            model._fused_inference = False
            layered = [model(rays, False, True, 2.0, 6.0), model(rays, True, False, 2.0, 6.0, t_rand=tr, u=u)]
        ref, aux = orc.nerf_forward(sd, rays_cpu, False, True, 2.0, 6.0, return_aux=True, **kw)
        ok = torch.stack([a["raw_sigma"][:, -1, 0].abs() for a in aux]).min(0).values > 0.05
        assert ok.float().mean() > 0.6
        for lvl in (0, 1):
            torch.testing.assert_close(fused[0][lvl][0].cpu()[ok], ref[lvl][0][ok], rtol=0, atol=2e-6)
            torch.testing.assert_close(fused[0][lvl][2].cpu()[ok], ref[lvl][2][ok], rtol=0, atol=2e-5)
            for a, b in zip(fused, layered):      # two engines, same weights
                torch.testing.assert_close(a[lvl][0][ok.to(dev)], b[lvl][0][ok.to(dev)], rtol=0, atol=2e-6)
        if (mn, mx, dv) == (0, 6, 2):
            okg = g["nerf_p_margin"] > 0.05
            for lvl, name in ((0, "coarse"), (1, "fine")):
                torch.testing.assert_close(fused[0][lvl][0].cpu()[okg], g[f"nerf_p_det_{name}_rgb"][okg], rtol=0, atol=2e-6)
    # more than 10 levels do not fit the slots: layer-wise engine
    big = NeRF(min_deg_point=0, max_deg_point=11, deg_view=4)
    assert big._general and not big._fused_inference


@pytest.mark.parametrize("degrees", [(0, 6, 2), (1, 8, 3), (-2, 8, 0)])
def test_training_at_other_degrees_on_the_fused_kernels(dev, degrees):
    """loss.backward() of NeRF(min_deg_point, max_deg_point, deg_view) with at most 10 / 4 levels on the FUSED training kernels
    (caller-encoded forward, weight gradients re-cut to the network's own column order): every parameter gradient as close to the
    oracle's fp64 autograd as the oracle's fp32 is, and equal (to summation-order level) to the layer-wise engine's."""
    import aon_amd.synthetic as syn
    from _gradcheck import assert_as_close_as_fp32, rel_l2
    from aon_amd.models.vanilla_nerf.model import NeRF

    mn, mx, dv = degrees
    gk = dict(min_deg_point=mn, max_deg_point=mx, deg_view=dv)
    n = 200
    sd = syn.make_general_nerf_state_dict(71, **gk)
    model = NeRF(**gk).to(dev)
    model.load_state_dict(sd)
    frame = syn.make_rays(24, 32, syn.look_at_pose(4.0, 60, 20), syn.focal_from_fovy(24))
    rays_cpu = {k: v[::3][:n].contiguous() for k, v in frame.items()}
    rays = {k: v.to(dev) for k, v in rays_cpu.items()}
    target = syn.seeded_uniform(72, n, 3)
    tr, u = syn.seeded_uniform(73, n, 65), syn.seeded_uniform(74, n, 128)

    def step():
        model.zero_grad()
        out = model(rays, True, True, 2.0, 6.0, t_rand=tr.to(dev), u=u.to(dev))
        loss = ((out[0][0] - target.to(dev)) ** 2).mean() + ((out[1][0] - target.to(dev)) ** 2).mean()
        loss.backward()
        return loss.item(), {k: p.grad.clone().cpu() for k, p in model.named_parameters()}

    loss, grads = step()
    for name, p in model.named_parameters():
        assert grads[name].shape == p.shape
    model._fused_training = False
    loss_l, grads_l = step()
    assert abs(loss - loss_l) < 2e-6

    def oracle_grads(dtype):
        sd_o = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in sd.items()}
        out = orc.nerf_forward(sd_o, {k: v.to(dtype) for k, v in rays_cpu.items()}, True, True, 2.0, 6.0, t_rand=tr.to(dtype), u=u.to(dtype), **gk)
        l = orc.img2mse(out[0][0], target.to(dtype)) + orc.img2mse(out[1][0], target.to(dtype))
        l.backward()
        return l.item(), {k: v.grad for k, v in sd_o.items()}

    (_, truth), (loss32, ref32) = oracle_grads(torch.float64), oracle_grads(torch.float32)
    assert abs(loss - loss32) < 2e-6
    assert_as_close_as_fp32(grads, truth, ref32, f"fused training at degrees {degrees}")
    assert_as_close_as_fp32(grads_l, truth, ref32, f"layer-wise training at degrees {degrees}")


def test_general_engine_chunking_is_invisible(dev):
    """A workspace smaller than the batch makes aon_grender_fwd walk ray chunks: same bits as one pass."""
    import aon_amd.synthetic as syn
    from aon_amd import ops
    from aon_amd.models.vanilla_nerf.model import NeRF

    kw = dict(min_deg_point=0, max_deg_point=5, deg_view=2)
    model = NeRF(**kw).to(dev)
    model._fused_inference = False        # this test is about the layer-wise engine's chunk walk
    model.load_state_dict(syn.make_general_nerf_state_dict(21, **kw))
    frame = syn.make_rays(20, 24, syn.look_at_pose(4.0, 60, 20), syn.focal_from_fovy(20))
    rays = {k: v.to(dev) for k, v in frame.items()}
    with torch.no_grad():
        one = model(rays, False, True, 2.0, 6.0)
        old = ops.G_CHUNK_RAYS
        try:
            ops.G_CHUNK_RAYS = 100
            ops._GWS_CACHE.clear()
            many = model(rays, False, True, 2.0, 6.0)
        finally:
            ops.G_CHUNK_RAYS = old
            ops._GWS_CACHE.clear()
    for lvl in (0, 1):
        for a, b in zip(one[lvl], many[lvl]):
            assert torch.equal(a, b)


def test_training_step_on_the_general_engine(dev):
    """loss.backward() through NeRF(min_deg_point, max_deg_point, deg_view, ...) on the layer-wise engine: every parameter gradient
    as close to the oracle's fp64 autograd as the oracle's fp32 autograd is (tests/_gradcheck.py); a second identical step gives
    identical gradients (no atomics)."""
    import aon_amd.synthetic as syn
    from _gradcheck import assert_as_close_as_fp32
    from aon_amd.models.vanilla_nerf.model import NeRF

    gk = dict(min_deg_point=0, max_deg_point=6, deg_view=2)
    n, nc, nf = 200, 40, 56
    kw = dict(num_coarse_samples=nc, num_fine_samples=nf, noise_std=0.2, **gk)
    sd = syn.make_general_nerf_state_dict(31, **gk)
    model = NeRF(**kw).to(dev)
    model._fused_inference = False        # (0, 6, 2) fits the fused kernels' slots: this test is about the layer-wise engine
    model.load_state_dict(sd)
    frame = syn.make_rays(24, 32, syn.look_at_pose(4.0, 60, 20), syn.focal_from_fovy(24))
    rays_cpu = {k: v[::3][:n].contiguous() for k, v in frame.items()}
    rays = {k: v.to(dev) for k, v in rays_cpu.items()}
    target = syn.seeded_uniform(32, n, 3)
    tr, u = syn.seeded_uniform(33, n, nc + 1), syn.seeded_uniform(34, n, nf)
    nz = [syn.seeded_uniform(35, n, nc + 1), syn.seeded_uniform(36, n, nc + 1 + nf)]

    def step():
        model.zero_grad()
        out = model(rays, True, True, 2.0, 6.0, t_rand=tr.to(dev), u=u.to(dev), noise=[z.to(dev) for z in nz])
        loss = ((out[0][0] - target.to(dev)) ** 2).mean() + ((out[1][0] - target.to(dev)) ** 2).mean()
        loss.backward()
        return loss.item(), {k: p.grad.clone() for k, p in model.named_parameters()}

    loss, grads = step()
    loss2, grads2 = step()
    assert loss == loss2 and all(torch.equal(grads[k], grads2[k]) for k in grads)

    def oracle_grads(dtype):
        sd_o = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in sd.items()}
        out = orc.nerf_forward(sd_o, {k: v.to(dtype) for k, v in rays_cpu.items()}, True, True, 2.0, 6.0, t_rand=tr.to(dtype), u=u.to(dtype),
                               noise=[z.to(dtype) for z in nz], **kw)
        l = orc.img2mse(out[0][0], target.to(dtype)) + orc.img2mse(out[1][0], target.to(dtype))
        l.backward()
        return l.item(), {k: v.grad for k, v in sd_o.items()}

    (_, truth), (loss32, ref32) = oracle_grads(torch.float64), oracle_grads(torch.float32)
    assert abs(loss - loss32) < 2e-6
    assert_as_close_as_fp32({k: v.cpu() for k, v in grads.items()}, truth, ref32, "general engine")


def test_general_mlp_gradients_deep_geometry(dev):
    """The backward of a geometry with a skip concatenation, a three-layer view branch and widths off the tile grid, through the
    whole path with one level: every parameter gradient against the oracle's fp64 autograd."""
    import aon_amd.synthetic as syn
    from _gradcheck import assert_as_close_as_fp32
    from aon_amd import ops
    from aon_amd.autograd import RenderGeneral

    kw = dict(min_deg_point=0, max_deg_point=4, deg_view=2, netdepth=6, netwidth=136, netdepth_condition=3, netwidth_condition=72, skip_layer=3)
    geom = ops.MlpGeometry(**kw)
    sd = syn.make_general_nerf_state_dict(41, prefixes=("coarse_mlp",), **kw)
    n, nc = 150, 30
    frame = syn.make_rays(24, 32, syn.look_at_pose(4.0, 60, 20), syn.focal_from_fovy(24))
    rays_cpu = {k: v[::5][:n].contiguous() for k, v in frame.items()}
    target = syn.seeded_uniform(42, n, 3)
    params = [sd[f"coarse_mlp.{nm}"].to(dev).requires_grad_(True) for nm in geom.param_order]
    flat = RenderGeneral.apply(rays_cpu["rays_o"].to(dev), rays_cpu["rays_d"].to(dev), rays_cpu["viewdirs"].to(dev), 2.0, 6.0, True, 1, None, None,
                               geom, ops.RenderOpts(num_coarse_samples=nc), None, *params)
    loss = ((flat[0] - target.to(dev)) ** 2).mean()
    loss.backward()

    def oracle_grads(dtype):
        sd_o = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in sd.items()}
        out = orc.nerf_forward(sd_o, {k: v.to(dtype) for k, v in rays_cpu.items()}, False, True, 2.0, 6.0, num_levels=1, num_coarse_samples=nc,
                               skip_layer=3, min_deg_point=0, max_deg_point=4, deg_view=2)
        l = orc.img2mse(out[0][0], target.to(dtype))
        l.backward()
        return l.item(), {k: v.grad for k, v in sd_o.items()}

    (_, truth), (loss32, ref32) = oracle_grads(torch.float64), oracle_grads(torch.float32)
    assert abs(loss.item() - loss32) < 2e-6
    assert_as_close_as_fp32({f"coarse_mlp.{nm}": p.grad.cpu() for nm, p in zip(geom.param_order, params)}, truth, ref32, "deep geometry")


def test_invalid_geometry_is_rejected():
    from aon_amd._lib import AonError
    from aon_amd.models.vanilla_nerf.model import NeRFMLP

    with pytest.raises(AonError, match="last trunk layer"):
        NeRFMLP(0, 10, 4, netdepth=5, skip_layer=4)     # the reference's forward raises on this one too (257 != 256 + 63 inputs)
    NeRFMLP(0, 10, 4, netdepth=6, skip_layer=4)


def test_edge_batches_on_every_route(dev):
    """num_levels = 1, one ray and an empty batch through the three routes (fused, fused with zero-weight slots, layer-wise engine),
    inference and (n >= 1) a training step: shapes, finiteness, the oracle's values."""
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model import NeRF

    frame = syn.make_rays(8, 8, syn.look_at_pose(4.0, 60, 20), syn.focal_from_fovy(8))
    for gk in (dict(), dict(min_deg_point=0, max_deg_point=6, deg_view=2), dict(min_deg_point=0, max_deg_point=12, deg_view=5)):
        sd = syn.make_general_nerf_state_dict(81, **gk)
        for levels in (1, 2):
            model = NeRF(num_levels=levels, num_coarse_samples=20, num_fine_samples=30, **gk).to(dev)
            model.load_state_dict(sd)
            for n in (0, 1, 5):
                rays_cpu = {k: v[:n].contiguous() for k, v in frame.items()}
                rays = {k: v.to(dev) for k, v in rays_cpu.items()}
                with torch.no_grad():
                    out = model(rays, False, True, 2.0, 6.0)
                assert len(out) == levels and all(o[0].shape == (n, 3) and o[1].shape == (n,) and o[2].shape == (n,) for o in out)
                if n == 0:
                    continue
                ref = orc.nerf_forward(sd, rays_cpu, False, True, 2.0, 6.0, num_levels=levels, num_coarse_samples=20, num_fine_samples=30, **gk)
                for lvl in range(levels):
                    torch.testing.assert_close(out[lvl][0].cpu(), ref[lvl][0], rtol=0, atol=5e-5)
                model.zero_grad()
                o2 = model(rays, False, True, 2.0, 6.0)
                sum(o[0].sum() for o in o2).backward()
                used = [p for name, p in model.named_parameters() if levels == 2 or name.startswith("coarse")]
                assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in used)
