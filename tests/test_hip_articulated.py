"""GPU parity of the articulated path (model_autodecoder.py NeRFMLP / NeRF_AE_Art,
code_library.py) against the oracle and the golden vectors produced by the imported reference.

Tolerances: MLP raw outputs 5e-5 (rgb) against oracle/reference (fp32, different summation order; the latent columns
are folded into bias vectors on the device).  End to end, on every ray (softplus keeps sigma > 0, so the far-plane
alpha is always 1 and the vanilla path's sign discontinuity does not exist here): PSNR >= 70 dB, every value within
1e-3 and >= 99 % within 2e-4.  The articulated path is intrinsically more sensitive than the vanilla one: a 1e-6
difference in the deformed point is multiplied by 2^9 inside the positional encoding that follows the deformation MLP
(the reference's own fp32-vs-fp64 spread on such a field is 2e-3)."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nerf_oracle as orc  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def assert_render_close(a, b, what=""):
    err = (a - b).abs()
    mse = torch.mean((a - b) ** 2).item()
    psnr = -10.0 * torch.log10(torch.tensor(max(mse, 1e-20))).item()
    frac = (err <= 2e-4).double().mean().item()
    assert psnr >= 70.0 and err.max().item() <= 1e-3 and frac >= 0.99, f"{what}: psnr {psnr:.1f} dB, max {err.max().item():.2e}, frac<=2e-4 {frac:.4f}"


@pytest.fixture(scope="module")
def art_sd():
    import aon_amd.synthetic as syn

    return syn.make_art_state_dict(seed=0, density_scale=30.0)


def _lat(g, tag, dev=None):
    d = {"density": g[f"lat_{tag}_density"], "color": g[f"lat_{tag}_color"], "articulation": g[f"lat_{tag}_articulation"]}
    return d if dev is None else {k: v.to(dev) for k, v in d.items()}


def test_code_library(dev, golden):
    import aon_amd.synthetic as syn
    from aon_amd.models.code_library import CodeLibraryArticulated

    g = golden("g11_nerf_ae_art")
    lib = CodeLibraryArticulated(types.SimpleNamespace(N_max_objs=2, N_obj_code_length=128)).to(dev)
    lib.load_state_dict(syn.make_code_library_state(seed=0, n_max_objs=2))
    train = lib({"instance_id": torch.tensor([1], device=dev), "articulation_id": torch.tensor([3], device=dev)})
    test = lib({"instance_id": torch.tensor([0], device=dev), "articulation_id": torch.tensor([7], device=dev)}, is_test=True)
    for k in ("density", "color", "articulation"):
        assert torch.equal(train[k].cpu(), _lat(g, "train")[k])
        assert torch.equal(test[k].cpu(), _lat(g, "test")[k])
    assert lib.get_interpolated_articulations(device=dev).shape == (19, 32)


def test_art_mlp_vs_reference_and_oracle(dev, golden, art_sd):
    import aon_amd.synthetic as syn
    from aon_amd import ops
    from aon_amd.models.vanilla_nerf.model_autodecoder import NeRF_AE_Art

    g = golden("g11_nerf_ae_art")
    model = NeRF_AE_Art().to(dev)
    model.load_state_dict(art_sd)
    lat = _lat(g, "train", dev)
    with torch.no_grad():
        rgb, sig = model.fine_mlp(g["mlp_pos"].to(dev), g["mlp_viewdirs_enc"].to(dev), lat)
    torch.testing.assert_close(rgb.cpu(), g["mlp_raw_rgb"], rtol=5e-5, atol=5e-5)
    torch.testing.assert_close(sig.cpu(), g["mlp_raw_sigma"], rtol=5e-5, atol=2e-3)  # density head x30
    # fused cast+deform+encode entry point vs the oracle on fresh rays, ragged sizes, both MLPs, test-time latents
    lat_t_cpu = _lat(g, "test")
    for n, S, seed, lvl in ((1, 65, 1, "coarse"), (33, 193, 2, "fine"), (130, 2, 3, "fine")):
        rays = syn.random_rays(n, seed=seed)
        t = torch.sort(torch.rand(n, S, generator=torch.Generator().manual_seed(seed)) * 4 + 2, dim=-1).values
        pos = orc.cast_rays(t, rays["rays_o"], rays["rays_d"])
        venc = orc.pos_enc(rays["viewdirs"], 0, 4)
        rgb_o, sig_o = orc.art_mlp(art_sd, f"{lvl}_mlp.", pos, venc, lat_t_cpu)
        mlp = getattr(model, f"{lvl}_mlp")
        small = mlp.prepared({k: v.to(dev) for k, v in lat_t_cpu.items()})
        raw = ops.art_mlp_fwd(mlp.packed(), small, rays["rays_o"].to(dev), rays["rays_d"].to(dev), rays["viewdirs"].to(dev), t.to(dev)).cpu()
        torch.testing.assert_close(raw[..., :3], rgb_o, rtol=5e-5, atol=5e-5)
        torch.testing.assert_close(raw[..., 3:], sig_o, rtol=5e-5, atol=2e-3)
        raw_p = ops.art_mlp_fwd_pos(mlp.packed(), small, pos.to(dev), venc.to(dev)).cpu()
        torch.testing.assert_close(raw_p, raw, rtol=5e-5, atol=2e-3)


def test_nerf_ae_art_forward(dev, golden, art_sd):
    from aon_amd.models.vanilla_nerf.model_autodecoder import NeRF_AE_Art

    g = golden("g11_nerf_ae_art")
    model = NeRF_AE_Art().to(dev)
    model.load_state_dict(art_sd)
    rays_cpu = {k: g[k] for k in ("rays_o", "rays_d", "viewdirs")}
    rays = {k: v.to(dev) for k, v in rays_cpu.items()}
    cases = (("det", "train", dict(randomized=False, white_bkgd=True), {}),
             ("tst_nowb", "test", dict(randomized=False, white_bkgd=False), {}),
             ("rnd", "train", dict(randomized=True, white_bkgd=True), dict(t_rand=g["t_rand"], u=g["u"])))
    for tag, lat_tag, kw, draws in cases:
        with torch.no_grad():
            out = model(rays, kw["randomized"], kw["white_bkgd"], g["near"], g["far"], _lat(g, lat_tag, dev),
                        **{k: v.to(dev) for k, v in draws.items()})
        ref = orc.nerf_ae_art_forward(art_sd, rays_cpu, near=g["near"], far=g["far"], latents=_lat(g, lat_tag), **kw, **draws)
        for lvl, name in ((0, "coarse"), (1, "fine")):
            rgb, acc, depth = (x.cpu() for x in out[lvl])
            assert_render_close(rgb, ref[lvl][0], f"{tag}/{name} rgb vs oracle")
            torch.testing.assert_close(acc, ref[lvl][1], rtol=0, atol=2e-4)
            # depth = sum w*t.  Coarse level: tight.  Fine level: ill-conditioned on this sharp (density x30) random
            # field -- a 1-ulp change of a coarse weight moves inverse-CDF draws across thin high-density shells; the
            # ORACLE ITSELF differs between fp32 and fp64 by 0.06 (deterministic) / 0.28 (randomized) in fine depth on
            # these very rays while its rgb agrees to 2e-4.  So: most rays tight, worst ray inside that spread.
            # Measured round 2 against the reference's outputs (tests/diag/diag_tolerances.py): coarse depth 1.8e-5; fine depth
            # 1.3e-2 deterministic, 0.17 randomized (p99 3e-2); on the smooth field of test_hip_smooth.py the same kernels hold 2e-5.
            derr = (depth - ref[lvl][2]).abs()
            if lvl == 0:
                assert derr.max().item() <= 1e-4
                torch.testing.assert_close(rgb, ref[lvl][0], rtol=0, atol=2e-6)
            else:
                assert (derr <= 5e-3).double().mean().item() >= 0.9 and derr.max().item() <= (0.3 if kw["randomized"] else 5e-2)
            assert_render_close(rgb, g[f"{tag}_{name}_rgb"], f"{tag}/{name} rgb vs reference")   # the reference's own output
            torch.testing.assert_close(acc, g[f"{tag}_{name}_acc"], rtol=0, atol=2e-4)
    # latents matter (different articulation code -> different image) and grad mode is refused loudly
    with torch.no_grad():
        a = model(rays, False, True, 2.0, 6.0, _lat(g, "train", dev))[1][0]
        b = model(rays, False, True, 2.0, 6.0, _lat(g, "test", dev))[1][0]
    assert (a - b).abs().max().item() > 1e-3
    # grad mode runs the HIP training path (tests/test_hip_training_art.py)
    out = model(rays, False, True, 2.0, 6.0, _lat(g, "train", dev))
    assert out[1][0].requires_grad


def test_articulated_frame_320x240_properties(dev, art_sd):
    """BASELINE config 4 size (320x240 articulated, 1 GPU): chunk invariance, determinism, ranges."""
    import aon_amd.synthetic as syn
    from aon_amd import ops
    from aon_amd.models.vanilla_nerf.model_autodecoder import NeRF_AE_Art

    H, W = 240, 320
    model = NeRF_AE_Art().to(dev)
    model.load_state_dict(art_sd)
    lib = syn.make_code_library_state(seed=0, n_max_objs=1)
    lat = {k: v.to(dev) for k, v in orc.code_library(lib, torch.tensor([0]), torch.tensor([4])).items()}
    ro, vd = ops.raygen(syn.look_at_pose(), H, W, syn.focal_from_fovy(H), device=dev)
    rays = {"rays_o": ro, "rays_d": vd, "viewdirs": vd}
    with torch.no_grad():
        full = model(rays, False, True, 2.0, 6.0, lat)
        again = model(rays, False, True, 2.0, 6.0, lat)
        sl = slice(30_000, 33_840)
        part = model({k: v[sl] for k, v in rays.items()}, False, True, 2.0, 6.0, lat)
    for lvl in (0, 1):
        rgb, acc, depth = full[lvl]
        assert torch.equal(rgb, again[lvl][0])
        assert torch.equal(part[lvl][0], rgb[sl]) and torch.equal(part[lvl][2], depth[sl])
        assert torch.isfinite(rgb).all() and torch.isfinite(depth).all()
        assert acc.min().item() >= 1.0 - 1e-5 and acc.max().item() <= 1.0 + 1e-5   # softplus sigma > 0 -> alpha_last = 1
        assert rgb.min().item() >= -0.001 - 1e-5 and rgb.max().item() <= 1.001 + 1e-4
    pick = torch.arange(0, H * W, 601)
    rays_cpu = {k: v[pick.to(dev)].cpu() for k, v in rays.items()}
    ref = orc.nerf_ae_art_forward(art_sd, rays_cpu, False, True, 2.0, 6.0, {k: v.cpu() for k, v in lat.items()})
    assert_render_close(full[1][0][pick.to(dev)].cpu(), ref[1][0], "320x240 strided sample vs oracle")
