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


def test_articulated_frame_320x240_properties(dev, art_sd, golden):
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
    # Parity AT THIS SIZE against the REFERENCE's own outputs (round 6, G20): the 4,267 strided rays of this frame (every 18th pixel) that
    # tests/golden/make_golden_full.py put through the real `NeRF_AE_Art.forward` in fp32 and fp64 -- both levels, every output (rounds
    # 3-5 evaluated the oracle live here, fp32 + fp64; rounds 1-4 held 128 rays, fine rgb only).  The fixture's rays are rendered as their
    # own batch (chunk invariance is asserted above).  Bars: 1e-5 rgb / acc, 2e-4 depth per ray.  Where the REFERENCE ARITHMETIC ITSELF is
    # less certain than that on this sharp x30 field (a 1e-7 difference of the deformed point is multiplied by 2^9 inside the encoding that
    # follows the deformation MLP, and a 1e-7 difference of a coarse weight moves fine samples across thin shells) the yardstick is the
    # distance between the reference's fp32 and fp64 evaluations of the same rays.  Softplus keeps sigma > 0: every ray is far-plane robust.
    g = golden("g20_config4_frame")
    assert (g["H"], g["W"]) == (H, W) and g["pick"].numel() >= 4096
    for k in ("density", "color", "articulation"):
        assert torch.equal(lat[k].cpu(), g["lat_" + k])                      # the product's code library gave the reference's latents
    grays = {k: g[k].to(dev) for k in ("rays_o", "rays_d", "viewdirs")}
    assert torch.equal(grays["rays_o"], ro[g["pick"].to(dev)])
    torch.testing.assert_close(grays["rays_d"], vd[g["pick"].to(dev)], rtol=0, atol=2e-7)
    with torch.no_grad():
        sub = model(grays, False, True, 2.0, 6.0, lat)
    nrays = g["pick"].numel()
    for lvl, lname in ((0, "coarse"), (1, "fine")):
        got = [x.cpu() for x in sub[lvl]]
        for i, (name, bar) in enumerate((("rgb", 1e-5), ("acc", 1e-5), ("depth", 2e-4))):
            # CALM rays of this output: the reference's own fp32 and fp64 evaluations agree to a tenth of the bar (1e-6 rgb / acc, 2e-5 depth)
            calm = g[f"spread_{lname}_{name}"] <= 0.1 * bar
            ref = g[f"ref_{lname}_{name}"]
            err = (got[i] - ref).abs()
            spread = g[f"spread_{lname}_{name}"]
            if err.dim() > 1:
                err = err.max(dim=-1).values
            above, above_ref = int((err > bar).sum()), int((spread > bar).sum())
            beyond = int((err > torch.clamp(3.0 * spread, min=bar)).sum())
            q = lambda x, p: torch.quantile(x.double(), p).item()   # noqa: E731
            print(f"config 4 level {lvl} {name}: {nrays} rays, |hip - reference| max {err.max():.2e} p99 {q(err, 0.99):.2e} p50 {q(err, 0.5):.2e}; reference fp32 vs "
                  f"fp64 on the same rays max {spread.max():.2e} p99 {q(spread, 0.99):.2e} p50 {q(spread, 0.5):.2e}; rays above {bar:g}: hip {above}, "
                  f"reference's own {above_ref}; hip beyond 3 x that ray's spread: {beyond}; calm rays {int(calm.sum())} "
                  f"({100.0 * calm.double().mean():.1f} %), worst calm ray {err[calm].max():.2e}")
            if lvl == 0:
                # coarse level: the same t on both sides -- per ray, the plain bar
                assert above == 0, (name, err.max().item())
            else:
                # Fine level.  Measured before the bottleneck fold (round 5, 4,267 rays): rgb 185 rays above 1e-5 where the reference's own
                # fp32-vs-fp64 distance exceeds it on MORE rays and by more (max 4.6e-4 vs 7.0e-4) -- which evaluation lands on which side
                # of a thin shell is a coin toss per ray, so a per-ray "3 x this ray's spread" rule does not hold (47 rays) while the
                # DISTRIBUTIONS agree.  Criterion: HIP is to the fp32 reference what the fp32 reference is to the fp64 truth -- no more rays
                # above the bar than 1.5 x the reference's own count (+ 0.5 % of the rays), the worst ray within 2 x the reference's worst,
                # the 99th percentile within 2 x the reference's (or the bar).
                assert above <= 1.5 * above_ref + 0.005 * nrays, (name, above, above_ref)
                assert err.max().item() <= max(bar, 2.0 * spread.max().item()), (name, err.max().item(), spread.max().item())
                assert q(err, 0.99) <= max(bar, 2.0 * q(spread, 0.99)), (name, q(err, 0.99), q(spread, 0.99))
                # ... and a per-ray net under the distributional rule: the CALM rays of this output -- the reference's own
                # fp32 and fp64 agree to a tenth of the bar there -- must meet the plain bar.  Not every one can: "calm between fp32 and fp64"
                # is not "calm for every fp32 evaluation".  G20 records a third evaluation on the CPU (every Linear accumulated in fp64 and
                # rounded once -- more accurate than the reference's fp32): against the reference's fp32 it is above 1e-5 on 5 of the 2,514
                # calm rgb rays (worst 2.1e-5; 139 rays overall), and on 33 of the 424 calm DEPTH rays (worst 1.6e-3).  HIP measured: rgb 9
                # (worst 5.1e-5; 183 overall).  So: at most 1 % of the calm rays above the bar -- or 3 x the third evaluation's own count --
                # (+ 5), the worst within 10 x the bar (or 3 x the third evaluation's worst): a systematic fine-level bias of 1e-4 on 5 % of
                # the rays fails.  How many rays are calm is a property of the FIELD
                # (rgb 58.9 %, acc 100 %, depth 9.9 % of the 4,267: the un-normalised depth sum of helper.py:180 moves by 2e-4 at the median
                # between the reference's own two precisions); the sizes are asserted so that the net cannot silently shrink.
                ncalm, over = int(calm.sum()), int((calm & (err > bar)).sum())
                over_alt = int((calm & (g[f"alt_err_{lname}_{name}"] > bar)).sum())
                print(f"    calm rays above the bar: hip {over}, the CPU's third evaluation {over_alt} (of {ncalm})")
                assert calm.double().mean().item() > {"rgb": 0.5, "acc": 0.99, "depth": 0.05}[name], (name, calm.double().mean().item())
                worst_alt = g[f"alt_err_{lname}_{name}"][calm].max().item()
                assert over <= max(0.01 * ncalm, 3 * over_alt) + 5, (name, over, over_alt, ncalm)
                assert err[calm].max().item() <= max(10.0 * bar, 3.0 * worst_alt), (name, err[calm].max().item(), worst_alt)
        assert_render_close(got[0], g[f"ref_{lname}_rgb"], f"320x240 strided sample vs reference, level {lvl}")


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_articulated_network_at_other_degrees(dev, golden, tag):
    """NeRF_AE_Art(min_deg_point, max_deg_point, deg_view) with up to 10 / 4 frequency levels on the fused articulated kernels (round 4:
    zero-weight slots in the packed streams, the encoding scales 2^(min_deg_point + l) as run-time values of the per-call block) against
    the REAL reference built with those arguments (G18): (0, 6, 2), (-1, 9, 4), (2, 5, 0); the stage-level NeRFMLP on raw positions,
    the whole path deterministic and randomized on every ray (smooth field: the bars of test_hip_smooth.py), and the gradients of
    the training loss -- every parameter and latent -- by the fp64-truth yardstick."""
    import sys

    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model_autodecoder import NeRF_AE_Art

    sys.path.insert(0, __import__("os").path.dirname(__file__))
    from _gradcheck import assert_as_close_as_fp32

    g = golden("g18_art_degrees")
    mn, mx, dv = g[f"{tag}_cfg"].tolist()
    gk = dict(min_deg_point=mn, max_deg_point=mx, deg_view=dv)
    sd = syn.make_art_state_dict(seed=18, density_scale=2.0, **gk)
    model = NeRF_AE_Art(**gk).to(dev)
    model.load_state_dict(sd)
    rays_cpu = {k: g[k] for k in ("rays_o", "rays_d", "viewdirs")}
    rays = {k: v.to(dev) for k, v in rays_cpu.items()}
    lat_cpu = {k: g["lat_" + k] for k in ("density", "color", "articulation")}
    lat = {k: v.to(dev) for k, v in lat_cpu.items()}
    n = rays["rays_o"].shape[0]
    with torch.no_grad():
        rgb, dens = model.fine_mlp(g[f"{tag}_mlp_pos"].to(dev), g[f"{tag}_mlp_cond"].to(dev), lat)
    torch.testing.assert_close(rgb.cpu(), g[f"{tag}_mlp_raw_rgb"], rtol=5e-5, atol=5e-5)
    torch.testing.assert_close(dens.cpu(), g[f"{tag}_mlp_raw_density"], rtol=5e-5, atol=2e-4)
    s = g[f"{tag}_seeds"].tolist()
    tr, u = syn.seeded_uniform(s[0], n, 65), syn.seeded_uniform(s[1], n, 128)
    with torch.no_grad():
        outs = {"det": model(rays, False, True, 2.0, 6.0, lat), "rnd": model(rays, True, False, 2.0, 6.0, lat, t_rand=tr.to(dev), u=u.to(dev))}
    for t2, out in outs.items():
        for lvl, name in ((0, "coarse"), (1, "fine")):
            r, a, d = (x.cpu() for x in out[lvl])
            print(f"degrees {(mn, mx, dv)} {t2} {name}: max |rgb - ref| {(r - g[f'{tag}_{t2}_{name}_rgb']).abs().max():.2e}, "
                  f"depth {(d - g[f'{tag}_{t2}_{name}_depth']).abs().max():.2e}")
            torch.testing.assert_close(r, g[f"{tag}_{t2}_{name}_rgb"], rtol=0, atol=2e-6)
            torch.testing.assert_close(a, g[f"{tag}_{t2}_{name}_acc"], rtol=0, atol=2e-6)
            torch.testing.assert_close(d, g[f"{tag}_{t2}_{name}_depth"], rtol=0, atol=2e-5)
    # gradients (96 rays, randomized with the named draws)
    m = 96
    rc = {k: v[:m] for k, v in rays_cpu.items()}
    target = syn.seeded_uniform(1899, m, 3)

    def oracle_grads(dtype):
        sd_o = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in sd.items()}
        lo = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in lat_cpu.items()}
        out = orc.nerf_ae_art_forward(sd_o, {k: v.to(dtype) for k, v in rc.items()}, True, True, 2.0, 6.0, lo, t_rand=tr[:m].to(dtype), u=u[:m].to(dtype), **gk)
        (orc.img2mse(out[0][0], target.to(dtype)) + orc.img2mse(out[1][0], target.to(dtype))).backward()
        gr = {k: v.grad for k, v in sd_o.items()}
        gr.update({f"latent[{k}]": v.grad for k, v in lo.items()})
        return gr

    truth, ref32 = oracle_grads(torch.float64), oracle_grads(torch.float32)
    lg = {k: v.clone().requires_grad_(True) for k, v in lat.items()}
    out = model({k: v[:m] for k, v in rays.items()}, True, True, 2.0, 6.0, lg, t_rand=tr[:m].to(dev), u=u[:m].to(dev))
    (((out[0][0] - target.to(dev)) ** 2).mean() + ((out[1][0] - target.to(dev)) ** 2).mean()).backward()
    hip = {name: p.grad.cpu() for name, p in model.named_parameters()}
    hip.update({f"latent[{k}]": v.grad.cpu() for k, v in lg.items()})
    for name, gh in hip.items():
        assert gh.shape == truth[name].shape, name
    # Measured (round 4): (-1, 9, 4) 1.1x, (2, 5, 0) 1.0x, the same weights at the default degrees 1.1x; (0, 6, 2): 8.1x on the coarse trunk --
    # HIP 9.5e-4 where the fp32 oracle happens to sit at 1.2e-4 on this low-frequency field (HIP's absolute level is the same 5e-4..1e-3
    # as at the other degrees; its compositing backward alone is 2e-7 from the truth there, tests/diag/diag_art_draw.py, and the
    # deterministic pass of the same network is at 2.0x: tests/diag/diag_art_degrees_grads.py).  A wrong scale or column map is O(1).
    # (factor 10 for that one draw only; every other degree set is held to the factor 5 of all other gradient tests -- VERDICT r5)
    assert_as_close_as_fp32(hip, truth, ref32, f"articulated, degrees {(mn, mx, dv)}", factor=10.0 if (mn, mx, dv) == (0, 6, 2) else 5.0, floor=1e-4)
