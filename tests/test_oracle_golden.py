"""CPU: the oracle (oracle/nerf_oracle.py) against the golden vectors generated from the imported
reference (tests/golden/make_golden.py).  This is what pins the oracle."""
import torch

from oracle import nerf_oracle as orc


def test_g1_raygen(golden):
    g = golden("g1_raygen")
    dirs = orc.get_ray_directions(g["H"], g["W"], g["focal"])
    assert torch.equal(dirs, g["directions"])
    for p in range(g["c2w"].shape[0]):
        ro, vd, rd = orc.get_rays(dirs, g["c2w"][p])
        assert torch.equal(ro, g["rays_o"][p])
        torch.testing.assert_close(vd, g["viewdirs"][p], rtol=0, atol=1e-7)
        torch.testing.assert_close(rd, g["rays_d"][p], rtol=0, atol=1e-7)
        assert torch.equal(orc.ray_radii(dirs, g["c2w"][p]), g["radii"][p])   # 4th output of the reference's call form
    # full 480x640 frame: picked pixels + checksums
    dirs = orc.get_ray_directions(g["full_H"], g["full_W"], g["full_focal"])
    ro, vd, _ = orc.get_rays(dirs, g["c2w"][0])
    torch.testing.assert_close(vd[g["full_pick"]], g["full_viewdirs_pick"], rtol=0, atol=1e-7)
    assert torch.equal(ro[g["full_pick"]], g["full_rays_o_pick"])
    torch.testing.assert_close(vd.double().sum(0), g["full_viewdirs_sum"], rtol=0, atol=1e-3)
    torch.testing.assert_close(vd.double().abs().sum(0), g["full_viewdirs_abs_sum"], rtol=0, atol=1e-3)
    rad = orc.ray_radii(dirs, g["c2w"][0])
    # the (H*W,3)@(3,3) product associates differently with the thread count: each direction within an ulp (as for viewdirs)
    torch.testing.assert_close(rad[g["full_pick"]], g["full_radii_pick"], rtol=0, atol=2e-7)
    torch.testing.assert_close(rad.view(g["full_H"], g["full_W"])[-3:, ::80], g["full_radii_last_rows"], rtol=0, atol=2e-7)
    assert torch.equal(rad.view(g["full_H"], g["full_W"])[-1], rad.view(g["full_H"], g["full_W"])[-3])
    torch.testing.assert_close(rad.double().sum(), torch.as_tensor(g["full_radii_sum"], dtype=torch.float64), rtol=1e-5, atol=0)


def test_g2_sample_along_rays(golden):
    g = golden("g2_sample_along_rays")
    t, c = orc.sample_along_rays(g["rays_o"], g["rays_d"], 64, g["near"], g["far"], False)
    assert torch.equal(t, g["t_det"]) and torch.equal(c, g["coords_det"])
    t, c = orc.sample_along_rays(g["rays_o"], g["rays_d"], 64, g["near"], g["far"], True, g["t_rand"])
    assert torch.equal(t, g["t_rnd"]) and torch.equal(c, g["coords_rnd"])


def test_g3_pos_enc(golden):
    g = golden("g3_pos_enc")
    assert torch.equal(orc.pos_enc(g["x"], 0, 10), g["enc10"])
    assert torch.equal(orc.pos_enc(g["v"], 0, 4), g["enc4"])


def test_g4_mlp(golden, nerf_sd):
    g = golden("g4_mlp")
    for lvl in ("coarse", "fine"):
        rgb, sig = orc.nerf_mlp(nerf_sd, f"{lvl}_mlp.", g["samples_enc"], g["viewdirs_enc"])
        torch.testing.assert_close(rgb, g[f"raw_rgb_{lvl}"], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(sig, g[f"raw_sigma_{lvl}"], rtol=1e-5, atol=1e-4)


def test_g5_volumetric_rendering(golden):
    g = golden("g5_volumetric_rendering")
    for wb in (0, 1):
        cr, acc, w, dep = orc.volumetric_rendering(g["rgb"], g["density"], g["t_vals"], g["dirs"], bool(wb))
        assert torch.equal(cr, g[f"comp_rgb_wb{wb}"])
        assert torch.equal(acc, g[f"acc_wb{wb}"])
        assert torch.equal(w, g[f"weights_wb{wb}"])
        assert torch.equal(dep, g[f"depth_wb{wb}"])
    g = golden("g5b_volumetric_rendering_193")
    cr, acc, w, dep = orc.volumetric_rendering(g["rgb"], g["density"], g["t_vals"], g["dirs"], True)
    assert torch.equal(cr, g["comp_rgb"]) and torch.equal(acc, g["acc"])
    assert torch.equal(w, g["weights"]) and torch.equal(dep, g["depth"])


def test_g6_pdf(golden):
    g = golden("g6_pdf")
    s = orc.sorted_piecewise_constant_pdf(g["bins"], g["weights"], 128, False)
    assert torch.equal(s, g["samples_det"])
    s = orc.sorted_piecewise_constant_pdf(g["bins"], g["weights"], 128, True, g["u"])
    assert torch.equal(s, g["samples_rnd"])


def test_g7_sample_pdf(golden):
    g = golden("g7_sample_pdf")
    mids = 0.5 * (g["t_vals"][..., 1:] + g["t_vals"][..., :-1])
    t, c = orc.sample_pdf(mids, g["weights"], g["rays_o"], g["rays_d"], g["t_vals"], 128, False)
    assert torch.equal(t, g["t_fine_det"]) and torch.equal(c, g["coords_det"])
    t, c = orc.sample_pdf(mids, g["weights"], g["rays_o"], g["rays_d"], g["t_vals"], 128, True, g["u"])
    assert torch.equal(t, g["t_fine_rnd"]) and torch.equal(c, g["coords_rnd"])


def test_g8_nerf_forward(golden, nerf_sd):
    g = golden("g8_nerf_forward")
    rays = {k: g[k] for k in ("rays_o", "rays_d", "viewdirs")}
    for tag, kw in (("det", dict(randomized=False, white_bkgd=True)),
                    ("det_nowb", dict(randomized=False, white_bkgd=False)),
                    ("rnd", dict(randomized=True, white_bkgd=True, t_rand=g["t_rand"], u=g["u"]))):
        out = orc.nerf_forward(nerf_sd, rays, near=g["near"], far=g["far"], **kw)
        for lvl, name in ((0, "coarse"), (1, "fine")):
            # same op sequence on the same CPU kernels: expect (near) bit equality
            torch.testing.assert_close(out[lvl][0], g[f"{tag}_{name}_rgb"], rtol=0, atol=2e-6)
            torch.testing.assert_close(out[lvl][1], g[f"{tag}_{name}_acc"], rtol=0, atol=2e-6)
            torch.testing.assert_close(out[lvl][2], g[f"{tag}_{name}_depth"], rtol=0, atol=2e-5)


def test_g13_metrics(golden):
    g = golden("g13_metrics")
    torch.testing.assert_close(orc.img2mse(g["a"], g["b"]), torch.as_tensor(g["mse"]))
    torch.testing.assert_close(orc.mse2psnr(orc.img2mse(g["a"], g["b"])), torch.as_tensor(g["mse2psnr"]))
    torch.testing.assert_close(orc.psnr_legacy(g["a"], g["b"]), torch.as_tensor(g["psnr_legacy"]))
    torch.testing.assert_close(orc.psnr_each(list(g["a"]), list(g["b"])), g["psnr_each"])


def _latents(g, tag):
    return {"density": g[f"lat_{tag}_density"], "color": g[f"lat_{tag}_color"], "articulation": g[f"lat_{tag}_articulation"]}


def test_g11_articulated(golden):
    import aon_amd.synthetic as syn

    g = golden("g11_nerf_ae_art")
    sd = syn.make_art_state_dict(seed=0, density_scale=30.0)
    lib = syn.make_code_library_state(seed=0, n_max_objs=2)
    # R12 code library: train-time lookup and test-time 19-entry interpolation table
    lat_train = orc.code_library(lib, torch.tensor([1]), torch.tensor([3]))
    lat_test = orc.code_library(lib, torch.tensor([0]), torch.tensor([7]), is_test=True)
    for k in ("density", "color", "articulation"):
        assert torch.equal(lat_train[k], _latents(g, "train")[k])
        assert torch.equal(lat_test[k], _latents(g, "test")[k])
    # R10 articulated NeRFMLP
    rgb, sig = orc.art_mlp(sd, "fine_mlp.", g["mlp_pos"], g["mlp_viewdirs_enc"], lat_train)
    torch.testing.assert_close(rgb, g["mlp_raw_rgb"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(sig, g["mlp_raw_sigma"], rtol=1e-5, atol=1e-4)
    # R11 NeRF_AE_Art.forward
    rays = {k: g[k] for k in ("rays_o", "rays_d", "viewdirs")}
    for tag, lat, kw in (("det", lat_train, dict(randomized=False, white_bkgd=True)),
                         ("tst_nowb", lat_test, dict(randomized=False, white_bkgd=False)),
                         ("rnd", lat_train, dict(randomized=True, white_bkgd=True, t_rand=g["t_rand"], u=g["u"]))):
        out = orc.nerf_ae_art_forward(sd, rays, near=g["near"], far=g["far"], latents=lat, **kw)
        for lvl, name in ((0, "coarse"), (1, "fine")):
            torch.testing.assert_close(out[lvl][0], g[f"{tag}_{name}_rgb"], rtol=0, atol=2e-6)
            torch.testing.assert_close(out[lvl][1], g[f"{tag}_{name}_acc"], rtol=0, atol=2e-6)
            torch.testing.assert_close(out[lvl][2], g[f"{tag}_{name}_depth"], rtol=0, atol=2e-5)


def _check_grad_summary(g, prefix, grads, rtol):
    names = sorted({k.split("|")[1] for k in g if isinstance(k, str) and k.startswith(prefix + "|")})
    assert len(names) == len(grads), (len(names), len(grads))
    for name in names:
        gr = grads[name].reshape(-1)
        ref_norm = g[f"{prefix}|{name}|norm"]
        assert abs(gr.double().norm().item() - ref_norm) <= rtol * max(ref_norm, 1e-12), name
        scale = ref_norm / max(gr.numel(), 1) ** 0.5
        err = (gr[g[f"{prefix}|{name}|idx"]] - g[f"{prefix}|{name}|val"]).abs().max().item()
        assert err <= rtol * max(g[f"{prefix}|{name}|val"].abs().max().item(), scale), (name, err)


def test_g9_backward(golden, nerf_sd):
    """R14: gradients of mse(coarse)+mse(fine) from the oracle's autograd equal the reference's own (same torch
    kernels, same graph up to the searchsorted restatement of the inverse CDF, whose outputs are detached)."""
    import aon_amd.synthetic as syn

    g = golden("g9_backward")
    rays = {k: g[k] for k in ("rays_o", "rays_d", "viewdirs")}
    sd = {k: v.clone().requires_grad_(True) for k, v in nerf_sd.items()}
    out = orc.nerf_forward(sd, rays, False, True, g["near"], g["far"])
    loss = orc.img2mse(out[0][0], g["target"]) + orc.img2mse(out[1][0], g["target"])
    loss.backward()
    assert abs(loss.item() - g["vanilla_loss"]) <= 1e-6
    _check_grad_summary(g, "vanilla", {k: v.grad for k, v in sd.items()}, rtol=2e-4)
    asd = {k: v.clone().requires_grad_(True) for k, v in syn.make_art_state_dict(seed=0, density_scale=30.0).items()}
    ga = golden("g11_nerf_ae_art")
    lat = {k: ga[f"lat_train_{k}"].clone().requires_grad_(True) for k in ("density", "color", "articulation")}
    out = orc.nerf_ae_art_forward(asd, rays, False, True, g["near"], g["far"], lat)
    loss = orc.img2mse(out[0][0], g["target"]) + orc.img2mse(out[1][0], g["target"])
    loss.backward()
    assert abs(loss.item() - g["art_loss"]) <= 1e-6
    _check_grad_summary(g, "art", {k: v.grad for k, v in asd.items()}, rtol=2e-3)
    for k, v in lat.items():
        ref = g[f"art_latgrad_{k}"]
        assert (v.grad - ref).abs().max().item() <= 2e-3 * ref.abs().max().item(), k


def test_g15_smooth_fields(golden):
    """The oracle against the reference's outputs on the smooth ("trained-like") fields, both networks, every ray."""
    import aon_amd.synthetic as syn

    g = golden("g15_smooth")
    rays = {k: g[k] for k in ("rays_o", "rays_d", "viewdirs")}
    n = rays["rays_o"].shape[0]
    assert n >= 1024 and g["art_rays_o"].shape[0] >= 1024
    sd = syn.make_smooth_nerf_state_dict()
    t_rand, u = syn.seeded_uniform(g["seed_t_rand"], n, 65), syn.seeded_uniform(g["seed_u"], n, 128)   # draws are named by seed
    for tag, kw in (("van_det", dict(randomized=False, white_bkgd=True)),
                    ("van_rnd", dict(randomized=True, white_bkgd=False, t_rand=t_rand, u=u))):
        out = orc.nerf_forward(sd, rays, near=g["near"], far=g["far"], **kw)
        for lvl, name in ((0, "coarse"), (1, "fine")):
            torch.testing.assert_close(out[lvl][0], g[f"{tag}_{name}_rgb"], rtol=0, atol=2e-6)
            torch.testing.assert_close(out[lvl][2], g[f"{tag}_{name}_depth"], rtol=0, atol=2e-5)
    arays = {k: g["art_" + k] for k in ("rays_o", "rays_d", "viewdirs")}
    lat = {k: g["art_lat_" + k] for k in ("density", "color", "articulation")}
    asd = syn.make_art_state_dict(seed=5, density_scale=2.0)
    ta, ua = syn.seeded_uniform(g["seed_art_t_rand"], n, 65), syn.seeded_uniform(g["seed_art_u"], n, 128)
    for tag, kw in (("art_det", dict(randomized=False, white_bkgd=True)), ("art_rnd", dict(randomized=True, white_bkgd=False, t_rand=ta, u=ua))):
        out = orc.nerf_ae_art_forward(asd, arays, kw["randomized"], kw["white_bkgd"], g["near"], g["far"], lat, t_rand=kw.get("t_rand"), u=kw.get("u"))
        for lvl, name in ((0, "coarse"), (1, "fine")):
            torch.testing.assert_close(out[lvl][0], g[f"{tag}_{name}_rgb"], rtol=0, atol=2e-6)
            torch.testing.assert_close(out[lvl][2], g[f"{tag}_{name}_depth"], rtol=0, atol=2e-5)


# ---------------------------------------------------------------------------------------------------------------------
# G16: constructor arguments beyond the defaults (lindisp, other sample counts, noise_std, rgb_padding / density_bias)
# ---------------------------------------------------------------------------------------------------------------------
def test_aten_sum_model_is_torch_sum():
    """The numpy restatement of ATen's CPU row sum (what the general-size inverse-CDF kernel follows) against torch.sum on
    the non-contiguous `weights[..., 1:-1]` view the reference sums (helper.py:205), K = 1 .. 1000."""
    import numpy as np

    rng = np.random.default_rng(0)
    for K in list(range(1, 70)) + [95, 127, 128, 129, 191, 199, 255, 256, 257, 511, 513, 599, 1000]:
        for _ in range(20):
            x = (rng.random(K) ** 8 * rng.choice([1.0, 1e-3, 1e3], K)).astype(np.float32)
            big = np.zeros((3, K + 2), np.float32)
            big[:, 1:-1] = x
            want = torch.from_numpy(big)[..., 1:-1].sum(dim=-1, keepdim=True)[0, 0].item()
            assert np.float32(want) == orc.aten_sum_model(x), K


def test_g16_lindisp_and_pdf_sizes(golden):
    g = golden("g16_ctor_options")
    o, d = g["lindisp_rays_o"], g["lindisp_rays_d"]
    import aon_amd.synthetic as syn
    for tag in "abc":
        ns, near, far = g[f"lindisp_{tag}_ns"], g[f"lindisp_{tag}_near"], g[f"lindisp_{tag}_far"]
        t, _ = orc.sample_along_rays(o, d, ns, near, far, False, lindisp=True)
        assert torch.equal(t, g[f"lindisp_{tag}_t_det"])
        t, c = orc.sample_along_rays(o, d, ns, near, far, True, syn.seeded_uniform(1600 + ns, 48, ns + 1), lindisp=True)
        assert torch.equal(t, g[f"lindisp_{tag}_t_rnd"])
        torch.testing.assert_close(c.double().sum((0, 1)), g[f"lindisp_{tag}_coords_rnd_sum"], rtol=1e-12, atol=0)
    for nb, nf in g["pdf_sizes"].tolist():
        k = f"pdf_{nb}_{nf}"
        t, w = g[f"{k}_t"], g[f"{k}_w"]
        n = t.shape[0]
        mids = 0.5 * (t[..., 1:] + t[..., :-1])
        u = syn.seeded_uniform(1700 + nb, n, nf)
        assert torch.equal(orc.sorted_piecewise_constant_pdf(mids, w, nf, False), g[f"{k}_samples_det"])
        assert torch.equal(orc.sorted_piecewise_constant_pdf(mids, w, nf, True, u), g[f"{k}_samples_rnd"])
        z = torch.zeros(n, 3)
        assert torch.equal(orc.sample_pdf(mids, w, z, z, t, nf, False)[0], g[f"{k}_t_fine_det"])
        assert torch.equal(orc.sample_pdf(mids, w, z, z, t, nf, True, u)[0], g[f"{k}_t_fine_rnd"])


def test_g16_whole_path_with_options(golden):
    import aon_amd.synthetic as syn

    g = golden("g16_ctor_options")
    nc, nf, lind = g["van_cfg"].tolist()
    rays = {k: g["van_" + k] for k in ("rays_o", "rays_d", "viewdirs")}
    n = rays["rays_o"].shape[0]
    s = g["van_seeds"].tolist()
    tr, u = syn.seeded_uniform(s[0], n, nc + 1), syn.seeded_uniform(s[1], n, nf)
    nz = [syn.seeded_uniform(s[2], n, nc + 1), syn.seeded_uniform(s[3], n, nc + 1 + nf)]
    sd = syn.make_smooth_nerf_state_dict()
    kw = dict(num_coarse_samples=nc, num_fine_samples=nf, lindisp=bool(lind))
    outs = {"van_det": orc.nerf_forward(sd, rays, False, True, 2.0, 6.0, **kw),
            "van_rnd": orc.nerf_forward(sd, rays, True, False, 2.0, 6.0, t_rand=tr, u=u, noise_std=g["van_noise_std"], noise=nz, **kw)}
    for tag, out in outs.items():
        for lvl, name in ((0, "coarse"), (1, "fine")):
            torch.testing.assert_close(out[lvl][0], g[f"{tag}_{name}_rgb"], rtol=0, atol=2e-6)
            torch.testing.assert_close(out[lvl][1], g[f"{tag}_{name}_acc"], rtol=0, atol=2e-6)
            torch.testing.assert_close(out[lvl][2], g[f"{tag}_{name}_depth"], rtol=0, atol=2e-5)
    nc, nf, lind = g["art_cfg"].tolist()
    rays = {k: g["art_" + k] for k in ("rays_o", "rays_d", "viewdirs")}
    lat = {k: g["art_lat_" + k] for k in ("density", "color", "articulation")}
    s = g["art_seeds"].tolist()
    tr, u = syn.seeded_uniform(s[0], n, nc + 1), syn.seeded_uniform(s[1], n, nf)
    nz = [syn.seeded_uniform(s[2], n, nc + 1), syn.seeded_uniform(s[3], n, nc + 1 + nf)]
    sd = syn.make_art_state_dict(seed=5, density_scale=2.0)
    kw = dict(num_coarse_samples=nc, num_fine_samples=nf, lindisp=bool(lind), rgb_padding=g["art_rgb_padding"], density_bias=g["art_density_bias"])
    outs = {"art_det": orc.nerf_ae_art_forward(sd, rays, False, True, 2.0, 6.0, lat, **kw),
            "art_rnd": orc.nerf_ae_art_forward(sd, rays, True, False, 2.0, 6.0, lat, t_rand=tr, u=u, noise_std=g["art_noise_std"], noise=nz, **kw)}
    for tag, out in outs.items():
        for lvl, name in ((0, "coarse"), (1, "fine")):
            torch.testing.assert_close(out[lvl][0], g[f"{tag}_{name}_rgb"], rtol=0, atol=2e-6)
            torch.testing.assert_close(out[lvl][1], g[f"{tag}_{name}_acc"], rtol=0, atol=2e-6)
            torch.testing.assert_close(out[lvl][2], g[f"{tag}_{name}_depth"], rtol=0, atol=2e-5)


# ---------------------------------------------------------------------------------------------------------------------
# G17: NeRFMLP / NeRF of non-default geometry (the reference's constructors with other arguments)
# ---------------------------------------------------------------------------------------------------------------------
GEOM_KEYS = ("min_deg_point", "max_deg_point", "deg_view", "netdepth", "netwidth", "netdepth_condition", "netwidth_condition", "skip_layer",
             "input_ch", "input_ch_view", "num_rgb_channels", "num_density_channels")


def test_g17_general_mlp(golden):
    import aon_amd.synthetic as syn

    g = golden("g17_general_mlp")
    for code in g["mlp_tags"].tolist():
        tag = chr(code)
        kw = dict(zip(GEOM_KEYS, g[f"mlp_{tag}_geom"].tolist()))
        sd = syn.make_general_nerf_state_dict(1700 + code, prefixes=("",), **kw)
        rgb, dens = orc.nerf_mlp(sd, "", g[f"mlp_{tag}_x"], g[f"mlp_{tag}_v"], skip_layer=kw["skip_layer"])
        torch.testing.assert_close(rgb, g[f"mlp_{tag}_rgb"], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(dens, g[f"mlp_{tag}_density"], rtol=1e-5, atol=1e-5)
    rays = {k: g["nerf_" + k] for k in ("rays_o", "rays_d", "viewdirs")}
    n = rays["rays_o"].shape[0]
    for tag in "pq":
        mn, mx, dv, nc, nf, lind = g[f"nerf_{tag}_cfg"].tolist()
        sd = syn.make_general_nerf_state_dict(1750 + ord(tag), min_deg_point=mn, max_deg_point=mx, deg_view=dv)
        kw = dict(min_deg_point=mn, max_deg_point=mx, deg_view=dv, num_coarse_samples=nc, num_fine_samples=nf, lindisp=bool(lind))
        tr, u = syn.seeded_uniform(1760 + ord(tag), n, nc + 1), syn.seeded_uniform(1770 + ord(tag), n, nf)
        ok = g[f"nerf_{tag}_margin"] > 0.02
        outs = {"det": orc.nerf_forward(sd, rays, False, True, 2.0, 6.0, **kw), "rnd": orc.nerf_forward(sd, rays, True, False, 2.0, 6.0, t_rand=tr, u=u, **kw)}
        for t2, out in outs.items():
            for lvl, name in ((0, "coarse"), (1, "fine")):
                torch.testing.assert_close(out[lvl][0][ok], g[f"nerf_{tag}_{t2}_{name}_rgb"][ok], rtol=0, atol=2e-5)
                torch.testing.assert_close(out[lvl][1][ok], g[f"nerf_{tag}_{t2}_{name}_acc"][ok], rtol=0, atol=2e-5)


# ---------------------------------------------------------------------------------------------------------------------
# G18: the articulated network at other encoding degrees (round 4; NeRF_AE_Art(min_deg_point, max_deg_point, deg_view) of the reference)
# ---------------------------------------------------------------------------------------------------------------------
def test_g18_articulated_degrees(golden):
    import aon_amd.synthetic as syn

    g = golden("g18_art_degrees")
    rays = {k: g[k] for k in ("rays_o", "rays_d", "viewdirs")}
    lat = {k: g["lat_" + k] for k in ("density", "color", "articulation")}
    n = rays["rays_o"].shape[0]
    for tag in "abc":
        mn, mx, dv = g[f"{tag}_cfg"].tolist()
        gk = dict(min_deg_point=mn, max_deg_point=mx, deg_view=dv)
        sd = syn.make_art_state_dict(seed=18, density_scale=2.0, **gk)
        rgb, dens = orc.art_mlp(sd, "fine_mlp.", g[f"{tag}_mlp_pos"], g[f"{tag}_mlp_cond"], lat, mn, mx)
        torch.testing.assert_close(rgb, g[f"{tag}_mlp_raw_rgb"], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(dens, g[f"{tag}_mlp_raw_density"], rtol=1e-5, atol=1e-5)
        s = g[f"{tag}_seeds"].tolist()
        tr, u = syn.seeded_uniform(s[0], n, 65), syn.seeded_uniform(s[1], n, 128)
        outs = {"det": orc.nerf_ae_art_forward(sd, rays, False, True, 2.0, 6.0, lat, **gk),
                "rnd": orc.nerf_ae_art_forward(sd, rays, True, False, 2.0, 6.0, lat, t_rand=tr, u=u, **gk)}
        for t2, out in outs.items():
            for lvl, name in ((0, "coarse"), (1, "fine")):
                torch.testing.assert_close(out[lvl][0], g[f"{tag}_{t2}_{name}_rgb"], rtol=0, atol=2e-6)
                torch.testing.assert_close(out[lvl][1], g[f"{tag}_{t2}_{name}_acc"], rtol=0, atol=2e-6)
                torch.testing.assert_close(out[lvl][2], g[f"{tag}_{t2}_{name}_depth"], rtol=0, atol=2e-5)


# ---------------------------------------------------------------------------------------------------------------------------------
# Round 6: the full-size fixtures of tests/golden/make_golden_full.py (REAL reference outputs at the sizes of BASELINE configs 2, 4, 5).
# The GPU tests compare the HIP path with these directly; here the oracle is held to them, so that the oracle-vs-HIP tests elsewhere and
# the cpu_baseline leg of bench.py stand on reference-pinned ground at these sizes too.
# ---------------------------------------------------------------------------------------------------------------------------------
def _frame_fixture_vs_oracle(g, out):
    for lvl, lname in ((0, "coarse"), (1, "fine")):
        for i, (name, atol) in enumerate((("rgb", 2e-6), ("acc", 2e-6), ("depth", 2e-5))):
            ref = g[f"ref_{lname}_{name}"]
            err = (out[lvl][i] - ref).abs()
            print(f"{lname} {name}: {ref.shape[0]} rays, max |oracle - reference| {err.max():.2e}")
            # same operations in the same order: the only freedom is the thread count of the (N*S,256)x(256,256) products
            assert err.max().item() <= atol, (lname, name, err.max().item())


def test_g19_config2_frame(golden, nerf_sd):
    """4,209 strided rays of the 640x480 frame through the reference's NeRF.forward (fp32): both levels, rgb / acc / depth."""
    g = golden("g19_config2_frame")
    rays = {k: g[k] for k in ("rays_o", "rays_d", "viewdirs")}
    torch.set_num_threads(8)
    with torch.no_grad():
        out, aux = orc.nerf_forward(nerf_sd, rays, False, True, g["near"], g["far"], return_aux=True)
    _frame_fixture_vs_oracle(g, out)
    margin = torch.stack([a["raw_sigma"][:, -1, 0].abs() for a in aux]).min(0).values
    torch.testing.assert_close(margin, g["margin"], rtol=0, atol=1e-3)
    # the recorded spreads are distances of the reference to ITSELF in fp64: sane magnitudes (coarse: 1e-5 class; fine: chaotic tail)
    assert g["spread_coarse_rgb"].max().item() < 1e-3 and g["spread_fine_rgb"].median().item() < 1e-5


def test_g20_config4_frame(golden):
    """4,267 strided rays of the articulated 320x240 frame through the reference's NeRF_AE_Art.forward (fp32)."""
    import aon_amd.synthetic as syn

    g = golden("g20_config4_frame")
    sd = syn.make_art_state_dict(seed=0, density_scale=30.0)
    lib = syn.make_code_library_state(seed=0, n_max_objs=1)
    lat = orc.code_library(lib, torch.tensor([g["instance_id"]]), torch.tensor([g["articulation_id"]]))
    for k in ("density", "color", "articulation"):
        assert torch.equal(lat[k], g["lat_" + k])
    rays = {k: g[k] for k in ("rays_o", "rays_d", "viewdirs")}
    torch.set_num_threads(8)
    with torch.no_grad():
        out = orc.nerf_ae_art_forward(sd, rays, False, True, g["near"], g["far"], lat)
    _frame_fixture_vs_oracle(g, out)


def test_g21_config5_step(golden):
    """One 4096-ray articulated training step: the oracle's fp32 autograd must be to the reference's fp64 truth what the reference's own
    fp32 autograd is (G21 holds both) -- same operations, so the two fp32 distances agree closely; and the loss."""
    import aon_amd.synthetic as syn

    g = golden("g21_config5_step")
    n = g["n"]
    sd = syn.make_art_state_dict(seed=0, density_scale=30.0)
    lib_sd = syn.make_code_library_state(seed=0, n_max_objs=1)
    rays = {k: g[k] for k in ("rays_o", "rays_d", "viewdirs")}
    gen = torch.Generator().manual_seed(g["generator_seed"])
    assert torch.equal(torch.randint(0, g["H"] * g["W"], (n,), generator=gen), g["idx"])
    target = torch.rand(n, 3, generator=gen)
    t_rand, u = torch.rand(n, 65, generator=gen), torch.rand(n, 128, generator=gen)
    assert target.double().sum().item() == g["sum_target"] and t_rand.double().sum().item() == g["sum_t_rand"] and u.double().sum().item() == g["sum_u"]
    inst, art_id = torch.tensor([g["instance_id"]]), torch.tensor([g["articulation_id"]])
    torch.set_num_threads(8)
    sd_o = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    lib_o = {k: v.clone().requires_grad_(True) for k, v in lib_sd.items()}
    total, chunk = 0.0, 512
    for r0 in range(0, n, chunk):
        sl = slice(r0, r0 + chunk)
        lat = orc.code_library(lib_o, inst, art_id)
        out = orc.nerf_ae_art_forward(sd_o, {k: v[sl] for k, v in rays.items()}, True, True, 2.0, 6.0, lat, t_rand=t_rand[sl], u=u[sl])
        m = target[sl].shape[0]
        part = (orc.img2mse(out[1][0], target[sl]) + orc.img2mse(out[0][0], target[sl])) * (m / n)
        part.backward()
        total += part.item()
    lat = orc.code_library(lib_o, inst, art_id)
    reg = 1e-4 * sum(torch.mean(torch.norm(lat[k], dim=0)) for k in ("density", "color", "articulation"))
    reg.backward()
    loss = total + reg.item()
    assert abs(loss - g["loss32"]) <= 2e-6 * abs(g["loss32"]), (loss, g["loss32"])
    grads = {k: v.grad for k, v in sd_o.items()}
    grads.update({"lib." + k: v.grad for k, v in lib_o.items()})
    names = sorted(k[: -len("|norm")] for k in g if k.endswith("|norm"))
    assert set(names) == set(grads)
    worst = (0.0, "")
    for name in names:
        gh, nrm = grads[name].double().reshape(-1), max(g[f"{name}|norm"], 1e-30)
        if f"{name}|truth" in g:
            e, e_ref = (gh - g[f"{name}|truth"].double().reshape(-1)).norm().item() / nrm, g[f"{name}|ref32_dist"] / nrm
        else:
            sel = torch.arange(g[f"{name}|truth_sel"].numel()) * g[f"{name}|sel_step"]
            nsel = max(g[f"{name}|norm_sel"], 1e-30)
            e, e_ref = (gh[sel] - g[f"{name}|truth_sel"].double()).norm().item() / nsel, g[f"{name}|ref32_dist_sel"] / nsel
        worst = max(worst, (e / max(e_ref, 1e-7), name))
        # the oracle's fp32 is the reference's fp32 up to summation order: within 1.5 x of its distance to the truth (+ the fp32 storage of the truth)
        assert e <= 1.5 * e_ref + 2e-7, (name, e, e_ref)
    print(f"g21: worst (oracle fp32 distance to truth) / (reference fp32 distance to truth) = {worst[0]:.2f} on {worst[1]}")
