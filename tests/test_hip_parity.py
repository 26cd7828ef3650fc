"""GPU parity tests: the HIP path (through the C ABI, via aon_amd.ops / the drop-in modules) against the CPU
oracle on identical seeded inputs and against the committed golden vectors.

Tolerances (fp32 path; stated per SURVEY 7 "parity hazards" / BASELINE.md 3):
  * index / selection / sort work (stratified t, merge-sort): bit-exact
  * sin-based encoding: 2.5e-7 abs (device sin_f32 vs the host libm, both ~1 ulp)
  * MLP raw outputs: 2e-5 abs + 2e-5 rel on rgb, sigma scaled by the density head (different fp32 summation order)
  * compositing on identical inputs: 2e-6 abs; inverse CDF and sort-merge: BIT-EXACT against the reference's draws (the
    kernel reproduces torch's CPU summation orders, see aon_render.hip:torch_sum63 and the double-accumulated cumsum)
  * end to end: PSNR(HIP, oracle) >= 70 dB and >= 99.9 % of values within 1e-3; measured values are far tighter
    and asserted at 2e-4 on rays whose far-plane density is robustly signed.
"""
import math

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


@pytest.fixture(scope="module")
def ops(dev):
    from aon_amd import ops as _ops

    return _ops


@pytest.fixture(scope="module")
def packed(ops, dev, nerf_sd):
    out = {}
    for lvl in ("coarse", "fine"):
        params = {k[len(lvl) + 5:]: v.to(dev) for k, v in nerf_sd.items() if k.startswith(lvl + "_mlp.")}
        out[lvl] = ops.pack_vanilla_mlp(params)
    return out


def frac_within(a, b, atol):
    return (torch.abs(a - b) <= atol).double().mean().item()


# ------------------------------------------------------------------ R1/R2
def test_raygen_matches_oracle_and_golden(ops, dev, golden):
    g = golden("g1_raygen")
    H, W, focal = g["H"], g["W"], g["focal"]
    dirs = ops.ray_directions(H, W, focal, device=dev)
    assert torch.equal(dirs.cpu(), g["directions"])
    for p in range(g["c2w"].shape[0]):
        ro, vd = ops.raygen(g["c2w"][p], H, W, focal, device=dev)
        assert torch.equal(ro.cpu(), g["rays_o"][p])
        torch.testing.assert_close(vd.cpu(), g["viewdirs"][p], rtol=0, atol=2e-7)
        ro2, vd2 = ops.get_rays(dirs, g["c2w"][p])
        assert torch.equal(vd2, vd) and torch.equal(ro2, ro)
    # full 640x480 frame against golden picks / checksums, and a sub-range equals the same slice of the frame
    Hf, Wf, ff = g["full_H"], g["full_W"], g["full_focal"]
    ro, vd = ops.raygen(g["c2w"][0], Hf, Wf, ff, device=dev)
    torch.testing.assert_close(vd.cpu()[g["full_pick"]], g["full_viewdirs_pick"], rtol=0, atol=2e-7)
    torch.testing.assert_close(vd.double().sum(0).cpu(), g["full_viewdirs_sum"], rtol=0, atol=5e-3)
    torch.testing.assert_close(vd.double().abs().sum(0).cpu(), g["full_viewdirs_abs_sum"], rtol=0, atol=5e-3)
    torch.testing.assert_close(vd.norm(dim=-1).cpu(), torch.ones(Hf * Wf), rtol=0, atol=2e-7)
    _, vd_part = ops.raygen(g["c2w"][0], Hf, Wf, ff, 38_400, 76_800, device=dev)
    assert torch.equal(vd_part, vd[38_400:76_800])


def test_get_rays_reference_call_form(dev, golden):
    """The literal call of every reference dataset (datasets/sapien.py:102,145; sapien_multi.py:301,343):
    ``rays_o, view_dirs, rays_d, radii = get_rays(directions, c2w, output_view_dirs=True, output_radii=True)``
    through the name-compatible mirror, against G1 (the reference's own four outputs)."""
    from aon_amd.datasets.ray_utils import get_ray_directions, get_rays

    g = golden("g1_raygen")
    directions = get_ray_directions(g["H"], g["W"], g["focal"], device=dev)
    for p in range(g["c2w"].shape[0]):
        c2w = g["c2w"][p]
        rays_o, view_dirs, rays_d, radii = get_rays(directions, c2w, output_view_dirs=True, output_radii=True)
        assert torch.equal(rays_o.cpu(), g["rays_o"][p])
        torch.testing.assert_close(view_dirs.cpu(), g["viewdirs"][p], rtol=0, atol=2e-7)
        torch.testing.assert_close(rays_d.cpu(), g["rays_d"][p], rtol=0, atol=2e-7)
        assert rays_d.data_ptr() == view_dirs.data_ptr()   # the reference's rays_d IS its viewdirs storage (ray_utils.py:146-147)
        assert radii.shape == (g["H"] * g["W"],)
        # fp32: differences of O(1) world directions (each within an ulp of the CPU matmul's) -> 2e-7 absolute on ~4e-3 values
        torch.testing.assert_close(radii.cpu(), g["radii"][p], rtol=0, atol=2e-7)
    # full 640x480 frame: picks, the copied last row (image row H-1 <- row H-3) and the checksum
    Hf, Wf = g["full_H"], g["full_W"]
    dirs_f = get_ray_directions(Hf, Wf, g["full_focal"], device=dev)
    out = get_rays(dirs_f, g["c2w"][0], output_view_dirs=True, output_radii=True)
    assert len(out) == 4
    rad = out[3].cpu()
    torch.testing.assert_close(rad[g["full_pick"]], g["full_radii_pick"], rtol=0, atol=2e-7)
    torch.testing.assert_close(rad.view(Hf, Wf)[-3:, ::80], g["full_radii_last_rows"], rtol=0, atol=2e-7)
    assert torch.equal(rad.view(Hf, Wf)[-1], rad.view(Hf, Wf)[-3])
    torch.testing.assert_close(rad.double().sum(), torch.as_tensor(g["full_radii_sum"], dtype=torch.float64), rtol=1e-5, atol=0)
    # the other call forms keep the reference's arity
    assert len(get_rays(dirs_f, g["c2w"][0])) == 2 and len(get_rays(dirs_f, g["c2w"][0], output_view_dirs=True)) == 3


# ------------------------------------------------------------------ R13
def test_metrics_product_side_vs_golden(dev, golden):
    """R13: the PRODUCT's metric functions (helper.img2mse / mse2psnr, LitModel.psnr_legacy / psnr_each / mse) on device
    tensors against G13, the reference's own outputs (helper.py:17-22, models/interface.py:54-74)."""
    from aon_amd.models.interface import LitModel
    from aon_amd.models.vanilla_nerf import helper

    g = golden("g13_metrics")
    a, b = g["a"].to(dev), g["b"].to(dev)
    mse = helper.img2mse(a, b)
    assert mse.is_cuda
    # fp32 mean over 3,840 elements: the CPU and device reductions associate differently -> 2e-6 relative (PSNR: 2e-5 dB)
    torch.testing.assert_close(mse.cpu(), torch.as_tensor(g["mse"]), rtol=2e-6, atol=0)
    torch.testing.assert_close(helper.mse2psnr(mse).cpu(), torch.as_tensor(g["mse2psnr"]), rtol=2e-6, atol=2e-5)
    lit = LitModel()
    torch.testing.assert_close(lit.mse(a, b).cpu(), torch.as_tensor(g["mse"]), rtol=2e-6, atol=0)
    torch.testing.assert_close(lit.psnr_legacy(a, b).cpu(), torch.as_tensor(g["psnr_legacy"]), rtol=2e-6, atol=2e-5)
    torch.testing.assert_close(lit.psnr_each(list(a), list(b)).cpu(), g["psnr_each"], rtol=2e-6, atol=2e-5)
    # the training loss of model.py:271-273 is img2mse(coarse) + img2mse(fine): same function, sum of two
    torch.testing.assert_close((helper.img2mse(a, b) + helper.img2mse(b, a)).cpu(), 2 * torch.as_tensor(g["mse"]), rtol=2e-6, atol=0)


def test_train_loss_two_launches_vs_torch_lines(dev, golden):
    """R13, round 5: helper.train_loss (aon_train_loss_fwd / _bwd: the loss lines of model.py:271-273 and model_autodecoder.py:455-466 as
    two launches) against the same lines written in torch, the reference's formulation, on the G13 images and on a 4096-ray batch:
    values to 2e-6 relative (fp64 sums here, fp32 tree sums there), gradients of the rendered colours to 1e-6 relative per element
    (same operations in the same order: observed equal), codes likewise, incl. a zero entry (norm_backward's masked_fill) and a loss
    scaled before backward (grad_loss != 1)."""
    from aon_amd.models.vanilla_nerf import helper

    g = golden("g13_metrics")
    gen = torch.Generator().manual_seed(3)
    cases = [(g["a"].reshape(-1, 3).to(dev), g["b"].reshape(-1, 3).to(dev), g["b"].flip(0).reshape(-1, 3).to(dev)),
             (torch.rand(4096, 3, generator=gen).to(dev), torch.rand(4096, 3, generator=gen).to(dev), torch.rand(4096, 3, generator=gen).to(dev))]
    for rgb_c0, rgb_f0, target in cases:
        for with_codes in (False, True):
            codes0 = [torch.randn(1, 128, generator=gen), torch.randn(1, 128, generator=gen), torch.randn(1, 32, generator=gen)] if with_codes else []
            if with_codes:
                codes0[0][0, 5] = 0.0
            res = {}
            for form in ("torch", "fused"):
                rc, rf = rgb_c0.clone().requires_grad_(True), rgb_f0.clone().requires_grad_(True)
                codes = [c.clone().to(dev).requires_grad_(True) for c in codes0]
                rendered = [(rc, None, None), (rf, None, None)]
                if form == "torch":
                    loss0, loss1 = helper.img2mse(rc, target), helper.img2mse(rf, target)
                    loss = loss1 + loss0
                    reg = torch.zeros((), device=dev)
                    if with_codes:
                        reg = 1e-4 * (torch.mean(torch.norm(codes[0], dim=0)) + torch.mean(torch.norm(codes[1], dim=0)) + torch.mean(torch.norm(codes[2], dim=0)))
                        loss = loss + reg
                    stats = torch.stack([loss0, loss1, reg, loss, helper.mse2psnr(loss0), helper.mse2psnr(loss1)]).detach()
                else:
                    loss, stats = helper.train_loss(rendered, target, codes, 1e-4)
                    assert not stats.requires_grad and loss.shape == ()
                (loss * 3.0).backward()
                res[form] = (loss.detach(), stats[:6], rc.grad, rf.grad, [c.grad for c in codes])
            t, f = res["torch"], res["fused"]
            torch.testing.assert_close(f[0], t[0], rtol=2e-6, atol=0)
            torch.testing.assert_close(f[1], t[1], rtol=2e-6, atol=2e-5)
            for a, b in zip((f[2], f[3]) + tuple(f[4]), (t[2], t[3]) + tuple(t[4])):
                torch.testing.assert_close(a, b, rtol=1e-6, atol=0)
            if with_codes:
                assert f[4][0][0, 5].item() == 0.0
    # one level (num_levels = 1): loss0 = 0, no coarse gradient
    rf = cases[1][1].clone().requires_grad_(True)
    loss, stats = helper.train_loss([(rf, None, None)], cases[1][2])
    loss.backward()
    torch.testing.assert_close(loss.detach(), helper.img2mse(rf.detach(), cases[1][2]), rtol=2e-6, atol=0)
    assert stats[0].item() == 0.0 and torch.isfinite(rf.grad).all()


# ------------------------------------------------------------------ R3
def test_sample_along_rays_bit_exact(ops, dev, golden):
    g = golden("g2_sample_along_rays")
    o, d = g["rays_o"].to(dev), g["rays_d"].to(dev)
    t, c = ops.sample_along_rays(o, d, 64, g["near"], g["far"])
    assert torch.equal(t.cpu(), g["t_det"]) and torch.equal(c.cpu(), g["coords_det"])
    t, c = ops.sample_along_rays(o, d, 64, g["near"], g["far"], t_rand=g["t_rand"].to(dev))
    assert torch.equal(t.cpu(), g["t_rnd"]) and torch.equal(c.cpu(), g["coords_rnd"])
    assert torch.equal(ops.cast_rays(t, o, d).cpu(), g["coords_rnd"])
    # other sample counts follow torch.linspace's two-sided formula; when 1/(S-1) is inexact the host's vectorised
    # linspace may fuse start + step*i, so allow 1 ulp there (the reference path only ever uses S = 65: exact)
    for ns in (1, 7, 100):
        t_or, c_or = orc.sample_along_rays(g["rays_o"], g["rays_d"], ns, 0.5, 3.25, False)
        t, c = ops.sample_along_rays(o, d, ns, 0.5, 3.25)
        torch.testing.assert_close(t.cpu(), t_or.contiguous(), rtol=0, atol=2.5e-7)
        torch.testing.assert_close(c.cpu(), c_or, rtol=0, atol=1e-6)
    # the t-only form the render / training paths use (four elements per thread, 16-byte stores) gives the same bits as the
    # per-element kernel, deterministic and randomized, at ray counts that leave every kind of tail
    t_ref, _ = ops.sample_along_rays(o, d, 64, g["near"], g["far"], t_rand=g["t_rand"].to(dev))
    t4, none = ops.sample_along_rays(o, d, 64, g["near"], g["far"], t_rand=g["t_rand"].to(dev), want_coords=False)
    assert none is None and torch.equal(t4, t_ref) and torch.equal(t4.cpu(), g["t_rnd"])
    gen = torch.Generator().manual_seed(17)
    for n in (1, 2, 3, 5, 63, 1001):
        oo, dd = torch.randn(n, 3, generator=gen).to(dev), torch.randn(n, 3, generator=gen).to(dev)
        for ns in (64, 6, 2):
            tr = torch.rand(n, ns + 1, generator=gen).to(dev)
            for rnd in (None, tr):
                a, _ = ops.sample_along_rays(oo, dd, ns, 2.0, 6.0, t_rand=rnd)
                b, _ = ops.sample_along_rays(oo, dd, ns, 2.0, 6.0, t_rand=rnd, want_coords=False)
                assert torch.equal(a, b), (n, ns, rnd is None)


# ------------------------------------------------------------------ R4
def test_pos_enc(ops, dev, golden):
    g = golden("g3_pos_enc")
    e10 = ops.pos_enc(g["x"].to(dev), 0, 10).cpu()
    assert torch.equal(e10[:, :3], g["x"])
    torch.testing.assert_close(e10, g["enc10"], rtol=0, atol=2.5e-7)
    torch.testing.assert_close(ops.pos_enc(g["v"].to(dev), 0, 4).cpu(), g["enc4"], rtol=0, atol=2.5e-7)
    # large arguments (|x| * 2^9 ~ 3000) against an fp64 evaluation of the same fp32 arguments
    x = (torch.rand(4096, 3, generator=torch.Generator().manual_seed(3)) * 12 - 6)
    e = ops.pos_enc(x.to(dev), 0, 10).cpu()
    scales = torch.tensor([2.0 ** l for l in range(10)])
    xb = (x[:, None, :] * scales[:, None]).reshape(x.shape[0], -1)
    ref = torch.sin(torch.cat([xb, xb + orc.HALF_PI_F32], -1).double())
    assert (e[:, 3:].double() - ref).abs().max().item() < 1.5e-7


# ------------------------------------------------------------------ R5
def test_mlp_on_golden_encodings(ops, dev, golden, packed):
    g = golden("g4_mlp")
    for lvl in ("coarse", "fine"):
        raw = ops.mlp_fwd_enc(packed[lvl], g["samples_enc"].to(dev), g["viewdirs_enc"].to(dev)).cpu()
        torch.testing.assert_close(raw[..., :3], g[f"raw_rgb_{lvl}"], rtol=2e-5, atol=2e-5)
        torch.testing.assert_close(raw[..., 3:], g[f"raw_sigma_{lvl}"], rtol=2e-5, atol=6e-4)  # density head x30


def test_mlp_fused_encode_vs_oracle(ops, dev, nerf_sd, packed):
    import aon_amd.synthetic as syn

    for n, S, seed in ((1, 65, 1), (37, 65, 2), (50, 193, 3), (128, 1, 4)):
        rays = syn.random_rays(n, seed=seed)
        t = torch.sort(torch.rand(n, S, generator=torch.Generator().manual_seed(seed)) * 4 + 2, dim=-1).values
        enc = orc.pos_enc(orc.cast_rays(t, rays["rays_o"], rays["rays_d"]), 0, 10)
        venc = orc.pos_enc(rays["viewdirs"], 0, 4)
        rgb_o, sig_o = orc.nerf_mlp(nerf_sd, "fine_mlp.", enc, venc)
        raw = ops.mlp_fwd(packed["fine"], rays["rays_o"].to(dev), rays["rays_d"].to(dev), rays["viewdirs"].to(dev), t.to(dev)).cpu()
        torch.testing.assert_close(raw[..., :3], rgb_o, rtol=5e-5, atol=5e-5)
        torch.testing.assert_close(raw[..., 3:], sig_o, rtol=5e-5, atol=2e-3)
        # the stage-level entry point on the oracle's encodings agrees with the fused one
        raw_e = ops.mlp_fwd_enc(packed["fine"], enc.to(dev), venc.to(dev)).cpu()
        torch.testing.assert_close(raw_e, raw, rtol=5e-5, atol=2e-3)


def test_mlp_transpose_detecting(ops, dev):
    """Asymmetric single-weight probes: every (layer, out, in) position of the packed stream is addressed right."""
    import aon_amd.synthetic as syn

    rng = np.random.Generator(np.random.PCG64(11))
    layout = syn.vanilla_mlp_layout()
    sd = syn.make_mlp_state(rng, layout, 1.0)
    # make everything positive and small so ReLUs stay open and the network is (almost) linear, then compare
    sd = {k: (v.abs() * 0.05 if k.endswith("weight") else v.abs() * 0.01) for k, v in sd.items()}
    sd = {"m." + k: v for k, v in sd.items()}
    n, S = 5, 65
    enc = torch.rand(n, S, 63, generator=torch.Generator().manual_seed(5))
    venc = torch.rand(n, 27, generator=torch.Generator().manual_seed(6))
    rgb_o, sig_o = orc.nerf_mlp(sd, "m.", enc, venc)
    pk = ops.pack_vanilla_mlp({k[2:]: v.to(dev) for k, v in sd.items()})
    raw = ops.mlp_fwd_enc(pk, enc.to(dev), venc.to(dev)).cpu()
    torch.testing.assert_close(raw[..., :3], rgb_o, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(raw[..., 3:], sig_o, rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------ R8
def test_volumetric_rendering(ops, dev, golden):
    for name, variants in (("g5_volumetric_rendering", (0, 1)), ("g5b_volumetric_rendering_193", (None,))):
        g = golden(name)
        for wb in variants:
            sfx = "" if wb is None else f"_wb{wb}"
            white = True if wb is None else bool(wb)
            cr, acc, w, dep = ops.volumetric_rendering(g["rgb"].to(dev), g["density"].to(dev), g["t_vals"].to(dev),
                                                       g["dirs"].to(dev), white)
            torch.testing.assert_close(cr.cpu(), g["comp_rgb" + sfx], rtol=0, atol=2e-6)
            torch.testing.assert_close(acc.cpu(), g["acc" + sfx], rtol=0, atol=2e-6)
            torch.testing.assert_close(w.cpu(), g["weights" + sfx], rtol=0, atol=1e-6)
            torch.testing.assert_close(dep.cpu(), g["depth" + sfx], rtol=0, atol=1e-5)


def test_composite_raw_activations(ops, dev):
    gen = torch.Generator().manual_seed(9)
    n, S = 33, 193
    raw = torch.randn(n, S, 4, generator=gen) * 3
    t = torch.sort(torch.rand(n, S, generator=gen) * 4 + 2, dim=-1).values
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)
    cr_o, acc_o, w_o, dep_o = orc.volumetric_rendering(torch.sigmoid(raw[..., :3]), torch.relu(raw[..., 3:]), t, d, True)
    cr, acc, w, dep = ops.composite_raw(raw.to(dev), t.to(dev), d.to(dev), True, ops.ACT_VANILLA)
    torch.testing.assert_close(cr.cpu(), cr_o, rtol=0, atol=2e-6)
    torch.testing.assert_close(w.cpu(), w_o, rtol=0, atol=1e-6)
    torch.testing.assert_close(dep.cpu(), dep_o, rtol=0, atol=1e-5)
    # articulated activations (model_autodecoder.py:321-323)
    rgb_a = torch.sigmoid(raw[..., :3]) * (1 + 2 * 0.001) - 0.001
    sig_a = torch.nn.functional.softplus(raw[..., 3:] - 1.0)
    cr_o, acc_o, w_o, dep_o = orc.volumetric_rendering(rgb_a, sig_a, t, d, False)
    cr, acc, w, dep = ops.composite_raw(raw.to(dev), t.to(dev), d.to(dev), False, ops.ACT_ARTICULATED)
    torch.testing.assert_close(cr.cpu(), cr_o, rtol=0, atol=2e-6)
    torch.testing.assert_close(acc.cpu(), acc_o, rtol=0, atol=2e-6)
    # NaN density -> depth = +inf (helper.py:182)
    raw[0, 5, 3] = float("nan")
    _, _, _, dep = ops.composite_raw(raw.to(dev), t.to(dev), d.to(dev), True, ops.ACT_NONE)
    assert torch.isinf(dep[0]).item() and dep[0].item() > 0


# ------------------------------------------------------------------ R6/R7
def _oracle_cdf(bins, weights):
    """fp64 evaluation of the reference's padded pdf -> 64-entry cdf (helper.py:206-222)."""
    w = weights.double()
    ws = w.sum(-1, keepdim=True)
    pad = torch.clamp(1e-5 - ws, min=0)
    w = w + pad / w.shape[-1]
    pdf = w / (ws + pad)
    cdf = torch.clamp(torch.cumsum(pdf[..., :-1], -1), max=1.0)
    return torch.cat([torch.zeros_like(cdf[..., :1]), cdf, torch.ones_like(cdf[..., :1])], -1)


def _cdf_of(samples, bins, cdf):
    """F(s): the piecewise-linear CDF evaluated at the drawn positions (fp64).  Inverse-CDF sampling promises
    F(sample) = u; inside a zero-probability (flat) zone every position has the same F, so this check is immune to
    the one legitimate ambiguity of the reference's selection rule."""
    b = bins.double()
    s = samples.double().clamp(b[..., :1], b[..., -1:])
    idx = (torch.searchsorted(b, s.contiguous(), right=True) - 1).clamp(0, 62)
    b0, b1 = torch.gather(b, -1, idx), torch.gather(b, -1, idx + 1)
    c0, c1 = torch.gather(cdf, -1, idx), torch.gather(cdf, -1, idx + 1)
    frac = torch.where(b1 > b0, (s - b0) / (b1 - b0), torch.zeros_like(s))
    return c0 + frac * (c1 - c0)


def _check_draws(samples, ref, bins, weights, u):
    """(a) BIT-EXACT against the reference's own draws: the kernel reproduces torch's CPU arithmetic of helper.py:203-243
    operation for operation -- the 63-term weight sum in ATen's vectorised association, the cumsum with its double running
    sum rounded per prefix, correctly rounded divisions, the mask/max/min selection as a right-bisect;
    (b) the probability-space invariant F(sample) = u, as well as the reference's own draws satisfy it."""
    assert torch.equal(samples, ref), f"{int((samples != ref).sum())} of {samples.numel()} draws differ, max {float((samples - ref).abs().max()):.3e}"
    cdf = _oracle_cdf(bins, weights)
    F, Fr = _cdf_of(samples, bins, cdf), _cdf_of(ref, bins, cdf)
    uu = u.double().expand_as(F)
    assert (F - uu).abs().max().item() <= 1e-6 + (Fr - uu).abs().max().item()   # never further from u than the reference's own fp32 draws


def test_inverse_cdf(ops, dev, golden):
    g = golden("g6_pdf")
    bins, w = g["bins"], g["weights"]
    s = ops.sorted_piecewise_constant_pdf(bins.to(dev), w.to(dev)).cpu()
    _check_draws(s, g["samples_det"], bins, w, orc.deterministic_u(128))
    s = ops.sorted_piecewise_constant_pdf(bins.to(dev), w.to(dev), u=g["u"].to(dev)).cpu()
    _check_draws(s, g["samples_rnd"], bins, w, g["u"])
    # u = 1.0 (the last deterministic draw rounds to exactly 1) collapses onto the last bin edge
    s_det = ops.sorted_piecewise_constant_pdf(bins.to(dev), w.to(dev)).cpu()
    assert torch.equal(s_det[:, -1], bins[:, -1])


def test_inverse_cdf_bit_exact_on_many_rows(ops, dev):
    """20,000 seeded rows (peaky, flat, tiny, all-zero and single-bin weights; random and gridded u) against the oracle,
    which test_oracle_golden.py holds bit-exact to the reference: every draw equal, no tolerance."""
    gen = torch.Generator().manual_seed(11)
    n = 20000
    w = torch.rand(n, 63, generator=gen) ** 8
    w[:200] = 0.0                                           # padding branch: uniform pdf
    w[200:400] *= 1e-7                                      # sum below the 1e-5 floor: padding + data
    w[400:600, 10:50] = 0.0                                 # flat cdf zones
    w[600:800] = 0.0
    w[600:800, 31] = 1.0                                    # one bin
    w[800:1000] = torch.rand(200, 63, generator=gen) * 30   # large unnormalised weights
    # weights over 17 decades: prefixes that are NOT exact in a double (the kernel's tree-scan fast path must then prove itself
    # far from a float rounding boundary, or fall back to the index-order chain)
    w[1000:4000] = torch.rand(3000, 63, generator=gen) * torch.exp(-40 * torch.rand(3000, 63, generator=gen))
    w[4000:4200, 20:] *= 1e-30                              # a surface followed by vanishing weights
    t = torch.sort(torch.rand(n, 65, generator=gen) * 4 + 2, -1).values
    bins = 0.5 * (t[:, 1:] + t[:, :-1])
    u = torch.rand(n, 128, generator=gen)
    for uu, rnd in ((None, False), (u, True)):
        got = ops.sorted_piecewise_constant_pdf(bins.to(dev), w.to(dev), u=None if uu is None else uu.to(dev)).cpu()
        want = orc.sorted_piecewise_constant_pdf(bins, w, 128, rnd, uu)
        assert torch.equal(got, want), f"{int((got != want).sum())} of {got.numel()} draws differ"
        tf = ops.sample_pdf_t(t.to(dev), w.to(dev), u=None if uu is None else uu.to(dev), bins=bins.to(dev)).cpu()
        assert torch.equal(tf, torch.sort(torch.cat([t, want], -1), -1).values)
    # sorted random draws take the rank-merge fast path; bins unrelated to t_coarse make its rank walk start from a wrong guess
    us = torch.sort(u, -1).values
    want = orc.sorted_piecewise_constant_pdf(bins, w, 128, True, us)
    tf = ops.sample_pdf_t(t.to(dev), w.to(dev), u=us.to(dev), bins=bins.to(dev)).cpu()
    assert torch.equal(tf, torch.sort(torch.cat([t, want], -1), -1).values)
    bins2 = torch.sort(torch.rand(n, 64, generator=gen) * 6 + 1, -1).values
    want2 = orc.sorted_piecewise_constant_pdf(bins2, w, 128, True, us)
    tf2 = ops.sample_pdf_t(t.to(dev), w.to(dev), u=us.to(dev), bins=bins2.to(dev)).cpu()
    assert torch.equal(tf2, torch.sort(torch.cat([t, want2], -1), -1).values)


def test_sample_pdf_merge(ops, dev, golden):
    g = golden("g7_sample_pdf")
    t, w = g["t_vals"].to(dev), g["weights"].to(dev)
    for u, tag in ((None, "det"), (g["u"].to(dev), "rnd")):
        tf = ops.sample_pdf_t(t, w, u=u)
        assert (tf[:, 1:] >= tf[:, :-1]).all()
        assert torch.equal(tf.cpu(), g[f"t_fine_{tag}"])          # the reference's own sample_pdf output, bit for bit
        mids = 0.5 * (g["t_vals"][..., 1:] + g["t_vals"][..., :-1])
        smp = ops.sorted_piecewise_constant_pdf(mids.to(dev), w, u=u)
        assert torch.equal(tf, torch.sort(torch.cat([t, smp], -1), -1).values)
        _check_draws(smp.cpu(), orc.sorted_piecewise_constant_pdf(mids, g["weights"], 128, u is not None, None if u is None else g["u"]),
                     mids, g["weights"], orc.deterministic_u(128) if u is None else g["u"])
    # full coarse-weights form (weights[...,1:-1] taken by stride) equals the dense (n,63) form
    wfull = torch.cat([torch.rand(64, 1), g["weights"], torch.rand(64, 1)], -1)
    assert torch.equal(ops.sample_pdf_t(t, wfull.to(dev)), ops.sample_pdf_t(t, w))


# ------------------------------------------------------------------ R9 end to end
def _robust_rays(aux, margin=2e-2):
    """Rays whose far-plane (1e10-long interval) density is robustly signed: alpha_last in {0,1} is decided by the
    sign of raw sigma at the last sample, a genuine discontinuity of the reference's math."""
    ok = torch.ones(aux[0]["raw_sigma"].shape[0], dtype=torch.bool)
    for a in aux:
        ok &= a["raw_sigma"][:, -1, 0].abs() > margin
    return ok


def _psnr(a, b):
    return -10.0 * math.log10(max(torch.mean((a - b) ** 2).item(), 1e-20))


def test_fused_coarse_level_equals_the_stage_kernels(ops, dev, nerf_sd):
    """aon_composite_pdf (coarse compositing + inverse CDF + merge in one kernel, model.py:160-173) gives the same BITS as
    aon_composite followed by aon_sample_pdf -- both activations, shared and per-ray u, degenerate rows -- and the whole-path
    entry points render the same bits with the fusion on and off."""
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model import NeRF

    gen = torch.Generator().manual_seed(21)
    n = 777
    raw = torch.randn(n, 65, 4, generator=gen) * 3
    raw[:40, :, 3] = -5.0                      # empty rays: all-zero weights -> padded pdf
    raw[40:80, 20, 3] = 80.0                   # one opaque shell
    raw[80:90, :, 3] = 50.0                    # opaque from the first sample
    raw[90:100, 64, 3] = -1.0                  # far sample transparent
    t = torch.sort(torch.rand(n, 65, generator=gen) * 4 + 2, dim=-1).values
    t[100:110] = torch.linspace(2.0, 6.0, 65)
    t[110:115, 10:14] = t[110:115, 10:11]      # repeated t: zero-length intervals / equal bins
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1) * (0.5 + torch.rand(n, 1, generator=gen))
    u_rand = torch.rand(n, 128, generator=gen)
    raw, t, d, u_rand = raw.to(dev), t.to(dev), d.to(dev), u_rand.to(dev)
    for act, white in ((ops.ACT_VANILLA, True), (ops.ACT_ARTICULATED, False)):
        for u in (None, u_rand):
            cr, acc, w, dep = ops.composite_raw(raw, t, d, white, act)
            tf = ops.sample_pdf_t(t, w, u)
            cr2, acc2, w2, dep2, tf2 = ops.composite_pdf(raw, t, d, white, act, u, want_weights=True)
            assert torch.equal(cr, cr2) and torch.equal(acc, acc2) and torch.equal(dep, dep2) and torch.equal(w, w2)
            assert torch.equal(tf, tf2), f"{int((tf != tf2).sum())} of {tf.numel()} fine t differ"
            cr3, _, w3, _, tf3 = ops.composite_pdf(raw, t, d, white, act, u)   # weights never written
            assert w3 is None and torch.equal(cr, cr3) and torch.equal(tf, tf3)
    # whole path: fusion on (default) == fusion off, deterministic and randomized
    model = NeRF().to(dev)
    model.load_state_dict(nerf_sd)
    rays = {k: v.to(dev) for k, v in syn.random_rays(1500, seed=3).items()}
    try:
        for randomized in (False, True):
            outs = []
            for on in (True, False):
                ops.set_coarse_fusion(on)
                torch.manual_seed(5)
                with torch.no_grad():
                    outs.append(model(rays, randomized, True, 2.0, 6.0))
            for lvl in (0, 1):
                for a, b in zip(outs[0][lvl], outs[1][lvl]):
                    assert torch.equal(a, b)
    finally:
        ops.set_coarse_fusion(True)


def test_compositing_at_every_block_shape(ops, dev):
    """Sample counts around the 64-sample block boundaries (the last sample is evaluated outside the blocks; 65 and 193 are
    compiled as constants, the rest take the run-time path), packed and unpacked inputs, against the oracle."""
    gen = torch.Generator().manual_seed(31)
    for S in (1, 2, 3, 63, 64, 65, 66, 128, 129, 130, 193, 194):
        n = 37
        raw = torch.randn(n, S, 4, generator=gen) * 3
        t = torch.sort(torch.rand(n, S, generator=gen) * 4 + 2, dim=-1).values
        d = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)
        rgb, sig = torch.sigmoid(raw[..., :3]), torch.relu(raw[..., 3:])
        cr_o, acc_o, w_o, dep_o = orc.volumetric_rendering(rgb, sig, t, d, True)
        for packed in (True, False):
            if packed:
                cr, acc, w, dep = ops.composite_raw(raw.to(dev), t.to(dev), d.to(dev), True, ops.ACT_VANILLA)
            else:
                cr, acc, w, dep = ops.volumetric_rendering(rgb.to(dev), sig.to(dev), t.to(dev), d.to(dev), True)
            torch.testing.assert_close(cr.cpu(), cr_o, rtol=0, atol=2e-6)
            torch.testing.assert_close(acc.cpu(), acc_o, rtol=0, atol=2e-6)
            torch.testing.assert_close(w.cpu(), w_o, rtol=0, atol=1e-6)
            torch.testing.assert_close(dep.cpu(), dep_o, rtol=0, atol=1e-5)


def test_render_is_independent_of_the_internal_chunking(ops, dev, nerf_sd):
    """aon_render_fwd walks the rays in chunks as large as the workspace admits (a whole frame with the default workspace): a
    workspace for 1,000 rays renders 2,500 rays in three chunks -- same bits as one chunk, deterministic and randomized."""
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model import NeRF

    model = NeRF().to(dev)
    model.load_state_dict(nerf_sd)
    rays = {k: v.to(dev) for k, v in syn.random_rays(2500, seed=11).items()}
    keep = ops.MAX_CHUNK_RAYS
    try:
        outs = []
        for chunk in (keep, 1000):
            ops.MAX_CHUNK_RAYS = chunk
            ops._WS_CACHE.clear()
            torch.manual_seed(3)
            with torch.no_grad():
                outs.append((model(rays, False, True, 2.0, 6.0), model(rays, True, True, 2.0, 6.0)))
        for a, b in zip(outs[0], outs[1]):
            for lvl in (0, 1):
                for x, y in zip(a[lvl], b[lvl]):
                    assert torch.equal(x, y)
    finally:
        ops.MAX_CHUNK_RAYS = keep
        ops._WS_CACHE.clear()


def test_wave_reductions_of_the_compositing_kernel(ops, dev):
    """The four-at-a-time lane-swap reduction (wave_sum4: v_permlane32_swap / v_permlane16_swap) routes every value to its own
    total: channels with very different magnitudes must not leak into each other, and the sums must match fp64 sums of the
    kernel's own weights."""
    gen = torch.Generator().manual_seed(4)
    n, S = 64, 65
    rgb = torch.rand(n, S, 3, generator=gen) * torch.tensor([1.0, 1e3, 1e-3])
    sig = torch.rand(n, S, 1, generator=gen) * 0.5
    t = torch.sort(torch.rand(n, S, generator=gen) * 4 + 2, dim=-1).values
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)
    cr, acc, w, dep = ops.volumetric_rendering(rgb.to(dev), sig.to(dev), t.to(dev), d.to(dev), False)
    w64 = w.double().cpu()
    torch.testing.assert_close(cr.double().cpu(), (w64[..., None] * rgb.double()).sum(1), rtol=2e-6, atol=1e-9)
    torch.testing.assert_close(acc.double().cpu(), w64.sum(1), rtol=2e-6, atol=1e-9)
    torch.testing.assert_close(dep.double().cpu(), (w64 * t.double()).sum(1), rtol=2e-6, atol=1e-9)


@pytest.mark.parametrize("white", [True, False])
def test_nerf_forward_vs_oracle_and_golden(dev, golden, nerf_sd, white):
    from aon_amd.models.vanilla_nerf.model import NeRF

    g = golden("g8_nerf_forward")
    model = NeRF().to(dev)
    model.load_state_dict(nerf_sd)
    rays_cpu = {k: g[k] for k in ("rays_o", "rays_d", "viewdirs")}
    rays = {k: v.to(dev) for k, v in rays_cpu.items()}
    rays["target"] = torch.zeros_like(rays["rays_o"])  # extra keys are ignored like in the reference
    with torch.no_grad():
        out = model(rays, False, white, g["near"], g["far"])
    ref, aux = orc.nerf_forward(nerf_sd, rays_cpu, False, white, g["near"], g["far"], return_aux=True)
    ok = _robust_rays(aux)
    assert ok.double().mean() > 0.8
    tag = "det" if white else "det_nowb"
    for lvl, name in ((0, "coarse"), (1, "fine")):
        rgb, acc, depth = (x.cpu() for x in out[lvl])
        # stated end-to-end tolerance on ALL rays
        assert _psnr(rgb, ref[lvl][0]) >= 70.0
        assert frac_within(rgb, ref[lvl][0], 1e-3) >= 0.999 or ok.double().mean() < 0.999
        # tight check on robustly-signed rays, against the oracle and against the reference's own outputs
        # (round 2, with the inverse CDF bit-exact: measured 9.5e-7 rgb / 3.4e-5 fine depth on this sharp x30 field; the bounds
        # were 2e-4 / 2e-3 in round 1.  The smooth fixture of test_hip_smooth.py holds 2e-6 / 1e-5 on every ray.)
        torch.testing.assert_close(rgb[ok], ref[lvl][0][ok], rtol=0, atol=1e-5)
        torch.testing.assert_close(acc[ok], ref[lvl][1][ok], rtol=0, atol=1e-5)
        torch.testing.assert_close(depth[ok], ref[lvl][2][ok], rtol=0, atol=2e-4)
        torch.testing.assert_close(rgb[ok], g[f"{tag}_{name}_rgb"][ok], rtol=0, atol=1e-5)
        torch.testing.assert_close(depth[ok], g[f"{tag}_{name}_depth"][ok], rtol=0, atol=2e-4)
        # a ray whose far raw sigma sits inside the margin must still land on one of the two legal values of the reference's
        # step function at the far plane (helper.py:163): here, on the reference's own output or differ only by that step
        if (~ok).any():
            bad = (rgb[~ok] - g[f"{tag}_{name}_rgb"][~ok]).abs().amax(-1)
            assert ((bad <= 1e-5) | (bad >= 1e-3)).all(), bad


RND_ATOL = 1e-5   # measured on MI355X (round 3): see the printed maxima


def test_nerf_forward_randomized(dev, golden, nerf_sd):
    from aon_amd.models.vanilla_nerf.model import NeRF

    g = golden("g8_nerf_forward")
    model = NeRF().to(dev)
    model.load_state_dict(nerf_sd)
    rays_cpu = {k: g[k] for k in ("rays_o", "rays_d", "viewdirs")}
    rays = {k: v.to(dev) for k, v in rays_cpu.items()}
    with torch.no_grad():
        out = model(rays, True, True, g["near"], g["far"], t_rand=g["t_rand"].to(dev), u=g["u"].to(dev))
    ref, aux = orc.nerf_forward(nerf_sd, rays_cpu, True, True, g["near"], g["far"], t_rand=g["t_rand"], u=g["u"], return_aux=True)
    ok = _robust_rays(aux)
    for lvl, name in ((0, "coarse"), (1, "fine")):
        rgb = out[lvl][0].cpu()
        assert _psnr(rgb, ref[lvl][0]) >= 70.0
        print(f"randomized {name}: max |rgb - oracle| on robust rays {(rgb[ok] - ref[lvl][0][ok]).abs().max():.2e}, "
              f"vs the reference's own output {(rgb[ok] - g[f'rnd_{name}_rgb'][ok]).abs().max():.2e}")
        # round 3: the bound follows the measured level like the deterministic case (round 2 still carried round 1's 2e-4)
        torch.testing.assert_close(rgb[ok], ref[lvl][0][ok], rtol=0, atol=RND_ATOL)
        torch.testing.assert_close(rgb[ok], g[f"rnd_{name}_rgb"][ok], rtol=0, atol=RND_ATOL)
    # without supplied draws the module draws its own: different result, still a valid render
    with torch.no_grad():
        out2 = model(rays, True, True, g["near"], g["far"])
    assert torch.isfinite(out2[1][0]).all() and not torch.equal(out2[1][0], out[1][0])


def test_coarse_only_and_edge_sizes(ops, dev, nerf_sd, packed):
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model import NeRF

    model1 = NeRF(num_levels=1).to(dev)
    model1.load_state_dict(nerf_sd)
    model2 = NeRF().to(dev)
    model2.load_state_dict(nerf_sd)
    for n in (1, 2, 127, 129, 1000):
        rays = {k: v.to(dev) for k, v in syn.random_rays(n, seed=n).items()}
        with torch.no_grad():
            o1 = model1(rays, False, True, 2.0, 6.0)
            o2 = model2(rays, False, True, 2.0, 6.0)
        assert len(o1) == 1 and len(o2) == 2
        assert torch.equal(o1[0][0], o2[0][0]) and o2[1][0].shape == (n, 3) and o2[1][2].shape == (n,)
    # empty batch
    rays = {k: torch.empty(0, 3, device=dev) for k in ("rays_o", "rays_d", "viewdirs")}
    with torch.no_grad():
        o = model2(rays, False, True, 2.0, 6.0)
    assert o[1][0].shape == (0, 3)
    # CPU tensors are rejected loudly (no host fallback)
    with pytest.raises(RuntimeError):
        ops.pos_enc(torch.zeros(4, 3), 0, 10)
    # gradient mode runs the HIP training path (tests/test_hip_training.py): outputs carry a graph to the parameters
    out = model2({k: v.to(dev) for k, v in syn.random_rays(4, seed=0).items()}, False, True, 2.0, 6.0)
    assert out[1][0].requires_grad and out[1][0].grad_fn is not None


def test_config1_coarse_only_frame_vs_oracle(ops, dev, nerf_sd):
    """BASELINE config 1 -- 320x240, `num_levels=1` (65 coarse evaluations per ray, model.py:149), the reference's own CPU-runnable
    case -- against the oracle run in the same mode (round 2 only compared the HIP path with itself): 1e-5 rgb / acc on a strided
    sample of the frame (far-plane-robust rays), plus the size-independent properties on the whole frame."""
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model import NeRF

    H, W = 240, 320
    model = NeRF(num_levels=1).to(dev)
    model.load_state_dict(nerf_sd)
    ro, vd = ops.raygen(syn.look_at_pose(), H, W, syn.focal_from_fovy(H), device=dev)
    rays = {"rays_o": ro, "rays_d": vd, "viewdirs": vd}
    with torch.no_grad():
        full = model(rays, False, True, 2.0, 6.0)
        again = model(rays, False, True, 2.0, 6.0)
        sl = slice(30_000, 33_840)   # one reference-sized chunk (opt.py:103)
        part = model({k: v[sl] for k, v in rays.items()}, False, True, 2.0, 6.0)
        nowb = model({k: v[sl] for k, v in rays.items()}, False, False, 2.0, 6.0)
    assert len(full) == 1 and full[0][0].shape == (H * W, 3)
    rgb, acc, depth = full[0]
    assert torch.equal(rgb, again[0][0]) and torch.equal(depth, again[0][2])                      # determinism
    assert torch.equal(part[0][0], rgb[sl]) and torch.equal(part[0][1], acc[sl]) and torch.equal(part[0][2], depth[sl])   # chunk invariance
    assert torch.isfinite(rgb).all() and acc.min().item() >= 0.0 and acc.max().item() <= 1.0 + 1e-5
    torch.testing.assert_close(part[0][0], nowb[0][0] + (1.0 - nowb[0][1])[:, None], rtol=0, atol=1e-6)   # helper.py:187-188
    pick = torch.arange(0, H * W, 37)
    rays_cpu = {k: v[pick.to(dev)].cpu() for k, v in rays.items()}
    ref, aux = orc.nerf_forward(nerf_sd, rays_cpu, False, True, 2.0, 6.0, num_levels=1, return_aux=True)
    assert len(ref) == 1
    ok = _robust_rays(aux)
    assert ok.double().mean() > 0.8
    got = [x[pick.to(dev)].cpu() for x in full[0]]
    print(f"config 1: {int(ok.sum())}/{ok.numel()} robust rays, max |rgb - oracle| {(got[0][ok] - ref[0][0][ok]).abs().max():.2e}, "
          f"depth {(got[2][ok] - ref[0][2][ok]).abs().max():.2e}")
    torch.testing.assert_close(got[0][ok], ref[0][0][ok], rtol=0, atol=1e-5)
    torch.testing.assert_close(got[1][ok], ref[0][1][ok], rtol=0, atol=1e-5)
    torch.testing.assert_close(got[2][ok], ref[0][2][ok], rtol=0, atol=2e-4)
    assert _psnr(got[0], ref[0][0]) >= 70.0


def test_full_frame_properties(ops, dev, nerf_sd, golden):
    """BASELINE config 2 size (640x480, 65+193): size-independent properties -- chunking invariance (a contiguous
    ray range renders to the same bits alone or inside the frame), determinism, range, white-background identity."""
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model import NeRF

    H, W = 480, 640
    model = NeRF().to(dev)
    model.load_state_dict(nerf_sd)
    ro, vd = ops.raygen(syn.look_at_pose(), H, W, syn.focal_from_fovy(H), device=dev)
    rays = {"rays_o": ro, "rays_d": vd, "viewdirs": vd}
    with torch.no_grad():
        full = model(rays, False, True, 2.0, 6.0)
        again = model(rays, False, True, 2.0, 6.0)
        sl = slice(100_000, 103_840)  # one reference-sized chunk (opt.py:103) from the middle of the frame
        part = model({k: v[sl] for k, v in rays.items()}, False, True, 2.0, 6.0)
        nowb = model({k: v[sl] for k, v in rays.items()}, False, False, 2.0, 6.0)
    for lvl in (0, 1):
        rgb, acc, depth = full[lvl]
        assert torch.equal(rgb, again[lvl][0]) and torch.equal(depth, again[lvl][2])
        assert torch.equal(part[lvl][0], rgb[sl]) and torch.equal(part[lvl][1], acc[sl]) and torch.equal(part[lvl][2], depth[sl])
        assert torch.isfinite(rgb).all() and torch.isfinite(acc).all()
        assert acc.min().item() >= 0.0 and acc.max().item() <= 1.0 + 1e-5
        assert rgb.min().item() >= -1e-5 and rgb.max().item() <= 1.0 + 1e-4
        d_ok = depth[torch.isfinite(depth)]
        assert d_ok.min().item() >= 0.0 and d_ok.max().item() <= 6.0 + 1e-3
        # white_bkgd only adds (1 - acc) (helper.py:187-188)
        torch.testing.assert_close(part[lvl][0], nowb[lvl][0] + (1.0 - nowb[lvl][1])[:, None], rtol=0, atol=1e-6)
    # Parity AT THIS SIZE against the REFERENCE's own outputs (round 6, G19): the 4,209 strided rays of this frame (every 73rd pixel) that
    # tests/golden/make_golden_full.py put through the real `NeRF.forward` in fp32 and in fp64 -- both levels, every output.  (Rounds 3-5
    # evaluated the oracle live on the GPU host here, in fp32 and fp64: one link longer and a minute of host time.)  The fixture's rays
    # are the reference's get_rays output for those pixels; they are rendered as their own batch (a contiguous range renders to the same
    # bits alone or inside the frame: asserted above).  Far-plane-robust rays only (helper.py:163; margin recorded in the fixture).
    # Bar per ray: 1e-5 rgb / acc, 2e-4 depth -- or, where the REFERENCE ARITHMETIC ITSELF is less certain than that on this sharp
    # x30 field, 3 x the distance between the reference's fp32 and fp64 evaluations of that ray (a 1e-7 difference of a coarse weight
    # moves fine-level samples across thin dense shells; measured round 4: 1 of 4,169 rays at 1.26e-5 rgb / 6.1e-4 depth, level 1).
    # The number of rays that needed the wider bar is printed and bounded (<= 1 %).
    g = golden("g19_config2_frame")
    assert (g["H"], g["W"]) == (H, W) and g["pick"].numel() >= 4096
    grays = {k: g[k].to(dev) for k in ("rays_o", "rays_d", "viewdirs")}
    # the fixture's rays ARE this frame's: the device ray generator gives the same origins (bit-equal) and directions (2e-7)
    assert torch.equal(grays["rays_o"], ro[g["pick"].to(dev)])
    torch.testing.assert_close(grays["rays_d"], vd[g["pick"].to(dev)], rtol=0, atol=2e-7)
    with torch.no_grad():
        sub = model(grays, False, True, 2.0, 6.0)
    ok = g["margin"] > 2e-2
    assert ok.double().mean() > 0.8
    for lvl, lname in ((0, "coarse"), (1, "fine")):
        got = [x.cpu() for x in sub[lvl]]
        widened = 0
        for i, (name, bar) in enumerate((("rgb", 1e-5), ("acc", 1e-5), ("depth", 2e-4))):
            ref = g[f"ref_{lname}_{name}"]
            err = (got[i] - ref).abs()
            spread = g[f"spread_{lname}_{name}"]
            if err.dim() > 1:
                err = err.max(dim=-1).values
            widened = max(widened, int((ok & (err > bar)).sum()))
            bad = ok & (err > torch.clamp(3.0 * spread, min=bar))
            print(f"config 2 level {lvl} {name}: {int(ok.sum())}/{ok.numel()} robust rays, max |hip - reference| {err[ok].max():.2e} "
                  f"(reference fp32 vs fp64 on the same ray set: {spread[ok].max():.2e}), rays above {bar:g}: {int((ok & (err > bar)).sum())}")
            assert not bad.any(), (lvl, name, err[bad].max().item(), spread[bad].max().item())
        assert widened <= 0.01 * int(ok.sum()), (lvl, widened)
        assert _psnr(got[0], g[f"ref_{lname}_rgb"]) >= 70.0


def test_volumetric_rendering_nocs_branch(dev, golden):
    """helper.volumetric_rendering(..., nocs=...) (helper.py:191-193): (comp_rgb, acc, weights, comp_nocs) against the reference's
    formula on G5's inputs (the reference function itself is run in tests/golden only for the nocs=None form)."""
    from aon_amd.models.vanilla_nerf import helper

    g = golden("g5_volumetric_rendering")
    gen = torch.Generator().manual_seed(3)
    nocs = torch.rand(g["rgb"].shape, generator=gen)
    out = helper.volumetric_rendering(g["rgb"].to(dev), g["density"].to(dev), g["t_vals"].to(dev), g["dirs"].to(dev), True, nocs=nocs.to(dev))
    assert len(out) == 4
    torch.testing.assert_close(out[0].cpu(), g["comp_rgb_wb1"], rtol=0, atol=2e-6)
    torch.testing.assert_close(out[2].cpu(), g["weights_wb1"], rtol=0, atol=2e-6)
    want = (g["weights_wb1"][..., None] * nocs).sum(dim=-2)
    torch.testing.assert_close(out[3].cpu(), want, rtol=0, atol=2e-6)
