"""GPU parity of the backward path against torch.autograd run on the CPU oracle.

Tolerances: composite backward on identical inputs 2e-5 relative to the largest gradient of the ray batch; stored
activation planes 2e-5; parameter gradients of the whole path: relative L2 error <= 2e-3 per tensor (fp32, different
summation order over ~10^4-10^5 samples), 1e-2 for fine-level tensors on the sharp density x30 field (ill-conditioned,
see test_hip_articulated)."""
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
    return (torch.linalg.norm((a - b).double()) / torch.linalg.norm(b.double()).clamp_min(1e-30)).item()


@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("white", [False, True])
def test_composite_backward(dev, act, white):
    from aon_amd import ops

    gen = torch.Generator().manual_seed(100 + act)
    for n, S in ((37, 65), (21, 193)):
        raw = (torch.randn(n, S, 4, generator=gen) * 2).requires_grad_(True)
        if act == 0:
            raw = torch.cat([torch.rand(n, S, 3, generator=gen), torch.relu(torch.randn(n, S, 1, generator=gen) * 3)], -1).requires_grad_(True)
        t = torch.sort(torch.rand(n, S, generator=gen) * 4 + 2, dim=-1).values
        d = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1)
        if act == 1:
            rgb, sig = torch.sigmoid(raw[..., :3]), torch.relu(raw[..., 3:])
        elif act == 2:
            rgb, sig = torch.sigmoid(raw[..., :3]) * 1.002 - 0.001, torch.nn.functional.softplus(raw[..., 3:] - 1.0)
        else:
            rgb, sig = raw[..., :3], raw[..., 3:]
        comp, acc, w, depth = orc.volumetric_rendering(rgb, sig, t, d, white)
        g_rgb, g_acc, g_depth = torch.randn(n, 3, generator=gen), torch.randn(n, generator=gen), torch.randn(n, generator=gen)
        ((comp * g_rgb).sum() + (acc * g_acc).sum() + (depth * g_depth).sum()).backward()
        Np = ops.padded_samples(n * S)
        d_raw = ops.composite_bwd(raw.detach().to(dev), t.to(dev), d.to(dev), g_rgb.to(dev), g_acc.to(dev), g_depth.to(dev), white, act, Np)
        assert d_raw.shape == (Np, 4) and (d_raw[n * S:] == 0).all()
        got = d_raw[: n * S].reshape(n, S, 4).cpu()
        scale = raw.grad.abs().amax(dim=(1, 2), keepdim=True).clamp_min(1e-12)
        assert ((got - raw.grad).abs() / scale).max().item() < 2e-5
        # rgb-only gradient (the training loss, model.py:271-273) with null g_acc / g_depth
        raw.grad = None
        comp2 = orc.volumetric_rendering(rgb.detach() if False else (torch.sigmoid(raw[..., :3]) if act == 1 else (torch.sigmoid(raw[..., :3]) * 1.002 - 0.001 if act == 2 else raw[..., :3])),
                                         (torch.relu(raw[..., 3:]) if act == 1 else (torch.nn.functional.softplus(raw[..., 3:] - 1.0) if act == 2 else raw[..., 3:])),
                                         t, d, white)[0]
        (comp2 * g_rgb).sum().backward()
        d_raw = ops.composite_bwd(raw.detach().to(dev), t.to(dev), d.to(dev), g_rgb.to(dev), None, None, white, act, Np)
        got = d_raw[: n * S].reshape(n, S, 4).cpu()
        scale = raw.grad.abs().amax(dim=(1, 2), keepdim=True).clamp_min(1e-12)
        assert ((got - raw.grad).abs() / scale).max().item() < 2e-5


def _layer_activations(sd, prefix, enc, venc):
    """Oracle-side per-layer activations of model.py:95-120 (post-ReLU), for the plane check."""
    import torch.nn.functional as F

    n, s, _ = enc.shape
    x = enc.reshape(-1, 63)
    inputs, acts = x, []
    for idx in range(8):
        x = F.relu(F.linear(x, sd[f"{prefix}pts_linears.{idx}.weight"], sd[f"{prefix}pts_linears.{idx}.bias"]))
        acts.append(x)
        if idx == 4:
            x = torch.cat([x, inputs], -1)
    bott = F.linear(acts[7], sd[f"{prefix}bottleneck_layer.weight"], sd[f"{prefix}bottleneck_layer.bias"])
    cond = venc[:, None, :].expand(n, s, 27).reshape(-1, 27)
    hv = F.relu(F.linear(torch.cat([bott, cond], -1), sd[f"{prefix}views_linear.0.weight"], sd[f"{prefix}views_linear.0.bias"]))
    return acts, bott, hv


def test_forward_train_planes(dev, nerf_sd, fold_form):
    import aon_amd.synthetic as syn
    from aon_amd import ops

    params = {k[len("fine_mlp."):]: v.to(dev) for k, v in nerf_sd.items() if k.startswith("fine_mlp.")}
    packed = ops.pack_vanilla_mlp(params)
    n, S = 9, 65  # 585 samples -> Np = 640
    rays = syn.random_rays(n, seed=5)
    t = torch.sort(torch.rand(n, S, generator=torch.Generator().manual_seed(5)) * 4 + 2, dim=-1).values
    args = [rays[k].to(dev) for k in ("rays_o", "rays_d", "viewdirs")] + [t.to(dev)]
    raw, planes, masks = ops.mlp_fwd_train(packed, *args)
    assert torch.equal(raw, ops.mlp_fwd(packed, *args))  # same arithmetic as the inference kernel
    assert planes.shape == (640 // 32, 2528 // 4, 32, 4)      # step-major: [step][row / 4][sample][row % 4] (include/aon_hip.h)
    enc = orc.pos_enc(orc.cast_rays(t, rays["rays_o"], rays["rays_d"]), 0, 10)
    venc = orc.pos_enc(rays["viewdirs"], 0, 4)
    acts, bott, hv = _layer_activations(nerf_sd, "fine_mlp.", enc, venc)
    pl = ops.plane_rows_view(planes).cpu()[:, : n * S]    # (rows, samples) view of the step-major planes
    torch.testing.assert_close(pl[0:63].T, enc.reshape(-1, 63), rtol=0, atol=2.5e-7)
    for l in range(8):
        torch.testing.assert_close(pl[64 + 256 * l: 64 + 256 * (l + 1)].T, acts[l], rtol=2e-5, atol=2e-5)
    if fold_form == "literal":   # (the folded form has no bottleneck output: views_linear[0] reads H7 through W' = W_v0[:, :256] W_b)
        torch.testing.assert_close(pl[2112:2368].T, bott, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(pl[2368:2395].T, venc[:, None, :].expand(n, S, 27).reshape(-1, 27), rtol=0, atol=2.5e-7)
    torch.testing.assert_close(pl[2400:2528].T, hv, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("randomized,white,density_scale", [(False, True, 30.0), (True, False, 5.0)])
def test_training_step_gradients_vs_oracle_autograd(dev, randomized, white, density_scale):
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model import NeRF

    sd = syn.make_nerf_state_dict(seed=3, density_scale=density_scale)
    n = 150
    rays_cpu = syn.random_rays(n, seed=8)
    gen = torch.Generator().manual_seed(8)
    target = torch.rand(n, 3, generator=gen)
    t_rand, u = torch.rand(n, 65, generator=gen), torch.rand(n, 128, generator=gen)
    kw = dict(t_rand=t_rand, u=u) if randomized else {}
    # oracle: autograd through the CPU restatement (the reference's training_step loss, model.py:271-273)
    sd_o = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out_o = orc.nerf_forward(sd_o, rays_cpu, randomized, white, 2.0, 6.0, **kw)
    loss_o = orc.img2mse(out_o[0][0], target) + orc.img2mse(out_o[1][0], target)
    loss_o.backward()
    # HIP path through the drop-in module
    model = NeRF().to(dev)
    model.load_state_dict(sd)
    rays = {k: v.to(dev) for k, v in rays_cpu.items()}
    out = model(rays, randomized, white, 2.0, 6.0, **{k: v.to(dev) for k, v in kw.items()})
    loss = torch.mean((out[0][0] - target.to(dev)) ** 2) + torch.mean((out[1][0] - target.to(dev)) ** 2)
    loss.backward()
    assert abs(loss.item() - loss_o.item()) <= 2e-5 * max(1.0, abs(loss_o.item()))
    # Coarse level: identical sample positions on both sides -> the kernels themselves are compared (measured 3e-6).
    # Fine level: the 128 inverse-CDF positions inherit 1-ulp differences of the coarse weights, so the two sides evaluate
    # the network at slightly different points; that input sensitivity (not the backward kernels, which
    # test_level_backward_with_shared_samples pins at 2e-5 for S = 193) bounds the agreement at the 1e-3 level.
    errs = {}
    for name, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
        errs[name] = rel_l2(p.grad.cpu(), sd_o[name].grad)
    print({k: f"{v:.1e}" for k, v in errs.items()})
    for name, err in errs.items():
        tol = 1e-2 if name.startswith("fine_mlp") else 2e-3  # coarse: 3e-6 measured unless a far-plane sigma sign flips
        assert err <= tol, f"{name}: relative L2 gradient error {err:.3e} > {tol}"


@pytest.mark.parametrize("n,S", [(40, 193), (3, 65)])
def test_level_backward_with_shared_samples(dev, nerf_sd, n, S, fold_form):
    """One level in isolation with the SAME sample positions on both sides: forward-train -> composite -> loss ->
    composite_bwd -> bwd chain -> wgrad against autograd on the oracle.  Pins the fine-level (S = 193) kernels tightly."""
    import aon_amd.synthetic as syn
    from aon_amd import ops

    prefix = "fine_mlp."
    params = {k[len(prefix):]: v.to(dev) for k, v in nerf_sd.items() if k.startswith(prefix)}
    packed, packed_bwd = ops.pack_vanilla_mlp(params), ops.pack_vanilla_mlp_bwd(params)
    rays = syn.random_rays(n, seed=12)
    gen = torch.Generator().manual_seed(12)
    t = torch.sort(torch.rand(n, S, generator=gen) * 4 + 2, dim=-1).values
    target = torch.rand(n, 3, generator=gen)
    sd_o = {k: v.clone().requires_grad_(True) for k, v in nerf_sd.items() if k.startswith(prefix)}
    enc = orc.pos_enc(orc.cast_rays(t, rays["rays_o"], rays["rays_d"]), 0, 10)
    raw_rgb, raw_sig = orc.nerf_mlp(sd_o, prefix, enc, orc.pos_enc(rays["viewdirs"], 0, 4))
    comp = orc.volumetric_rendering(torch.sigmoid(raw_rgb), torch.relu(raw_sig), t, rays["rays_d"], True)[0]
    orc.img2mse(comp, target).backward()
    o, d, v, tt = (x.to(dev) for x in (rays["rays_o"], rays["rays_d"], rays["viewdirs"], t))
    raw, planes, masks = ops.mlp_fwd_train(packed, o, d, v, tt)
    rgb = ops.composite_raw(raw, tt, d, True, ops.ACT_VANILLA)[0]
    g_rgb = 2.0 * (rgb - target.to(dev)) / (n * 3)
    d_raw = ops.composite_bwd(raw, tt, d, g_rgb, None, None, True, ops.ACT_VANILLA, ops.plane_samples(planes))
    dplanes = ops.mlp_bwd_chain(packed_bwd, packed, d_raw, masks, planes.shape)
    assert ops.lib.aon_stream_is_folded(ops._ptr(packed_bwd)) == (fold_form == "folded")
    grads = ops.vanilla_wgrad(planes, dplanes, d_raw, packed_bwd)
    for name, g in grads.items():
        err = rel_l2(g.cpu(), sd_o[prefix + name].grad)
        assert err <= 2e-5, f"{name}: relative L2 gradient error {err:.3e}"


def test_training_decreases_loss_and_is_deterministic(dev):
    """A few Adam steps (the reference's optimiser, model.py:386-389) on a fixed ray batch: loss goes down, repeated runs
    are bit-identical (the weight-gradient reduction is atomics-free), parameters are re-packed after every step."""
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model import NeRF

    def run():
        sd = syn.make_nerf_state_dict(seed=4, density_scale=5.0)
        model = NeRF().to(dev)
        model.load_state_dict(sd)
        opt = torch.optim.Adam(model.parameters(), lr=5e-4, betas=(0.9, 0.999))
        rays = {k: v.to(dev) for k, v in syn.random_rays(256, seed=9).items()}
        target = torch.rand(256, 3, generator=torch.Generator().manual_seed(9)).to(dev)
        t_rand = torch.rand(256, 65, generator=torch.Generator().manual_seed(10)).to(dev)
        u = torch.rand(256, 128, generator=torch.Generator().manual_seed(11)).to(dev)
        losses = []
        for _ in range(6):
            opt.zero_grad()
            out = model(rays, True, True, 2.0, 6.0, t_rand=t_rand, u=u)
            loss = torch.mean((out[0][0] - target) ** 2) + torch.mean((out[1][0] - target) ** 2)
            loss.backward()
            opt.step()
            losses.append(loss.item())
        return losses

    a, b = run(), run()
    assert a == b
    assert a[-1] < a[0] and all(x == x for x in a)


def test_gradients_vs_reference_golden(dev, golden, nerf_sd):
    """G9: gradients the REFERENCE's autograd produced (tests/golden/g9_backward.npz) for mse(coarse)+mse(fine) on 64
    rays, vanilla and articulated, against the HIP backward through the drop-in modules.  Per parameter: gradient
    norm and 48 seeded entries (relative L2 over them).  Tolerances as in the oracle-autograd tests above (fine level and the articulated path
    inherit the inverse-CDF / deformation input sensitivity), relative to the parameter's gradient scale."""
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model import NeRF
    from aon_amd.models.vanilla_nerf.model_autodecoder import NeRF_AE_Art

    g = golden("g9_backward")
    rays = {k: g[k].to(dev) for k in ("rays_o", "rays_d", "viewdirs")}
    target = g["target"].to(dev)

    def check(prefix, named, tol_of):
        bad = []
        for name, p in named:
            gr = p.grad.detach().reshape(-1).cpu()
            ref_norm = g[f"{prefix}|{name}|norm"]
            tol = tol_of(name)
            rms = ref_norm / gr.numel() ** 0.5
            err_norm = abs(gr.double().norm().item() - ref_norm) / max(ref_norm, 1e-12)
            val = g[f"{prefix}|{name}|val"].double()
            # relative L2 over the 48 sampled entries (gradient entries are heavy-tailed: sparse ReLU activations), with
            # the RMS entry as the floor of the scale; 48 draws of a tol-sized relative error -> allow 3x
            err_val = (gr[g[f"{prefix}|{name}|idx"]].double() - val).norm().item() / max(val.norm().item(), 48 ** 0.5 * rms, 1e-12)
            if err_norm > tol or err_val > 3 * tol:
                bad.append((name, f"{err_norm:.1e}", f"{err_val:.1e}"))
        assert not bad, bad

    model = NeRF().to(dev)
    model.load_state_dict(nerf_sd)
    out = model(rays, False, True, g["near"], g["far"])
    loss = torch.mean((out[0][0] - target) ** 2) + torch.mean((out[1][0] - target) ** 2)
    loss.backward()
    assert abs(loss.item() - g["vanilla_loss"]) <= 2e-5
    # measured round 2 (tests/diag/diag_tolerances.py): coarse 6.9e-7 (norm) / 1.5e-5 (entries); fine 2.2e-4 / 2.5e-3
    check("vanilla", model.named_parameters(), lambda n: 3e-3 if n.startswith("fine_mlp") else 1e-4)

    ga = golden("g11_nerf_ae_art")
    amodel = NeRF_AE_Art().to(dev)
    amodel.load_state_dict(syn.make_art_state_dict(seed=0, density_scale=30.0))
    lat = {k: ga[f"lat_train_{k}"].to(dev).requires_grad_(True) for k in ("density", "color", "articulation")}
    out = amodel(rays, False, True, g["near"], g["far"], lat)
    loss = torch.mean((out[0][0] - target) ** 2) + torch.mean((out[1][0] - target) ** 2)
    loss.backward()
    assert abs(loss.item() - g["art_loss"]) <= 1e-4
    # measured: coarse 1.2e-4 (norm) / 3.1e-4 (entries).  The articulated FINE level and the latents are not held to G9 here: on this
    # sharp x30 field two fp32 evaluations of them sit 4.6e-3 .. 4.7e-2 apart (fine samples move across thin shells), and a fixed
    # 2e-2 / 3e-2 bar (rounds 1-4) could hide a tenfold regression inside itself.  Their check is the fp64 yardstick on
    # these very inputs, test_g9_inputs_by_the_fp64_yardstick below (G9 == the oracle's fp32 autograd: tests/test_oracle_golden.py).
    check("art", [(n, p) for n, p in amodel.named_parameters() if n.startswith("coarse_mlp")], lambda n: 1e-3)
    assert all(v.grad is not None and torch.isfinite(v.grad).all() for v in lat.values())


@pytest.mark.parametrize("net", ["vanilla", "articulated"])
def test_g9_inputs_by_the_fp64_yardstick(dev, golden, nerf_sd, net):
    """The 2e-2 / 3e-2 bars of the G9 comparison above (sharp x30 field, articulated fine level and latents) are distances between
    TWO fp32 evaluations.  This test prices them: same rays, weights, latents and target, truth = the oracle's autograd in fp64,
    yardstick = the oracle's own fp32 autograd (the reference's arithmetic, pinned to G9 by tests/test_oracle_golden.py); every HIP
    gradient must be as close to the truth as the reference's fp32 is x 5 (tests/_gradcheck.py, floor 1e-4, head biases 2e-5).
    The ratios are printed (state them)."""
    import aon_amd.synthetic as syn
    from _gradcheck import assert_as_close_as_fp32
    from aon_amd.models.vanilla_nerf.model import NeRF
    from aon_amd.models.vanilla_nerf.model_autodecoder import NeRF_AE_Art

    g = golden("g9_backward")
    rays_cpu = {k: g[k] for k in ("rays_o", "rays_d", "viewdirs")}
    target = g["target"]
    art = net == "articulated"
    if art:
        ga = golden("g11_nerf_ae_art")
        sd = syn.make_art_state_dict(seed=0, density_scale=30.0)
        lat0 = {k: ga[f"lat_train_{k}"] for k in ("density", "color", "articulation")}
        model = NeRF_AE_Art().to(dev)
    else:
        sd, lat0, model = nerf_sd, None, NeRF().to(dev)
    model.load_state_dict(sd)

    def oracle_grads(dtype):
        sd_o = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in sd.items()}
        r = {k: v.to(dtype) for k, v in rays_cpu.items()}
        lat = None if lat0 is None else {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in lat0.items()}
        out = orc.nerf_ae_art_forward(sd_o, r, False, True, g["near"], g["far"], lat) if art else orc.nerf_forward(sd_o, r, False, True, g["near"], g["far"])
        (orc.img2mse(out[0][0], target.to(dtype)) + orc.img2mse(out[1][0], target.to(dtype))).backward()
        gr = {k: v.grad for k, v in sd_o.items()}
        if lat is not None:
            gr.update({f"latent[{k}]": v.grad for k, v in lat.items()})
        return gr

    truth, ref32 = oracle_grads(torch.float64), oracle_grads(torch.float32)
    rays = {k: v.to(dev) for k, v in rays_cpu.items()}
    if art:
        lat = {k: v.detach().clone().to(dev).requires_grad_(True) for k, v in lat0.items()}
        out = model(rays, False, True, g["near"], g["far"], lat)
    else:
        out = model(rays, False, True, g["near"], g["far"])
    (torch.mean((out[0][0] - target.to(dev)) ** 2) + torch.mean((out[1][0] - target.to(dev)) ** 2)).backward()
    hip = {name: p.grad.cpu() for name, p in model.named_parameters()}
    if art:
        hip.update({f"latent[{k}]": lat[k].grad.cpu() for k in lat})
    assert_as_close_as_fp32(hip, truth, ref32, f"G9 inputs, {net}", factor=5.0, floor=1e-4)


def test_training_trajectory_vs_oracle(dev):
    """Three Adam steps with identical batches and draws: the HIP path (module forward/backward + torch.optim.Adam on the
    device) against the oracle (CPU autograd + the same optimiser).  Losses agree per step; after the steps the parameters
    have moved by ~3 * lr = 1.5e-3 each and agree to 2 % of that movement on average (measured 0.6 %)."""
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model import NeRF

    n, steps, lr = 128, 3, 5e-4
    sd = syn.make_nerf_state_dict(seed=6, density_scale=5.0)
    rays_cpu = syn.random_rays(n, seed=12)
    gen = torch.Generator().manual_seed(12)
    target = torch.rand(n, 3, generator=gen)
    draws = [(torch.rand(n, 65, generator=gen), torch.rand(n, 128, generator=gen)) for _ in range(steps)]
    # oracle side
    sd_o = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt_o = torch.optim.Adam(list(sd_o.values()), lr=lr, betas=(0.9, 0.999))
    losses_o = []
    for t_rand, u in draws:
        opt_o.zero_grad()
        out = orc.nerf_forward(sd_o, rays_cpu, True, True, 2.0, 6.0, t_rand=t_rand, u=u)
        loss = orc.img2mse(out[0][0], target) + orc.img2mse(out[1][0], target)
        loss.backward()
        opt_o.step()
        losses_o.append(loss.item())
    # HIP side
    model = NeRF().to(dev)
    model.load_state_dict(sd)
    opt = torch.optim.Adam(model.parameters(), lr=lr, betas=(0.9, 0.999))
    rays = {k: v.to(dev) for k, v in rays_cpu.items()}
    losses = []
    for t_rand, u in draws:
        opt.zero_grad()
        out = model(rays, True, True, 2.0, 6.0, t_rand=t_rand.to(dev), u=u.to(dev))
        loss = torch.mean((out[0][0] - target.to(dev)) ** 2) + torch.mean((out[1][0] - target.to(dev)) ** 2)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    for a, b in zip(losses, losses_o):
        assert abs(a - b) <= 5e-5 * max(1.0, abs(b)), (losses, losses_o)
    # Adam normalises the step: |delta| ~ lr per step wherever the gradient is not tiny, so compare against the movement
    worst = 0.0
    for name, p in model.named_parameters():
        moved = (sd_o[name].detach() - sd[name]).abs().max().item()
        diff = (p.detach().cpu() - sd_o[name].detach()).abs()
        # sign flips of near-zero gradients move single entries by up to 2*lr per step; bound the bulk and the worst entry
        assert diff.mean().item() <= 0.02 * max(moved, 1e-6), (name, diff.mean().item(), moved)
        assert diff.max().item() <= 2.2 * steps * lr, (name, diff.max().item())
        worst = max(worst, diff.mean().item() / max(moved, 1e-6))
    print("worst mean-difference / movement:", worst)


def test_two_call_step_equals_the_staged_entry_points(dev, nerf_sd):
    """The training step through aon_render_fwd_train / aon_render_bwd (what the drop-in module runs) against the same step
    assembled from the stage-level entry points: outputs and all 48 gradients bit for bit; and the two-stream backward equals
    the single-stream one."""
    import aon_amd.synthetic as syn
    from aon_amd import ops
    from aon_amd.models.vanilla_nerf.model import NeRF

    n = 200
    rays = {k: v.to(dev) for k, v in syn.random_rays(n, seed=31).items()}
    gen = torch.Generator().manual_seed(31)
    t_rand, u, target = (torch.rand(n, 65, generator=gen).to(dev), torch.rand(n, 128, generator=gen).to(dev), torch.rand(n, 3, generator=gen).to(dev))

    def fused(overlap):
        ops.set_bwd_overlap(overlap)
        model = NeRF().to(dev)
        model.load_state_dict(nerf_sd)
        out = model(rays, True, False, 2.0, 6.0, t_rand=t_rand, u=u)
        (torch.mean((out[0][0] - target) ** 2) + torch.mean((out[1][0] - target) ** 2)).backward()
        return [o.detach() for lvl in out for o in lvl], {k: p.grad.clone() for k, p in model.named_parameters()}

    try:
        outs_a, grads_a = fused(True)
        outs_b, grads_b = fused(False)
    finally:
        ops.set_bwd_overlap(True)
    assert all(torch.equal(a, b) for a, b in zip(outs_a, outs_b))
    assert all(torch.equal(grads_a[k], grads_b[k]) for k in grads_a)
    # staged: the same launches driven from Python
    outs_s, grads_s = [], {}
    t_vals = weights = None
    for lvl, name in enumerate(("coarse_mlp", "fine_mlp")):
        params = {k[len(name) + 1:]: v.to(dev) for k, v in nerf_sd.items() if k.startswith(name + ".")}
        pf, pb = ops.pack_vanilla_mlp(params), ops.pack_vanilla_mlp_bwd(params)
        t_vals = ops.sample_along_rays(rays["rays_o"], rays["rays_d"], 64, 2.0, 6.0, t_rand, want_coords=False)[0] if lvl == 0 else ops.sample_pdf_t(t_vals, weights, u)
        raw, planes, masks = ops.mlp_fwd_train(pf, rays["rays_o"], rays["rays_d"], rays["viewdirs"], t_vals)
        rgb, acc, weights, depth = ops.composite_raw(raw, t_vals, rays["rays_d"], False, ops.ACT_VANILLA, want_weights=True)
        outs_s += [rgb, acc, depth]
        g_rgb = 2.0 * (rgb - target) / (3 * n)
        d_raw = ops.composite_bwd(raw, t_vals, rays["rays_d"], g_rgb.contiguous(), None, None, False, ops.ACT_VANILLA, ops.plane_samples(planes))
        dpl = ops.mlp_bwd_chain(pb, pf, d_raw, masks, planes.shape)
        for k, v in ops.vanilla_wgrad(planes, dpl, d_raw, pb).items():
            grads_s[f"{name}.{k}"] = v
    assert all(torch.equal(a, b) for a, b in zip(outs_a, outs_s))
    for k in grads_a:
        assert torch.equal(grads_a[k], grads_s[k]), k


def _to_step_major(rows_view):
    """(rows, Np) -> the step-major plane tensor (Np/32, rows/4, 32, 4) of include/aon_hip.h."""
    rows, Np = rows_view.shape
    return rows_view.reshape(rows // 4, 4, Np // 32, 32).permute(2, 0, 3, 1).contiguous()


@pytest.mark.parametrize("net,n_samples", [("vanilla", 640), ("vanilla", 4096 * 3 + 128), ("articulated", 1152), ("articulated", 9984)])
def test_grouped_wgrad_against_fp64_matmul(dev, net, n_samples, fold_form):
    """The grouped weight-gradient launch on RANDOM step-major planes against dW = dZ . H^T in fp64, every parameter of the network
    -- all five job kinds (256x256, 128x128, 256x64, 128x256, 128x32), the head / bias / first-deformation-layer reductions, the
    second stage, the latent outer products and latent gradients -- independent of the forward and the chain (round 3; until now
    the weight gradients were only checked through whole-path gradients).  Also: the layout helpers are inverse to each other,
    sample counts that leave trailing workgroups of a layer with a short range, and bit-equal repeats (no atomics)."""
    import aon_amd.synthetic as syn
    from aon_amd import ops

    gen = torch.Generator().manual_seed(n_samples)
    Np = n_samples
    assert Np % 128 == 0
    art = net == "articulated"
    rows = 3456 if art else 2528
    P = torch.randn(rows, Np, generator=gen)
    D = torch.randn(rows, Np, generator=gen) * 0.1
    d_raw = torch.randn(Np, 4, generator=gen) * 0.1
    planes, dplanes = _to_step_major(P).to(dev), _to_step_major(D).to(dev)
    assert torch.equal(ops.plane_rows_view(planes).cpu(), P)
    P64, D64, R64 = P.double(), D.double(), d_raw.double()

    def close(name, got, want, tol=2e-5):
        err = (got.double().cpu() - want).abs().max().item() / max(want.abs().max().item(), 1e-30)
        assert err <= tol, (name, err)

    # Folded form (round 5): the bottleneck rows of the planes are not read; dW' = dZ_v0 . H7^T and db' are un-folded with the raw
    # W_v0[:, :256], W_b, b_b (fp64 products): dW_b = W_v0h^T dW', db_b = W_v0h^T db', dW_v0h = dW' W_b^T + db' (x) b_b.
    folded = fold_form == "folded"

    def unfolded(dhv, h7, Wv_h, Wb, bb):
        dWf, dbf = dhv @ h7.T, dhv.sum(1)
        return Wv_h.T @ dWf, Wv_h.T @ dbf, dWf @ Wb.T + torch.outer(dbf, bb)

    if not art:
        vsd = {k[len("fine_mlp."):]: v.to(dev) for k, v in syn.make_nerf_state_dict(seed=7, density_scale=1.0).items() if k.startswith("fine_mlp.")}
        pb = ops.pack_vanilla_mlp_bwd(vsd)
        g = ops.vanilla_wgrad(planes, dplanes, d_raw.to(dev), pb)
        g2 = ops.vanilla_wgrad(planes, dplanes, d_raw.to(dev), pb)
        assert all(torch.equal(g[k], g2[k]) for k in g)
        E, VE = P64[0:63], P64[2368:2395]
        H = lambda l: P64[64 + 256 * l: 64 + 256 * (l + 1)]
        dZ = lambda l: D64[64 + 256 * l: 64 + 256 * (l + 1)]
        close("pts_linears.0.weight", g["pts_linears.0.weight"], dZ(0) @ E.T)
        close("pts_linears.0.bias", g["pts_linears.0.bias"], dZ(0).sum(1))
        for l in range(1, 8):
            want = dZ(l) @ H(l - 1).T
            if l == 5:
                want = torch.cat([want, dZ(5) @ E.T], 1)
            close(f"pts_linears.{l}.weight", g[f"pts_linears.{l}.weight"], want)
            close(f"pts_linears.{l}.bias", g[f"pts_linears.{l}.bias"], dZ(l).sum(1))
        dbot, dhv, bott, hv = D64[2112:2368], D64[2400:2528], P64[2112:2368], P64[2400:2528]
        if folded:
            Wd = {k: v.double().cpu() for k, v in vsd.items()}
            dWb, dbb, dWvh = unfolded(dhv, H(7), Wd["views_linear.0.weight"][:, :256], Wd["bottleneck_layer.weight"], Wd["bottleneck_layer.bias"])
            close("bottleneck_layer.weight", g["bottleneck_layer.weight"], dWb)
            close("bottleneck_layer.bias", g["bottleneck_layer.bias"], dbb)
            close("views_linear.0.weight", g["views_linear.0.weight"], torch.cat([dWvh, dhv @ VE.T], 1))
        else:
            close("bottleneck_layer.weight", g["bottleneck_layer.weight"], dbot @ H(7).T)
            close("bottleneck_layer.bias", g["bottleneck_layer.bias"], dbot.sum(1))
            close("views_linear.0.weight", g["views_linear.0.weight"], torch.cat([dhv @ bott.T, dhv @ VE.T], 1))
        close("views_linear.0.bias", g["views_linear.0.bias"], dhv.sum(1))
        close("density_layer.weight", g["density_layer.weight"], (H(7) @ R64[:, 3:4]).T)
        close("density_layer.bias", g["density_layer.bias"], R64[:, 3].sum(0, keepdim=True))
        close("rgb_layer.weight", g["rgb_layer.weight"], (hv @ R64[:, :3]).T)
        close("rgb_layer.bias", g["rgb_layer.bias"], R64[:, :3].sum(0))
        return
    sd = {k[len("fine_mlp."):]: v.to(dev) for k, v in syn.make_art_state_dict(seed=2, density_scale=1.0).items() if k.startswith("fine_mlp.")}
    lat = {"density": torch.randn(1, 128, generator=gen).to(dev), "color": torch.randn(1, 128, generator=gen).to(dev),
           "articulation": torch.randn(1, 32, generator=gen).to(dev)}
    dxp = torch.randn(Np, 4, generator=gen) * 0.1
    pb = ops.pack_art_mlp_bwd(sd)
    g, gl = ops.art_wgrad(planes, dplanes, d_raw.to(dev), dxp.to(dev), sd, lat, packed_bwd=pb)
    g2, gl2 = ops.art_wgrad(planes, dplanes, d_raw.to(dev), dxp.to(dev), sd, lat, packed_bwd=pb)
    assert all(torch.equal(g[k], g2[k]) for k in g) and all(torch.equal(gl[k], gl2[k]) for k in gl)
    X64 = dxp.double()
    d_ = lambda l: slice(32 + 128 * l, 32 + 128 * (l + 1))
    h_ = lambda l: slice(608 + 256 * l, 608 + 256 * (l + 1))
    v_ = lambda l: slice(2944 + 128 * l, 2944 + 128 * (l + 1))
    E, bot, VE, pos = P64[544:607], slice(2656, 2912), P64[2912:2939], P64[0:3]
    shape, app, artc = (lat[k].double().cpu().reshape(-1) for k in ("density", "color", "articulation"))
    db0 = D64[d_(0)].sum(1)
    close("deformations_linear.0.weight", g["deformations_linear.0.weight"],
          torch.cat([D64[d_(0)] @ pos.T, torch.outer(db0, shape), torch.outer(db0, artc)], 1))
    close("deformations_linear.0.bias", g["deformations_linear.0.bias"], db0)
    for l in range(1, 4):
        close(f"deformations_linear.{l}.weight", g[f"deformations_linear.{l}.weight"], D64[d_(l)] @ P64[d_(l - 1)].T)
        close(f"deformations_linear.{l}.bias", g[f"deformations_linear.{l}.bias"], D64[d_(l)].sum(1))
        close(f"views_linear.{l}.weight", g[f"views_linear.{l}.weight"], D64[v_(l)] @ P64[v_(l - 1)].T)
        close(f"views_linear.{l}.bias", g[f"views_linear.{l}.bias"], D64[v_(l)].sum(1))
    close("deformation_layer.weight", g["deformation_layer.weight"], (P64[d_(3)] @ X64[:, :3]).T)
    close("deformation_layer.bias", g["deformation_layer.bias"], X64[:, :3].sum(0))
    db_t0, db_t5, db_v0 = D64[h_(0)].sum(1), D64[h_(5)].sum(1), D64[v_(0)].sum(1)
    close("pts_linears.0.weight", g["pts_linears.0.weight"], torch.cat([D64[h_(0)] @ E.T, torch.outer(db_t0, shape)], 1))
    for l in range(1, 8):
        want = D64[h_(l)] @ P64[h_(l - 1)].T
        if l == 5:
            want = torch.cat([want, D64[h_(5)] @ E.T, torch.outer(db_t5, shape)], 1)
        close(f"pts_linears.{l}.weight", g[f"pts_linears.{l}.weight"], want)
        close(f"pts_linears.{l}.bias", g[f"pts_linears.{l}.bias"], D64[h_(l)].sum(1))
    W = {k: v.double().cpu() for k, v in sd.items()}
    if folded:
        dWb, dbb, dWvh = unfolded(D64[v_(0)], P64[h_(7)], W["views_linear.0.weight"][:, :256], W["bottleneck_layer.weight"], W["bottleneck_layer.bias"])
        close("bottleneck_layer.weight", g["bottleneck_layer.weight"], dWb)
        close("bottleneck_layer.bias", g["bottleneck_layer.bias"], dbb)
        close("views_linear.0.weight", g["views_linear.0.weight"], torch.cat([dWvh, D64[v_(0)] @ VE.T, torch.outer(db_v0, app)], 1))
    else:
        close("bottleneck_layer.weight", g["bottleneck_layer.weight"], D64[bot] @ P64[h_(7)].T)
        close("bottleneck_layer.bias", g["bottleneck_layer.bias"], D64[bot].sum(1))
        close("views_linear.0.weight", g["views_linear.0.weight"], torch.cat([D64[v_(0)] @ P64[bot].T, D64[v_(0)] @ VE.T, torch.outer(db_v0, app)], 1))
    close("views_linear.0.bias", g["views_linear.0.bias"], db_v0)
    close("density_layer.weight", g["density_layer.weight"], (P64[h_(7)] @ R64[:, 3:4]).T)
    close("rgb_layer.weight", g["rgb_layer.weight"], (P64[v_(3)] @ R64[:, :3]).T)
    close("rgb_layer.bias", g["rgb_layer.bias"], R64[:, :3].sum(0))
    close("latent density", gl["density"], W["deformations_linear.0.weight"][:, 3:131].T @ db0 + W["pts_linears.0.weight"][:, 63:191].T @ db_t0 +
          W["pts_linears.5.weight"][:, 319:447].T @ db_t5)
    close("latent color", gl["color"], W["views_linear.0.weight"][:, 283:411].T @ db_v0)
    close("latent articulation", gl["articulation"], W["deformations_linear.0.weight"][:, 131:163].T @ db0)


@pytest.mark.parametrize("net", ["vanilla", "articulated", "other_degrees"])
def test_forward_overlap_is_bit_identical(dev, net):
    """The training forward in its three schedules -- merged (round 4 default: coarse(A) | fine(A) + coarse(B) | fine(B), the middle
    launch carrying two networks; forced here, small batches would not gain a round), two ray halves on two library streams
    (aon_set_fwd_overlap, round 3) and one launch per level on one stream -- gives the same outputs and, through the planes,
    decision bits and raw records it leaves, the same gradients, bit for bit; ragged ray counts (splits are on multiples of 128 rays;
    below 256 rays there is none)."""
    import aon_amd.synthetic as syn
    from aon_amd import ops
    from aon_amd.models.vanilla_nerf.model import NeRF
    from aon_amd.models.vanilla_nerf.model_autodecoder import NeRF_AE_Art

    for n in (257, 600, 1000):
        frame = syn.make_rays(32, 40, syn.look_at_pose(4.0, 60, 20), syn.focal_from_fovy(32))
        rays = {k: v[:n].contiguous().to(dev) for k, v in frame.items()}
        target = syn.seeded_uniform(5, n, 3).to(dev)
        tr, u = syn.seeded_uniform(6, n, 65).to(dev), syn.seeded_uniform(7, n, 128).to(dev)
        lat = None
        if net == "articulated":
            model = NeRF_AE_Art().to(dev)
            model.load_state_dict(syn.make_art_state_dict(seed=5, density_scale=2.0))
            lib = syn.make_code_library_state(seed=0, n_max_objs=2)
            lat = {"density": lib["embedding_instance_shape.weight"][1:2].to(dev), "color": lib["embedding_instance_appearance.weight"][1:2].to(dev),
                   "articulation": lib["embedding_instance_articulation.weight"][3:4].to(dev)}
        elif net == "other_degrees":
            gk = dict(min_deg_point=0, max_deg_point=6, deg_view=2)
            model = NeRF(**gk).to(dev)
            model.load_state_dict(syn.make_general_nerf_state_dict(9, **gk))
        else:
            model = NeRF().to(dev)
            model.load_state_dict(syn.make_smooth_nerf_state_dict())
        res = []
        try:
            # (and the backward chains merged or per level; merged: the chain-independent head reductions beside the chain or behind it)
            for merge, on, bwd_merge, early in ((2, True, True, True), (0, True, False, True), (0, False, True, False)):
                ops.set_fwd_merge(merge)
                ops.set_fwd_overlap(on)
                ops.set_bwd_merge(bwd_merge)
                ops.set_bwd_early_heads(early)
                model.zero_grad()
                out = model(rays, True, True, 2.0, 6.0, lat, t_rand=tr, u=u) if lat is not None else model(rays, True, True, 2.0, 6.0, t_rand=tr, u=u)
                (((out[0][0] - target) ** 2).mean() + ((out[1][0] - target) ** 2).mean() + out[1][2].mean() * 1e-3).backward()
                res.append([x.detach().clone() for lvl in out for x in lvl] + [p.grad.clone() for p in model.parameters()])
        finally:
            ops.set_fwd_overlap(True)
            ops.set_fwd_merge(True)
            ops.set_bwd_merge(True)
            ops.set_bwd_early_heads(True)
        for a, b, c in zip(*res):
            assert torch.equal(a, b) and torch.equal(a, c)


def test_transposed_streams_packed_on_a_side_stream_same_gradients(dev, nerf_sd, monkeypatch):
    """Round 5: the drop-in modules pack the levels' transposed streams on a side torch stream beside the forward
    (models/vanilla_nerf/model.py packed_bwd_aside; AON_PACK_ASIDE=0 packs in line).  Same buffers, same kernels: every output and
    gradient bit-equal, over several steps with an optimiser step in between (the side stream must see the updated parameters)."""
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model import NeRF

    n = 300
    rays = {k: v.to(dev) for k, v in syn.random_rays(n, seed=77).items()}
    target = syn.seeded_uniform(78, n, 3).to(dev)
    res = {}
    for aside in ("1", "0"):
        monkeypatch.setenv("AON_PACK_ASIDE", aside)
        model = NeRF().to(dev)
        model.load_state_dict(nerf_sd)
        opt = torch.optim.SGD(model.parameters(), lr=1e-3)
        out_all = []
        for step in range(3):
            tr, u = syn.seeded_uniform(80 + step, n, 65).to(dev), syn.seeded_uniform(90 + step, n, 128).to(dev)
            opt.zero_grad()
            out = model(rays, True, True, 2.0, 6.0, t_rand=tr, u=u)
            (((out[0][0] - target) ** 2).mean() + ((out[1][0] - target) ** 2).mean()).backward()
            out_all += [x.detach().clone() for lvl in out for x in lvl] + [p.grad.clone() for p in model.parameters()]
            opt.step()
        res[aside] = out_all
    assert all(torch.equal(a, b) for a, b in zip(res["1"], res["0"]))
