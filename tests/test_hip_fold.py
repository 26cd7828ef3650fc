"""Round 5: bottleneck_layer folded into views_linear[0] (models/vanilla_nerf/model.py:105-118, model_autodecoder.py:222-237).

The reference's bottleneck has no activation, so W_v0[:, :256] (W_b h + b_b) == (W_v0[:, :256] W_b) h + W_v0[:, :256] b_b: the fused
kernels run ONE 256 -> 128 layer W' (default) where the literal graph runs two.  Every other GPU test runs the folded form (it is the
default) against the reference's goldens and the oracle at its unchanged bar; this file holds what is specific to the fold:
  * W' / b' in the packed buffers are the correctly rounded fp64 products of the fp32 parameters;
  * the LITERAL form (aon_set_bottleneck_fold(0)) is still there and still meets the reference bars (the A/B partner);
  * the two forms agree with each other far inside those bars, forward and gradients, vanilla and articulated;
  * a buffer keeps the form it was packed in whatever the switch says later, and buffers of two forms in one call are refused."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nerf_oracle as orc  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.fixture()
def ops():
    from aon_amd import ops as _ops

    before = _ops.bottleneck_fold()
    yield _ops
    _ops.set_bottleneck_fold(before)


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _vanilla_params(nerf_sd, lvl, dev):
    return {k[len(lvl) + 5:]: v.to(dev) for k, v in nerf_sd.items() if k.startswith(lvl + "_mlp.")}


def test_folded_weights_are_the_rounded_fp64_product(ops, dev, nerf_sd):
    """W' (128 x 256) and b' (128) as the pack call leaves them behind the folded stream (csrc/aon_common.h kFoldTmpOff) against
    W_v0[:, :256] @ W_b and W_v0[:, :256] @ b_b + b_v0 evaluated in fp64 and rounded once: equal, or one ulp apart where the fp64 sums'
    own last bits decide a tie."""
    ops.set_bottleneck_fold(True)
    p = _vanilla_params(nerf_sd, "fine", dev)
    packed = ops.pack_vanilla_mlp(p)
    assert ops.lib.aon_stream_is_folded(ops._ptr(packed)) == 1
    off = 60 * 32768 + 9 * 16384
    tmp = packed[off: off + (128 * 256 + 128) * 4].view(torch.float32).cpu()
    Wv, Wb = p["views_linear.0.weight"].double().cpu(), p["bottleneck_layer.weight"].double().cpu()
    want_w = (Wv[:, :256] @ Wb).float()
    want_b = (Wv[:, :256] @ p["bottleneck_layer.bias"].double().cpu() + p["views_linear.0.bias"].double().cpu()).float()
    got_w, got_b = tmp[: 128 * 256].reshape(128, 256), tmp[128 * 256:]
    ulp = lambda x: torch.maximum(x.abs(), torch.tensor(1e-30)) * 2.0 ** -23   # noqa: E731
    assert ((got_w - want_w).abs() <= ulp(want_w)).all() and (got_w == want_w).double().mean() > 0.999
    assert ((got_b - want_b).abs() <= ulp(want_b)).all()


def test_literal_form_still_meets_the_reference_bars(ops, dev, golden, nerf_sd):
    """aon_set_bottleneck_fold(0): the two-layer kernels of rounds 1-4 against G4 (the reference's own outputs) and the oracle."""
    import aon_amd.synthetic as syn

    ops.set_bottleneck_fold(False)
    g = golden("g4_mlp")
    for lvl in ("coarse", "fine"):
        packed = ops.pack_vanilla_mlp(_vanilla_params(nerf_sd, lvl, dev))
        assert ops.lib.aon_stream_is_folded(ops._ptr(packed)) == 0
        raw = ops.mlp_fwd_enc(packed, g["samples_enc"].to(dev), g["viewdirs_enc"].to(dev)).cpu()
        torch.testing.assert_close(raw[..., :3], g[f"raw_rgb_{lvl}"], rtol=2e-5, atol=2e-5)
        torch.testing.assert_close(raw[..., 3:], g[f"raw_sigma_{lvl}"], rtol=2e-5, atol=6e-4)
    from aon_amd.models.vanilla_nerf.model import NeRF

    model = NeRF().to(dev)
    model.load_state_dict(nerf_sd)
    g8 = golden("g8_nerf_forward")
    rays = {k: g8[k].to(dev) for k in ("rays_o", "rays_d", "viewdirs")}
    with torch.no_grad():
        out = model(rays, False, True, g8["near"], g8["far"])
    ref, aux = orc.nerf_forward(nerf_sd, {k: g8[k] for k in ("rays_o", "rays_d", "viewdirs")}, False, True, g8["near"], g8["far"], return_aux=True)
    ok = torch.ones(g8["rays_o"].shape[0], dtype=torch.bool)
    for a in aux:
        ok &= a["raw_sigma"][:, -1, 0].abs() > 2e-2
    for lvl in (0, 1):
        torch.testing.assert_close(out[lvl][0].cpu()[ok], ref[lvl][0][ok], rtol=0, atol=1e-5 if lvl == 0 else 2e-4)
    del syn


@pytest.mark.parametrize("net", ["vanilla", "articulated"])
def test_two_forms_agree_forward_and_gradients(ops, dev, net):
    """Same weights, rays, draws through the folded and the literal kernels: the MLP's raw outputs agree to 1e-5 of their scale, the
    rendered levels to 2e-6 on a smooth field (where nothing is chaotic, tests/test_hip_smooth.py), and every parameter (and latent)
    gradient of the training loss to 2e-4 relative L2 -- a wrong fold (a transposed factor, a missing b_b term, an un-folding that
    forgets db' (x) b_b) is O(1).  Measured values are printed."""
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model import NeRF
    from aon_amd.models.vanilla_nerf.model_autodecoder import NeRF_AE_Art

    n = 384
    frame = syn.make_rays(24, 32, syn.look_at_pose(4.0, 40, 25), syn.focal_from_fovy(24))
    rays = {k: v[:n].contiguous().to(dev) for k, v in frame.items()}
    target = syn.seeded_uniform(51, n, 3).to(dev)
    tr, u = syn.seeded_uniform(52, n, 65).to(dev), syn.seeded_uniform(53, n, 128).to(dev)
    art = net == "articulated"
    if art:
        sd = syn.make_art_state_dict(seed=18, density_scale=2.0)
        lib = syn.make_code_library_state(seed=0, n_max_objs=2)
        lat0 = {"density": lib["embedding_instance_shape.weight"][1:2], "color": lib["embedding_instance_appearance.weight"][1:2],
                "articulation": lib["embedding_instance_articulation.weight"][3:4]}
    else:
        sd = syn.make_smooth_nerf_state_dict()
    # make the bottleneck bias matter (the synthetic state dicts keep nn.Linear's small default biases)
    for k in list(sd):
        if k.endswith("bottleneck_layer.bias"):
            sd[k] = sd[k] + 0.05 * syn.seeded_uniform(77, *sd[k].shape) - 0.025
    res = {}
    for form in ("folded", "literal"):
        ops.set_bottleneck_fold(form == "folded")
        model = (NeRF_AE_Art() if art else NeRF()).to(dev)
        model.load_state_dict(sd)
        lat = {k: v.clone().to(dev).requires_grad_(True) for k, v in lat0.items()} if art else None
        args = (rays, True, True, 2.0, 6.0) + ((lat,) if art else ())
        with torch.no_grad():
            det = model(*((rays, False, True, 2.0, 6.0) + ((lat,) if art else ())))
        out = model(*args, t_rand=tr, u=u)
        loss = ((out[0][0] - target) ** 2).mean() + ((out[1][0] - target) ** 2).mean()
        loss.backward()
        grads = {k: p.grad.clone() for k, p in model.named_parameters()}
        if art:
            grads.update({f"latent[{k}]": v.grad.clone() for k, v in lat.items()})
        res[form] = ([x.detach() for lvl in det for x in lvl], [x.detach() for lvl in out for x in lvl], loss.item(), grads)
    outs_f, outs_l = res["folded"][0] + res["folded"][1], res["literal"][0] + res["literal"][1]
    worst_out = max((a - b).abs().max().item() for i, (a, b) in enumerate(zip(outs_f, outs_l)) if i % 3 != 2)      # rgb, acc
    worst_depth = max((a - b).abs().max().item() for i, (a, b) in enumerate(zip(outs_f, outs_l)) if i % 3 == 2)
    worst = max((rel_l2(res["folded"][3][k], res["literal"][3][k]), k) for k in res["literal"][3])
    print(f"{net}: folded vs literal: worst rendered rgb / acc {worst_out:.2e}, depth {worst_depth:.2e}, loss {res['folded'][2]:.8f} vs "
          f"{res['literal'][2]:.8f}, worst gradient rel L2 {worst[0]:.2e} on {worst[1]}")
    assert worst_out <= (5e-5 if art else 5e-6) and worst_depth <= (5e-4 if art else 5e-5)
    assert abs(res["folded"][2] - res["literal"][2]) <= 1e-6 * max(1.0, abs(res["literal"][2]))
    for k in res["literal"][3]:
        e = rel_l2(res["folded"][3][k], res["literal"][3][k])
        assert e <= (1e-3 if art else 2e-4), (k, e)


def test_form_is_a_property_of_the_buffer(ops, dev, nerf_sd):
    """A stream packed in one form runs in that form whatever the switch says afterwards (bit-equal results); the training backward
    refuses a forward stream and a transposed stream of two forms."""
    import aon_amd.synthetic as syn

    p = _vanilla_params(nerf_sd, "fine", dev)
    rays = {k: v.to(dev) for k, v in syn.random_rays(70, seed=4).items()}
    t = torch.sort(torch.rand(70, 65, generator=torch.Generator().manual_seed(4)) * 4 + 2, dim=-1).values.to(dev)
    ops.set_bottleneck_fold(True)
    pf_fold = ops.pack_vanilla_mlp(p)
    a = ops.mlp_fwd(pf_fold, rays["rays_o"], rays["rays_d"], rays["viewdirs"], t)
    ops.set_bottleneck_fold(False)
    b = ops.mlp_fwd(pf_fold, rays["rays_o"], rays["rays_d"], rays["viewdirs"], t)       # still the folded kernel
    assert torch.equal(a, b)
    pf_lit = ops.pack_vanilla_mlp(p)
    c = ops.mlp_fwd(pf_lit, rays["rays_o"], rays["rays_d"], rays["viewdirs"], t)
    assert not torch.equal(a, c) and (a - c).abs().max().item() <= 2e-5 * max(1.0, a.abs().max().item())
    pb_lit = ops.pack_vanilla_mlp_bwd(p)
    from aon_amd.models.vanilla_nerf.model import NeRF

    model = NeRF().to(dev)
    model.load_state_dict(nerf_sd)
    ops.set_bottleneck_fold(True)
    out = model(rays, False, True, 2.0, 6.0)              # forward packed folded ...
    ops.set_bottleneck_fold(False)
    out[1][0].sum().backward()                            # ... and its backward runs folded too: the streams were packed at forward time
    assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in model.parameters())
    # buffers of two forms in one call: refused
    with pytest.raises(Exception, match="different forms"):
        ws_out, ws, geo = ops.render_fwd_train(pf_fold, pf_fold, rays["rays_o"], rays["rays_d"], rays["viewdirs"], 2.0, 6.0, True, 2, None, None)
        g = [torch.zeros(70, 3, device=dev)] * 2
        ops.render_bwd(ws, [pb_lit, pb_lit], [pf_fold, pf_fold], rays["rays_d"], True, 2, g, [None, None], [None, None], geometry=geo)


def test_per_ray_view_bias_gives_the_chunk_forms_bits(ops, dev, nerf_sd):
    """Round 5: whole-path calls of the folded vanilla network start the view layer's accumulators from b' + W_v0[:, 256:] ve of the RAY
    (aon_set_view_bias, default on) where the chunk form runs the view-encoding chunk per sample.  The per-ray kernel performs the
    chunk's fused multiply-adds in the chunk's order, so the two forms -- and the stage-level MLP call, which keeps the chunk -- agree
    bit for bit: inference and training forward, ragged ray counts, rays whose samples straddle wave tiles."""
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model import NeRF

    ops.set_bottleneck_fold(True)
    model = NeRF().to(dev)
    model.load_state_dict(nerf_sd)
    try:
        for n in (1, 37, 640, 1500):
            rays = {k: v.to(dev) for k, v in syn.random_rays(n, seed=60 + n).items()}
            tr, u = syn.seeded_uniform(61, n, 65).to(dev), syn.seeded_uniform(62, n, 128).to(dev)
            res = []
            for on in (True, False):
                ops.set_view_bias(on)
                with torch.no_grad():
                    det = model(rays, False, True, 2.0, 6.0)
                model.zero_grad()
                out = model(rays, True, True, 2.0, 6.0, t_rand=tr, u=u)
                (out[0][0].sum() + out[1][0].sum()).backward()
                res.append([x.detach().clone() for lvl in det + out for x in lvl] + [p.grad.clone() for p in model.parameters()])
            for a, b in zip(*res):
                assert torch.equal(a, b)
        # the stage-level call (chunk form) against the same samples through the per-ray form of the whole path
        ops.set_view_bias(True)
        n = 300
        rays = {k: v.to(dev) for k, v in syn.random_rays(n, seed=9).items()}
        packed = model.fine_mlp.packed()
        t = torch.sort(torch.rand(n, 65, generator=torch.Generator().manual_seed(4)) * 4 + 2, dim=-1).values.to(dev)
        raw = ops.mlp_fwd(packed, rays["rays_o"], rays["rays_d"], rays["viewdirs"], t)
        vb = ops.view_bias(packed, rays["viewdirs"])
        # b' + W_v0[:, 256:] ve against torch in fp64: the kernel's 28-term fp32 chains stay within a few ulp of the exact value
        p = dict(model.fine_mlp.named_parameters())
        from oracle import nerf_oracle as orc2
        venc = orc2.pos_enc(rays["viewdirs"].cpu(), 0, 4).double()
        Wv, Wb = p["views_linear.0.weight"].detach().double().cpu(), p["bottleneck_layer.weight"].detach().double().cpu()
        want = venc @ Wv[:, 256:].T + (Wv[:, :256] @ p["bottleneck_layer.bias"].detach().double().cpu() + p["views_linear.0.bias"].detach().double().cpu())
        assert (vb.cpu().double() - want).abs().max().item() <= 2e-6 * max(1.0, want.abs().max().item())
        assert raw.shape == (n, 65, 4) and torch.isfinite(raw).all()
        del Wb
    finally:
        ops.set_view_bias(True)


def test_per_ray_view_bias_articulated(ops, dev):
    """... and the articulated network: views_linear.0's effective bias (appearance latent and W_v0[:, :256] b_b folded in per call) plus its
    view-encoding term per ray; inference, training forward and every gradient bit-equal to the chunk form."""
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model_autodecoder import NeRF_AE_Art

    ops.set_bottleneck_fold(True)
    model = NeRF_AE_Art().to(dev)
    model.load_state_dict(syn.make_art_state_dict(seed=18, density_scale=2.0))
    lib = syn.make_code_library_state(seed=0, n_max_objs=2)
    lat0 = {"density": lib["embedding_instance_shape.weight"][1:2], "color": lib["embedding_instance_appearance.weight"][1:2],
            "articulation": lib["embedding_instance_articulation.weight"][3:4]}
    try:
        for n in (5, 700):
            rays = {k: v.to(dev) for k, v in syn.random_rays(n, seed=80 + n).items()}
            tr, u = syn.seeded_uniform(81, n, 65).to(dev), syn.seeded_uniform(82, n, 128).to(dev)
            res = []
            for on in (True, False):
                ops.set_view_bias(on)
                lat = {k: v.clone().to(dev).requires_grad_(True) for k, v in lat0.items()}
                with torch.no_grad():
                    det = model(rays, False, True, 2.0, 6.0, lat)
                model.zero_grad()
                out = model(rays, True, True, 2.0, 6.0, lat, t_rand=tr, u=u)
                (out[0][0].sum() + out[1][0].sum()).backward()
                res.append([x.detach().clone() for lvl in det + out for x in lvl] + [p.grad.clone() for p in model.parameters()] + [v.grad.clone() for v in lat.values()])
            for a, b in zip(*res):
                assert torch.equal(a, b)
    finally:
        ops.set_view_bias(True)


@pytest.mark.parametrize("degrees", [(0, 10, 4), (1, 8, 3)])
@pytest.mark.parametrize("folded", [True, False])
def test_one_call_pack_of_a_training_step_writes_the_same_bytes(ops, dev, folded, degrees):
    """aon_art_pack_step (round 6: both networks' forward streams, per-call blocks and transposed streams in one C call, the four fp64 fold
    products as ONE launch in front) against the six separate calls on zeroed buffers: every byte equal, in both forms; NULL transposed
    streams are skipped; a misaligned buffer is refused."""
    import aon_amd.synthetic as syn
    from aon_amd import _lib
    from aon_amd.models.vanilla_nerf.model_autodecoder import NeRF_AE_Art

    dk = dict(min_deg_point=degrees[0], max_deg_point=degrees[1], deg_view=degrees[2])
    model = NeRF_AE_Art(**dk).to(dev)
    model.load_state_dict(syn.make_art_state_dict(seed=23, density_scale=2.0, **dk))
    cl = syn.make_code_library_state(seed=1, n_max_objs=2)
    lat = {"density": cl["embedding_instance_shape.weight"][0:1].to(dev), "color": cl["embedding_instance_appearance.weight"][1:2].to(dev),
           "articulation": cl["embedding_instance_articulation.weight"][2:3].to(dev)}
    sizes = [int(_lib.lib.aon_art_packed_bytes()), int(_lib.lib.aon_art_small_bytes()), int(_lib.lib.aon_art_bwd_packed_bytes())]
    zeros = lambda: [tuple(torch.zeros(n, dtype=torch.uint8, device=dev) for n in sizes) for _ in range(2)]   # noqa: E731
    ops.set_bottleneck_fold(folded)
    try:
        mlps = [model.coarse_mlp, model.fine_mlp]
        P = [dict(m.named_parameters()) for m in mlps]
        sep = zeros()
        for (pk, sm, bw), prm in zip(sep, P):
            ops.art_prepare(prm, lat, out=sm, degrees=degrees)
            ops.pack_art_mlp(prm, out=pk, degrees=degrees)
            ops.pack_art_mlp_bwd(prm, out=bw, degrees=degrees)
        one = ops.art_pack_step(P[0], P[1], lat, degrees=degrees, out=zeros())
        for a, b in zip(sep, one):
            for x, y in zip(a, b):
                assert torch.equal(x, y)
                assert x._aon_form == y._aon_form == int(_lib.lib.aon_stream_form(y.data_ptr())) == (1 if folded else 0)
        fwd_only = [(pk, sm, None) for pk, sm, _ in zeros()]
        got = ops.art_pack_step(P[0], P[1], lat, degrees=degrees, out=fwd_only)
        assert all(g[2] is None and torch.equal(g[0], s[0]) and torch.equal(g[1], s[1]) for g, s in zip(got, sep))
        crooked = zeros()
        crooked[1] = (torch.zeros(sizes[0] + 4, dtype=torch.uint8, device=dev)[4:], crooked[1][1], crooked[1][2])
        with pytest.raises(_lib.AonError, match="16-byte"):
            ops.art_pack_step(P[0], P[1], lat, degrees=degrees, out=crooked)
    finally:
        ops.set_bottleneck_fold(True)


@pytest.mark.parametrize("degrees", [(0, 10, 4), (1, 8, 3)])
@pytest.mark.parametrize("folded", [True, False])
def test_one_call_pack_of_a_vanilla_training_step_writes_the_same_bytes(ops, dev, folded, degrees):
    """aon_vanilla_pack_step (round 6: both networks' forward and transposed streams in one C call, the eight fp64 fold products as ONE launch
    in front) against the four separate calls on zeroed buffers: every byte equal, in both forms, at the default and at other encoding
    degrees; NULL transposed streams are skipped; a misaligned buffer is refused."""
    import aon_amd.synthetic as syn
    from aon_amd import _lib

    sizes = [ops.packed_bytes(), int(_lib.lib.aon_bwd_packed_bytes())]
    zeros = lambda: [tuple(torch.zeros(n, dtype=torch.uint8, device=dev) for n in sizes) for _ in range(2)]   # noqa: E731
    sd = syn.make_nerf_state_dict(seed=29, density_scale=2.0, pos_size=3 + 6 * (degrees[1] - degrees[0]), view_pos_size=3 + 6 * degrees[2])
    P = [{k[len(pre):]: v.to(dev) for k, v in sd.items() if k.startswith(pre)} for pre in ("coarse_mlp.", "fine_mlp.")]
    ops.set_bottleneck_fold(folded)
    try:
        sep = zeros()
        for (pk, bw), prm in zip(sep, P):
            ops.pack_vanilla_mlp(prm, out=pk, degrees=degrees)
            ops.pack_vanilla_mlp_bwd(prm, out=bw, degrees=degrees)
        one = ops.vanilla_pack_step(P[0], P[1], degrees=degrees, out=zeros())
        for a, b in zip(sep, one):
            for x, y in zip(a, b):
                assert torch.equal(x, y)
                assert x._aon_form == y._aon_form == int(_lib.lib.aon_stream_form(y.data_ptr())) == (1 if folded else 0)
        got = ops.vanilla_pack_step(P[0], P[1], degrees=degrees, out=[(pk, None) for pk, _ in zeros()])
        assert all(g[1] is None and torch.equal(g[0], s[0]) for g, s in zip(got, sep))
        crooked = zeros()
        crooked[1] = (torch.zeros(sizes[0] + 8, dtype=torch.uint8, device=dev)[8:], crooked[1][1])
        with pytest.raises(_lib.AonError, match="16-byte"):
            ops.vanilla_pack_step(P[0], P[1], degrees=degrees, out=crooked)
    finally:
        ops.set_bottleneck_fold(True)
