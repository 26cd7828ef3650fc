"""torch.autograd glue for the HIP render path (SURVEY 8(a) R14, 8(b) "Ownership"): ``NeRF.forward`` under
``torch.is_grad_enabled()`` returns tensors whose graph reaches the module's parameters, so the reference's
``training_step`` (``loss.backward()``, model.py:264-282) works unchanged.  Forward and backward are both HIP kernels
(no eager-PyTorch math): fused forward with activation planes -> composite backward -> fused data-gradient chain ->
grouped split-N weight-gradient GEMMs, each direction ONE C call (aon_render_fwd_train / aon_render_bwd).  Gradients reach only the MLP parameters: the inverse-CDF draws are detached
(helper.py:249) and rays are data."""
from __future__ import annotations

import torch

from . import ops
from .arena import arena_of, grad_views


def _arena_plan(ctx, params):
    """Round 6: when every parameter of this forward lives in one ParamArena (aon_amd/arena.py), its backward writes the parameter
    gradients straight into the arena's gradient slots and returns views of them -- autograd adopts a returned tensor as `.grad` without
    copying -- so that the optimiser and the data-parallel mean run on one flat buffer.  Decided here (forward): the arena and a claim on
    the slots (see ParamArena.claim); re-checked in the backward (`_arena_grads`)."""
    ctx.arena_plan = None
    ao = arena_of(params)
    if ao is None:
        return
    tok = ao[0].claim(ao[1])
    if tok is not None:
        ctx.arena_plan = (ao, tok, list(params))


def _arena_grads(ctx, per_level_shapes):
    """-> per level list of gradient-slot views, or None: only if the plan was granted AND no parameter holds a gradient yet (a second
    backward before zero_grad must ADD to the first one's gradients, which live in those very slots)."""
    plan = getattr(ctx, "arena_plan", None)
    if plan is None:
        return None
    (arena, offs), tok, params = plan
    if tok.done or not arena.intact() or any(p.grad is not None for p in params):
        return None
    flat_shapes = [shp for lvl in per_level_shapes for shp in lvl]
    views = grad_views((arena, offs), flat_shapes)
    out, k = [], 0
    for lvl in per_level_shapes:
        out.append(views[k: k + len(lvl)])
        k += len(lvl)
    return out


def _arena_done(ctx):
    plan = getattr(ctx, "arena_plan", None)
    if plan is not None:
        plan[1].done = True
        ctx.arena_plan = None


def _wait_packed(packs_bwd, device):
    """The transposed streams may have been packed on a side stream (models.vanilla_nerf.model.packed_bwd_aside): the stream this backward
    runs on -- not necessarily the forward's -- waits for that pack."""
    seen = set()
    for b in packs_bwd:
        ev = getattr(b, "_aon_ready", None)
        if ev is not None and id(ev) not in seen:
            seen.add(id(ev))
            torch.cuda.current_stream(device).wait_event(ev)


def _check_not_released(ctx):
    """The activation planes (10-14 KB per sample) are handed back after the first backward, like autograd's own saved
    tensors without retain_graph: a second backward through the same forward gets the error autograd would give."""
    if getattr(ctx, "released", False):
        raise RuntimeError("aon render: the activation workspace of this forward was released by its first backward "
                           "(a second backward / retain_graph=True is not supported; run the forward again)")


class RenderVanilla(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rays_o, rays_d, viewdirs, near, far, white_bkgd, num_levels, t_rand, u, packs, opts, noise, *params):
        # packs: [(packed_fwd, packed_bwd)] per level; params: 24 tensors per level in ops.VANILLA_PARAM_ORDER.  The whole
        # forward is ONE C call (aon_render_fwd_train); the backward chain reads the forward stream for its head weights.
        ctx.rays_d = rays_d
        ctx.white_bkgd = white_bkgd
        ctx.num_levels = num_levels
        ctx.set_materialize_grads(False)   # acc / depth carry no gradient in training: None, not four zero-filled tensors per step
        levels, ws, ctx.geometry = ops.render_fwd_train(packs[0][0], packs[1][0] if num_levels == 2 else None, rays_o, rays_d, viewdirs, near, far,
                                                        white_bkgd, num_levels, t_rand, u, opts=opts, noise=noise)
        ctx.fused = (ws, [pk[1] for pk in packs], [pk[0] for pk in packs])
        _arena_plan(ctx, params)
        ctx.param_shapes = [tuple(p.shape) for p in params]
        return tuple(x for lvl in levels for x in lvl)

    @staticmethod
    def backward(ctx, *gouts):
        _check_not_released(ctx)
        ws, packs_bwd, packs_fwd = ctx.fused    # the whole backward is ONE C call too (aon_render_bwd)
        _wait_packed(packs_bwd, ctx.rays_d.device)
        n_per = len(ops.VANILLA_PARAM_ORDER)
        slots = _arena_grads(ctx, [ctx.param_shapes[l * n_per: (l + 1) * n_per] for l in range(ctx.num_levels)])
        n = ctx.rays_d.shape[0]
        g_rgb = [gouts[3 * l] if gouts[3 * l] is not None else torch.zeros((n, 3), dtype=torch.float32, device=ctx.rays_d.device)
                 for l in range(ctx.num_levels)]
        per_level = ops.render_bwd(ws, packs_bwd, packs_fwd, ctx.rays_d, ctx.white_bkgd, ctx.num_levels, g_rgb,
                                   [gouts[3 * l + 1] for l in range(ctx.num_levels)], [gouts[3 * l + 2] for l in range(ctx.num_levels)],
                                   geometry=ctx.geometry, grads_out=slots)
        ctx.fused, ctx.released, ctx.geometry = None, True, None
        ops.pool_give(ws)      # (the backward's launches are enqueued: whoever takes the workspace next is ordered behind them)
        _arena_done(ctx)
        return (None,) * 12 + tuple(g[name] for g in per_level for name in ops.VANILLA_PARAM_ORDER)


class RenderLevelVanilla(torch.autograd.Function):
    """ONE level of NeRF.forward on given sample positions t (model.py:175-193: cast + encode + NeRFMLP + activations +
    volumetric_rendering), from the stage-level training entry points: aon_mlp_fwd_train -> aon_composite | aon_composite_bwd ->
    aon_mlp_bwd_chain -> aon_vanilla_wgrad.  The drop-in NeRF loops it for `num_levels > 2` in training (the two-call fused step
    covers one or two levels); the weights it returns feed the next level's inverse CDF and carry no gradient (helper.py:249)."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, viewdirs, t_vals, white_bkgd, packed_fwd, packed_bwd, *params):
        raw, planes, masks = ops.mlp_fwd_train(packed_fwd, rays_o, rays_d, viewdirs, t_vals)
        comp, acc, weights, depth = ops.composite_raw(raw, t_vals, rays_d, white_bkgd, ops.ACT_VANILLA, True)
        ctx.keep = (raw, planes, masks, t_vals, rays_d, packed_fwd, packed_bwd)
        ctx.white_bkgd = white_bkgd
        ctx.mark_non_differentiable(weights)
        return comp, acc, depth, weights

    @staticmethod
    def backward(ctx, g_rgb, g_acc, g_depth, _g_weights):
        _check_not_released(ctx)
        raw, planes, masks, t_vals, rays_d, packed_fwd, packed_bwd = ctx.keep
        n = t_vals.shape[0]
        if g_rgb is None:
            g_rgb = torch.zeros((n, 3), dtype=torch.float32, device=t_vals.device)
        d_raw = ops.composite_bwd(raw, t_vals, rays_d, g_rgb, g_acc, g_depth, ctx.white_bkgd, ops.ACT_VANILLA, ops.plane_samples(planes))
        dplanes = ops.mlp_bwd_chain(packed_bwd, packed_fwd, d_raw, masks, planes.shape)
        grads = ops.vanilla_wgrad(planes, dplanes, d_raw, packed_bwd)
        ctx.keep, ctx.released = None, True
        return (None,) * 7 + tuple(grads[name] for name in ops.VANILLA_PARAM_ORDER)


class RenderGeneral(torch.autograd.Function):
    """NeRF.forward with a NeRFMLP of non-default geometry: layer-wise GEMM engine (aon_grender_fwd_train / aon_grender_bwd)."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, viewdirs, near, far, white_bkgd, num_levels, t_rand, u, geom, opts, noise, *params):
        n_per = len(geom.param_order)
        ctx.geom, ctx.rays_d, ctx.white_bkgd, ctx.num_levels = geom, rays_d, white_bkgd, num_levels
        # The backward reads the parameter storages again (W^T of the data chain) while the activations in the workspace come from the
        # forward-time weights: save_for_backward, so that an in-place update between this forward and its backward (two live graphs
        # with an optimizer.step() between their backward calls) raises autograd's version error instead of silently mixing two
        # sets of weights (round 3 kept p.detach() aliases, which bypass the check).
        ctx.save_for_backward(*params)
        pd = [dict(zip(geom.param_order, [p.detach() for p in params[l * n_per: (l + 1) * n_per]])) for l in range(num_levels)]
        levels, ctx.ws, ctx.geometry = ops.grender_fwd_train(geom, pd[0], pd[1] if num_levels == 2 else None, rays_o, rays_d,
                                                             viewdirs, near, far, white_bkgd, num_levels, t_rand, u, opts=opts, noise=noise)
        return tuple(x for lvl in levels for x in lvl)

    @staticmethod
    def backward(ctx, *gouts):
        _check_not_released(ctx)
        n = ctx.rays_d.shape[0]
        g_rgb = [gouts[3 * l] if gouts[3 * l] is not None else torch.zeros((n, 3), dtype=torch.float32, device=ctx.rays_d.device)
                 for l in range(ctx.num_levels)]
        n_per = len(ctx.geom.param_order)
        saved = ctx.saved_tensors   # (raises if a parameter was modified in place since the forward)
        params = [dict(zip(ctx.geom.param_order, saved[l * n_per: (l + 1) * n_per])) for l in range(ctx.num_levels)]
        per_level = ops.grender_bwd(ctx.geom, ctx.ws, params, ctx.rays_d, ctx.white_bkgd, ctx.num_levels, g_rgb,
                                    [gouts[3 * l + 1] for l in range(ctx.num_levels)], [gouts[3 * l + 2] for l in range(ctx.num_levels)],
                                    ctx.geometry)
        ops.pool_give(ctx.ws)
        ctx.ws, ctx.released, ctx.geometry = None, True, None
        return (None,) * 12 + tuple(g[name] for g in per_level for name in ctx.geom.param_order)


class RenderArticulated(torch.autograd.Function):
    """NeRF_AE_Art.forward with gradients to the 2 x 40 MLP parameters and the three latents."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, viewdirs, near, far, white_bkgd, num_levels, t_rand, u, packs, opts, noise, lat_density, lat_color,
                lat_articulation, *params):
        # packs: per level (packed_fwd, small, packed_bwd); params: 40 tensors per level in ops.ART_PARAM_ORDER
        ctx.rays_d, ctx.white_bkgd, ctx.num_levels = rays_d, white_bkgd, num_levels
        ctx.set_materialize_grads(False)   # acc / depth carry no gradient in training: None, not four zero-filled tensors per step
        ctx.lat_shapes = (lat_density.shape, lat_color.shape, lat_articulation.shape)
        # parameters and latents are read again by the backward (latent columns: dW = db (x) latent, d latent = W^T db): saved the
        # autograd way, so an in-place update in between raises instead of mixing two sets of weights (as RenderGeneral)
        ctx.save_for_backward(lat_density, lat_color, lat_articulation, *params)
        levels, ws, ctx.geometry = ops.render_fwd_train(packs[0][0], packs[1][0] if num_levels == 2 else None, rays_o, rays_d, viewdirs, near, far,
                                                        white_bkgd, num_levels, t_rand, u, small_c=packs[0][1],
                                                        small_f=packs[1][1] if num_levels == 2 else None, opts=opts, noise=noise)
        ctx.fused = (ws, [pk[2] for pk in packs], [pk[1] for pk in packs])   # ONE C call (aon_art_render_fwd_train)
        _arena_plan(ctx, params)
        return tuple(x for lvl in levels for x in lvl)

    @staticmethod
    def backward(ctx, *gouts):
        n_per = len(ops.ART_PARAM_ORDER)
        _check_not_released(ctx)
        ws, packs_bwd, smalls = ctx.fused       # the whole backward in ONE C call (aon_art_render_bwd)
        _wait_packed(packs_bwd, ctx.rays_d.device)
        n = ctx.rays_d.shape[0]
        g_rgb = [gouts[3 * l] if gouts[3 * l] is not None else torch.zeros((n, 3), dtype=torch.float32, device=ctx.rays_d.device)
                 for l in range(ctx.num_levels)]
        saved = ctx.saved_tensors   # (raises if a parameter or latent was modified in place since the forward)
        latents = {"density": saved[0], "color": saved[1], "articulation": saved[2]}
        params = [dict(zip(ops.ART_PARAM_ORDER, saved[3 + l * n_per: 3 + (l + 1) * n_per])) for l in range(ctx.num_levels)]
        slots = _arena_grads(ctx, [[tuple(t.shape) for t in saved[3 + l * n_per: 3 + (l + 1) * n_per]] for l in range(ctx.num_levels)])
        per_level, g_lat = ops.art_render_bwd(ws, packs_bwd, smalls, ctx.rays_d, ctx.white_bkgd, ctx.num_levels, g_rgb,
                                              [gouts[3 * l + 1] for l in range(ctx.num_levels)], [gouts[3 * l + 2] for l in range(ctx.num_levels)],
                                              params, latents, geometry=ctx.geometry, grads_out=slots)
        ctx.fused, ctx.released, ctx.geometry = None, True, None
        ops.pool_give(ws)      # (the backward's launches are enqueued: whoever takes the workspace next is ordered behind them)
        _arena_done(ctx)
        lat = tuple(g_lat[k].reshape(shp) for k, shp in zip(("density", "color", "articulation"), ctx.lat_shapes))
        return (None,) * 12 + lat + tuple(g[name] for g in per_level for name in ops.ART_PARAM_ORDER)
