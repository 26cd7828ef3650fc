"""torch.autograd glue for the HIP render path (SURVEY 8(a) R14, 8(b) "Ownership"): ``NeRF.forward`` under
``torch.is_grad_enabled()`` returns tensors whose graph reaches the module's parameters, so the reference's
``training_step`` (``loss.backward()``, model.py:264-282) works unchanged.  Forward and backward are both HIP kernels
(no eager-PyTorch math): fused forward with activation planes -> composite backward -> fused data-gradient chain ->
split-N weight-gradient GEMMs.  Gradients reach only the MLP parameters: the inverse-CDF draws are detached
(helper.py:249) and rays are data."""
from __future__ import annotations

import torch

from . import ops


def _check_not_released(ctx):
    """The activation planes (10-14 KB per sample) are handed back after the first backward, like autograd's own saved
    tensors without retain_graph: a second backward through the same forward gets the error autograd would give."""
    if getattr(ctx, "released", False):
        raise RuntimeError("aon render: the activation workspace of this forward was released by its first backward "
                           "(a second backward / retain_graph=True is not supported; run the forward again)")


class RenderVanilla(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rays_o, rays_d, viewdirs, near, far, white_bkgd, num_levels, t_rand, u, packs, *params):
        # packs: [(packed_fwd, packed_bwd[, packed_bf16x3, packed_bwd_bf16x3])] per level; params: 24 tensors per level in
        # ops.VANILLA_PARAM_ORDER.  The optional elements select the bf16x3 training forward / backward chain; the fp32
        # forward stream is still read by the backward chain for its head weights.
        ctx.rays_d = rays_d
        ctx.white_bkgd = white_bkgd
        ctx.num_levels = num_levels
        if all(len(pk) == 2 for pk in packs):   # exact-fp32 engine: the whole forward is ONE C call (aon_render_fwd_train)
            levels, ws = ops.render_fwd_train(packs[0][0], packs[1][0] if num_levels == 2 else None, rays_o, rays_d, viewdirs, near, far,
                                              white_bkgd, num_levels, t_rand, u)
            ctx.fused = (ws, [pk[1] for pk in packs], [pk[0] for pk in packs])
            return tuple(x for lvl in levels for x in lvl)
        ctx.fused = None
        saved, outs = [], []
        t_vals = weights = None
        for lvl in range(num_levels):
            packed_fwd, packed_bwd = packs[lvl][:2]
            packed_bf = packs[lvl][2] if len(packs[lvl]) > 2 else None
            packed_bwd_bf = packs[lvl][3] if len(packs[lvl]) > 3 else None
            if lvl == 0:
                t_vals, _ = ops.sample_along_rays(rays_o, rays_d, 64, near, far, t_rand, want_coords=False)
            else:
                t_vals = ops.sample_pdf_t(t_vals, weights, u)
            if packed_bf is not None:
                raw, planes, masks = ops.mlp_fwd_train(packed_bf, rays_o, rays_d, viewdirs, t_vals, engine="bf16x3")
            else:
                raw, planes, masks = ops.mlp_fwd_train(packed_fwd, rays_o, rays_d, viewdirs, t_vals)
            rgb, acc, weights, depth = ops.composite_raw(raw, t_vals, rays_d, white_bkgd, ops.ACT_VANILLA, want_weights=True)
            outs += [rgb, acc, depth]
            saved.append((raw, t_vals, planes, masks, packed_fwd, packed_bwd, packed_bwd_bf))
        ctx.saved = saved
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        grads = []
        _check_not_released(ctx)
        if ctx.fused is not None:               # ... and so is the whole backward (aon_render_bwd)
            ws, packs_bwd, packs_fwd = ctx.fused
            n = ctx.rays_d.shape[0]
            g_rgb = [gouts[3 * l] if gouts[3 * l] is not None else torch.zeros((n, 3), dtype=torch.float32, device=ctx.rays_d.device)
                     for l in range(ctx.num_levels)]
            per_level = ops.render_bwd(ws, packs_bwd, packs_fwd, ctx.rays_d, ctx.white_bkgd, ctx.num_levels, g_rgb,
                                       [gouts[3 * l + 1] for l in range(ctx.num_levels)], [gouts[3 * l + 2] for l in range(ctx.num_levels)])
            ctx.fused, ctx.released = None, True
            return (None,) * 10 + tuple(g[name] for g in per_level for name in ops.VANILLA_PARAM_ORDER)
        for lvl in range(ctx.num_levels):
            raw, t_vals, planes, masks, packed_fwd, packed_bwd, packed_bwd_bf = ctx.saved[lvl]
            g_rgb, g_acc, g_depth = gouts[3 * lvl: 3 * lvl + 3]
            if g_rgb is None:
                g_rgb = torch.zeros((t_vals.shape[0], 3), dtype=torch.float32, device=t_vals.device)
            d_raw = ops.composite_bwd(raw, t_vals, ctx.rays_d, g_rgb.contiguous(), g_acc, g_depth, ctx.white_bkgd, ops.ACT_VANILLA,
                                      planes.shape[1])
            if packed_bwd_bf is not None:
                dplanes = ops.mlp_bwd_chain(packed_bwd_bf, packed_fwd, d_raw, masks, planes.shape, engine="bf16x3")
            else:
                dplanes = ops.mlp_bwd_chain(packed_bwd, packed_fwd, d_raw, masks, planes.shape)
            g = ops.vanilla_wgrad(planes, dplanes, d_raw)
            grads += [g[name] for name in ops.VANILLA_PARAM_ORDER]
            del dplanes
        ctx.saved, ctx.released = None, True
        return (None,) * 10 + tuple(grads)


class RenderArticulated(torch.autograd.Function):
    """NeRF_AE_Art.forward with gradients to the 2 x 40 MLP parameters and the three latents."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, viewdirs, near, far, white_bkgd, num_levels, t_rand, u, packs, lat_density, lat_color,
                lat_articulation, *params):
        # packs: per level (packed_fwd, small, packed_bwd); params: 40 tensors per level in ops.ART_PARAM_ORDER
        ctx.rays_d, ctx.white_bkgd, ctx.num_levels = rays_d, white_bkgd, num_levels
        ctx.latents = {"density": lat_density.detach(), "color": lat_color.detach(), "articulation": lat_articulation.detach()}
        ctx.lat_shapes = (lat_density.shape, lat_color.shape, lat_articulation.shape)
        ctx.params = [p.detach() for p in params]
        if all(len(pk) == 3 for pk in packs):   # exact-fp32 engine: ONE C call (aon_art_render_fwd_train)
            levels, ws = ops.render_fwd_train(packs[0][0], packs[1][0] if num_levels == 2 else None, rays_o, rays_d, viewdirs, near, far,
                                              white_bkgd, num_levels, t_rand, u, small_c=packs[0][1], small_f=packs[1][1] if num_levels == 2 else None)
            ctx.fused = (ws, [pk[2] for pk in packs], [pk[1] for pk in packs])
            return tuple(x for lvl in levels for x in lvl)
        ctx.fused = None
        saved, outs = [], []
        t_vals = weights = None
        for lvl in range(num_levels):
            packed_fwd, small, packed_bwd = packs[lvl][:3]
            packed_bf = packs[lvl][3] if len(packs[lvl]) > 3 else None   # selects the bf16x3 training forward
            packed_bwd_bf = packs[lvl][4] if len(packs[lvl]) > 4 else None   # ... and the bf16x3 backward chain
            if lvl == 0:
                t_vals, _ = ops.sample_along_rays(rays_o, rays_d, 64, near, far, t_rand, want_coords=False)
            else:
                t_vals = ops.sample_pdf_t(t_vals, weights, u)
            if packed_bf is not None:
                raw, planes, masks = ops.art_mlp_fwd_train(packed_bf, small, rays_o, rays_d, viewdirs, t_vals, engine="bf16x3")
            else:
                raw, planes, masks = ops.art_mlp_fwd_train(packed_fwd, small, rays_o, rays_d, viewdirs, t_vals)
            rgb, acc, weights, depth = ops.composite_raw(raw, t_vals, rays_d, white_bkgd, ops.ACT_ARTICULATED, want_weights=True)
            outs += [rgb, acc, depth]
            saved.append((raw, t_vals, planes, masks, small, packed_bwd, packed_bwd_bf))
        ctx.saved = saved
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        grads = []
        g_lat_tot = None
        n_per = len(ops.ART_PARAM_ORDER)
        _check_not_released(ctx)
        if ctx.fused is not None:               # the whole backward in ONE C call (aon_art_render_bwd)
            ws, packs_bwd, smalls = ctx.fused
            n = ctx.rays_d.shape[0]
            g_rgb = [gouts[3 * l] if gouts[3 * l] is not None else torch.zeros((n, 3), dtype=torch.float32, device=ctx.rays_d.device)
                     for l in range(ctx.num_levels)]
            params = [dict(zip(ops.ART_PARAM_ORDER, ctx.params[l * n_per: (l + 1) * n_per])) for l in range(ctx.num_levels)]
            per_level, g_lat = ops.art_render_bwd(ws, packs_bwd, smalls, ctx.rays_d, ctx.white_bkgd, ctx.num_levels, g_rgb,
                                                  [gouts[3 * l + 1] for l in range(ctx.num_levels)], [gouts[3 * l + 2] for l in range(ctx.num_levels)],
                                                  params, ctx.latents)
            ctx.fused, ctx.released = None, True
            lat = tuple(g_lat[k].reshape(shp) for k, shp in zip(("density", "color", "articulation"), ctx.lat_shapes))
            return (None,) * 10 + lat + tuple(g[name] for g in per_level for name in ops.ART_PARAM_ORDER)
        for lvl in range(ctx.num_levels):
            raw, t_vals, planes, masks, small, packed_bwd, packed_bwd_bf = ctx.saved[lvl]
            g_rgb, g_acc, g_depth = gouts[3 * lvl: 3 * lvl + 3]
            if g_rgb is None:
                g_rgb = torch.zeros((t_vals.shape[0], 3), dtype=torch.float32, device=t_vals.device)
            d_raw = ops.composite_bwd(raw, t_vals, ctx.rays_d, g_rgb.contiguous(), g_acc, g_depth, ctx.white_bkgd, ops.ACT_ARTICULATED,
                                      planes.shape[1])
            if packed_bwd_bf is not None:
                dplanes, dxp = ops.art_bwd_chain(packed_bwd_bf, small, d_raw, masks, planes, engine="bf16x3")
            else:
                dplanes, dxp = ops.art_bwd_chain(packed_bwd, small, d_raw, masks, planes)
            params = dict(zip(ops.ART_PARAM_ORDER, ctx.params[lvl * n_per: (lvl + 1) * n_per]))
            g, g_lat = ops.art_wgrad(planes, dplanes, d_raw, dxp, params, ctx.latents)
            grads += [g[name] for name in ops.ART_PARAM_ORDER]
            g_lat_tot = g_lat if g_lat_tot is None else {k: g_lat_tot[k] + g_lat[k] for k in g_lat}
            del dplanes
        ctx.saved, ctx.released = None, True
        lat = tuple(g_lat_tot[k].reshape(shp) for k, shp in zip(("density", "color", "articulation"), ctx.lat_shapes))
        return (None,) * 10 + lat + tuple(grads)
