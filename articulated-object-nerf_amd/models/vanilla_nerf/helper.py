"""Host-side mirror of the reference's ``models/vanilla_nerf/helper.py`` (same names, argument meaning and
return structure), every function backed by a HIP kernel through the C ABI (``aon_amd.ops``).

Differences a caller can observe, all additive:
  * the random draws the reference takes from ``torch.rand`` inside ``sample_along_rays`` (helper.py:126) and
    ``sorted_piecewise_constant_pdf`` (helper.py:227) can be supplied (``t_rand=`` / ``u=``) for reproducibility;
    when omitted and ``randomized`` is true they are drawn with ``torch.rand`` on the inputs' device.
``lindisp``, any ``num_samples`` and any ``float_min_eps`` are served (round 3): the reference geometry (64 bins, 128 draws)
by the specialised kernels, everything else by ``aon_sample_pdf_n`` -- same bits where they overlap.
"""
from __future__ import annotations

import math

import torch

from ... import ops


def img2mse(x, y):
    """helper.py:17-18"""
    return torch.mean((x - y) ** 2)


def mse2psnr(x):
    """helper.py:21-22"""
    return -10.0 * torch.log(x) / math.log(10.0)


class _TrainLoss(torch.autograd.Function):
    """loss0 + loss1 (+ reg) with their gradients as two launches (ops.train_loss_fwd / _bwd) instead of ~47 tiny torch kernels."""

    @staticmethod
    def forward(ctx, rgb_c, rgb_f, target, reg_scale, *latents):
        loss, stats = ops.train_loss_fwd(rgb_c, rgb_f, target, latents, reg_scale)
        ctx.save_for_backward(*[t for t in (rgb_c, rgb_f, target) + tuple(latents) if t is not None])
        ctx.layout = (rgb_c is not None, len(latents), reg_scale)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)     # the logged stats carry no gradient: None, not a zero-filled tensor per step
        return loss.reshape(()), stats

    @staticmethod
    def backward(ctx, grad_loss, _grad_stats):
        has_c, n_lat, reg_scale = ctx.layout
        if grad_loss is None:
            return (None,) * (4 + n_lat)
        saved = list(ctx.saved_tensors)
        rgb_c = saved.pop(0) if has_c else None
        rgb_f, target = saved.pop(0), saved.pop(0)
        d_c, d_f, d_l = ops.train_loss_bwd(rgb_c, rgb_f, target, saved, reg_scale, grad_loss.reshape(1))
        return (d_c, d_f, None, None) + tuple(d_l[:n_lat])


def train_loss(rendered, target, latents=(), reg_scale=1e-4):
    """The training steps' loss lines in one piece: model.py:271-273 (``loss0 + loss1``) and model_autodecoder.py:460-466
    (``+ 1e-4 * (mean ||shape|| + mean ||appearance|| + mean ||articulation||)``, ``torch.norm(code, dim=0)`` of the one-row codes).
    ``rendered``: the model's output list (one or two levels of ``(rgb, acc, depth)``).  Returns ``(loss, stats)``; ``loss`` carries the
    autograd graph to the rendered colours and the codes, ``stats`` = ``[loss0, loss1, reg, loss, psnr0, psnr1, 0, 0]`` (detached, for
    the logs: ``mse2psnr`` of the two levels included).  Codes with more than one row fall back to the reference's torch formula."""
    rgb_f = rendered[-1][0]
    rgb_c = rendered[0][0] if len(rendered) > 1 else None
    latents = tuple(latents)
    if any(c.dim() != 2 or c.shape[0] != 1 for c in latents) or len(latents) > 3 or len(rendered) > 2:
        loss_levels = [img2mse(r[0], target) for r in rendered]
        reg = reg_scale * sum(torch.mean(torch.norm(c, dim=0)) for c in latents) if latents else torch.zeros((), device=target.device)
        loss = sum(reversed(loss_levels)) + reg if latents else sum(loss_levels)
        l0 = loss_levels[0].detach() if len(rendered) > 1 else torch.zeros((), device=target.device)
        l1 = loss_levels[-1].detach()
        zero = torch.zeros((), device=target.device)
        return loss, torch.stack([l0, l1, reg.detach(), loss.detach(), mse2psnr(l0), mse2psnr(l1), zero, zero])
    return _TrainLoss.apply(rgb_c, rgb_f, target, float(reg_scale), *latents)


def cast_rays(t_vals, origins, directions):
    """helper.py:25-26"""
    return ops.cast_rays(t_vals, origins, directions)


def sample_along_rays(rays_o, rays_d, num_samples, near, far, randomized, lindisp, t_rand=None):
    """helper.py:106-133 -> (t_vals (N,num_samples+1), coords (N,num_samples+1,3))"""
    if randomized and t_rand is None:
        t_rand = torch.rand((rays_o.shape[0], num_samples + 1), device=rays_o.device)
    return ops.sample_along_rays(rays_o, rays_d, num_samples, near, far, t_rand if randomized else None, lindisp=bool(lindisp))


def pos_enc(x, min_deg, max_deg):
    """helper.py:136-140"""
    return ops.pos_enc(x, min_deg, max_deg)


def volumetric_rendering(rgb, density, t_vals, dirs, white_bkgd, nocs=None):
    """helper.py:157-195 -> (comp_rgb, acc, weights, depth), or with ``nocs`` (n,S,3) -> (comp_rgb, acc, weights, comp_nocs)
    (:191-193): comp_nocs = sum_s weights * nocs is the same compositing kernel run on `nocs` in place of `rgb` without the white
    background (the weights depend on density and t only)."""
    comp_rgb, acc, weights, depth = ops.volumetric_rendering(rgb, density, t_vals, dirs, white_bkgd)
    if nocs is not None:
        comp_nocs = ops.volumetric_rendering(nocs, density, t_vals, dirs, False)[0]
        return comp_rgb, acc, weights, comp_nocs
    return comp_rgb, acc, weights, depth


def sorted_piecewise_constant_pdf(bins, weights, num_samples, randomized, float_min_eps=2 ** -32, u=None):
    """helper.py:203-243 -> samples (N,num_samples)"""
    if randomized and u is None:
        u = torch.rand((bins.shape[0], num_samples), device=bins.device)
    if not randomized:   # helper.py:229; the default eps takes the cached vector
        u = None if float_min_eps == 2 ** -32 else torch.linspace(0.0, 1.0 - float_min_eps, num_samples).to(bins.device)
    return ops.sorted_piecewise_constant_pdf(bins, weights, u, num_samples=num_samples)


def sample_pdf(bins, weights, origins, directions, t_vals, num_samples, randomized, u=None):
    """helper.py:246-252 -> (t_vals (N, S + num_samples), coords (N, S + num_samples, 3))"""
    if randomized and u is None:
        u = torch.rand((bins.shape[0], num_samples), device=bins.device)
    if (t_vals.shape[1], bins.shape[1], num_samples) == (65, 64, 128):
        t_fine = ops.sample_pdf_t(t_vals, weights, u if randomized else None, bins=bins)
    else:
        t_fine = ops.sample_pdf_t_n(t_vals, weights, num_samples, u if randomized else None, bins=bins)
    return t_fine, ops.cast_rays(t_fine, origins, directions)


def get_learning_rate(optimizer):
    """helper.py:198-200: the learning rate of the (first) parameter group, as logged by the training steps."""
    for group in optimizer.param_groups:
        return group["lr"]


def get_parameters(models):
    """helper.py:143-154: flat parameter list of a module, or of a list / dict of modules."""
    if isinstance(models, dict):
        models = list(models.values())
    if not isinstance(models, (list, tuple)):
        models = [models]
    params = []
    for m in models:
        params += get_parameters(m) if isinstance(m, (list, tuple, dict)) else list(m.parameters())
    return params
