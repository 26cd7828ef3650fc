"""CPU oracle for the NeRF volume-rendering hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32) *restatement* of the reference algorithm
(zubair-irshad/articulated-object-nerf).  It is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.  The product
path (``articulated-object-nerf_amd``) never imports anything from ``oracle/`` and fails loudly when the
HIP library is missing.

Parity status: PINNED.  Every function below is checked against golden vectors produced by importing
the real reference in the build container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` runs the comparison on CPU).  The reference ships no tests or golden
vectors of its own, and one third-party function on the path
(``kornia.create_meshgrid``, kornia==0.6.1, reference ``requirements.txt:3``) is absent from the image;
its published semantics ((1,H,W,2), [...,0]=x=column index, [...,1]=y=row index, un-normalised) are
restated in ``get_ray_directions`` and in the stub used to generate the goldens.

Each function cites the reference file:line it follows (paths relative to the reference root).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

HALF_PI_F32 = torch.tensor(0.5 * math.pi, dtype=torch.float32)


# --------------------------------------------------------------------------------------------------
# R1 / R2  ray generation                                   datasets/ray_utils.py:71-90, 118-159
# --------------------------------------------------------------------------------------------------
def get_ray_directions(H: int, W: int, focal: float) -> torch.Tensor:
    """datasets/ray_utils.py:71-90.  Camera-space direction per pixel, no +0.5 pixel centre:
    ((i - W/2)/focal, -(j - H/2)/focal, -1), i = column, j = row.  (H,W,3) fp32."""
    j = torch.arange(H, dtype=torch.float32)[:, None].expand(H, W)
    i = torch.arange(W, dtype=torch.float32)[None, :].expand(H, W)
    return torch.stack([(i - W / 2) / focal, -(j - H / 2) / focal, -torch.ones(H, W)], -1)


def get_rays(directions: torch.Tensor, c2w: torch.Tensor):
    """datasets/ray_utils.py:118-159 with output_view_dirs=True.  Returns (rays_o, viewdirs, rays_d),
    each (H*W,3); the reference normalises ``rays_d`` in place through its ``viewdirs`` alias
    (:146-147), so rays_d == viewdirs (unit norm).  ``radii`` (:138-143): see ray_radii below."""
    rays_d = directions @ c2w[:, :3].T
    rays_d = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    rays_o = c2w[:, 3].expand(rays_d.shape)
    rays_d = rays_d.reshape(-1, 3)
    return rays_o.reshape(-1, 3), rays_d, rays_d


def ray_radii(directions: torch.Tensor, c2w: torch.Tensor) -> torch.Tensor:
    """datasets/ray_utils.py:138-143, the 4th output of get_rays(..., output_view_dirs=True, output_radii=True):
    row-to-row distance of the un-normalised world directions, last image row copied from row H-3
    (``dx[-2:-1]`` of the (H-1)-row difference), times 2 / sqrt(12) (the sqrt of an int8 tensor is fp32)."""
    d = directions @ c2w[:, :3].T
    dx = torch.sqrt(torch.sum((d[:-1, :, :] - d[1:, :, :]) ** 2, dim=-1))
    dx = torch.cat([dx, dx[-2:-1, :]], dim=0)
    radius = dx[..., None] * 2 / torch.sqrt(torch.tensor(12, dtype=torch.int8))
    return radius.reshape(-1)


# --------------------------------------------------------------------------------------------------
# R3  stratified sampling                                   models/vanilla_nerf/helper.py:25-26,106-133
# --------------------------------------------------------------------------------------------------
def cast_rays(t_vals, origins, directions):
    """helper.py:25-26 (multiply then add, no fused multiply-add)."""
    return origins[..., None, :] + t_vals[..., None] * directions[..., None, :]


def sample_along_rays(rays_o, rays_d, num_samples, near, far, randomized, t_rand=None, lindisp=False):
    """helper.py:106-133.  ``t_rand`` (N, num_samples+1) replaces the reference's ``torch.rand`` draw (:126) so
    randomized runs are reproducible.  lindisp (:116-117): ``1.0 / near`` is a Python (double) division before it meets
    the fp32 tensor."""
    n = rays_o.shape[0]
    s = torch.linspace(0.0, 1.0, num_samples + 1)
    if lindisp:
        t_vals = 1.0 / (1.0 / near * (1.0 - s) + 1.0 / far * s)
    else:
        t_vals = near * (1.0 - s) + far * s
    if randomized:
        mids = 0.5 * (t_vals[1:] + t_vals[:-1])
        upper = torch.cat([mids, t_vals[-1:]])
        lower = torch.cat([t_vals[:1], mids])
        t_vals = lower + (upper - lower) * t_rand
    else:
        t_vals = t_vals.expand(n, num_samples + 1)
    return t_vals, cast_rays(t_vals, rays_o, rays_d)


# --------------------------------------------------------------------------------------------------
# R4  positional encoding                                   models/vanilla_nerf/helper.py:136-140
# --------------------------------------------------------------------------------------------------
def pos_enc(x, min_deg, max_deg):
    """helper.py:136-140.  [x ; sin(2^l x) (scale-major, xyz-minor) ; sin(2^l x + fp32(pi/2))]."""
    scales = torch.tensor([2.0 ** l for l in range(min_deg, max_deg)], dtype=x.dtype)
    xb = (x[..., None, :] * scales[:, None]).reshape(*x.shape[:-1], -1)
    return torch.cat([x, torch.sin(torch.cat([xb, xb + HALF_PI_F32], dim=-1))], dim=-1)


# --------------------------------------------------------------------------------------------------
# R5  vanilla NeRFMLP                                       models/vanilla_nerf/model.py:95-120
# --------------------------------------------------------------------------------------------------
def nerf_mlp(sd: dict, prefix: str, x_enc, view_enc, netdepth: int | None = None, skip_layer: int = 4, netdepth_condition: int | None = None):
    """model.py:95-120.  ``sd`` maps '<prefix>pts_linears.0.weight' ... to tensors in nn.Linear (out,in)
    layout.  x_enc (N,S,pos_size), view_enc (N,view_pos_size) -> raw_rgb (N,S,C_rgb), raw_density (N,S,C_density).  Depths
    default to the number of layers the state dict holds; widths and encoding sizes are the tensors' shapes."""
    if netdepth is None:
        netdepth = sum(1 for k in sd if k.startswith(f"{prefix}pts_linears.") and k.endswith(".weight"))
    if netdepth_condition is None:
        netdepth_condition = sum(1 for k in sd if k.startswith(f"{prefix}views_linear.") and k.endswith(".weight"))
    n, s, feat = x_enc.shape
    x = x_enc.reshape(-1, feat)
    inputs = x
    for idx in range(netdepth):
        x = F.relu(F.linear(x, sd[f"{prefix}pts_linears.{idx}.weight"], sd[f"{prefix}pts_linears.{idx}.bias"]))
        if idx % skip_layer == 0 and idx > 0:
            x = torch.cat([x, inputs], dim=-1)
    raw_density = F.linear(x, sd[f"{prefix}density_layer.weight"], sd[f"{prefix}density_layer.bias"]).reshape(n, s, -1)
    bott = F.linear(x, sd[f"{prefix}bottleneck_layer.weight"], sd[f"{prefix}bottleneck_layer.bias"])
    cond = view_enc[:, None, :].expand(n, s, view_enc.shape[-1]).reshape(-1, view_enc.shape[-1])
    x = torch.cat([bott, cond], dim=-1)
    for idx in range(netdepth_condition):   # model.py:112-114
        x = F.relu(F.linear(x, sd[f"{prefix}views_linear.{idx}.weight"], sd[f"{prefix}views_linear.{idx}.bias"]))
    raw_rgb = F.linear(x, sd[f"{prefix}rgb_layer.weight"], sd[f"{prefix}rgb_layer.bias"]).reshape(n, s, -1)
    return raw_rgb, raw_density


# --------------------------------------------------------------------------------------------------
# R8  alpha compositing                                     models/vanilla_nerf/helper.py:157-195
# --------------------------------------------------------------------------------------------------
def volumetric_rendering(rgb, density, t_vals, dirs, white_bkgd):
    """helper.py:157-195.  rgb (N,S,3), density (N,S,1) (already activated), t_vals (N,S), dirs (N,3)
    -> comp_rgb (N,3), acc (N,), weights (N,S), depth (N,)."""
    eps = 1e-10
    dists = torch.cat([t_vals[..., 1:] - t_vals[..., :-1], torch.full_like(t_vals[..., :1], 1e10)], dim=-1)
    dists = dists * torch.norm(dirs[..., None, :], dim=-1)
    alpha = 1.0 - torch.exp(-density[..., 0] * dists)
    trans = torch.cat([torch.ones_like(alpha[..., :1]), torch.cumprod(1.0 - alpha[..., :-1] + eps, dim=-1)], dim=-1)
    weights = alpha * trans
    comp_rgb = (weights[..., None] * rgb).sum(dim=-2)
    depth = (weights * t_vals).sum(dim=-1)
    depth = torch.nan_to_num(depth, float("inf"))          # helper.py:182 (NaN -> +inf)
    depth = torch.clamp(depth, torch.min(depth), torch.max(depth))  # helper.py:183 (identity)
    acc = weights.sum(dim=-1)
    if white_bkgd:
        comp_rgb = comp_rgb + (1.0 - acc[..., None])
    return comp_rgb, acc, weights, depth


# --------------------------------------------------------------------------------------------------
# R6 / R7  hierarchical inverse-CDF sampling                models/vanilla_nerf/helper.py:203-252
# --------------------------------------------------------------------------------------------------
def deterministic_u(num_samples: int) -> torch.Tensor:
    """helper.py:229: linspace(0, 1 - 2^-32, num_samples) in fp32 (the last element rounds to 1.0)."""
    return torch.linspace(0.0, 1.0 - 2.0 ** -32, num_samples)


def sorted_piecewise_constant_pdf(bins, weights, num_samples, randomized, u=None):
    """helper.py:203-243 restated with a searchsorted instead of the (N,64,128) mask/max/min
    broadcast: for every u the reference picks (bin0,cdf0) = entry at the last index with cdf <= u
    and (bin1,cdf1) = entry at the first index with cdf > u (else the last entry).  ``u`` (N,num_samples)
    replaces ``torch.rand`` (:227) when randomized."""
    eps = 1e-5
    weight_sum = weights.sum(dim=-1, keepdim=True)
    padding = torch.fmax(torch.zeros_like(weight_sum), eps - weight_sum)
    weights = weights + padding / weights.shape[-1]
    weight_sum = weight_sum + padding
    pdf = weights / weight_sum
    cdf = torch.fmin(torch.ones_like(pdf[..., :-1]), torch.cumsum(pdf[..., :-1], dim=-1))
    one = list(cdf.shape[:-1]) + [1]   # (helper.py:216-221 builds these from the shape: with ONE weight cdf[..., :1] would be empty)
    cdf = torch.cat([torch.zeros(one, dtype=cdf.dtype), cdf, torch.ones(one, dtype=cdf.dtype)], dim=-1)
    if not randomized:
        u = deterministic_u(num_samples).expand(*cdf.shape[:-1], num_samples)
    u = u.contiguous()
    last = cdf.shape[-1] - 1
    idx = torch.searchsorted(cdf.contiguous(), u, right=True)
    i0 = (idx - 1).clamp(0, last)
    i1 = idx.clamp(0, last)
    cdf0, cdf1 = torch.gather(cdf, -1, i0), torch.gather(cdf, -1, i1)
    bin0, bin1 = torch.gather(bins, -1, i0), torch.gather(bins, -1, i1)
    t = torch.clip(torch.nan_to_num((u - cdf0) / (cdf1 - cdf0), 0), 0, 1)
    return bin0 + t * (bin1 - bin0)


def sample_pdf(bins, weights, origins, directions, t_vals, num_samples, randomized, u=None):
    """helper.py:246-252: inverse-CDF draw, sort-merge with the coarse t's, cast."""
    t_samples = sorted_piecewise_constant_pdf(bins, weights, num_samples, randomized, u).detach()  # helper.py:249
    t_vals = torch.sort(torch.cat([t_vals, t_samples], dim=-1), dim=-1).values
    return t_vals, cast_rays(t_vals, origins, directions)


def aten_sum_model(x) -> "np.float32":
    """What ``torch.sum`` over a contiguous fp32 row of K elements computes on the CPU (helper.py:205 ``weights.sum(dim=-1)``;
    ATen cpu/SumKernel.cpp, AVX2 build: 8-float vectors, ILP factor 4, four-level cascade of ``multi_row_sum``) restated in
    numpy, element by element.  The general-size inverse-CDF kernel (csrc/aon_render.hip ``aten_row_sum``) follows the same
    steps; tests/test_oracle_golden.py holds this model to torch.sum for K = 1 .. 1000."""
    import numpy as np

    f32 = np.float32
    x = np.asarray(x, f32)
    K = x.shape[0]

    def multi_row_sum(load, size, width):
        cl = 0
        while (1 << cl) < size:
            cl += 1
        level_power = max(4, cl // 4)
        level_step = 1 << level_power
        level_mask = level_step - 1
        acc = [[np.zeros(width, f32) for _ in range(4)] for _ in range(4)]
        i = 0
        while i + level_step <= size:
            for _ in range(level_step):
                for k in range(4):
                    acc[0][k] = acc[0][k] + load(i, k)
                i += 1
            for j in range(1, 4):
                for k in range(4):
                    acc[j][k] = acc[j][k] + acc[j - 1][k]
                    acc[j - 1][k] = np.zeros(width, f32)
                if (i & (level_mask << (j * level_power))) != 0:
                    break
        while i < size:
            for k in range(4):
                acc[0][k] = acc[0][k] + load(i, k)
            i += 1
        for j in range(1, 4):
            for k in range(4):
                acc[0][k] = acc[0][k] + acc[j][k]
        return acc[0]

    if K < 8:   # scalar_inner_sum -> row_sum on scalars
        size_ilp = K // 4
        part = multi_row_sum(lambda i, k: x[4 * i + k: 4 * i + k + 1], size_ilp, 1)
        for i in range(size_ilp * 4, K):
            part[0] = part[0] + x[i: i + 1]
        for k in range(1, 4):
            part[0] = part[0] + part[k]
        return part[0][0]
    vec = K // 8
    size_ilp = vec // 4
    part = multi_row_sum(lambda i, k: x[(4 * i + k) * 8: (4 * i + k) * 8 + 8], size_ilp, 8)
    for i in range(size_ilp * 4, vec):
        part[0] = part[0] + x[i * 8: i * 8 + 8]
    for k in range(1, 4):
        part[0] = part[0] + part[k]
    fin = f32(0)
    for k in range(vec * 8, K):
        fin = f32(fin + x[k])
    for k in range(8):
        fin = f32(fin + part[0][k])
    return fin


# --------------------------------------------------------------------------------------------------
# R9  NeRF.forward                                          models/vanilla_nerf/model.py:147-199
# --------------------------------------------------------------------------------------------------
def nerf_forward(sd, rays, randomized, white_bkgd, near, far, num_levels=2, min_deg_point=0,
                 max_deg_point=10, deg_view=4, num_coarse_samples=64, num_fine_samples=128,
                 t_rand=None, u=None, return_aux=False, lindisp=False, noise_std=0.0, noise=None, skip_layer=4):
    """model.py:147-199.  ``sd`` uses the reference's key names ('coarse_mlp.pts_linears.0.weight', ...).
    Returns [(comp_rgb, acc, depth)_coarse, (comp_rgb, acc, depth)_fine]; with ``return_aux`` also a
    dict of intermediates per level (t_vals, raw_rgb, raw_sigma, weights).  ``noise``: per-level (N,S,1) tensors replacing
    ``torch.rand_like(raw_sigma)`` (:184)."""
    ret, aux = [], []
    t_vals = weights = None
    for i_level in range(num_levels):
        if i_level == 0:
            t_vals, samples = sample_along_rays(rays["rays_o"], rays["rays_d"], num_coarse_samples, near, far,
                                                randomized, t_rand, lindisp)
            prefix = "coarse_mlp."
        else:
            t_mids = 0.5 * (t_vals[..., 1:] + t_vals[..., :-1])
            t_vals, samples = sample_pdf(t_mids, weights[..., 1:-1], rays["rays_o"], rays["rays_d"], t_vals,
                                         num_fine_samples, randomized, u)
            prefix = "fine_mlp."
        samples_enc = pos_enc(samples, min_deg_point, max_deg_point)
        viewdirs_enc = pos_enc(rays["viewdirs"], 0, deg_view)
        raw_rgb, raw_sigma = nerf_mlp(sd, prefix, samples_enc, viewdirs_enc, skip_layer=skip_layer)
        if noise_std > 0 and randomized:                                           # model.py:183-184
            raw_sigma = raw_sigma + noise[i_level].reshape(raw_sigma.shape) * noise_std
        rgb = torch.sigmoid(raw_rgb)
        sigma = F.relu(raw_sigma)
        comp_rgb, acc, weights, depth = volumetric_rendering(rgb, sigma, t_vals, rays["rays_d"], white_bkgd)
        ret.append((comp_rgb, acc, depth))
        aux.append({"t_vals": t_vals, "raw_rgb": raw_rgb, "raw_sigma": raw_sigma, "weights": weights})
    return (ret, aux) if return_aux else ret


# --------------------------------------------------------------------------------------------------
# R10  articulated NeRFMLP              models/vanilla_nerf/model_autodecoder.py:172-239
# --------------------------------------------------------------------------------------------------
def art_mlp(sd: dict, prefix: str, pos, view_enc, latents: dict, min_deg_point: int = 0, max_deg_point: int = 10):
    """model_autodecoder.py:172-239 (deformation_mlp=True, enc_after=True, embed_deg=False).
    pos (N,S,3) raw sample positions, view_enc (N,27), latents = {"density": (1,128), "color": (1,128),
    "articulation": (1,32)} broadcast to every sample (einops.repeat, :186-194)."""
    n, s, _ = pos.shape
    p = pos.reshape(-1, 3)
    bn = p.shape[0]
    shape = latents["density"].expand(bn, -1)
    app = latents["color"].expand(bn, -1)
    art = latents["articulation"].expand(bn, -1)
    lin = lambda name, x: F.linear(x, sd[f"{prefix}{name}.weight"], sd[f"{prefix}{name}.bias"])  # noqa: E731
    x = torch.cat([p, shape, art], -1)
    for i in range(4):
        x = F.relu(lin(f"deformations_linear.{i}", x))
    x = lin("deformation_layer", x) + p
    x = pos_enc(x, min_deg_point, max_deg_point)                      # model_autodecoder.py:207-212
    x = torch.cat([x, shape], -1)
    inputs = x
    for idx in range(8):
        x = F.relu(lin(f"pts_linears.{idx}", x))
        if idx % 4 == 0 and idx > 0:
            x = torch.cat([x, inputs], dim=-1)
    raw_density = lin("density_layer", x).reshape(n, s, 1)
    bott = lin("bottleneck_layer", x)
    cond = view_enc[:, None, :].expand(n, s, view_enc.shape[-1]).reshape(-1, view_enc.shape[-1])
    x = torch.cat([bott, cond, app], dim=-1)
    for i in range(4):
        x = F.relu(lin(f"views_linear.{i}", x))
    raw_rgb = lin("rgb_layer", x).reshape(n, s, 3)
    return raw_rgb, raw_density


# --------------------------------------------------------------------------------------------------
# R11  NeRF_AE_Art.forward              models/vanilla_nerf/model_autodecoder.py:278-337
# --------------------------------------------------------------------------------------------------
def nerf_ae_art_forward(sd, rays, randomized, white_bkgd, near, far, latents, num_levels=2, t_rand=None, u=None,
                        return_aux=False, num_coarse_samples=64, num_fine_samples=128, lindisp=False, noise_std=0.0, noise=None,
                        rgb_padding=0.001, density_bias=-1.0, min_deg_point=0, max_deg_point=10, deg_view=4):
    ret, aux = [], []
    t_vals = weights = None
    for i_level in range(num_levels):
        if i_level == 0:
            t_vals, samples = sample_along_rays(rays["rays_o"], rays["rays_d"], num_coarse_samples, near, far, randomized, t_rand, lindisp)
            prefix = "coarse_mlp."
        else:
            t_mids = 0.5 * (t_vals[..., 1:] + t_vals[..., :-1])
            t_vals, samples = sample_pdf(t_mids, weights[..., 1:-1], rays["rays_o"], rays["rays_d"], t_vals, num_fine_samples, randomized, u)
            prefix = "fine_mlp."
        viewdirs_enc = pos_enc(rays["viewdirs"], 0, deg_view)                      # :315
        raw_rgb, raw_sigma = art_mlp(sd, prefix, samples, viewdirs_enc, latents, min_deg_point, max_deg_point)  # samples un-encoded (:306-307)
        if noise_std > 0 and randomized:                                           # :318-319
            raw_sigma = raw_sigma + noise[i_level].reshape(raw_sigma.shape) * noise_std
        rgb = torch.sigmoid(raw_rgb) * (1 + 2 * rgb_padding) - rgb_padding        # :321-322
        sigma = F.softplus(raw_sigma + density_bias)                               # :323
        comp_rgb, acc, weights, depth = volumetric_rendering(rgb, sigma, t_vals, rays["rays_d"], white_bkgd)
        ret.append((comp_rgb, acc, depth))
        aux.append({"t_vals": t_vals, "raw_rgb": raw_rgb, "raw_sigma": raw_sigma, "weights": weights})
    return (ret, aux) if return_aux else ret


# --------------------------------------------------------------------------------------------------
# R12  CodeLibraryArticulated           models/code_library.py:36-71
# --------------------------------------------------------------------------------------------------
def code_library(sd, instance_id, articulation_id, is_test=False):
    """Embedding lookups; at test time the 10 learned articulation codes are expanded to 19 = 10 + 9 mid-points
    (code_library.py:55-71) and indexed by articulation_id."""
    out = {"density": sd["embedding_instance_shape.weight"][instance_id],
           "color": sd["embedding_instance_appearance.weight"][instance_id]}
    art = sd["embedding_instance_articulation.weight"]
    if is_test:
        table = torch.zeros(19, art.shape[1])
        table[0::2] = art
        table[1::2] = (art[:-1] + art[1:]) / 2
        out["articulation"] = table[articulation_id]
    else:
        out["articulation"] = art[articulation_id]
    return out


# --------------------------------------------------------------------------------------------------
# R13  losses / metrics                   helper.py:17-22, models/interface.py:54-74
# --------------------------------------------------------------------------------------------------
def img2mse(x, y):
    return torch.mean((x - y) ** 2)


def mse2psnr(x):
    return -10.0 * torch.log(x) / math.log(10.0)


def psnr_legacy(pred, gt):
    return -10.0 * torch.log10(torch.mean((pred - gt) ** 2))


def psnr_each(preds, gts):
    out = []
    for p, g in zip(preds, gts):
        mse = torch.mean((torch.clip(p, 0, 1) - torch.clip(g, 0, 1)) ** 2)
        out.append(-10.0 * torch.log(mse) / math.log(10.0))
    return torch.stack(out)
