"""Deterministic synthetic inputs for tests and benchmarks (no dataset, no checkpoint, no network).

Everything here is generated from ``numpy.random.Generator(PCG64(seed))`` so that the very same
weights / poses / rays can be rebuilt bit-for-bit on any machine (the golden fixtures under
``tests/golden`` were produced from these generators and the imported reference; see
``tests/golden/make_golden.py``).

Camera model and near/far follow the reference's data generator and dataset:
  * fovy = 35 deg  -> focal = 0.5*H/tan(17.5 deg) (= 761.18 at H=480)   datagen/data_gen.py:60-67
  * look-at-origin OpenGL poses on a sphere of radius 4                  datasets/sapien_multi.py:29-72
  * near = 2.0, far = 6.0                                                datasets/sapien.py:72-73
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch

NEAR = 2.0
FAR = 6.0


def focal_from_fovy(H: int, fovy_deg: float = 35.0) -> float:
    return 0.5 * H / math.tan(0.5 * math.radians(fovy_deg))


def look_at_pose(radius: float = 4.0, azim_deg: float = 30.0, elev_deg: float = 30.0) -> torch.Tensor:
    """(3,4) fp32 camera-to-world matrix, OpenGL convention (camera looks down -z, +y up),
    positioned on a sphere of ``radius`` around the origin and looking at the origin."""
    az, el = math.radians(azim_deg), math.radians(elev_deg)
    eye = np.array(
        [radius * math.cos(el) * math.cos(az), radius * math.cos(el) * math.sin(az), radius * math.sin(el)]
    )
    fwd = -eye / np.linalg.norm(eye)  # camera -z axis points at the origin
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    true_up = np.cross(right, fwd)
    c2w = np.stack([right, true_up, -fwd, eye], axis=1)  # columns: x, y, z axes, origin
    return torch.from_numpy(c2w.astype(np.float32))


def _uniform(rng: np.random.Generator, shape, bound: float) -> torch.Tensor:
    return torch.from_numpy(rng.uniform(-bound, bound, size=shape).astype(np.float32))


# (name, out_features, in_features, init) for one vanilla NeRFMLP; reference models/vanilla_nerf/model.py:39-93
def vanilla_mlp_layout(netwidth: int = 256, netwidth_condition: int = 128, pos_size: int = 63,
                       view_pos_size: int = 27, netdepth: int = 8, skip_layer: int = 4):
    layout = [("pts_linears.0", netwidth, pos_size, "xavier")]
    for idx in range(netdepth - 1):
        fan_in = netwidth + pos_size if (idx % skip_layer == 0 and idx > 0) else netwidth
        layout.append((f"pts_linears.{idx + 1}", netwidth, fan_in, "xavier"))
    layout.append(("views_linear.0", netwidth_condition, netwidth + view_pos_size, "kaiming"))
    layout.append(("bottleneck_layer", netwidth, netwidth, "xavier"))
    layout.append(("density_layer", 1, netwidth, "xavier"))
    layout.append(("rgb_layer", 3, netwidth_condition, "xavier"))
    return layout


def make_mlp_state(rng: np.random.Generator, layout, density_scale: float = 1.0) -> "OrderedDict[str, torch.Tensor]":
    """Weights with the same distributions as the reference's initialisers (xavier-uniform on every
    weight except views_linear.0 which keeps nn.Linear's default kaiming-uniform(a=sqrt(5)); biases keep
    nn.Linear's default U(-1/sqrt(fan_in), 1/sqrt(fan_in))).  ``density_scale`` multiplies
    ``density_layer.weight`` to give the random field non-trivial opacity."""
    sd = OrderedDict()
    for name, fan_out, fan_in, kind in layout:
        if kind == "xavier":
            bound = math.sqrt(6.0 / (fan_in + fan_out))
        else:  # kaiming_uniform_(a=sqrt(5)) -> bound = 1/sqrt(fan_in)
            bound = 1.0 / math.sqrt(fan_in)
        w = _uniform(rng, (fan_out, fan_in), bound)
        if name == "density_layer":
            w = w * density_scale
        sd[name + ".weight"] = w
        sd[name + ".bias"] = _uniform(rng, (fan_out,), 1.0 / math.sqrt(fan_in))
    return sd


def make_nerf_state_dict(seed: int = 0, density_scale: float = 30.0, **layout_kw) -> "OrderedDict[str, torch.Tensor]":
    """State dict with the reference's key names for ``NeRF`` (coarse_mlp.* / fine_mlp.*)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    layout = vanilla_mlp_layout(**layout_kw)
    sd = OrderedDict()
    for prefix in ("coarse_mlp", "fine_mlp"):
        for k, v in make_mlp_state(rng, layout, density_scale).items():
            sd[f"{prefix}.{k}"] = v
    return sd


def general_mlp_layout(min_deg_point=0, max_deg_point=10, deg_view=4, netdepth=8, netwidth=256, netdepth_condition=1,
                       netwidth_condition=128, skip_layer=4, input_ch=3, input_ch_view=3, num_rgb_channels=3, num_density_channels=1):
    """Layer list of NeRFMLP(...) for ANY constructor arguments (model.py:40-93), in the module's parameter order."""
    pos_size = ((max_deg_point - min_deg_point) * 2 + 1) * input_ch
    view_pos_size = (deg_view * 2 + 1) * input_ch_view
    layout = [("pts_linears.0", netwidth, pos_size, "xavier")]
    for idx in range(netdepth - 1):
        fan_in = netwidth + pos_size if (idx % skip_layer == 0 and idx > 0) else netwidth
        layout.append((f"pts_linears.{idx + 1}", netwidth, fan_in, "xavier"))
    layout.append(("views_linear.0", netwidth_condition, netwidth + view_pos_size, "kaiming"))
    layout += [(f"views_linear.{i}", netwidth_condition, netwidth_condition, "xavier") for i in range(1, netdepth_condition)]
    layout.append(("bottleneck_layer", netwidth, netwidth, "xavier"))
    layout.append(("density_layer", num_density_channels, netwidth, "xavier"))
    layout.append(("rgb_layer", num_rgb_channels, netwidth_condition, "xavier"))
    return layout


def make_general_nerf_state_dict(seed: int, density_scale: float = 2.0, density_bias: float = 0.75, prefixes=("coarse_mlp", "fine_mlp"), **geometry):
    """Smooth-field weights (see make_smooth_nerf_state_dict) for a NeRFMLP of any geometry; keys '<prefix>.<name>' (or bare names
    with prefixes=("",))."""
    rng = np.random.Generator(np.random.PCG64(seed))
    layout = general_mlp_layout(**geometry)
    sd = OrderedDict()
    for prefix in prefixes:
        for k, v in make_mlp_state(rng, layout, density_scale).items():
            if k == "density_layer.bias":
                v = v + density_bias
            sd[f"{prefix}.{k}" if prefix else k] = v
    return sd


# articulated NeRFMLP (models/vanilla_nerf/model_autodecoder.py:60-170, default geometry)
def make_smooth_nerf_state_dict(seed: int = 5, density_scale: float = 2.0, density_bias: float = 0.75):
    """A "trained-like" smooth field for tight end-to-end parity (density x1-3 instead of the x30 of the
    structure fixtures): small density gain and a positive density bias, so the far sample's raw sigma -- whose sign alone
    decides the 1e10-long last interval (helper.py:163) -- sits well away from zero on almost every ray."""
    sd = make_nerf_state_dict(seed=seed, density_scale=density_scale)
    for lvl in ("coarse_mlp", "fine_mlp"):
        sd[f"{lvl}.density_layer.bias"] = sd[f"{lvl}.density_layer.bias"] + density_bias
    return sd


def art_mlp_layout(min_deg_point: int = 0, max_deg_point: int = 10, deg_view: int = 4):
    P, V = 3 + 6 * (max_deg_point - min_deg_point), 3 + 6 * deg_view      # model_autodecoder.py:91,127-129
    layout = [("deformations_linear.0", 128, 163, "xavier")]
    layout += [(f"deformations_linear.{i}", 128, 128, "xavier") for i in (1, 2, 3)]
    layout.append(("deformation_layer", 3, 128, "xavier"))
    layout.append(("pts_linears.0", 256, P + 128, "xavier"))
    for idx in range(7):
        layout.append((f"pts_linears.{idx + 1}", 256, 256 + P + 128 if idx == 4 else 256, "xavier"))
    layout.append(("views_linear.0", 128, 256 + V + 128, "kaiming"))
    layout += [(f"views_linear.{i}", 128, 128, "xavier") for i in (1, 2, 3)]
    layout.append(("bottleneck_layer", 256, 256, "xavier"))
    layout.append(("density_layer", 1, 256, "xavier"))
    layout.append(("rgb_layer", 3, 128, "xavier"))
    return layout


def make_art_state_dict(seed: int = 0, density_scale: float = 30.0, **degrees) -> "OrderedDict[str, torch.Tensor]":
    """State dict with the reference's key names for ``NeRF_AE_Art`` (coarse_mlp.* / fine_mlp.*); ``degrees``: min_deg_point,
    max_deg_point, deg_view of the network (defaults 0, 10, 4)."""
    rng = np.random.Generator(np.random.PCG64(seed + 1000))
    sd = OrderedDict()
    for prefix in ("coarse_mlp", "fine_mlp"):
        for k, v in make_mlp_state(rng, art_mlp_layout(**degrees), density_scale).items():
            sd[f"{prefix}.{k}"] = v
    return sd


def make_code_library_state(seed: int = 0, n_max_objs: int = 2, code_len: int = 128):
    """CodeLibraryArticulated weights (models/code_library.py:20-34): xavier-uniform embeddings."""
    rng = np.random.Generator(np.random.PCG64(seed + 2000))
    sd = OrderedDict()
    for name, rows, cols in (("embedding_instance_shape", n_max_objs, code_len),
                             ("embedding_instance_appearance", n_max_objs, code_len),
                             ("embedding_instance_articulation", 10, 32)):
        sd[name + ".weight"] = _uniform(rng, (rows, cols), math.sqrt(6.0 / (rows + cols)))
    return sd


def make_rays(H: int, W: int, c2w: torch.Tensor | None = None, focal: float | None = None):
    """CPU/torch construction of the per-ray record of one frame (row-major pixel order) following
    datasets/ray_utils.py:71-90,118-159 semantics; returns dict(rays_o, rays_d, viewdirs) fp32 (H*W,3)."""
    if c2w is None:
        c2w = look_at_pose()
    if focal is None:
        focal = focal_from_fovy(H)
    j, i = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    dirs = torch.stack([(i - W / 2) / focal, -(j - H / 2) / focal, -torch.ones_like(i)], -1)
    rays_d = dirs @ c2w[:, :3].T
    rays_d = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    rays_d = rays_d.reshape(-1, 3).contiguous()
    rays_o = c2w[:, 3].expand(rays_d.shape).contiguous()
    return {"rays_o": rays_o, "rays_d": rays_d, "viewdirs": rays_d.clone()}


def seeded_uniform(seed: int, *shape) -> torch.Tensor:
    """U[0,1) fp32 draws from PCG64(seed): the stratified-sampling / inverse-CDF draws of the larger fixtures are named by
    their seed instead of being stored (tests/golden/make_golden.py and the tests regenerate the same bits)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.random(shape, dtype=np.float32))


def random_rays(n: int, seed: int = 0, radius: float = 4.0):
    """n rays from random camera positions on the radius-4 sphere pointing roughly at the origin
    (unit-norm directions, as the datasets deliver them)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    o = rng.normal(size=(n, 3))
    o = radius * o / np.linalg.norm(o, axis=1, keepdims=True)
    tgt = rng.uniform(-0.6, 0.6, size=(n, 3))
    d = tgt - o
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    o = torch.from_numpy(o.astype(np.float32))
    d = torch.from_numpy(d.astype(np.float32))
    d = d / torch.norm(d, dim=-1, keepdim=True)
    return {"rays_o": o, "rays_d": d, "viewdirs": d.clone()}
