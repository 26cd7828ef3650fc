"""Tensor-level wrappers over the C ABI: torch is used for device memory and streams only.

Every function takes/returns CUDA(HIP) fp32 tensors, allocates its outputs, and enqueues on torch's current
stream.  CPU tensors are rejected: there is no host fallback.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check, lib

ACT_NONE, ACT_VANILLA, ACT_ARTICULATED = 0, 1, 2


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor is on {t.device}; the HIP path needs a cuda (ROCm) tensor and has no CPU fallback")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    return t.contiguous()


def _tag(out: torch.Tensor) -> torch.Tensor:
    """A freshly packed stream / per-call block remembers the form it was packed in (bottleneck folded or literal) ON THE TENSOR: the
    launchers look the form up by address (csrc/aon_fold.hip), and `_pk` re-declares it from here on every call, so an address the caching
    allocator hands out again can never carry an earlier tenant's form."""
    out._aon_form = int(lib.aon_stream_form(out.data_ptr()))
    return out


def _pk(t: torch.Tensor | None) -> C.c_void_p:
    """Pointer of a packed weight stream / per-call block for a C call, with its form re-declared.  A tensor without a form is not one
    this binding packed (a clone or a moved copy of a packed buffer has none): refused here instead of being run under a guessed form."""
    if t is None:
        return None
    form = getattr(t, "_aon_form", None)
    if form is None:
        raise ValueError("not a packed stream of this binding (tensors returned by ops.pack_* / ops.art_prepare carry their form; a clone or "
                         "copy does not: use ops.clone_packed)")
    check(lib.aon_declare_stream_form(t.data_ptr(), form), "aon_declare_stream_form")
    return C.c_void_p(t.data_ptr())


def clone_packed(t: torch.Tensor) -> torch.Tensor:
    """A copy of a packed stream / per-call block that keeps its form (plain `.clone()` yields a buffer every call refuses)."""
    out = t.clone()
    out._aon_form = t._aon_form
    check(lib.aon_declare_stream_form(out.data_ptr(), out._aon_form), "aon_declare_stream_form")
    return out


def _ptr(t: torch.Tensor | None) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _c2w_host(c2w) -> "C.Array":
    c = torch.as_tensor(c2w, dtype=torch.float32).detach().cpu().reshape(-1)
    if c.numel() < 12:
        raise ValueError("c2w must hold at least (3,4) values")
    return (C.c_float * 12)(*c[:12].tolist())


# ------------------------------------------------------------------ R1/R2 ray generation
def raygen(c2w, H: int, W: int, focal: float, pix_begin: int = 0, pix_end: int | None = None, device=None):
    """Fused get_ray_directions + get_rays for row-major pixels [pix_begin, pix_end).
    Returns (rays_o, viewdirs) with shapes (n,3); the reference's rays_d is the same tensor as viewdirs."""
    pix_end = H * W if pix_end is None else pix_end
    n = pix_end - pix_begin
    dev = torch.device("cuda") if device is None else torch.device(device)
    rays_o = torch.empty((n, 3), dtype=torch.float32, device=dev)
    viewdirs = torch.empty((n, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib.aon_raygen(_c2w_host(c2w), H, W, float(focal), pix_begin, pix_end, _ptr(rays_o), _ptr(viewdirs), None,
                             _stream()), "aon_raygen")
    return rays_o, viewdirs


# ------------------------------------------------------------------ R13 training losses, two launches
def _loss_args(rgb_c, rgb_f, target, latents):
    rgb_f, target = _f32(rgb_f, "rgb_fine"), _f32(target, "target")
    rgb_c = None if rgb_c is None else _f32(rgb_c, "rgb_coarse")
    n = rgb_f.numel() // 3
    if target.numel() != 3 * n or (rgb_c is not None and rgb_c.numel() != 3 * n):
        raise ValueError("train_loss: rgb / target sizes differ")
    lats = [None, None, None]
    for k, t in enumerate(latents or ()):
        lats[k] = None if t is None else _f32(t, "latent")
    lens = (C.c_int * 3)(*[0 if t is None else t.numel() for t in lats])
    return rgb_c, rgb_f, target, n, lats, lens


def train_loss_fwd(rgb_c, rgb_f, target, latents=(), reg_scale: float = 1e-4):
    """-> (loss (1,), stats (8,) = {loss0, loss1, reg, loss, psnr0, psnr1, 0, 0}); `latents`: up to three one-row codes."""
    rgb_c, rgb_f, target, n, lats, lens = _loss_args(rgb_c, rgb_f, target, latents)
    stats = torch.empty(8, dtype=torch.float32, device=rgb_f.device)
    loss = torch.empty(1, dtype=torch.float32, device=rgb_f.device)
    with torch.cuda.device(rgb_f.device):
        check(lib.aon_train_loss_fwd(_ptr(rgb_c), _ptr(rgb_f), _ptr(target), n, _ptr_array(lats), lens, float(reg_scale), _ptr(stats), _ptr(loss), _stream()),
              "aon_train_loss_fwd")
    return loss, stats


def train_loss_bwd(rgb_c, rgb_f, target, latents, reg_scale: float, grad_loss):
    """-> (d_rgb_c or None, d_rgb_f, [d_latent or None] * 3)"""
    rgb_c, rgb_f, target, n, lats, lens = _loss_args(rgb_c, rgb_f, target, latents)
    go = _f32(grad_loss, "grad_loss")
    d_c = None if rgb_c is None else torch.empty_like(rgb_c)
    d_f = torch.empty_like(rgb_f)
    d_l = [None if t is None else torch.empty_like(t) for t in lats]
    with torch.cuda.device(rgb_f.device):
        check(lib.aon_train_loss_bwd(_ptr(rgb_c), _ptr(rgb_f), _ptr(target), n, _ptr_array(lats), lens, float(reg_scale), _ptr(go), _ptr(d_c), _ptr(d_f),
                                     _ptr_array(d_l), _stream()), "aon_train_loss_bwd")
    return d_c, d_f, d_l


def ray_directions(H: int, W: int, focal: float, device=None):
    dev = torch.device("cuda") if device is None else torch.device(device)
    out = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib.aon_ray_directions(H, W, float(focal), _ptr(out), _stream()), "aon_ray_directions")
    return out


def get_rays(directions: torch.Tensor, c2w):
    d = _f32(directions, "directions")
    n = d.numel() // 3
    rays_o = torch.empty((n, 3), dtype=torch.float32, device=d.device)
    viewdirs = torch.empty((n, 3), dtype=torch.float32, device=d.device)
    with torch.cuda.device(d.device):
        check(lib.aon_get_rays(_ptr(d), _c2w_host(c2w), n, _ptr(rays_o), _ptr(viewdirs), None, _stream()), "aon_get_rays")
    return rays_o, viewdirs


def ray_radii(directions: torch.Tensor, c2w):
    """radii (H*W,) of get_rays(..., output_radii=True) (ray_utils.py:138-143) from (H,W,3) camera-space directions."""
    d = _f32(directions, "directions")
    if d.dim() != 3 or d.shape[-1] != 3 or d.shape[0] < 3:
        raise ValueError(f"output_radii needs directions of shape (H>=3, W, 3) like the reference (it differences image rows), got {tuple(d.shape)}")
    H, W = int(d.shape[0]), int(d.shape[1])
    radii = torch.empty((H * W,), dtype=torch.float32, device=d.device)
    with torch.cuda.device(d.device):
        check(lib.aon_ray_radii(_ptr(d), _c2w_host(c2w), H, W, _ptr(radii), _stream()), "aon_ray_radii")
    return radii


# ------------------------------------------------------------------ R3 sampling
def cast_rays(t_vals, origins, directions):
    t, o, d = _f32(t_vals, "t_vals"), _f32(origins, "origins"), _f32(directions, "directions")
    n, S = t.shape
    coords = torch.empty((n, S, 3), dtype=torch.float32, device=t.device)
    with torch.cuda.device(t.device):
        check(lib.aon_cast_rays(_ptr(t), _ptr(o), _ptr(d), n, S, _ptr(coords), _stream()), "aon_cast_rays")
    return coords


def sample_along_rays(rays_o, rays_d, num_samples: int, near: float, far: float, t_rand=None, want_coords=True, lindisp=False):
    o, d = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d")
    n, S = o.shape[0], num_samples + 1
    tr = None if t_rand is None else _f32(t_rand, "t_rand")
    if tr is not None and tuple(tr.shape) != (n, S):
        raise ValueError(f"t_rand must be ({n},{S}), got {tuple(tr.shape)}")
    t_vals = torch.empty((n, S), dtype=torch.float32, device=o.device)
    coords = torch.empty((n, S, 3), dtype=torch.float32, device=o.device) if want_coords else None
    with torch.cuda.device(o.device):
        if lindisp:   # helper.py:117 evaluates 1.0 / near in Python double precision; ctypes rounds it to fp32 once, as torch does
            check(lib.aon_sample_along_rays_ex(_ptr(o), _ptr(d), n, S, float(near), float(far), 1, 1.0 / near, 1.0 / far, _ptr(tr),
                                               _ptr(t_vals), _ptr(coords), _stream()), "aon_sample_along_rays_ex")
        else:
            check(lib.aon_sample_along_rays(_ptr(o), _ptr(d), n, S, float(near), float(far), _ptr(tr), _ptr(t_vals), _ptr(coords),
                                            _stream()), "aon_sample_along_rays")
    return t_vals, coords


# ------------------------------------------------------------------ R4 positional encoding (stage-level)
def pos_enc(x, min_deg: int, max_deg: int):
    xc = _f32(x, "x")
    if xc.shape[-1] != 3:
        raise ValueError("pos_enc expects (...,3)")
    n = xc.numel() // 3
    F = 3 + 6 * (max_deg - min_deg)
    out = torch.empty((*xc.shape[:-1], F), dtype=torch.float32, device=xc.device)
    with torch.cuda.device(xc.device):
        check(lib.aon_pos_enc(_ptr(xc), n, min_deg, max_deg, _ptr(out), _stream()), "aon_pos_enc")
    return out


# ------------------------------------------------------------------ R5 MLP
VANILLA_PARAM_ORDER = (
    [f"pts_linears.{i}.{k}" for i in range(8) for k in ("weight", "bias")]
    + [f"{m}.{k}" for m in ("views_linear.0", "bottleneck_layer", "density_layer", "rgb_layer") for k in ("weight", "bias")]
)
VANILLA_PARAM_SHAPES = {
    "pts_linears.0.weight": (256, 63), "pts_linears.5.weight": (256, 319), "views_linear.0.weight": (128, 283),
    "bottleneck_layer.weight": (256, 256), "density_layer.weight": (1, 256), "rgb_layer.weight": (3, 128),
    "views_linear.0.bias": (128,), "bottleneck_layer.bias": (256,), "density_layer.bias": (1,), "rgb_layer.bias": (3,),
}
for _i in (1, 2, 3, 4, 6, 7):
    VANILLA_PARAM_SHAPES[f"pts_linears.{_i}.weight"] = (256, 256)
for _i in range(8):
    VANILLA_PARAM_SHAPES[f"pts_linears.{_i}.bias"] = (256,)


def packed_bytes() -> int:
    return int(lib.aon_mlp_packed_bytes())


def pack_vanilla_mlp(params: dict, out: torch.Tensor | None = None, degrees=(0, 10, 4)) -> torch.Tensor:
    """params: name -> tensor with the reference's NeRFMLP parameter names (no prefix).  Returns the packed
    uint8 weight stream consumed by mlp_fwd / render_fwd (re-pack whenever the parameters change).  ``degrees`` =
    (min_deg_point, max_deg_point, deg_view) of a default-size network with at most 10 / 4 frequency levels: the slots of the
    missing levels get zero weight (aon_pack_vanilla_mlp_deg)."""
    mn, mx, dv = degrees
    shapes = dict(VANILLA_PARAM_SHAPES)
    P, V = 3 + 6 * (mx - mn), 3 + 6 * dv
    shapes.update({"pts_linears.0.weight": (256, P), "pts_linears.5.weight": (256, 256 + P), "views_linear.0.weight": (128, 256 + V)})
    tensors = []
    for name in VANILLA_PARAM_ORDER:
        t = _f32(params[name].detach(), name)
        if tuple(t.shape) != shapes[name]:
            raise ValueError(f"{name}: shape {tuple(t.shape)} != {shapes[name]} (the fused kernels take the reference's default "
                             f"NeRFMLP sizes with encoding degrees {tuple(degrees)})")
        tensors.append(t)
    dev = tensors[0].device
    if out is None:
        out = torch.empty(packed_bytes(), dtype=torch.uint8, device=dev)
    arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    with torch.cuda.device(dev):
        if tuple(degrees) == (0, 10, 4):
            check(lib.aon_pack_vanilla_mlp(arr, _ptr(out), _stream()), "aon_pack_vanilla_mlp")
        else:
            check(lib.aon_pack_vanilla_mlp_deg(arr, mn, mx, dv, _ptr(out), _stream()), "aon_pack_vanilla_mlp_deg")
    return _tag(out)


def mlp_fwd(packed, rays_o, rays_d, viewdirs, t_vals):
    """cast_rays + pos_enc + NeRFMLP.forward fused.  Returns raw (n,S,4) = (raw_rgb, raw_density)."""
    o, d, v, t = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _f32(viewdirs, "viewdirs"), _f32(t_vals, "t_vals")
    n, S = t.shape
    raw = torch.empty((n, S, 4), dtype=torch.float32, device=t.device)
    with torch.cuda.device(t.device):
        check(lib.aon_mlp_fwd(_pk(packed), _ptr(o), _ptr(d), _ptr(v), _ptr(t), n, S, _ptr(raw), _stream()), "aon_mlp_fwd")
    return raw


def mlp_fwd_enc(packed, samples_enc, viewdirs_enc):
    x, c = _f32(samples_enc, "samples_enc"), _f32(viewdirs_enc, "viewdirs_enc")
    n, S, F = x.shape
    if F != 63 or tuple(c.shape) != (n, 27):
        raise ValueError("mlp_fwd_enc expects samples_enc (n,S,63) and viewdirs_enc (n,27)")
    raw = torch.empty((n, S, 4), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.aon_mlp_fwd_enc(_pk(packed), _ptr(x), _ptr(c), n, S, _ptr(raw), _stream()), "aon_mlp_fwd_enc")
    return raw


# ------------------------------------------------------------------ constructor arguments beyond the defaults
class RenderOpts:
    """Sampler / activation arguments of ``NeRF.__init__`` (model.py:124-135) and ``NeRF_AE_Art.__init__``
    (model_autodecoder.py:241-257) -> ``aon_render_opts``.  Python numbers become fp32 exactly where torch would round them."""

    def __init__(self, num_coarse_samples=64, num_fine_samples=128, lindisp=False, noise_std=0.0, rgb_padding=0.001, density_bias=-1.0,
                 degrees=(0, 10, 4)):
        self.degrees = tuple(int(x) for x in degrees)   # (min_deg_point, max_deg_point, deg_view) of a default-size network on the fused kernels
        self.num_coarse_samples, self.num_fine_samples = int(num_coarse_samples), int(num_fine_samples)
        self.lindisp, self.noise_std = bool(lindisp), float(noise_std)
        self.rgb_padding, self.density_bias = float(rgb_padding), float(density_bias)
        if not 2 <= self.num_coarse_samples <= 1023 or self.num_fine_samples < 1:
            raise ValueError("num_coarse_samples must be in [2, 1023] and num_fine_samples >= 1")

    @property
    def Sc(self) -> int:
        return self.num_coarse_samples + 1

    @property
    def Sf(self) -> int:
        return self.num_coarse_samples + 1 + self.num_fine_samples

    def S(self, level: int) -> int:
        return self.Sc if level == 0 else self.Sf

    def c_struct(self, near: float, far: float, noise=None):
        """-> (aon_render_opts, tensors to keep alive).  ``noise``: per-level (n,S) uniform draws or None (model.py:184)."""
        st = _lib.RenderOptsC()
        lib.aon_render_opts_init(C.byref(st))
        st.num_coarse_samples, st.num_fine_samples, st.lindisp = self.num_coarse_samples, self.num_fine_samples, int(self.lindisp)
        if self.lindisp:
            st.inv_near, st.inv_far = 1.0 / near, 1.0 / far      # double -> fp32 once (helper.py:117)
        st.rgb_scale, st.rgb_shift, st.sigma_bias = 1 + 2 * self.rgb_padding, self.rgb_padding, self.density_bias
        st.min_deg_point, st.max_deg_point, st.deg_view = self.degrees
        keep = []
        if noise is not None and self.noise_std > 0:
            st.noise_std = self.noise_std
            for lvl, field in enumerate(("noise_c", "noise_f")):
                if lvl < len(noise) and noise[lvl] is not None:
                    t = _f32(noise[lvl], "noise")
                    keep.append(t)
                    setattr(st, field, t.data_ptr())
        return st, keep


DEFAULT_OPTS = RenderOpts()


def _opts(opts):
    return DEFAULT_OPTS if opts is None else opts


# ------------------------------------------------------------------ R8 compositing
def _composite(rgb_t, rgb_stride, sig_t, sig_off, sig_stride, t_vals, dirs, white_bkgd, act, want_weights):
    t, d = _f32(t_vals, "t_vals"), _f32(dirs, "dirs")
    n, S = t.shape
    dev = t.device
    comp = torch.empty((n, 3), dtype=torch.float32, device=dev)
    acc = torch.empty((n,), dtype=torch.float32, device=dev)
    depth = torch.empty((n,), dtype=torch.float32, device=dev)
    weights = torch.empty((n, S), dtype=torch.float32, device=dev) if want_weights else None
    sig_ptr = C.c_void_p(sig_t.data_ptr() + 4 * sig_off)
    with torch.cuda.device(dev):
        check(lib.aon_composite(_ptr(rgb_t), rgb_stride, sig_ptr, sig_stride, _ptr(t), _ptr(d), n, S, int(bool(white_bkgd)), act,
                                _ptr(comp), _ptr(acc), _ptr(depth), _ptr(weights), _stream()), "aon_composite")
    return comp, acc, weights, depth


def volumetric_rendering(rgb, density, t_vals, dirs, white_bkgd):
    """helper.volumetric_rendering on activated rgb (n,S,3) / density (n,S,1)."""
    r, s = _f32(rgb, "rgb"), _f32(density, "density")
    return _composite(r, 3, s, 0, 1, t_vals, dirs, white_bkgd, ACT_NONE, True)


def composite_raw(raw, t_vals, dirs, white_bkgd, act=ACT_VANILLA, want_weights=True, opts=None, noise=None):
    """Activation + compositing of the fused kernel's packed raw (n,S,4).  ``opts`` / ``noise`` (n,S): the activation scalars
    and the density noise of a non-default constructor (aon_composite_ex)."""
    r = _f32(raw, "raw")
    if opts is None and noise is None:
        return _composite(r, 4, r, 3, 4, t_vals, dirs, white_bkgd, act, want_weights)
    t, d = _f32(t_vals, "t_vals"), _f32(dirs, "dirs")
    n, S = t.shape
    dev = t.device
    st, keep = _opts(opts).c_struct(1.0, 1.0, [noise])
    comp = torch.empty((n, 3), dtype=torch.float32, device=dev)
    acc = torch.empty((n,), dtype=torch.float32, device=dev)
    depth = torch.empty((n,), dtype=torch.float32, device=dev)
    weights = torch.empty((n, S), dtype=torch.float32, device=dev) if want_weights else None
    with torch.cuda.device(dev):
        check(lib.aon_composite_ex(_ptr(r), 4, C.c_void_p(r.data_ptr() + 12), 4, _ptr(t), _ptr(d), n, S, int(bool(white_bkgd)), act,
                                   C.byref(st), _ptr(comp), _ptr(acc), _ptr(depth), _ptr(weights), _stream()), "aon_composite_ex")
    return comp, acc, weights, depth


# ------------------------------------------------------------------ R6/R7 inverse CDF
_U_CACHE: dict = {}


def deterministic_u(device, num_samples: int = 128) -> torch.Tensor:
    """helper.py:229: torch.linspace(0, 1 - 2**-32, num_samples) (fp32; the last element rounds to exactly 1.0).
    Computed once on the host and copied, so it is the same vector the reference builds."""
    key = (str(device), num_samples)
    if key not in _U_CACHE:
        _U_CACHE[key] = torch.linspace(0.0, 1.0 - 2.0 ** -32, num_samples).to(device)
    return _U_CACHE[key]


def _u_args(u, n, device, nf: int = 128):
    if u is None:
        return deterministic_u(device, nf), 0
    uu = _f32(u, "u")
    if tuple(uu.shape) == (nf,):
        return uu, 0
    if tuple(uu.shape) != (n, nf):
        raise ValueError(f"u must be ({nf},) or ({n},{nf}), got {tuple(uu.shape)}")
    return uu, nf


def sorted_piecewise_constant_pdf(bins, weights, u=None, num_samples: int = 128):
    """helper.sorted_piecewise_constant_pdf: bins (n,nb), weights (n,nb-1) -> num_samples draws per ray.  The reference
    geometry (64 bins, 128 draws) runs the specialised kernel, anything else aon_sample_pdf_n (same bits where they overlap)."""
    b, w = _f32(bins, "bins"), _f32(weights, "weights")
    n, nb = b.shape
    if tuple(w.shape) != (n, nb - 1) or nb < 2:
        raise ValueError(f"weights must be ({n},{nb - 1}) for bins ({n},{nb})")
    uu, us = _u_args(u, n, b.device, num_samples)
    samples = torch.empty((n, num_samples), dtype=torch.float32, device=b.device)
    with torch.cuda.device(b.device):
        if (nb, num_samples) == (64, 128):
            check(lib.aon_sample_pdf(_ptr(b), _ptr(w), 63, None, _ptr(uu), us, n, _ptr(samples), None, _stream()), "aon_sample_pdf")
        else:
            check(lib.aon_sample_pdf_n(_ptr(b), _ptr(w), nb - 1, None, _ptr(uu), us, n, nb, num_samples, 0, _ptr(samples), None, _stream()),
                  "aon_sample_pdf_n")
    return samples


def sample_pdf_t_n(t_coarse, coarse_weights, num_samples: int, u=None, bins=None, force_generic=False):
    """sample_pdf_t for any sizes: t_coarse (n,nt), the full coarse weights (n,nt) (the pdf uses weights[...,1:-1]) or explicit
    (bins (n,nb), weights (n,nb-1)) -> t_fine (n, nt + num_samples), through aon_sample_pdf_n."""
    t, w = _f32(t_coarse, "t_coarse"), _f32(coarse_weights, "weights")
    n, nt = t.shape
    b = None if bins is None else _f32(bins, "bins")
    nb = nt - 1 if b is None else b.shape[1]
    if tuple(w.shape) == (n, nb + 1):
        w_ptr, w_stride = C.c_void_p(w.data_ptr() + 4), nb + 1
    elif tuple(w.shape) == (n, nb - 1):
        w_ptr, w_stride = _ptr(w), nb - 1
    else:
        raise ValueError(f"weights must be ({n},{nb + 1}) or ({n},{nb - 1})")
    uu, us = _u_args(u, n, t.device, num_samples)
    t_fine = torch.empty((n, nt + num_samples), dtype=torch.float32, device=t.device)
    with torch.cuda.device(t.device):
        check(lib.aon_sample_pdf_n(_ptr(b), w_ptr, w_stride, _ptr(t), _ptr(uu), us, n, nb, num_samples, nt, None, _ptr(t_fine), _stream()),
              "aon_sample_pdf_n")
    return t_fine


def sample_pdf_t(t_coarse, coarse_weights, u=None, bins=None):
    """t_fine (n,193) = sort(cat[t_coarse, inverse-CDF samples]) from the full coarse weights (n,65)
    (the pdf uses weights[...,1:-1]) -- or from explicit (bins (n,64), weights (n,63))."""
    t = _f32(t_coarse, "t_coarse")
    w = _f32(coarse_weights, "weights")
    n = t.shape[0]
    if tuple(t.shape) != (n, 65):
        raise ValueError("t_coarse must be (n,65)")
    if tuple(w.shape) == (n, 65):
        w_ptr, w_stride = C.c_void_p(w.data_ptr() + 4), 65
    elif tuple(w.shape) == (n, 63):
        w_ptr, w_stride = _ptr(w), 63
    else:
        raise ValueError("weights must be (n,65) or (n,63)")
    b = None if bins is None else _f32(bins, "bins")
    uu, us = _u_args(u, n, t.device)
    t_fine = torch.empty((n, 193), dtype=torch.float32, device=t.device)
    with torch.cuda.device(t.device):
        check(lib.aon_sample_pdf(_ptr(b), w_ptr, w_stride, _ptr(t), _ptr(uu), us, n, None, _ptr(t_fine), _stream()), "aon_sample_pdf")
    return t_fine


def composite_pdf(raw, t_coarse, dirs, white_bkgd, act=ACT_VANILLA, u=None, want_weights=False):
    """Coarse level of NeRF.forward in ONE kernel (model.py:160-173): compositing of the packed raw (n,65,4) records and the
    fine level's sampling -> (comp_rgb, acc, weights | None, depth, t_fine (n,193)).  Same bits as composite_raw followed by
    sample_pdf_t."""
    r, t, d = _f32(raw, "raw"), _f32(t_coarse, "t_coarse"), _f32(dirs, "dirs")
    n, dev = t.shape[0], t.device
    if tuple(t.shape) != (n, 65) or r.numel() != n * 65 * 4:
        raise ValueError("composite_pdf: t_coarse must be (n,65) and raw (n,65,4)")
    uu, us = _u_args(u, n, dev)
    comp = torch.empty((n, 3), dtype=torch.float32, device=dev)
    acc = torch.empty((n,), dtype=torch.float32, device=dev)
    depth = torch.empty((n,), dtype=torch.float32, device=dev)
    weights = torch.empty((n, 65), dtype=torch.float32, device=dev) if want_weights else None
    t_fine = torch.empty((n, 193), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib.aon_composite_pdf(_ptr(r), _ptr(t), _ptr(d), n, int(bool(white_bkgd)), act, _ptr(uu), us, _ptr(comp), _ptr(acc),
                                    _ptr(depth), _ptr(weights), _ptr(t_fine), _stream()), "aon_composite_pdf")
    return comp, acc, weights, depth, t_fine


def set_coarse_fusion(on: bool) -> None:
    """Whole-path entry points: coarse compositing + inverse CDF as one kernel (default) or as the two stage kernels."""
    check(lib.aon_set_coarse_fusion(int(bool(on))), "aon_set_coarse_fusion")


# ------------------------------------------------------------------ R9 whole path
_WS_CACHE: dict = {}
# rays rendered per internal chunk: a whole 640x480 frame (307,200 rays) in one pass -- 4,892 B/ray of per-ray buffers (sample
# records, t, weights, the per-ray view bias) = 1.5 GB of the 288 GB, so that every per-ray kernel is ONE launch per frame (round 2: 65,536-ray chunks, five 30-60 us launches
# each of which spent a fifth of its time ramping up and draining)
MAX_CHUNK_RAYS = 327680


def _workspace(device, n_rays: int, st=None) -> torch.Tensor:
    chunk = MAX_CHUNK_RAYS
    if st is not None and (st.min_deg_point, st.max_deg_point, st.deg_view) != (0, 10, 4):
        chunk = 65536      # the encodings of a chunk are materialised (51 KB per ray) for other degrees
    need = int(lib.aon_render_workspace_bytes_ex(min(n_rays, chunk), None if st is None else C.byref(st)))
    if need < 0:
        check(need, "aon_render_workspace_bytes_ex")
    key = str(device)
    ws = _WS_CACHE.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        _WS_CACHE[key] = ws
    return ws


def _check_noise(noise, n, op, num_levels):
    if noise is None:
        return None
    out = []
    for lvl in range(num_levels):
        t = noise[lvl] if lvl < len(noise) else None
        if t is not None and t.numel() != n * op.S(lvl):
            raise ValueError(f"noise[{lvl}] must hold ({n},{op.S(lvl)}) values, got {tuple(t.shape)}")
        out.append(t)
    return out


def render_fwd(packed_coarse, packed_fine, rays_o, rays_d, viewdirs, near, far, white_bkgd, num_levels=2, t_rand=None, u=None,
               opts=None, noise=None):
    """NeRF.forward: returns [(rgb, acc, depth)_coarse, (rgb, acc, depth)_fine] (fine omitted if num_levels == 1).
    ``opts`` (RenderOpts): non-default sample counts / lindisp / noise_std; ``noise``: per-level (n,S) uniform draws."""
    o, d, v = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _f32(viewdirs, "viewdirs")
    n, dev = o.shape[0], o.device
    op = _opts(opts)
    tr = None if t_rand is None else _f32(t_rand, "t_rand")
    if tr is not None and tuple(tr.shape) != (n, op.Sc):
        raise ValueError(f"t_rand must be ({n},{op.Sc})")
    uu, us = _u_args(u, n, dev, op.num_fine_samples) if num_levels == 2 else (None, 0)
    outs = []
    for _ in range(num_levels):
        outs.append((torch.empty((n, 3), dtype=torch.float32, device=dev), torch.empty((n,), dtype=torch.float32, device=dev),
                     torch.empty((n,), dtype=torch.float32, device=dev)))
    fine = outs[1] if num_levels == 2 else (None, None, None)
    st, keep = op.c_struct(near, far, _check_noise(noise, n, op, num_levels))
    ws = _workspace(dev, n, st)
    with torch.cuda.device(dev):
        check(lib.aon_render_fwd_ex(_pk(packed_coarse), _pk(packed_fine), _ptr(o), _ptr(d), _ptr(v), n, float(near), float(far),
                                    int(bool(white_bkgd)), num_levels, _ptr(tr), _ptr(uu), us,
                                    _ptr(outs[0][0]), _ptr(outs[0][1]), _ptr(outs[0][2]), _ptr(fine[0]), _ptr(fine[1]), _ptr(fine[2]),
                                    _ptr(ws), ws.numel(), _stream(), C.byref(st)), "aon_render_fwd")
    return outs


# ------------------------------------------------------------------ R10/R11 articulated network
ART_PARAM_ORDER = (
    [f"deformations_linear.{i}.{k}" for i in range(4) for k in ("weight", "bias")]
    + [f"deformation_layer.{k}" for k in ("weight", "bias")]
    + [f"pts_linears.{i}.{k}" for i in range(8) for k in ("weight", "bias")]
    + [f"views_linear.{i}.{k}" for i in range(4) for k in ("weight", "bias")]
    + [f"{m}.{k}" for m in ("bottleneck_layer", "density_layer", "rgb_layer") for k in ("weight", "bias")]
)
ART_PARAM_SHAPES = {
    "deformations_linear.0.weight": (128, 163), "deformation_layer.weight": (3, 128), "deformation_layer.bias": (3,),
    "pts_linears.0.weight": (256, 191), "pts_linears.5.weight": (256, 447), "views_linear.0.weight": (128, 411),
    "bottleneck_layer.weight": (256, 256), "bottleneck_layer.bias": (256,), "density_layer.weight": (1, 256),
    "density_layer.bias": (1,), "rgb_layer.weight": (3, 128), "rgb_layer.bias": (3,),
}
for _i in (1, 2, 3):
    ART_PARAM_SHAPES[f"deformations_linear.{_i}.weight"] = (128, 128)
    ART_PARAM_SHAPES[f"views_linear.{_i}.weight"] = (128, 128)
for _i in range(4):
    ART_PARAM_SHAPES[f"deformations_linear.{_i}.bias"] = (128,)
    ART_PARAM_SHAPES[f"views_linear.{_i}.bias"] = (128,)
for _i in (1, 2, 3, 4, 6, 7):
    ART_PARAM_SHAPES[f"pts_linears.{_i}.weight"] = (256, 256)
for _i in range(8):
    ART_PARAM_SHAPES[f"pts_linears.{_i}.bias"] = (256,)


def art_param_shapes(degrees=(0, 10, 4)) -> dict:
    """Parameter shapes of the articulated NeRFMLP built with (min_deg_point, max_deg_point, deg_view) and default widths
    (model_autodecoder.py:60-170): the three concatenating layers follow P = 3 + 6 (max - min), V = 3 + 6 deg_view."""
    lo, hi, dv = (int(x) for x in degrees)
    if not (0 <= hi - lo <= 10 and 0 <= dv <= 4 and -32 <= lo and hi <= 32):
        raise NotImplementedError(f"articulated NeRFMLP with degrees {tuple(degrees)}: the kernels hold up to 10 position and 4 view frequency levels")
    P, V = 3 + 6 * (hi - lo), 3 + 6 * dv
    shapes = dict(ART_PARAM_SHAPES)
    shapes["pts_linears.0.weight"] = (256, P + 128)
    shapes["pts_linears.5.weight"] = (256, 256 + P + 128)
    shapes["views_linear.0.weight"] = (128, 256 + V + 128)
    return shapes


def _art_param_array(params: dict, degrees=(0, 10, 4)):
    tensors = []
    shapes = ART_PARAM_SHAPES if tuple(degrees) == (0, 10, 4) else art_param_shapes(degrees)
    for name in ART_PARAM_ORDER:
        t = _f32(params[name].detach(), name)
        if tuple(t.shape) != shapes[name]:
            raise ValueError(f"{name}: shape {tuple(t.shape)} != {shapes[name]} (the articulated NeRFMLP has HIP kernels for the "
                             f"reference's default widths at degrees {tuple(degrees)})")
        tensors.append(t)
    return tensors, (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def pack_art_mlp(params: dict, out: torch.Tensor | None = None, degrees=(0, 10, 4)) -> torch.Tensor:
    """Packed weight stream of one articulated NeRFMLP (re-pack whenever the parameters change); other ``degrees``: zero weight in the
    63 / 27-wide slots of the levels the network lacks (aon_pack_art_mlp_deg)."""
    tensors, arr = _art_param_array(params, degrees)
    dev = tensors[0].device
    if out is None:
        out = torch.empty(int(lib.aon_art_packed_bytes()), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(lib.aon_pack_art_mlp_deg(arr, int(degrees[0]), int(degrees[1]), int(degrees[2]), _ptr(out), _stream()), "aon_pack_art_mlp_deg")
    return _tag(out)


def _latent(latents: dict, key: str, width: int) -> torch.Tensor:
    t = _f32(latents[key].detach(), f"latents[{key!r}]").reshape(-1)
    if t.numel() != width:
        raise ValueError(f"latents[{key!r}] must hold {width} values (one instance per call, like the reference's "
                         f"'n1 c -> (n1 n2) c' broadcast), got {tuple(latents[key].shape)}")
    return t


def art_prepare(params: dict, latents: dict, out: torch.Tensor | None = None, degrees=(0, 10, 4)) -> torch.Tensor:
    """Per-call block: small vectors + biases with the three latents folded in (latents keys as the reference:
    'density' (1,128), 'color' (1,128), 'articulation' (1,32)) + the ten encoding scales of ``degrees``."""
    tensors, arr = _art_param_array(params, degrees)
    dev = tensors[0].device
    shape, app, art = _latent(latents, "density", 128), _latent(latents, "color", 128), _latent(latents, "articulation", 32)
    if out is None:
        out = torch.empty(int(lib.aon_art_small_bytes()), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(lib.aon_art_prepare_deg(arr, _ptr(shape), _ptr(app), _ptr(art), int(degrees[0]), int(degrees[1]), int(degrees[2]), _ptr(out), _stream()),
              "aon_art_prepare_deg")
    return _tag(out)


def art_pack_step(params_c: dict, params_f: dict, latents: dict, degrees=(0, 10, 4), with_bwd: bool = True, out=None):
    """pack_art_mlp + art_prepare (+ pack_art_mlp_bwd) of the coarse and the fine network in ONE C call (aon_art_pack_step): the four fp64
    fold products as one launch in front.  -> [(packed, small, packed_bwd or None)] per level, tagged with their form; fresh buffers unless
    ``out`` = [(packed, small, packed_bwd or None)] x 2 hands in uint8 buffers of the library's sizes."""
    tens, arrs = [], []
    for params in (params_c, params_f):
        t, arr = _art_param_array(params, degrees)
        tens.append(t)
        arrs.append(arr)
    shape, app, art = _latent(latents, "density", 128), _latent(latents, "color", 128), _latent(latents, "articulation", 32)
    dev = tens[0][0].device
    new = lambda nbytes: torch.empty(int(nbytes), dtype=torch.uint8, device=dev)   # noqa: E731
    if out is not None:
        pk, sm, bw = ([o[i] for o in out] for i in range(3))
        for got, want in zip(pk + sm + bw, [lib.aon_art_packed_bytes()] * 2 + [lib.aon_art_small_bytes()] * 2 + [lib.aon_art_bwd_packed_bytes()] * 2):
            if got is not None and (got.dtype != torch.uint8 or got.numel() != int(want) or got.device != dev or not got.is_contiguous()):
                raise ValueError(f"art_pack_step: out buffers must be contiguous uint8 tensors of the library's sizes on {dev}")
    else:
        pk = [new(lib.aon_art_packed_bytes()) for _ in range(2)]
        sm = [new(lib.aon_art_small_bytes()) for _ in range(2)]
        bw = [new(lib.aon_art_bwd_packed_bytes()) if with_bwd else None for _ in range(2)]
    with torch.cuda.device(dev):
        check(lib.aon_art_pack_step(arrs[0], arrs[1], _ptr(shape), _ptr(app), _ptr(art), int(degrees[0]), int(degrees[1]), int(degrees[2]),
                                    _ptr(pk[0]), _ptr(sm[0]), _ptr(bw[0]), _ptr(pk[1]), _ptr(sm[1]), _ptr(bw[1]), _stream()), "aon_art_pack_step")
    return [(_tag(pk[l]), _tag(sm[l]), None if bw[l] is None else _tag(bw[l])) for l in range(2)]


def vanilla_pack_step(params_c: dict, params_f: dict, degrees=(0, 10, 4), with_bwd: bool = True, out=None):
    """pack_vanilla_mlp (+ pack_vanilla_mlp_bwd) of the coarse and the fine network in ONE C call (aon_vanilla_pack_step): the eight fp64
    fold products as one launch in front.  -> [(packed, packed_bwd or None)] per level, tagged with their form; fresh buffers unless ``out`` =
    [(packed, packed_bwd or None)] x 2 hands in uint8 buffers of the library's sizes."""
    shapes = vanilla_param_shapes(degrees)
    arrs, keep = [], []
    for params in (params_c, params_f):
        tensors = [_f32(params[name].detach(), name) for name in VANILLA_PARAM_ORDER]
        for name, t in zip(VANILLA_PARAM_ORDER, tensors):
            if tuple(t.shape) != shapes[name]:
                raise ValueError(f"{name}: shape {tuple(t.shape)} != {shapes[name]} for encoding degrees {tuple(degrees)}")
        keep.append(tensors)
        arrs.append((C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors]))
    dev = keep[0][0].device
    sizes = (packed_bytes(), int(lib.aon_bwd_packed_bytes()))
    if out is not None:
        pk, bw = ([o[i] for o in out] for i in range(2))
        for got, want in zip(pk + bw, [sizes[0]] * 2 + [sizes[1]] * 2):
            if got is not None and (got.dtype != torch.uint8 or got.numel() != want or got.device != dev or not got.is_contiguous()):
                raise ValueError(f"vanilla_pack_step: out buffers must be contiguous uint8 tensors of the library's sizes on {dev}")
    else:
        pk = [torch.empty(sizes[0], dtype=torch.uint8, device=dev) for _ in range(2)]
        bw = [torch.empty(sizes[1], dtype=torch.uint8, device=dev) if with_bwd else None for _ in range(2)]
    with torch.cuda.device(dev):
        check(lib.aon_vanilla_pack_step(arrs[0], arrs[1], int(degrees[0]), int(degrees[1]), int(degrees[2]), _ptr(pk[0]), _ptr(bw[0]), _ptr(pk[1]), _ptr(bw[1]),
                                        _stream()), "aon_vanilla_pack_step")
    return [(_tag(pk[l]), None if bw[l] is None else _tag(bw[l])) for l in range(2)]


def art_mlp_fwd(packed, small, rays_o, rays_d, viewdirs, t_vals):
    o, d, v, t = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _f32(viewdirs, "viewdirs"), _f32(t_vals, "t_vals")
    n, S = t.shape
    raw = torch.empty((n, S, 4), dtype=torch.float32, device=t.device)
    with torch.cuda.device(t.device):
        check(lib.aon_art_mlp_fwd(_pk(packed), _pk(small), _ptr(o), _ptr(d), _ptr(v), _ptr(t), n, S, _ptr(raw), _stream()),
              "aon_art_mlp_fwd")
    return raw


def art_mlp_fwd_pos(packed, small, pos, viewdirs_enc):
    x, c = _f32(pos, "pos"), _f32(viewdirs_enc, "viewdirs_enc")
    n, S, F = x.shape
    if F != 3 or tuple(c.shape) != (n, 27):
        raise ValueError("art_mlp_fwd_pos expects pos (n,S,3) and viewdirs_enc (n,27)")
    raw = torch.empty((n, S, 4), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.aon_art_mlp_fwd_pos(_pk(packed), _pk(small), _ptr(x), _ptr(c), n, S, _ptr(raw), _stream()), "aon_art_mlp_fwd_pos")
    return raw


def art_render_fwd(packed_c, small_c, packed_f, small_f, rays_o, rays_d, viewdirs, near, far, white_bkgd, num_levels=2,
                   t_rand=None, u=None, opts=None, noise=None):
    """NeRF_AE_Art.forward: [(rgb, acc, depth)_coarse, (rgb, acc, depth)_fine]."""
    o, d, v = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _f32(viewdirs, "viewdirs")
    n, dev = o.shape[0], o.device
    op = _opts(opts)
    tr = None if t_rand is None else _f32(t_rand, "t_rand")
    if tr is not None and tuple(tr.shape) != (n, op.Sc):
        raise ValueError(f"t_rand must be ({n},{op.Sc})")
    uu, us = _u_args(u, n, dev, op.num_fine_samples) if num_levels == 2 else (None, 0)
    outs = []
    for _ in range(num_levels):
        outs.append((torch.empty((n, 3), dtype=torch.float32, device=dev), torch.empty((n,), dtype=torch.float32, device=dev),
                     torch.empty((n,), dtype=torch.float32, device=dev)))
    fine = outs[1] if num_levels == 2 else (None, None, None)
    st, keep = op.c_struct(near, far, _check_noise(noise, n, op, num_levels))
    ws = _workspace(dev, n, st)
    with torch.cuda.device(dev):
        check(lib.aon_art_render_fwd_ex(_pk(packed_c), _pk(small_c), _pk(packed_f), _pk(small_f), _ptr(o), _ptr(d), _ptr(v), n,
                 float(near), float(far), int(bool(white_bkgd)), num_levels, _ptr(tr), _ptr(uu), us,
                 _ptr(outs[0][0]), _ptr(outs[0][1]), _ptr(outs[0][2]), _ptr(fine[0]), _ptr(fine[1]), _ptr(fine[2]),
                 _ptr(ws), ws.numel(), _stream(), C.byref(st)), "aon_art_render_fwd")
    return outs


# ------------------------------------------------------------------ R14 training (vanilla)
def vanilla_param_shapes(degrees=(0, 10, 4)) -> dict:
    """Shapes of a default-size NeRFMLP with the given encoding degrees (the fused kernels' networks)."""
    mn, mx, dv = degrees
    P, V = 3 + 6 * (mx - mn), 3 + 6 * dv
    shapes = dict(VANILLA_PARAM_SHAPES)
    shapes.update({"pts_linears.0.weight": (256, P), "pts_linears.5.weight": (256, 256 + P), "views_linear.0.weight": (128, 256 + V)})
    return shapes


def pack_vanilla_mlp_bwd(params: dict, out: torch.Tensor | None = None, degrees=(0, 10, 4)) -> torch.Tensor:
    """Transposed weight stream for the backward data chain (re-pack whenever the parameters change)."""
    shapes = vanilla_param_shapes(degrees)
    tensors = [_f32(params[name].detach(), name) for name in VANILLA_PARAM_ORDER]
    for name, t in zip(VANILLA_PARAM_ORDER, tensors):
        if tuple(t.shape) != shapes[name]:
            raise ValueError(f"{name}: shape {tuple(t.shape)} != {shapes[name]} for encoding degrees {tuple(degrees)}")
    dev = tensors[0].device
    if out is None:
        out = torch.empty(int(lib.aon_bwd_packed_bytes()), dtype=torch.uint8, device=dev)
    arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    with torch.cuda.device(dev):
        if tuple(degrees) == (0, 10, 4):
            check(lib.aon_pack_vanilla_mlp_bwd(arr, _ptr(out), _stream()), "aon_pack_vanilla_mlp_bwd")
        else:
            check(lib.aon_pack_vanilla_mlp_bwd_deg(arr, degrees[0], degrees[1], degrees[2], _ptr(out), _stream()), "aon_pack_vanilla_mlp_bwd_deg")
    return _tag(out)


def padded_samples(n_samples: int) -> int:
    return (n_samples + 127) // 128 * 128


# Training planes are STEP-MAJOR (include/aon_hip.h, csrc/aon_mlp_core.h): a tensor of shape (Np / 32, rows / 4, 32, 4) =
# [step of 32 samples][feature row / 4][sample in step][feature row % 4].
def _new_planes(rows: int, Np: int, device) -> torch.Tensor:
    return torch.empty((Np // 32, rows // 4, 32, 4), dtype=torch.float32, device=device)


def plane_samples(planes: torch.Tensor) -> int:
    """Np, the padded sample count of a plane buffer."""
    return planes.shape[0] * 32


def plane_rows_view(planes: torch.Tensor) -> torch.Tensor:
    """(rows, Np) view-copy of step-major planes: row f, column s = feature row f of sample s (tests, diagnostics)."""
    nst, ng = planes.shape[0], planes.shape[1]
    return planes.permute(1, 3, 0, 2).reshape(ng * 4, nst * 32)


def mlp_fwd_train(packed, rays_o, rays_d, viewdirs, t_vals):
    """Fused forward that also stores the step-major activation planes and the ReLU bit masks
    -> (raw (n,S,4), planes (Np/32, rows/4, 32, 4), masks)."""
    o, d, v, t = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _f32(viewdirs, "viewdirs"), _f32(t_vals, "t_vals")
    n, S = t.shape
    Np = padded_samples(n * S)
    raw = torch.empty((n, S, 4), dtype=torch.float32, device=t.device)
    # every row the backward consumes is fully written by the kernel (padded columns included); the encoding pad rows
    # (63, and 27..31 of the view block) only ever feed gradient columns that are never emitted -> no zero fill needed
    planes = _new_planes(int(lib.aon_train_plane_rows()), Np, t.device)
    masks = torch.empty(int(lib.aon_train_mask_bytes(Np)), dtype=torch.uint8, device=t.device)
    with torch.cuda.device(t.device):
        check(lib.aon_mlp_fwd_train(_pk(packed), _ptr(o), _ptr(d), _ptr(v), _ptr(t), n, S, _ptr(raw), _ptr(planes), _ptr(masks), _stream()), "aon_mlp_fwd_train")
    return raw, planes, masks


def composite_bwd(raw, t_vals, dirs, g_rgb, g_acc, g_depth, white_bkgd, act, Np: int):
    """-> d_raw (Np,4): dL/d(raw rgb, raw sigma), zero in the padded tail."""
    r, t, d, g = _f32(raw, "raw"), _f32(t_vals, "t_vals"), _f32(dirs, "dirs"), _f32(g_rgb, "g_rgb")
    ga = None if g_acc is None else _f32(g_acc, "g_acc")
    gd = None if g_depth is None else _f32(g_depth, "g_depth")
    n, S = t.shape
    d_raw = torch.zeros((Np, 4), dtype=torch.float32, device=t.device)
    with torch.cuda.device(t.device):
        check(lib.aon_composite_bwd(_ptr(r), _ptr(t), _ptr(d), _ptr(g), _ptr(ga), _ptr(gd), n, S, int(bool(white_bkgd)), act,
                                    _ptr(d_raw), _stream()), "aon_composite_bwd")
    return d_raw


def mlp_bwd_chain(packed_bwd, packed_fwd, d_raw, masks, plane_shape):
    """-> dplanes (layout of the forward planes): pre-activation gradient planes (rows of the encodings are not written / not used)."""
    dplanes = torch.empty(tuple(plane_shape), dtype=torch.float32, device=d_raw.device)
    Np = plane_shape[0] * 32
    with torch.cuda.device(d_raw.device):
        check(lib.aon_mlp_bwd_chain(_pk(packed_bwd), _pk(packed_fwd), _ptr(d_raw), _ptr(masks), _ptr(dplanes), Np, _stream()), "aon_mlp_bwd_chain")
    return dplanes


_WG_WS: dict = {}


def vanilla_wgrad(planes, dplanes, d_raw, packed_bwd):
    """-> dict name -> gradient (the reference's NeRFMLP parameter names / shapes).  ``packed_bwd`` (required): the transposed stream the
    chain of these planes ran with -- its form (bottleneck folded or literal) is the planes' form, and the folded form's buffer carries the
    raw weights the un-folding reads.  (No default: with the fold on by default, "None = literal planes" would silently mis-read folded ones.)"""
    if packed_bwd is None:
        raise ValueError("vanilla_wgrad: pass the transposed stream (pack_vanilla_mlp_bwd) the backward chain of these planes ran with")
    dev = planes.device
    key = str(dev)
    need = int(lib.aon_wgrad_workspace_bytes())
    if key not in _WG_WS or _WG_WS[key].numel() < need:
        _WG_WS[key] = torch.empty(need, dtype=torch.uint8, device=dev)
    ws = _WG_WS[key]
    grads = {name: torch.empty(VANILLA_PARAM_SHAPES[name], dtype=torch.float32, device=dev) for name in VANILLA_PARAM_ORDER}
    arr = (C.c_void_p * len(VANILLA_PARAM_ORDER))(*[grads[n].data_ptr() for n in VANILLA_PARAM_ORDER])
    with torch.cuda.device(dev):
        check(lib.aon_vanilla_wgrad(_ptr(planes), _ptr(dplanes), _ptr(d_raw), plane_samples(planes), arr, _ptr(ws), ws.numel(), _stream(),
                                    _pk(packed_bwd)), "aon_vanilla_wgrad")
    return grads


# ------------------------------------------------------------------ R14 training (articulated)
def pack_art_mlp_bwd(params: dict, out: torch.Tensor | None = None, degrees=(0, 10, 4)) -> torch.Tensor:
    tensors, arr = _art_param_array(params, degrees)
    dev = tensors[0].device
    if out is None:
        out = torch.empty(int(lib.aon_art_bwd_packed_bytes()), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(lib.aon_pack_art_mlp_bwd_deg(arr, int(degrees[0]), int(degrees[1]), int(degrees[2]), _ptr(out), _stream()), "aon_pack_art_mlp_bwd_deg")
    return _tag(out)


def art_mlp_fwd_train(packed, small, rays_o, rays_d, viewdirs, t_vals):
    o, d, v, t = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _f32(viewdirs, "viewdirs"), _f32(t_vals, "t_vals")
    n, S = t.shape
    Np = padded_samples(n * S)
    raw = torch.empty((n, S, 4), dtype=torch.float32, device=t.device)
    planes = _new_planes(int(lib.aon_art_train_plane_rows()), Np, t.device)
    masks = torch.empty(int(lib.aon_art_train_mask_bytes(Np)), dtype=torch.uint8, device=t.device)
    with torch.cuda.device(t.device):
        check(lib.aon_art_mlp_fwd_train(_pk(packed), _pk(small), _ptr(o), _ptr(d), _ptr(v), _ptr(t), n, S, _ptr(raw), _ptr(planes), _ptr(masks), _stream()),
              "aon_art_mlp_fwd_train")
    return raw, planes, masks


def art_bwd_chain(packed_bwd, small, d_raw, masks, planes):
    """-> (dplanes (layout of the forward planes), dxp (Np,4) = dL/d deformed position)."""
    dplanes = torch.empty(planes.shape, dtype=torch.float32, device=planes.device)
    Np = plane_samples(planes)
    dxp = torch.empty((Np, 4), dtype=torch.float32, device=planes.device)
    with torch.cuda.device(planes.device):
        check(lib.aon_art_bwd_chain(_pk(packed_bwd), _pk(small), _ptr(d_raw), _ptr(masks), _ptr(planes), _ptr(dplanes), _ptr(dxp), Np, _stream()),
              "aon_art_bwd_chain")
    return dplanes, dxp


def art_wgrad(planes, dplanes, d_raw, dxp, params: dict, latents: dict, degrees=(0, 10, 4), *, packed_bwd):
    """-> (dict name -> parameter gradient, dict latent key -> gradient (flat)).  ``packed_bwd`` (required keyword): as vanilla_wgrad."""
    if packed_bwd is None:
        raise ValueError("art_wgrad: pass packed_bwd= the transposed stream (pack_art_mlp_bwd) the backward chain of these planes ran with")
    dev = planes.device
    key = str(dev)
    need = int(lib.aon_wgrad_workspace_bytes())
    if key not in _WG_WS or _WG_WS[key].numel() < need:
        _WG_WS[key] = torch.empty(need, dtype=torch.uint8, device=dev)
    ws = _WG_WS[key]
    tensors, parr = _art_param_array(params, degrees)
    shape, app, art = _latent(latents, "density", 128), _latent(latents, "color", 128), _latent(latents, "articulation", 32)
    shapes = art_param_shapes(degrees)
    grads = {name: torch.empty(shapes[name], dtype=torch.float32, device=dev) for name in ART_PARAM_ORDER}
    garr = (C.c_void_p * len(ART_PARAM_ORDER))(*[grads[n].data_ptr() for n in ART_PARAM_ORDER])
    g_lat = {"density": torch.empty(128, device=dev), "color": torch.empty(128, device=dev), "articulation": torch.empty(32, device=dev)}
    with torch.cuda.device(dev):
        check(lib.aon_art_wgrad_deg(_ptr(planes), _ptr(dplanes), _ptr(d_raw), _ptr(dxp), plane_samples(planes), parr, _ptr(shape), _ptr(app),
                                    _ptr(art), garr, _ptr(g_lat["density"]), _ptr(g_lat["color"]), _ptr(g_lat["articulation"]), _ptr(ws),
                                    ws.numel(), _stream(), int(degrees[0]), int(degrees[1]), int(degrees[2]), _pk(packed_bwd)), "aon_art_wgrad_deg")
    return grads, g_lat


# ------------------------------------------------------------------ bottleneck_layer folded into views_linear[0] (round 5)
def set_bottleneck_fold(on: bool) -> None:
    """Form of the streams / per-call blocks the pack calls build from now on: folded (default: W' = W_v0[:, :256] W_b, one 256 -> 128
    layer where the reference's graph runs 256 -> 256 then 256 -> 128, model.py:109-114) or the literal two layers.  Buffers already
    packed keep their form (include/aon_hip.h)."""
    check(lib.aon_set_bottleneck_fold(int(bool(on))), "aon_set_bottleneck_fold")


def bottleneck_fold() -> bool:
    return bool(lib.aon_get_bottleneck_fold())


# ------------------------------------------------------------------ R14 training step in two C calls
def set_bwd_overlap(on: bool) -> None:
    """Two-level backward on two library streams (default on) or everything on the caller's stream."""
    check(lib.aon_set_bwd_overlap(2 if on == 2 else int(bool(on))), "aon_set_bwd_overlap")   # 2 (measurements): head reductions on side streams in the merged form


def _sized(nbytes: int, what: str, device) -> torch.Tensor:
    if nbytes < 0:
        check(nbytes, what)
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def set_fwd_overlap(on: bool) -> None:
    """Training forward of two levels as two ray halves on two library streams (default on) or everything on the caller's stream."""
    check(lib.aon_set_fwd_overlap(int(bool(on))), "aon_set_fwd_overlap")


def set_view_bias(on: bool) -> None:
    """Whole-path calls of the folded vanilla network: the view-encoding term as a per-ray bias (default on; same bits as the chunk form)."""
    check(lib.aon_set_view_bias(int(bool(on))), "aon_set_view_bias")


def view_bias_enabled() -> bool:
    """True when the whole-path calls of folded networks use the per-ray view bias (needs the fold: bottleneck_fold())."""
    return bool(lib.aon_get_view_bias()) and bottleneck_fold()


def view_bias(packed: torch.Tensor, viewdirs: torch.Tensor) -> torch.Tensor:
    """(n,128) = b' + W_v0[:, 256:] pos_enc(viewdirs, 0, 4) of a folded vanilla stream: what the whole-path calls start the view layer from."""
    v = _f32(viewdirs, "viewdirs")
    n = v.numel() // 3
    out = torch.empty((n, 128), dtype=torch.float32, device=v.device)
    with torch.cuda.device(v.device):
        check(lib.aon_view_bias(_pk(packed), _ptr(v), n, _ptr(out), _stream()), "aon_view_bias")
    return out


def set_bwd_early_heads(on: bool) -> None:
    """Merged backward: the chain-independent head / bias reductions on a side stream beside the chain launch (default on; same bits)."""
    check(lib.aon_set_bwd_early_heads(int(bool(on))), "aon_set_bwd_early_heads")


def set_bwd_merge(on: bool) -> None:
    """Backward chains of the two levels as ONE persistent launch of two segments (default on; off: one launch per level, round 3)."""
    check(lib.aon_set_bwd_merge(int(bool(on))), "aon_set_bwd_merge")


def set_fwd_merge(on: bool) -> None:
    """Training forward of two levels as three persistent launches, coarse(A) | fine(A) + coarse(B) | fine(B) (default on; off: the
    forms of set_fwd_overlap)."""
    check(lib.aon_set_fwd_merge(2 if on == 2 else int(bool(on))), "aon_set_fwd_merge")   # 2 (tests): merge even when no round is saved


# The two big buffers of a training step -- the forward -> backward workspace (15 GB at 4096 articulated rays) and the backward's scratch
# (11 GB) -- are taken from / given back to a small POOL of this module instead of torch's caching allocator (round 6).  Through the allocator
# they were freed and re-requested every step, and while such a block sat free the step's own 2-7 MB allocations (weight streams, gradient
# buffers) could be carved out of it: the next request for the full size then found no block and the allocator went to the driver -- ONE
# hipMalloc of 11-15 GB, 80-125 ms of host time, once per process at a step that depends on timing (measured: 4 device mallocs and +13.9 GB
# reserved INSIDE the timed loop of every run of tools/train_bench.py; when the stall fell on an early step the device ran dry and the run
# averaged 33-40 ms per step instead of 30.4, tools/slowmode_probe.sh).  A pooled buffer is only handed to the stream it was returned on;
# at most two per (device, size) are kept; release_workspaces() drops them.
import collections as _collections

_TRAIN_POOL: "_collections.OrderedDict" = _collections.OrderedDict()
_TRAIN_POOL_KEEP = 2          # buffers kept per (device, size)
_TRAIN_POOL_SIZES = 6         # distinct (device, size) keys kept, least recently returned dropped first (a loop whose ray count changes every
                              # step must not pile up buffers: workspace + scratch of the full batch and of an epoch's tail batch are four keys)


def _pool_take(nbytes: int, what: str, device) -> torch.Tensor:
    if nbytes < 0:
        check(nbytes, what)
    dev = torch.device(device)
    if dev.type == "cuda":
        stream = torch.cuda.current_stream(dev).cuda_stream
        key = (dev.index if dev.index is not None else torch.cuda.current_device(), nbytes)
        free = _TRAIN_POOL.get(key)
        if free:
            _TRAIN_POOL.move_to_end(key)
            for i, (t, s) in enumerate(free):
                if s == stream:
                    del free[i]
                    return t
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def pool_give(t) -> None:
    """Hand a training workspace / scratch back (stream-ordered: everything enqueued on the current stream so far may still use it, whatever
    takes it next is enqueued behind).  Called by the backward once it has enqueued its last launch."""
    if t is None or not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.uint8 or t.dim() != 1:
        return
    key = (t.device.index, t.numel())
    free = _TRAIN_POOL.setdefault(key, [])
    _TRAIN_POOL.move_to_end(key)
    if len(free) < _TRAIN_POOL_KEEP:
        free.append((t, torch.cuda.current_stream(t.device).cuda_stream))
    while len(_TRAIN_POOL) > _TRAIN_POOL_SIZES:
        _TRAIN_POOL.popitem(last=False)


def train_workspace(device, n_rays: int, articulated: bool, num_levels: int = 2, st=None) -> torch.Tensor:
    """Workspace of one training forward/backward pair (it carries the forward's planes to the backward, so it is owned by the autograd
    graph until the backward gives it back to the pool: `pool_give`); sized by the levels in use."""
    return _pool_take(int(lib.aon_train_workspace_bytes_ex(n_rays, int(articulated), num_levels, None if st is None else C.byref(st))),
                      "aon_train_workspace_bytes_ex", device)


def train_scratch(device, n_rays: int, articulated: bool, num_levels: int = 2, st=None) -> torch.Tensor:
    """The backward's own temporaries (gradient planes, d_raw, weight-gradient partials): taken from the pool when the backward runs and
    given back right after, so a live graph pins the forward's workspace only."""
    return _pool_take(int(lib.aon_train_scratch_bytes_ex(n_rays, int(articulated), num_levels, None if st is None else C.byref(st))),
                      "aon_train_scratch_bytes_ex", device)


def _level_outs(n, dev, num_levels):
    outs = [(torch.empty((n, 3), dtype=torch.float32, device=dev), torch.empty((n,), dtype=torch.float32, device=dev),
             torch.empty((n,), dtype=torch.float32, device=dev)) for _ in range(num_levels)]
    return outs, (outs[1] if num_levels == 2 else (None, None, None))


def render_fwd_train(packed_c, packed_f, rays_o, rays_d, viewdirs, near, far, white_bkgd, num_levels, t_rand, u, small_c=None, small_f=None,
                     opts=None, noise=None):
    """NeRF.forward / NeRF_AE_Art.forward (small blocks given) under grad mode in ONE C call -> (outs, workspace, geometry).
    ``geometry`` = (aon_render_opts struct, tensors it points at): the backward of this forward must be given the same one."""
    o, d, v = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _f32(viewdirs, "viewdirs")
    n, dev = o.shape[0], o.device
    art = small_c is not None
    op = _opts(opts)
    tr = None if t_rand is None else _f32(t_rand, "t_rand")
    if tr is not None and tuple(tr.shape) != (n, op.Sc):
        raise ValueError(f"t_rand must be ({n},{op.Sc})")
    uu, us = _u_args(u, n, dev, op.num_fine_samples) if num_levels == 2 else (None, 0)
    outs, fine = _level_outs(n, dev, num_levels)
    st, keep = op.c_struct(near, far, _check_noise(noise, n, op, num_levels))
    ws = train_workspace(dev, n, art, num_levels, st)
    common = (_ptr(o), _ptr(d), _ptr(v), n, float(near), float(far), int(bool(white_bkgd)), num_levels, _ptr(tr), _ptr(uu), us,
              _ptr(outs[0][0]), _ptr(outs[0][1]), _ptr(outs[0][2]), _ptr(fine[0]), _ptr(fine[1]), _ptr(fine[2]), _ptr(ws), ws.numel(), _stream(),
              C.byref(st))
    with torch.cuda.device(dev):
        if art:
            check(lib.aon_art_render_fwd_train_ex(_pk(packed_c), _pk(small_c), _pk(packed_f), _pk(small_f), *common), "aon_art_render_fwd_train")
        else:
            check(lib.aon_render_fwd_train_ex(_pk(packed_c), _pk(packed_f), *common), "aon_render_fwd_train")
    return outs, ws, (st, keep)


def _grad_dicts(order, shapes, num_levels, dev, grads_out):
    """Per level {parameter name: tensor the C call writes the gradient into}: fresh tensors, or the caller's (checked: fp32, contiguous,
    right shape and device -- the kernels write every element, 16-byte vector stores included)."""
    if grads_out is None:
        return [{name: torch.empty(shapes[name], dtype=torch.float32, device=dev) for name in order} for _ in range(num_levels)]
    out = []
    for lvl in range(num_levels):
        d = {}
        for name, t in zip(order, grads_out[lvl]):
            if tuple(t.shape) != tuple(shapes[name]) or t.dtype != torch.float32 or t.device != dev or not t.is_contiguous() or (t.data_ptr() & 15):
                raise ValueError(f"gradient buffer for {name}: expected a contiguous 16-byte aligned float32 {tuple(shapes[name])} tensor on {dev}")
            d[name] = t
        out.append(d)
    return out


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[0 if t is None else t.data_ptr() for t in tensors])


def render_bwd(ws, packs_bwd, packs_fwd, rays_d, white_bkgd, num_levels, g_rgb, g_acc, g_depth, geometry=None, grads_out=None):
    """loss.backward() through render_fwd_train (vanilla): g_* = per-level lists (entries may be None except g_rgb)
    -> per-level dicts of the 24 parameter gradients (shapes of the network's own encoding degrees, read from `geometry`).
    `grads_out`: per level, the 24 tensors to write them into (views of a gradient arena, aon_amd/arena.py) instead of fresh ones."""
    d = _f32(rays_d, "rays_d")
    n, dev = d.shape[0], d.device
    st0 = None if geometry is None else geometry[0]
    shapes = vanilla_param_shapes((0, 10, 4) if st0 is None else (st0.min_deg_point, st0.max_deg_point, st0.deg_view))
    grads = _grad_dicts(VANILLA_PARAM_ORDER, shapes, num_levels, dev, grads_out)
    garr = [_ptr_array([g[nm] for nm in VANILLA_PARAM_ORDER]) for g in grads] + [None] * (2 - num_levels)
    pb, pf = list(packs_bwd) + [None] * (2 - num_levels), list(packs_fwd) + [None] * (2 - num_levels)
    keep = [None if t is None else _f32(t, "grad") for t in list(g_rgb) + list(g_acc) + list(g_depth)]
    k = num_levels
    st = None if geometry is None else geometry[0]
    scratch = train_scratch(dev, n, False, num_levels, st)
    with torch.cuda.device(dev):
        check(lib.aon_render_bwd_ex(_pk(pb[0]), _pk(pf[0]), _pk(pb[1]), _pk(pf[1]), _ptr(d), n, int(bool(white_bkgd)), num_levels,
                                    _ptr_array(keep[:k]), _ptr_array(keep[k:2 * k]), _ptr_array(keep[2 * k:3 * k]), garr[0], garr[1], _ptr(ws), ws.numel(),
                                    _ptr(scratch), scratch.numel(), _stream(), None if st is None else C.byref(st)), "aon_render_bwd")
    pool_give(scratch)
    return grads


def art_render_bwd(ws, packs_bwd, smalls, rays_d, white_bkgd, num_levels, g_rgb, g_acc, g_depth, params_per_level, latents: dict, geometry=None,
                   grads_out=None):
    """Articulated twin -> (per-level dicts of the 40 parameter gradients, dict of latent gradients summed over the levels).
    `grads_out`: as render_bwd."""
    d = _f32(rays_d, "rays_d")
    n, dev = d.shape[0], d.device
    st0 = None if geometry is None else geometry[0]
    degrees = (0, 10, 4) if st0 is None else (int(st0.min_deg_point), int(st0.max_deg_point), int(st0.deg_view))
    shapes = art_param_shapes(degrees)
    grads = _grad_dicts(ART_PARAM_ORDER, shapes, num_levels, dev, grads_out)
    garr = [_ptr_array([g[nm] for nm in ART_PARAM_ORDER]) for g in grads] + [None] * (2 - num_levels)
    tens, parr = [], []
    for params in params_per_level:
        t, arr = _art_param_array(params, degrees)
        tens.append(t)
        parr.append(arr)
    parr += [None] * (2 - num_levels)
    pb, sm = list(packs_bwd) + [None] * (2 - num_levels), list(smalls) + [None] * (2 - num_levels)
    shape, app, art = _latent(latents, "density", 128), _latent(latents, "color", 128), _latent(latents, "articulation", 32)
    g_lat = {"density": torch.empty(128, device=dev), "color": torch.empty(128, device=dev), "articulation": torch.empty(32, device=dev)}
    keep = [None if t is None else _f32(t, "grad") for t in list(g_rgb) + list(g_acc) + list(g_depth)]
    k = num_levels
    st = None if geometry is None else geometry[0]
    scratch = train_scratch(dev, n, True, num_levels, st)
    with torch.cuda.device(dev):
        check(lib.aon_art_render_bwd_ex(_pk(pb[0]), _pk(sm[0]), _pk(pb[1]), _pk(sm[1]), _ptr(d), n, int(bool(white_bkgd)), num_levels,
                                        _ptr_array(keep[:k]), _ptr_array(keep[k:2 * k]), _ptr_array(keep[2 * k:3 * k]), parr[0], parr[1],
                                        _ptr(shape), _ptr(app), _ptr(art), garr[0], garr[1], _ptr(g_lat["density"]), _ptr(g_lat["color"]),
                                        _ptr(g_lat["articulation"]), _ptr(ws), ws.numel(), _ptr(scratch), scratch.numel(), _stream(),
                                        None if st is None else C.byref(st)), "aon_art_render_bwd")
    pool_give(scratch)
    return grads, g_lat


# ------------------------------------------------------------------ NeRFMLP of any constructor geometry (csrc/aon_gmlp.hip)
class MlpGeometry:
    """The arguments of ``NeRFMLP.__init__`` (model.py:40-54) -> ``aon_mlp_geometry``; parameter names / shapes / order of the
    module that constructor builds."""

    FIELDS = ("min_deg_point", "max_deg_point", "deg_view", "netdepth", "netwidth", "netdepth_condition", "netwidth_condition",
              "skip_layer", "input_ch", "input_ch_view", "num_rgb_channels", "num_density_channels")
    DEFAULT = (0, 10, 4, 8, 256, 1, 128, 4, 3, 3, 3, 1)

    def __init__(self, min_deg_point=0, max_deg_point=10, deg_view=4, netdepth=8, netwidth=256, netdepth_condition=1,
                 netwidth_condition=128, skip_layer=4, input_ch=3, input_ch_view=3, num_rgb_channels=3, num_density_channels=1):
        vals = (min_deg_point, max_deg_point, deg_view, netdepth, netwidth, netdepth_condition, netwidth_condition, skip_layer, input_ch,
                input_ch_view, num_rgb_channels, num_density_channels)
        for name, v in zip(self.FIELDS, vals):
            setattr(self, name, int(v))
        self.pos_size = ((self.max_deg_point - self.min_deg_point) * 2 + 1) * self.input_ch
        self.view_pos_size = (self.deg_view * 2 + 1) * self.input_ch_view
        st = self.c_struct()
        if lib.aon_gmlp_param_count(C.byref(st)) < 0:     # the C side owns the validity rules
            check(-1, "NeRFMLP geometry")

    def as_tuple(self):
        return tuple(getattr(self, n) for n in self.FIELDS)

    @property
    def is_default(self) -> bool:
        return self.as_tuple() == self.DEFAULT

    @property
    def fits_fused_inference(self) -> bool:
        """Default widths / depths with at most 10 position and 4 view frequency levels: the fused inference kernels take it
        (zero-weight slots for the missing levels, aon_pack_vanilla_mlp_deg)."""
        return (self.as_tuple()[3:] == self.DEFAULT[3:] and 0 <= self.max_deg_point - self.min_deg_point <= 10
                and 0 <= self.deg_view <= 4)

    def c_struct(self):
        st = _lib.MlpGeometryC()
        for n in self.FIELDS:
            setattr(st, n, getattr(self, n))
        return st

    def cat_before(self, l: int) -> bool:
        """layer l reads cat([x, inputs]) (model.py:75-76)"""
        return l >= 2 and (l - 1) % self.skip_layer == 0

    @property
    def param_order(self):
        names = [f"pts_linears.{i}" for i in range(self.netdepth)] + [f"views_linear.{i}" for i in range(self.netdepth_condition)]
        names += ["bottleneck_layer", "density_layer", "rgb_layer"]
        return [f"{m}.{k}" for m in names for k in ("weight", "bias")]

    @property
    def param_shapes(self) -> dict:
        W, Wc, P, V = self.netwidth, self.netwidth_condition, self.pos_size, self.view_pos_size
        out = {}
        for l in range(self.netdepth):
            out[f"pts_linears.{l}.weight"] = (W, P if l == 0 else (W + P if self.cat_before(l) else W))
            out[f"pts_linears.{l}.bias"] = (W,)
        for i in range(self.netdepth_condition):
            out[f"views_linear.{i}.weight"] = (Wc, W + V if i == 0 else Wc)
            out[f"views_linear.{i}.bias"] = (Wc,)
        out.update({"bottleneck_layer.weight": (W, W), "bottleneck_layer.bias": (W,),
                    "density_layer.weight": (self.num_density_channels, W), "density_layer.bias": (self.num_density_channels,),
                    "rgb_layer.weight": (self.num_rgb_channels, Wc), "rgb_layer.bias": (self.num_rgb_channels,)})
        return out


def _gmlp_param_array(geom: MlpGeometry, params: dict):
    shapes = geom.param_shapes
    tensors = []
    for name in geom.param_order:
        t = _f32(params[name].detach(), name)
        if tuple(t.shape) != shapes[name]:
            raise ValueError(f"{name}: shape {tuple(t.shape)} != {shapes[name]} for NeRFMLP geometry {geom.as_tuple()}")
        tensors.append(t)
    return tensors, (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def gmlp_fwd(geom: MlpGeometry, params: dict, samples_enc, viewdirs_enc):
    """NeRFMLP.forward(x, condition) for any geometry -> (raw_rgb (n,S,C_rgb), raw_density (n,S,C_density))."""
    x, c = _f32(samples_enc, "samples_enc"), _f32(viewdirs_enc, "viewdirs_enc")
    n, S, F = x.shape
    if F != geom.pos_size or tuple(c.shape) != (n, geom.view_pos_size):
        raise ValueError(f"expected samples_enc (n,S,{geom.pos_size}) and viewdirs_enc (n,{geom.view_pos_size})")
    tensors, arr = _gmlp_param_array(geom, params)
    st = geom.c_struct()
    dev = x.device
    rgb = torch.empty((n, S, geom.num_rgb_channels), dtype=torch.float32, device=dev)
    dens = torch.empty((n, S, geom.num_density_channels), dtype=torch.float32, device=dev)
    ws = _sized(int(lib.aon_gmlp_workspace_bytes(C.byref(st), n * S)), "aon_gmlp_workspace_bytes", dev)
    with torch.cuda.device(dev):
        check(lib.aon_gmlp_fwd(C.byref(st), arr, _ptr(x), _ptr(c), n, S, _ptr(rgb), _ptr(dens), _ptr(ws), ws.numel(), _stream()), "aon_gmlp_fwd")
    return rgb, dens


# Workspace of the layer-wise engine's inference call: its activations live in HBM (~ (P + 3 W + 2 Wc) * 4 B per sample) and the C side
# chunks over rays to whatever workspace it is given, so the chunk is sized from a BYTE budget (round 3: a fixed 8,192 rays = ~7 GB
# at 256-wide networks, far more at the widths make_gg accepts, cached per device until release_workspaces(): ADVICE r3).  At 2 GB
# the default-width network gets ~2,400-ray chunks of 193 samples = 460 k-row GEMMs: still hundreds of 128-row tiles per launch.
G_WS_BUDGET_BYTES = 2 << 30
G_CHUNK_RAYS = 8192            # upper bound on rays per chunk
_GWS_CACHE: dict = {}


def _grender_chunk_rays(gst, st, n: int) -> tuple[int, int]:
    """(rays per chunk, workspace bytes) under G_WS_BUDGET_BYTES; never below 128 rays (then the budget is exceeded, as it must be)."""
    rays = max(1, min(n, G_CHUNK_RAYS))
    while True:
        need = int(lib.aon_grender_workspace_bytes(C.byref(gst), rays, C.byref(st)))
        if need < 0:
            check(need, "aon_grender_workspace_bytes")
        if need <= G_WS_BUDGET_BYTES or rays <= 128:
            return rays, need
        rays = max(128, min(rays // 2, int(rays * G_WS_BUDGET_BYTES / need)))


def grender_fwd(geom: MlpGeometry, params_c: dict, params_f, rays_o, rays_d, viewdirs, near, far, white_bkgd, num_levels=2, t_rand=None, u=None,
                opts=None, noise=None):
    """NeRF.forward with a NeRFMLP of any geometry (aon_grender_fwd)."""
    o, d, v = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _f32(viewdirs, "viewdirs")
    n, dev = o.shape[0], o.device
    op = _opts(opts)
    tr = None if t_rand is None else _f32(t_rand, "t_rand")
    if tr is not None and tuple(tr.shape) != (n, op.Sc):
        raise ValueError(f"t_rand must be ({n},{op.Sc})")
    uu, us = _u_args(u, n, dev, op.num_fine_samples) if num_levels == 2 else (None, 0)
    outs, fine = _level_outs(n, dev, num_levels)
    st, keep = op.c_struct(near, far, _check_noise(noise, n, op, num_levels))
    gst = geom.c_struct()
    tc, arr_c = _gmlp_param_array(geom, params_c)
    tf, arr_f = _gmlp_param_array(geom, params_f) if num_levels == 2 else (None, None)
    _, need = _grender_chunk_rays(gst, st, n)
    key = str(dev)
    ws = _GWS_CACHE.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        _GWS_CACHE[key] = ws
    with torch.cuda.device(dev):
        check(lib.aon_grender_fwd(C.byref(gst), arr_c, arr_f, _ptr(o), _ptr(d), _ptr(v), n, float(near), float(far), int(bool(white_bkgd)), num_levels,
                                  _ptr(tr), _ptr(uu), us, _ptr(outs[0][0]), _ptr(outs[0][1]), _ptr(outs[0][2]), _ptr(fine[0]), _ptr(fine[1]),
                                  _ptr(fine[2]), _ptr(ws), ws.numel(), _stream(), C.byref(st)), "aon_grender_fwd")
    return outs


def grender_fwd_train(geom: MlpGeometry, params_c: dict, params_f, rays_o, rays_d, viewdirs, near, far, white_bkgd, num_levels, t_rand, u,
                      opts=None, noise=None):
    """-> (outs, workspace, geometry): the forward of a training step; the workspace carries every layer's output to the backward."""
    o, d, v = _f32(rays_o, "rays_o"), _f32(rays_d, "rays_d"), _f32(viewdirs, "viewdirs")
    n, dev = o.shape[0], o.device
    op = _opts(opts)
    tr = None if t_rand is None else _f32(t_rand, "t_rand")
    if tr is not None and tuple(tr.shape) != (n, op.Sc):
        raise ValueError(f"t_rand must be ({n},{op.Sc})")
    uu, us = _u_args(u, n, dev, op.num_fine_samples) if num_levels == 2 else (None, 0)
    outs, fine = _level_outs(n, dev, num_levels)
    st, keep = op.c_struct(near, far, _check_noise(noise, n, op, num_levels))
    gst = geom.c_struct()
    tc, arr_c = _gmlp_param_array(geom, params_c)
    tf, arr_f = _gmlp_param_array(geom, params_f) if num_levels == 2 else (None, None)
    ws = _pool_take(int(lib.aon_grender_train_workspace_bytes(C.byref(gst), n, num_levels, C.byref(st))), "aon_grender_train_workspace_bytes", dev)   # (pooled: see _TRAIN_POOL)
    with torch.cuda.device(dev):
        check(lib.aon_grender_fwd_train(C.byref(gst), arr_c, arr_f, _ptr(o), _ptr(d), _ptr(v), n, float(near), float(far), int(bool(white_bkgd)),
                                        num_levels, _ptr(tr), _ptr(uu), us, _ptr(outs[0][0]), _ptr(outs[0][1]), _ptr(outs[0][2]), _ptr(fine[0]),
                                        _ptr(fine[1]), _ptr(fine[2]), _ptr(ws), ws.numel(), _stream(), C.byref(st)), "aon_grender_fwd_train")
    return outs, ws, (st, keep, uu)


def grender_bwd(geom: MlpGeometry, ws, params_per_level, rays_d, white_bkgd, num_levels, g_rgb, g_acc, g_depth, geometry):
    """loss.backward() through grender_fwd_train -> per-level dicts of parameter gradients."""
    d = _f32(rays_d, "rays_d")
    n, dev = d.shape[0], d.device
    gst = geom.c_struct()
    shapes, order = geom.param_shapes, geom.param_order
    grads = [{name: torch.empty(shapes[name], dtype=torch.float32, device=dev) for name in order} for _ in range(num_levels)]
    garr = [_ptr_array([g[nm] for nm in order]) for g in grads] + [None] * (2 - num_levels)
    tens, parr = [], []
    for params in params_per_level:
        t, arr = _gmlp_param_array(geom, params)
        tens.append(t)
        parr.append(arr)
    parr += [None] * (2 - num_levels)
    keep = [None if t is None else _f32(t, "grad") for t in list(g_rgb) + list(g_acc) + list(g_depth)]
    k = num_levels
    st = geometry[0]
    scratch = _pool_take(int(lib.aon_grender_train_scratch_bytes(C.byref(gst), n, num_levels, C.byref(st))), "aon_grender_train_scratch_bytes", dev)
    with torch.cuda.device(dev):
        check(lib.aon_grender_bwd(C.byref(gst), parr[0], parr[1], _ptr(d), n, int(bool(white_bkgd)), num_levels, _ptr_array(keep[:k]),
                                  _ptr_array(keep[k:2 * k]), _ptr_array(keep[2 * k:3 * k]), garr[0], garr[1], _ptr(ws), ws.numel(), _ptr(scratch),
                                  scratch.numel(), _stream(), C.byref(st)), "aon_grender_bwd")
    pool_give(scratch)
    return grads


def release_workspaces() -> None:
    """Drop the per-device workspace caches of the inference calls (fused path: up to 1.35 GB, or 3.3 GB with materialised
    encodings; layer-wise engine: G_WS_BUDGET_BYTES = 2 GB) back to torch's caching allocator.  They are re-made on the
    next call; and the pool of training workspaces / scratch buffers (`_TRAIN_POOL`: up to two of each size in use, 26 GB per pair at
    4096 articulated rays)."""
    _TRAIN_POOL.clear()
    _WS_CACHE.clear()
    _GWS_CACHE.clear()
    _WG_WS.clear()


# ------------------------------------------------------------------ measurement aid
def profile_begin() -> None:
    check(lib.aon_profile_begin(), "aon_profile_begin")


def profile_end():
    """-> (mlp_kernel_ms_total, launches, samples) for the fused-MLP launches since profile_begin()."""
    ms, launches, samples = C.c_double(0), C.c_int64(0), C.c_int64(0)
    check(lib.aon_profile_end(C.byref(ms), C.byref(launches), C.byref(samples)), "aon_profile_end")
    return ms.value, launches.value, samples.value


PROF_CLASSES = {"mlp_fwd": 0, "bwd_chain": 1, "wgrad": 2, "composite": 3, "sample_pdf": 4, "composite_bwd": 5, "composite_pdf": 6, "sample_t": 7}


def profile_classes() -> dict:
    """Per kernel class of the interval closed by the last profile_end(): {name: (ms_total, launches, units)}; units are
    samples for mlp_fwd / bwd_chain / wgrad and rays for the per-ray kernels."""
    out = {}
    for name, cls in PROF_CLASSES.items():
        ms, launches, units = C.c_double(0), C.c_int64(0), C.c_int64(0)
        check(lib.aon_profile_class(cls, C.byref(ms), C.byref(launches), C.byref(units)), "aon_profile_class")
        out[name] = (ms.value, launches.value, units.value)
    return out


# ------------------------------------------------------------------ the end of a training step on one parameter arena (csrc/aon_optim.hip)
def adam_step(params_flat, grads_flat, exp_avg, exp_avg_sq, begin: int, count: int, lr: float, beta1: float, beta2: float, eps: float, step: int) -> None:
    """torch.optim.Adam's update (the reference's optimizer: model.py:386-389) on elements [begin, begin + count) of four flat fp32 buffers of
    one layout, ONE launch.  `step` = the count after this update."""
    for name, t in (("params", params_flat), ("grads", grads_flat), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.dim() == 1):
            raise RuntimeError(f"adam_step: {name} must be a flat contiguous float32 cuda tensor (no CPU fallback)")
        if begin < 0 or count < 0 or begin + count > t.numel():
            raise ValueError(f"adam_step: range [{begin}, {begin + count}) outside {name} ({t.numel()} elements)")
    off = 4 * int(begin)
    with torch.cuda.device(params_flat.device):
        check(lib.aon_adam_step(C.c_void_p(params_flat.data_ptr() + off), C.c_void_p(grads_flat.data_ptr() + off), C.c_void_p(exp_avg.data_ptr() + off),
                                C.c_void_p(exp_avg_sq.data_ptr() + off), int(count), float(lr), float(beta1), float(beta2), float(eps), int(step), _stream()),
              "aon_adam_step")


def _code_library_args(tables, ids):
    rows = (C.c_int * 3)(*[int(t.shape[0]) for t in tables])
    dims = (C.c_int * 3)(*[int(t.shape[1]) for t in tables])
    idp = (C.c_void_p * 3)(*[i.data_ptr() for i in ids])
    return rows, dims, idp


def code_library_fwd(tables, ids):
    """CodeLibraryArticulated.forward (code_library.py:36-53) for ids of ONE element each: tables = (shape, appearance, articulation) weight
    matrices, ids = (instance_id, instance_id, articulation_id) int64 device tensors -> three (1, dim) rows, one launch."""
    tabs = [_f32(t, "table") for t in tables]
    idl = [i if (i.dtype == torch.int64 and i.is_cuda) else i.to(device=tabs[0].device, dtype=torch.int64) for i in ids]
    if any(i.numel() != 1 for i in idl):
        raise ValueError("code_library_fwd: one id per table (the reference's batch of one object in one state)")
    outs = [torch.empty((1, t.shape[1]), dtype=torch.float32, device=t.device) for t in tabs]
    rows, dims, idp = _code_library_args(tabs, idl)
    with torch.cuda.device(tabs[0].device):
        check(lib.aon_code_library_fwd(_ptr_array(tabs), idp, rows, dims, _ptr_array(outs), _stream()), "aon_code_library_fwd")
    return outs, idl


def code_library_bwd(g_rows, ids, shapes, outs=None):
    """The dense table gradients of the three lookups (what nn.Embedding's autograd produces), one launch; `outs`: tensors to write into
    (the gradient arena's views) or None."""
    dev = ids[0].device
    gr = [_f32(g, "g_row") if g is not None else torch.zeros((1, shp[1]), dtype=torch.float32, device=dev) for g, shp in zip(g_rows, shapes)]
    if outs is None:
        outs = [torch.empty(tuple(shp), dtype=torch.float32, device=dev) for shp in shapes]
    rows = (C.c_int * 3)(*[int(s[0]) for s in shapes])
    dims = (C.c_int * 3)(*[int(s[1]) for s in shapes])
    idp = (C.c_void_p * 3)(*[i.data_ptr() for i in ids])
    with torch.cuda.device(dev):
        check(lib.aon_code_library_bwd(_ptr_array(gr), idp, rows, dims, _ptr_array(outs), _stream()), "aon_code_library_bwd")
    return outs
