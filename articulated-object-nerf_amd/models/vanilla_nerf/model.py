"""Drop-in ``NeRFMLP`` / ``NeRF`` modules: same constructor defaults, attribute and parameter names (so the
reference's Lightning checkpoints load unchanged) and the same ``forward`` signatures and return structure as
``models/vanilla_nerf/model.py:39-199`` -- with ``forward`` running on the fused HIP kernels.

Two engines, chosen by the constructor arguments.  The reference's default NeRFMLP (8x256 trunk with a skip at layer 4,
1x128 view branch, 10/4 encoding degrees) -- the only one its CLI can build (``LitNeRF`` builds ``NeRF()``, model.py:218) --
runs on the fused register-resident kernels.  Every other ``NeRFMLP(...)`` geometry and every ``NeRF(min_deg_point,
max_deg_point, deg_view)`` runs on the layer-wise MFMA GEMM engine (``csrc/aon_gmlp.hip``): same C boundary, same module
interface, forward and backward.  Sample counts, ``lindisp`` and ``noise_std`` are runtime arguments of both.
"""
from __future__ import annotations

import os as _os

import torch
import torch.nn as nn
import torch.nn.init as init

from ... import ops
from ...autograd import RenderGeneral, RenderLevelVanilla, RenderVanilla


class NeRFMLP(nn.Module):
    """model.py:39-120.  ``forward(x, condition)``: x (N,S,63) encoded samples, condition (N,27) encoded view
    directions -> (raw_rgb (N,S,3), raw_density (N,S,1))."""

    def __init__(self, min_deg_point, max_deg_point, deg_view, netdepth: int = 8, netwidth: int = 256,
                 netdepth_condition: int = 1, netwidth_condition: int = 128, skip_layer: int = 4, input_ch: int = 3,
                 input_ch_view: int = 3, num_rgb_channels: int = 3, num_density_channels: int = 1):
        super().__init__()
        # validated by the C side (aon_gmlp_param_count): e.g. a skip concatenation after the LAST trunk layer is rejected, the
        # reference's own forward fails on it (density_layer is built for netwidth inputs)
        self.geometry = ops.MlpGeometry(min_deg_point, max_deg_point, deg_view, netdepth, netwidth, netdepth_condition, netwidth_condition,
                                        skip_layer, input_ch, input_ch_view, num_rgb_channels, num_density_channels)
        self.min_deg_point, self.max_deg_point, self.deg_view = min_deg_point, max_deg_point, deg_view
        self.netdepth, self.netwidth, self.skip_layer = netdepth, netwidth, skip_layer
        self.netdepth_condition, self.netwidth_condition = netdepth_condition, netwidth_condition
        self.num_rgb_channels, self.num_density_channels = num_rgb_channels, num_density_channels
        self.net_activation = nn.ReLU()
        pos_size = ((max_deg_point - min_deg_point) * 2 + 1) * input_ch
        view_pos_size = (deg_view * 2 + 1) * input_ch_view
        layers = [nn.Linear(pos_size, netwidth)]
        for idx in range(netdepth - 1):
            layers.append(nn.Linear(netwidth + pos_size if (idx % skip_layer == 0 and idx > 0) else netwidth, netwidth))
        for layer in layers:
            init.xavier_uniform_(layer.weight)
        self.pts_linears = nn.ModuleList(layers)
        views = [nn.Linear(netwidth + view_pos_size, netwidth_condition)]
        for _ in range(netdepth_condition - 1):
            layer = nn.Linear(netwidth_condition, netwidth_condition)
            init.xavier_uniform_(layer.weight)
            views.append(layer)
        self.views_linear = nn.ModuleList(views)
        self.bottleneck_layer = nn.Linear(netwidth, netwidth)
        self.density_layer = nn.Linear(netwidth, num_density_channels)
        self.rgb_layer = nn.Linear(netwidth_condition, num_rgb_channels)
        init.xavier_uniform_(self.bottleneck_layer.weight)
        init.xavier_uniform_(self.density_layer.weight)
        init.xavier_uniform_(self.rgb_layer.weight)
        self._streams = {}

    # Kernel-side weight streams.  They are rebuilt from the LIVE parameters on every call (one ~6 us HIP pack kernel,
    # 2.4 MB): no version/pointer key can go stale, so `p.data.copy_()`, `dist.broadcast(p.data)`, EMA or clipping code
    # that mutates parameters without bumping `p._version` is seen exactly as nn.Linear would see it.  `fresh=True`
    # (training) writes into a new buffer, because autograd saves the stream for backward and a later forward must not
    # overwrite what an earlier graph still needs; inference reuses one buffer per stream kind (stream-ordered).
    _PACKERS = {"fwd": "pack_vanilla_mlp", "bwd": "pack_vanilla_mlp_bwd"}

    def _pack(self, kind: str, fresh: bool, out: torch.Tensor | None = None) -> torch.Tensor:
        params = dict(self.named_parameters())
        dev = next(iter(params.values())).device
        if out is None:
            out = None if fresh else self._streams.get(kind)
        if out is not None and out.device != dev:
            out = None
        if not self.geometry.is_default:   # other degrees on the fused kernels (fits_fused_inference): zero-weight slots
            out = getattr(ops, self._PACKERS[kind])(params, out=out, degrees=(self.min_deg_point, self.max_deg_point, self.deg_view))
        else:
            out = getattr(ops, self._PACKERS[kind])(params, out=out)
        if not fresh:
            self._streams[kind] = out
        return out

    def packed(self, fresh: bool = False) -> torch.Tensor:
        """Forward weight stream of the fp32 kernels."""
        return self._pack("fwd", fresh)

    def packed_bwd(self, fresh: bool = False, out: torch.Tensor | None = None) -> torch.Tensor:
        """Transposed weight stream for the backward data chain (training only)."""
        return self._pack("bwd", fresh, out)

    _BWD_BYTES = "aon_bwd_packed_bytes"

    def new_bwd_buffer(self) -> torch.Tensor:
        return torch.empty(int(getattr(ops.lib, self._BWD_BYTES)()), dtype=torch.uint8, device=next(self.parameters()).device)

    def ordered_params(self):
        params = dict(self.named_parameters())
        return [params[name] for name in self.geometry.param_order]   # == ops.VANILLA_PARAM_ORDER for the default geometry

    def forward(self, x, condition):
        if not self.geometry.is_default:   # layer-wise engine, any geometry (inference; training goes through NeRF.forward)
            return ops.gmlp_fwd(self.geometry, dict(self.named_parameters()), x, condition)
        raw = ops.mlp_fwd_enc(self.packed(), x, condition)
        return raw[..., :3], raw[..., 3:4]


_SIDE_STREAMS: dict = {}


def pack_aside_mode() -> int:
    """AON_PACK_ASIDE: 0 (default since round 6) = every pack launch in line on the current stream; 1 = round 5: the transposed streams (and,
    round 6, the fine level's forward pack) on side streams, the transposed ones waited for AFTER the forward's launches; 2 = the same side
    streams, all of them waited for BEFORE the forward is launched.
    Why 0: measured with the workspace pool in place (ops._TRAIN_POOL), six alternating runs each on one box, the four variants are the same to
    0.02 ms per 30.4 ms step (30.401 / 30.403 / 30.404 / 30.421 ms: in line / mode 1 / mode 1 + the level-0 second stage aside / mode 2) --
    the 0.03-0.05 ms that rounds 5 and 6 first read off shorter A/Bs do not survive more repetitions -- and side streams around persistent
    launches are exactly where this code base has been burnt before (profiles/r04_backward_schedules.txt).  The modes stay for experiments."""
    import os

    v = os.environ.get("AON_PACK_ASIDE", "0")
    return int(v) if v in ("0", "1", "2") else 0


def packed_bwd_aside(mlps):
    """The levels' transposed weight streams (read by the BACKWARD only), packed on a side stream so that the two small launches per level
    (38 us each in a 31 ms step, profiles/r05_step_timeline.txt) run beside the forward's own pack kernels and launches instead of in front
    of them.  Returns (buffers, event): the buffers are allocated on the current stream; the caller makes the current stream wait for
    `event` once the forward is enqueued -- long before the backward reads them.  AON_PACK_ASIDE=0: packed in line, event None."""
    import os

    dev = next(mlps[0].parameters()).device
    if pack_aside_mode() == 0 or dev.type != "cuda":
        return [m.packed_bwd(True) for m in mlps], None
    cur = torch.cuda.current_stream(dev)
    outs = [m.new_bwd_buffer() for m in mlps]
    # one side stream per level (round 6: the two levels' fold -> pack chains beside each other; the last stream waits for the others,
    # so ONE event covers them all)
    sides = []
    for lvl in range(len(mlps)):
        side = _SIDE_STREAMS.get((dev, "bwd", lvl))
        if side is None:
            side = _SIDE_STREAMS[(dev, "bwd", lvl)] = torch.cuda.Stream(device=dev)
        sides.append(side)
    for side, m, o in zip(sides, mlps, outs):
        side.wait_stream(cur)        # the parameters as the optimiser step left them
        with torch.cuda.stream(side):
            m.packed_bwd(True, out=o)
    for side in sides[:-1]:
        sides[-1].wait_stream(side)
    side = sides[-1]
    ev = side.record_event()
    for o in outs:
        # (1) the buffers were allocated on the current stream but are written on the side stream: the caching allocator must not
        # hand their memory to current-stream work before the pack kernels are done, even if the caller drops them early (an exception
        # inside the forward); (2) the event travels WITH the buffer, so that a backward run on another stream than the forward waits too
        for sd in sides:
            o.record_stream(sd)
        o._aon_ready = ev
    return outs, ev


def run_aside(dev, key: str, fn):
    """`fn()` -> tuple of tensors, enqueued on a side stream of its own (one per device and `key`) behind everything the current stream
    holds; -> (tensors, event or None).  Round 6: the FINE level's pack / prepare launches (fold -> pack, ~48 us in a row) run beside the
    coarse level's instead of behind them; the caller makes the current stream wait for the event before the forward's C call.  The tensors
    are allocated on the side stream and used on the current one: recorded for it, so the caching allocator keeps them until both are done."""
    import os

    if pack_aside_mode() == 0 or dev.type != "cuda":
        return fn(), None
    cur = torch.cuda.current_stream(dev)
    side = _SIDE_STREAMS.get((dev, key))
    if side is None:
        side = _SIDE_STREAMS[(dev, key)] = torch.cuda.Stream(device=dev)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        outs = fn()
    ev = side.record_event()
    for o in outs:
        o.record_stream(cur)
    return outs, ev


class NeRF(nn.Module):
    """model.py:123-199.  ``forward(rays, randomized, white_bkgd, near, far)`` with ``rays`` a dict holding
    ``rays_o``, ``rays_d``, ``viewdirs`` (extra keys are ignored, as in the reference) returns
    ``[(comp_rgb, acc, depth)_coarse, (comp_rgb, acc, depth)_fine]``.

    ``t_rand`` (N,num_coarse+1) / ``u`` (N,num_fine) / ``noise`` (per level (N,S), read when ``noise_std > 0 and randomized``,
    model.py:183-184) optionally replace the reference's in-function ``torch.rand`` / ``torch.rand_like`` draws."""

    def __init__(self, num_levels: int = 2, min_deg_point: int = 0, max_deg_point: int = 10, deg_view: int = 4,
                 num_coarse_samples: int = 64, num_fine_samples: int = 128, use_viewdirs: bool = True,
                 noise_std: float = 0.0, lindisp: bool = False):
        super().__init__()
        if num_levels < 1:
            raise ValueError("num_levels must be >= 1")
        if num_levels > 2 and (min_deg_point, max_deg_point, deg_view) != (0, 10, 4):
            raise NotImplementedError("more than two levels are served for the default network only (stage-level calls, inference)")
        # sample counts, lindisp and noise_std are runtime arguments of the C calls (aon_render_opts); use_viewdirs is stored and
        # never read by the reference's forward (model.py:147-199 always encodes rays["viewdirs"])
        self._opts = ops.RenderOpts(num_coarse_samples, num_fine_samples, lindisp, noise_std)
        self.num_levels, self.min_deg_point, self.max_deg_point, self.deg_view = num_levels, min_deg_point, max_deg_point, deg_view
        self.num_coarse_samples, self.num_fine_samples = num_coarse_samples, num_fine_samples
        self.use_viewdirs, self.noise_std, self.lindisp = use_viewdirs, noise_std, lindisp
        self.rgb_activation = nn.Sigmoid()
        self.sigma_activation = nn.ReLU()
        self.coarse_mlp = NeRFMLP(min_deg_point, max_deg_point, deg_view)
        self.fine_mlp = NeRFMLP(min_deg_point, max_deg_point, deg_view)
        geom = self.coarse_mlp.geometry
        # other encoding degrees: on the fused kernels (inference AND training) when the levels fit their 63 / 27 input slots -- zero-weight
        # slots in the packed streams, encodings in the padded layout -- else on the layer-wise engine
        self._fused_inference = geom.fits_fused_inference
        self._general = not geom.is_default and not self._fused_inference
        self._fused_training = True      # (tests / measurements: False sends the training step of a padded-slot network to the layer-wise engine)
        if not geom.is_default and self._fused_inference:
            self._opts.degrees = (min_deg_point, max_deg_point, deg_view)

    def _draw_noise(self, noise, randomized, n, device):
        if not (self.noise_std > 0 and randomized):
            return None
        noise = list(noise) if noise is not None else []
        noise += [None] * (self.num_levels - len(noise))
        return [noise[lvl] if noise[lvl] is not None else torch.rand((n, self._opts.S(lvl)), device=device)
                for lvl in range(self.num_levels)]

    def _forward_many_levels(self, rays, randomized, white_bkgd, near, far, t_rand, u, noise, training=False):
        """model.py:147-199 for num_levels > 2 (every level after the first resamples from the previous level's t and weights
        with fine_mlp, :162-173), driven level by level through the stage-level C calls: fused cast + encode + MLP, compositing with
        the weights written, general-size inverse CDF.  ``u``: the first resampling's draws, or a list with one entry per
        resampling level.  ``training`` (round 4): every level is an autograd node built from the stage-level training calls
        (autograd.RenderLevelVanilla); fine_mlp's gradients add up over the levels that use it, as under the reference's autograd."""
        o, d, v = rays["rays_o"], rays["rays_d"], rays["viewdirs"]
        n = o.shape[0]
        us = list(u) if isinstance(u, (list, tuple)) else [u]
        t, _ = ops.sample_along_rays(o, d, self.num_coarse_samples, near, far, t_rand, want_coords=False, lindisp=self.lindisp)
        mlps = [self.coarse_mlp, self.fine_mlp]
        if training:
            # (limits of this unmerged path -- one launch set per level, no merged three-launch forward, no one-launch chain -- named up
            # front, not as AON_E_INVALID from inside loss.backward() after a full forward: ADVICE r4)
            s_last = self.num_coarse_samples + 1 + (self.num_levels - 1) * self.num_fine_samples
            if s_last > 512:
                raise NotImplementedError(f"training with num_levels={self.num_levels}: level {self.num_levels - 1} evaluates {s_last} samples per ray, "
                                          "the compositing backward (aon_composite_bwd) holds at most 512; inference has no such limit")
            if self.noise_std > 0 and randomized:
                raise NotImplementedError("training with num_levels > 2 AND density noise (noise_std > 0, randomized=True): the stage-level "
                                          "compositing backward takes no noise; inference with noise, and training with noise at num_levels <= 2, work")
            if n == 0:
                raise ValueError("empty ray batch in training mode")
            packs = [(m.packed(True), m.packed_bwd(True)) for m in mlps]
        else:
            packed = [m.packed() for m in mlps]
        ret, weights = [], None
        for lvl in range(self.num_levels):
            if lvl > 0:
                ul = us[lvl - 1] if lvl - 1 < len(us) else None
                if randomized and ul is None:
                    ul = torch.rand((n, self.num_fine_samples), device=o.device)
                t = ops.sample_pdf_t_n(t, weights, self.num_fine_samples, ul if randomized else None)
            k = min(lvl, 1)
            if training:
                comp, acc, depth, weights = RenderLevelVanilla.apply(o, d, v, t, bool(white_bkgd), packs[k][0], packs[k][1], *mlps[k].ordered_params())
            else:
                raw = ops.mlp_fwd(packed[k], o, d, v, t)
                nz = None
                if self.noise_std > 0 and randomized:
                    nz = noise[lvl] if (noise is not None and lvl < len(noise) and noise[lvl] is not None) else torch.rand(t.shape, device=o.device)
                comp, acc, weights, depth = ops.composite_raw(raw, t, d, white_bkgd, ops.ACT_VANILLA, True, opts=self._opts if nz is not None else None, noise=nz)
            ret.append((comp, acc, depth))
        return ret

    def forward(self, rays, randomized, white_bkgd, near, far, t_rand=None, u=None, noise=None):
        rays_o = rays["rays_o"]
        n = rays_o.shape[0]
        # the stratified / inverse-CDF draws may ride in the batch dict (keys "aon_t_rand", "aon_u": an extension, namespaced so that a user batch carrying its own "u" / "t_rand" is never misread -- the reference's forward
        # ignores extra keys, model.py:299-306 -- that makes a harness run reproducible: tests/test_hip_long_training.py)
        if t_rand is None:
            t_rand = rays.get("aon_t_rand")
        if u is None:
            u = rays.get("aon_u")
        if randomized:
            if t_rand is None:
                t_rand = torch.rand((n, self.num_coarse_samples + 1), device=rays_o.device)
            if u is None and self.num_levels == 2:
                u = torch.rand((n, self.num_fine_samples), device=rays_o.device)
        else:
            t_rand, u = None, None
        noise = self._draw_noise(noise, randomized, n, rays_o.device) if self.num_levels <= 2 else noise
        training = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if self.num_levels > 2:
            if self._general or not self.coarse_mlp.geometry.is_default:
                raise NotImplementedError("num_levels > 2 is served for the default network geometry / encoding degrees only")
            return self._forward_many_levels(rays, randomized, white_bkgd, near, far, t_rand, u, noise, training=training)
        layerwise = self._general or not self._fused_inference or (training and not self._fused_training and not self.coarse_mlp.geometry.is_default)
        if layerwise:
            geom = self.coarse_mlp.geometry
            mlps = [self.coarse_mlp, self.fine_mlp][: self.num_levels]
            if training:
                if n == 0:
                    raise ValueError("empty ray batch in training mode")
                params = [p for m in mlps for p in m.ordered_params()]
                flat = RenderGeneral.apply(rays_o, rays["rays_d"], rays["viewdirs"], float(near), float(far), bool(white_bkgd),
                                           self.num_levels, t_rand, u, geom, self._opts, noise, *params)
                return [tuple(flat[3 * i: 3 * i + 3]) for i in range(self.num_levels)]
            pd = [dict(m.named_parameters()) for m in mlps]
            outs = ops.grender_fwd(geom, pd[0], pd[1] if self.num_levels == 2 else None, rays_o, rays["rays_d"], rays["viewdirs"], near, far,
                                   white_bkgd, self.num_levels, t_rand, u, opts=self._opts, noise=noise)
            return [tuple(o) for o in outs]
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # training: fused forward that keeps the activation planes + HIP backward (autograd.RenderVanilla)
            if n == 0:
                raise ValueError("empty ray batch in training mode")
            mlps = [self.coarse_mlp, self.fine_mlp][: self.num_levels]
            bwd_ready = fine_ready = None
            degs = [(m.min_deg_point, m.max_deg_point, m.deg_view) for m in mlps]
            if len(mlps) == 2 and pack_aside_mode() == 0 and degs[0] == degs[1] and _os.environ.get("AON_PACK_STEP", "1") != "0":
                # round 6: both networks' forward and transposed streams in ONE C call -- the eight fp64 fold products as one launch in front
                # instead of four in a row with their pack kernels (aon_vanilla_pack_step; the same bytes in every buffer)
                packs = ops.vanilla_pack_step(dict(mlps[0].named_parameters()), dict(mlps[1].named_parameters()), degrees=degs[0])
            else:
                bwd, bwd_ready = packed_bwd_aside(mlps)
                fine_pk, fine_ready = run_aside(rays_o.device, "fine", lambda: (mlps[1].packed(True),)) if len(mlps) == 2 else (None, None)
                packs = [(mlps[0].packed(True), bwd[0])] + ([(fine_pk[0], bwd[1])] if len(mlps) == 2 else [])
            if fine_ready is not None:
                torch.cuda.current_stream(rays_o.device).wait_event(fine_ready)
            if bwd_ready is not None and pack_aside_mode() == 2:
                # no side-stream kernel may be in flight when the forward's persistent launches are dispatched (see pack_aside_mode)
                torch.cuda.current_stream(rays_o.device).wait_event(bwd_ready)
            params = [p for m in mlps for p in m.ordered_params()]
            try:
                flat = RenderVanilla.apply(rays_o, rays["rays_d"], rays["viewdirs"], float(near), float(far), bool(white_bkgd),
                                           self.num_levels, t_rand, u, packs, self._opts, noise, *params)
            finally:
                if bwd_ready is not None:   # (behind the forward's launches: free by then; also when the forward raised)
                    torch.cuda.current_stream(rays_o.device).wait_event(bwd_ready)
            return [tuple(flat[3 * i: 3 * i + 3]) for i in range(self.num_levels)]
        coarse = self.coarse_mlp.packed()
        fine = self.fine_mlp.packed() if self.num_levels == 2 else None
        outs = ops.render_fwd(coarse, fine, rays_o, rays["rays_d"], rays["viewdirs"], near, far,
                              white_bkgd, self.num_levels, t_rand, u, opts=self._opts, noise=noise)
        return [tuple(o) for o in outs]


# --------------------------------------------------------------------------------------------------------------------
# Harness-level equivalents of the reference's LightningModule methods, without pytorch-lightning.
# --------------------------------------------------------------------------------------------------------------------
import math
from collections import defaultdict
from types import SimpleNamespace

from . import helper
from ..interface import Harness


def _fused_adam(params) -> bool:
    """torch.optim.Adam's fused form when every parameter lives on a GPU (AON_FUSED_ADAM=0 in the environment: the foreach form, A/B)."""
    import os

    return os.environ.get("AON_FUSED_ADAM", "1") != "0" and all(p.is_cuda for p in params)


def build_adam(modules, lr: float):
    """``torch.optim.Adam(params, lr, betas=(0.9, 0.999))`` of the reference's ``configure_optimizers`` (model.py:386-389,
    model_autodecoder.py:604-606).  On a GPU (round 6): the parameters move into ONE flat arena and the optimizer is ``ArenaAdam`` -- the
    same update as ONE HIP launch, the HIP backward writes the gradients into the arena, the data-parallel mean reduces it in place
    (``aon_amd/arena.py``).  ``state_dict`` keeps torch.optim.Adam's layout.  AON_ARENA=0 in the environment (A/B) or CPU parameters:
    torch's own Adam (fused form on a GPU)."""
    import os

    params = [p for m in modules for p in m.parameters()]
    if os.environ.get("AON_ARENA", "1") != "0" and params and all(p.is_cuda and p.dtype == torch.float32 for p in params):
        from ...arena import ArenaAdam, ParamArena

        return ArenaAdam(ParamArena.for_modules(list(modules)), lr=lr, betas=(0.9, 0.999))   # (a second call reuses the arena the parameters live in)
    return torch.optim.Adam(params=params, lr=lr, betas=(0.9, 0.999), fused=_fused_adam(params))


class LitNeRF(Harness):
    """``models/vanilla_nerf/model.py:202-419`` minus Lightning: same method names, batch contracts and return
    structures for ``training_step`` (:256-282), ``render_rays`` (:295-321, fine level only, chunked by ``hparams.chunk``,
    logs val/psnr), ``render_rays_test`` (:323-348), ``validation_step`` (:353-375), ``test_step`` (:377-384),
    ``configure_optimizers`` (:386-389) and the learning-rate rule of ``optimizer_step`` (:391-419).
    Values the reference hands to ``self.log`` are collected in ``self.logged``.  near / far / white_bkgd, which the
    reference copies from its dataset in ``setup`` (:245-254), are constructor arguments here (dataset IO is out of scope)."""

    def __init__(self, hparams=None, lr_init: float = 5.0e-4, lr_final: float = 5.0e-6, lr_delay_steps: int = 2500,
                 lr_delay_mult: float = 0.01, randomized: bool = True, near: float = 2.0, far: float = 6.0, white_bkgd: bool = True,
                 model_kwargs: dict | None = None):
        super().__init__()
        self._init_harness(hparams, dict(chunk=3840, run_max_steps=100000, img_wh=(640, 480)))  # opt.py:103,112,17
        self.lr_init, self.lr_final, self.lr_delay_steps, self.lr_delay_mult = lr_init, lr_final, lr_delay_steps, lr_delay_mult
        self.randomized, self.near, self.far, self.white_bkgd = randomized, near, far, white_bkgd
        # the reference builds NeRF() (model.py:218); `model_kwargs` hands its constructor arguments through (sample counts, degrees, ...)
        self.model = NeRF(**(model_kwargs or {}))

    def training_step(self, batch, batch_idx):
        batch = {k: (v if k == "obj_idx" else v.squeeze(0)) for k, v in batch.items()}
        rendered = self.model(batch, self.randomized, self.white_bkgd, self.near, self.far)
        # model.py:271-279: loss0 + loss1 and the three logged values -- one launch forward, one backward (helper.train_loss)
        loss, stats = helper.train_loss(rendered, batch["target"])
        self.log("train/psnr1", stats[5])
        self.log("train/psnr0", stats[4])
        self.log("train/loss", stats[3])
        return loss

    @torch.no_grad()
    def render_rays(self, batch, batch_idx):
        B = batch["rays_o"].shape[0]
        ret = defaultdict(list)
        for i in range(0, B, self.hparams.chunk):
            chunk = {k: (v if k == "obj_idx" else v[i: i + self.hparams.chunk]) for k, v in batch.items()}
            out = self.model(chunk, False, self.white_bkgd, self.near, self.far)
            ret["comp_rgb"] += [out[1][0]]
            ret["acc"] += [out[1][1]]
            ret["depth"] += [out[1][2]]
        ret = {k: torch.cat(v, 0) for k, v in ret.items()}
        self.log("val/psnr", self.psnr_legacy(ret["comp_rgb"], batch["target"]).mean().item())
        return ret

    @torch.no_grad()
    def render_rays_test(self, batch, batch_idx):
        B = batch["rays_o"].shape[0]
        rgb = []
        for i in range(0, B, self.hparams.chunk):
            chunk = {k: v[i: i + self.hparams.chunk] for k, v in batch.items()}
            rgb.append(self.model(chunk, False, self.white_bkgd, self.near, self.far)[1][0])
        return {"target": batch["target"], "instance_mask": batch["instance_mask"], "rgb": torch.cat(rgb, 0)}

    def validation_step(self, batch, batch_idx):
        batch = {k: (v if k == "obj_idx" else v.squeeze(0)) for k, v in batch.items()}
        return self.render_rays(batch, batch_idx)  # the reference renders twice (:367,:375); once is the same result

    def test_step(self, batch, batch_idx):
        batch = {k: v.squeeze(0) if v.dim() > 0 and v.shape[0] == 1 else v for k, v in batch.items()}
        return self.render_rays_test(batch, batch_idx)

    def configure_optimizers(self):
        # model.py:386-389.  On a GPU the optimizer runs in its fused form -- one kernel for all 48 parameter tensors instead of the
        # foreach form's dozen multi-tensor launches: the same update rule, 1 ms of a 32 ms training step (round 5, tools/train_bench.py)
        return build_adam([self], self.lr_init)     # (round 6: ONE launch on a parameter arena; torch.optim.Adam's state layout)
