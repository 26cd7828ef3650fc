"""Drop-in articulated ``NeRFMLP`` / ``NeRF_AE_Art`` (reference ``models/vanilla_nerf/model_autodecoder.py:60-337``):
same constructor defaults, parameter names (``deformations_linear.*``, ``deformation_layer``, ``pts_linears.*``,
``views_linear.*``, ``bottleneck_layer``, ``density_layer``, ``rgb_layer`` under ``coarse_mlp`` / ``fine_mlp``) and
``forward`` signatures, running on the fused HIP kernels.  Only the reference's default geometry
(deformation_mlp=True, enc_after=True, embed_deg=False, 4x128 deformation and view branches) has kernels."""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.init as init

from ... import ops
from ...autograd import RenderArticulated


class NeRFMLP(nn.Module):
    """model_autodecoder.py:60-239.  ``forward(pos, condition, latents)``: pos (N,S,3) un-encoded sample positions,
    condition (N,27) encoded view dirs, latents {"density": (1,128), "color": (1,128), "articulation": (1,32)}
    -> (raw_rgb (N,S,3), raw_density (N,S,1))."""

    def __init__(self, min_deg_point, max_deg_point, deg_view, netdepth: int = 8, netwidth: int = 256,
                 netdepth_deformation=4, netwidth_deformation: int = 128, netdepth_condition: int = 4,
                 netwidth_condition: int = 128, shape_latent_dim=128, appearance_latent_dim=128,
                 articulation_latent_dim=32, skip_layer: int = 4, input_ch: int = 3, input_ch_view: int = 3,
                 num_rgb_channels: int = 3, num_density_channels: int = 1, deformation_mlp: bool = True,
                 enc_after: bool = True, embed_deg: bool = False):
        super().__init__()
        geometry = (min_deg_point, max_deg_point, deg_view, netdepth, netwidth, netdepth_deformation, netwidth_deformation,
                    netdepth_condition, netwidth_condition, shape_latent_dim, appearance_latent_dim, articulation_latent_dim,
                    skip_layer, input_ch, input_ch_view, num_rgb_channels, num_density_channels, deformation_mlp, enc_after,
                    embed_deg)
        if geometry[3:] != (8, 256, 4, 128, 4, 128, 128, 128, 32, 4, 3, 3, 3, 1, True, True, False):
            raise NotImplementedError(f"articulated NeRFMLP geometry {geometry} has no HIP kernel (the reference's default widths / latent sizes "
                                      "do, at any encoding degrees of up to 10 position and 4 view levels; enc_after=False and embed_deg=True "
                                      "change the network, model_autodecoder.py:95-103,181-184)")
        # round 4: other encoding degrees on the same kernels (zero-weight slots + run-time encoding scales, aon_pack_art_mlp_deg)
        self.degrees = (int(min_deg_point), int(max_deg_point), int(deg_view))
        ops.art_param_shapes(self.degrees)      # raises for more than 10 / 4 levels
        self.net_activation = nn.ReLU()
        self.enc_after, self.embed_deg, self.deformation_mlp = enc_after, embed_deg, deformation_mlp
        self.netdepth, self.netdepth_deformation, self.netdepth_condition, self.skip_layer = netdepth, 4, 4, skip_layer
        self.min_deg_point, self.max_deg_point, self.deg_view = min_deg_point, max_deg_point, deg_view
        self.num_rgb_channels, self.num_density_channels = num_rgb_channels, num_density_channels
        deform = [nn.Linear(3 + 128 + 32, 128)] + [nn.Linear(128, 128) for _ in range(3)]
        self.deformations_linear = nn.ModuleList(deform)
        self.deformation_layer = nn.Linear(128, 3)
        pos_size = ((max_deg_point - min_deg_point) * 2 + 1) * 3 + 128        # model_autodecoder.py:127-129
        view_pos_size = (deg_view * 2 + 1) * 3                                # :91
        pts = [nn.Linear(pos_size, 256)]
        for idx in range(7):
            pts.append(nn.Linear(256 + pos_size if (idx % skip_layer == 0 and idx > 0) else 256, 256))
        self.pts_linears = nn.ModuleList(pts)
        self.views_linear = nn.ModuleList([nn.Linear(256 + view_pos_size + 128, 128)] + [nn.Linear(128, 128) for _ in range(3)])
        self.bottleneck_layer = nn.Linear(256, 256)
        self.density_layer = nn.Linear(256, 1)
        self.rgb_layer = nn.Linear(128, 3)
        for m in list(self.deformations_linear) + [self.deformation_layer] + list(self.pts_linears) + \
                list(self.views_linear)[1:] + [self.bottleneck_layer, self.density_layer, self.rgb_layer]:
            init.xavier_uniform_(m.weight)  # views_linear[0] keeps the default init, like the reference (:147-151)
        self._streams = {}
        self._small = None

    # weight streams are rebuilt from the live parameters on every call (see vanilla NeRFMLP._pack: nothing can go stale)
    _PACKERS = {"fwd": "pack_art_mlp", "bwd": "pack_art_mlp_bwd"}

    def _pack(self, kind: str, fresh: bool, out: torch.Tensor | None = None) -> torch.Tensor:
        params = dict(self.named_parameters())
        dev = next(iter(params.values())).device
        if out is None:
            out = None if fresh else self._streams.get(kind)
        if out is not None and out.device != dev:
            out = None
        out = getattr(ops, self._PACKERS[kind])(params, out=out, degrees=self.degrees)
        if not fresh:
            self._streams[kind] = out
        return out

    def packed(self, fresh: bool = False) -> torch.Tensor:
        return self._pack("fwd", fresh)

    def packed_bwd(self, fresh: bool = False, out: torch.Tensor | None = None) -> torch.Tensor:
        return self._pack("bwd", fresh, out)

    def new_bwd_buffer(self) -> torch.Tensor:
        return torch.empty(int(ops.lib.aon_art_bwd_packed_bytes()), dtype=torch.uint8, device=next(self.parameters()).device)

    def ordered_params(self):
        params = dict(self.named_parameters())
        return [params[name] for name in ops.ART_PARAM_ORDER]

    def prepared(self, latents: dict) -> torch.Tensor:
        """Per-call latent-folded block (cheap: ~0.1 MFLOP); always rebuilt because latents are call arguments."""
        params = dict(self.named_parameters())
        dev = next(iter(params.values())).device
        out = self._small if (self._small is not None and self._small.device == dev) else None
        self._small = ops.art_prepare(params, latents, out=out, degrees=self.degrees)
        return self._small

    def forward(self, pos, condition, latents):
        if self.embed_deg:
            raise NotImplementedError
        dv = self.degrees[2]
        if dv != 4:   # the kernel reads the view encoding in its 27-wide slot layout [v ; sin block of 12 ; shifted block of 12]
            padded = condition.new_zeros((condition.shape[0], 27))
            padded[:, : 3 + 3 * dv] = condition[:, : 3 + 3 * dv]
            padded[:, 15: 15 + 3 * dv] = condition[:, 3 + 3 * dv:]
            condition = padded
        raw = ops.art_mlp_fwd_pos(self.packed(), self.prepared(latents), pos, condition)
        return raw[..., :3], raw[..., 3:4]


class NeRF_AE_Art(nn.Module):
    """model_autodecoder.py:242-337.  ``forward(rays, randomized, white_bkgd, near, far, latents, train=True)`` ->
    ``[(comp_rgb, acc, depth)_coarse, (comp_rgb, acc, depth)_fine]`` with rgb = sigmoid(raw)*(1+2*0.001)-0.001 and
    sigma = softplus(raw - 1)."""

    def __init__(self, num_levels: int = 2, min_deg_point: int = 0, max_deg_point: int = 10, deg_view: int = 4,
                 num_coarse_samples: int = 64, num_fine_samples: int = 128, use_viewdirs: bool = True,
                 noise_std: float = 0.0, lindisp: bool = False, rgb_padding: float = 0.001, density_bias: float = -1.0,
                 enc_after=True, embed_deg=False):
        super().__init__()
        if (enc_after, embed_deg) != (True, False) or num_levels not in (1, 2):
            raise NotImplementedError("enc_after=False / embed_deg=True change the network (model_autodecoder.py:95-103,181-184): only the "
                                      "reference's default articulated NeRFMLP has HIP kernels; num_levels must be 1 or 2")
        # sample counts, lindisp, noise_std, rgb_padding and density_bias are runtime arguments of the C calls (aon_render_opts)
        self._opts = ops.RenderOpts(num_coarse_samples, num_fine_samples, lindisp, noise_std, rgb_padding, density_bias,
                                    degrees=(min_deg_point, max_deg_point, deg_view))   # (read by the backward for the gradients' layout)
        self.use_viewdirs, self.noise_std, self.lindisp = use_viewdirs, noise_std, lindisp
        self.num_levels, self.min_deg_point, self.max_deg_point, self.deg_view = num_levels, min_deg_point, max_deg_point, deg_view
        self.num_coarse_samples, self.num_fine_samples = num_coarse_samples, num_fine_samples
        self.rgb_padding, self.density_bias, self.enc_after, self.embed_deg = rgb_padding, density_bias, enc_after, embed_deg
        self.rgb_activation = nn.Sigmoid()
        self.sigma_activation = nn.Softplus()
        self.coarse_mlp = NeRFMLP(min_deg_point, max_deg_point, deg_view)
        self.fine_mlp = NeRFMLP(min_deg_point, max_deg_point, deg_view)

    def forward(self, rays, randomized, white_bkgd, near, far, latents, train=True, t_rand=None, u=None, noise=None):
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
        if self.noise_std > 0 and randomized:   # model_autodecoder.py:318-319
            noise = list(noise) if noise is not None else []
            noise += [None] * (self.num_levels - len(noise))
            noise = [noise[lvl] if noise[lvl] is not None else torch.rand((n, self._opts.S(lvl)), device=rays_o.device)
                     for lvl in range(self.num_levels)]
        else:
            noise = None
        if torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters())
                                        or any(getattr(v, "requires_grad", False) for v in latents.values())):
            # training: HIP forward that keeps the activation planes + HIP backward (autograd.RenderArticulated);
            # the per-call block must not alias the cached inference buffer (it is saved for backward)
            mlps = [self.coarse_mlp, self.fine_mlp][: self.num_levels]
            bwd_ready = fine_ready = None
            if (len(mlps) == 2 and pack_aside_mode() == 0 and mlps[0].degrees == mlps[1].degrees
                    and os.environ.get("AON_PACK_STEP", "1") != "0"):
                # round 6: both networks' streams, per-call blocks and transposed streams in ONE C call -- the four fp64 fold products as one
                # launch in front instead of four in a row with their pack kernels (aon_art_pack_step; the same bytes in every buffer)
                packs = ops.art_pack_step(dict(mlps[0].named_parameters()), dict(mlps[1].named_parameters()), latents, degrees=mlps[0].degrees)
            else:
                bwd, bwd_ready = packed_bwd_aside(mlps)

                def level_pack(mlp):     # the per-call latent-folded block + the forward weight stream of one level (prepare | fold -> pack)
                    return ops.art_prepare(dict(mlp.named_parameters()), latents, degrees=mlp.degrees), mlp.packed(True)

                # (AON_PACK_ASIDE=1/2: the fine level's three launches on a side stream of their own, beside the coarse level's)
                fine, fine_ready = run_aside(rays_o.device, "fine", lambda: level_pack(mlps[1])) if len(mlps) == 2 else (None, None)
                small_c, pk_c = level_pack(mlps[0])
                packs = [(pk_c, small_c, bwd[0])] + ([(fine[1], fine[0], bwd[1])] if len(mlps) == 2 else [])
            if fine_ready is not None:
                torch.cuda.current_stream(rays_o.device).wait_event(fine_ready)
            if bwd_ready is not None and pack_aside_mode() == 2:
                # no side-stream kernel may be in flight when the forward's persistent launches are dispatched (see pack_aside_mode)
                torch.cuda.current_stream(rays_o.device).wait_event(bwd_ready)
            params = [p for mlp in mlps for p in mlp.ordered_params()]
            try:
                flat = RenderArticulated.apply(rays_o, rays["rays_d"], rays["viewdirs"], float(near), float(far), bool(white_bkgd),
                                               self.num_levels, t_rand, u, packs, self._opts, noise, latents["density"], latents["color"],
                                               latents["articulation"], *params)
            finally:
                if bwd_ready is not None:   # (behind the forward's launches: free by then; also when the forward raised)
                    torch.cuda.current_stream(rays_o.device).wait_event(bwd_ready)
            return [tuple(flat[3 * i: 3 * i + 3]) for i in range(self.num_levels)]
        two = self.num_levels == 2
        pc = self.coarse_mlp.packed()
        pf = self.fine_mlp.packed() if two else None
        outs = ops.art_render_fwd(pc, self.coarse_mlp.prepared(latents), pf, self.fine_mlp.prepared(latents) if two else None,
                                  rays_o, rays["rays_d"], rays["viewdirs"], near, far, white_bkgd, self.num_levels, t_rand, u,
                                  opts=self._opts, noise=noise)
        return [tuple(o) for o in outs]


# --------------------------------------------------------------------------------------------------------------------
from collections import defaultdict  # noqa: E402

from . import helper  # noqa: E402
from ..code_library import CodeLibraryArticulated  # noqa: E402
from ..interface import Harness  # noqa: E402
from .model import build_adam, pack_aside_mode, packed_bwd_aside, run_aside  # noqa: E402

_SCALAR_KEYS = ("deg", "instance_id", "articulation_id")


class LitNeRF_AutoDecoder(Harness):
    """``model_autodecoder.py:340-701`` minus Lightning: ``NeRF_AE_Art`` + ``CodeLibraryArticulated`` with the
    reference's ``training_step`` (:393-477: mse(coarse)+mse(fine) + 1e-4 * latent-norm regulariser), ``render_rays``
    (:479-513, fine level, chunked, logs val/psnr and the object-pixel PSNR), ``render_rays_test`` (:515-543),
    ``validation_step`` (:548-586, wandb image grid dropped), ``test_step`` (:588-605, test-time interpolated
    articulation codes), ``configure_optimizers`` (:607-609: one Adam over model + code library) and the
    learning-rate rule (:611-640).  near / far / white_bkgd come from the dataset in the reference's ``setup``
    (:359-391); ``setup(dataset)`` copies them the same way."""

    def __init__(self, hparams=None, lr_init: float = 5.0e-4, lr_final: float = 5.0e-6, lr_delay_steps: int = 2500,
                 lr_delay_mult: float = 0.01, randomized: bool = True, near: float = 2.0, far: float = 6.0, white_bkgd: bool = True,
                 model_kwargs: dict | None = None):
        super().__init__()
        self._init_harness(hparams, dict(chunk=3840, run_max_steps=100000, img_wh=(320, 240), N_max_objs=1, N_obj_code_length=128))
        self.lr_init, self.lr_final, self.lr_delay_steps, self.lr_delay_mult = lr_init, lr_final, lr_delay_steps, lr_delay_mult
        self.randomized, self.near, self.far, self.white_bkgd = randomized, near, far, white_bkgd
        self.model = NeRF_AE_Art(**(model_kwargs or {}))   # the reference builds NeRF_AE_Art() (model_autodecoder.py:352)
        self.code_library = CodeLibraryArticulated(self.hparams)

    def setup(self, dataset):
        self.near, self.far, self.white_bkgd = dataset.near, dataset.far, dataset.white_back

    @staticmethod
    def _unbatch(batch):
        return {k: (v if k in _SCALAR_KEYS else v.squeeze(0)) for k, v in batch.items()}

    def training_step(self, batch, batch_idx):
        batch = self._unbatch(batch)
        latents = self.code_library(batch)
        rendered = self.model(batch, self.randomized, self.white_bkgd, self.near, self.far, latents)
        # model_autodecoder.py:455-477: loss1 + loss0 + 1e-4 * (mean ||shape|| + mean ||appearance|| + mean ||articulation||) and the four
        # logged values -- one launch forward, one backward (helper.train_loss) where torch runs ~47
        loss, stats = helper.train_loss(rendered, batch["target"], (latents["density"], latents["color"], latents["articulation"]), 1e-4)
        self.log("train/psnr1", stats[5])
        self.log("train/psnr0", stats[4])
        self.log("train/loss", stats[3])
        self.log("train/loss/reg", stats[2])
        return loss

    def _render_chunks(self, batch, latents, skip=()):
        B = batch["rays_o"].shape[0]
        ret = defaultdict(list)
        for i in range(0, B, self.hparams.chunk):
            chunk = {k: v[i: i + self.hparams.chunk] for k, v in batch.items() if k not in skip and k not in _SCALAR_KEYS}
            out = self.model(chunk, False, self.white_bkgd, self.near, self.far, latents)
            ret["comp_rgb"] += [out[1][0]]
            ret["acc"] += [out[1][1]]
            ret["depth"] += [out[1][2]]
        return {k: torch.cat(v, 0) for k, v in ret.items()}

    @torch.no_grad()
    def render_rays(self, batch, latents):
        ret = self._render_chunks(batch, latents, skip=("img_wh", "src_imgs"))
        self.log("val/psnr", self.psnr_legacy(ret["comp_rgb"], batch["target"]).mean().item())
        mask = batch["instance_mask"].view(-1, 1).expand(-1, 3)
        self.log("val/psnr_obj", self.psnr_legacy(ret["comp_rgb"][mask], batch["target"][mask]).mean().item())
        return ret

    @torch.no_grad()
    def render_rays_test(self, batch, latents):
        ret = self._render_chunks(batch, latents, skip=("img_wh", "src_imgs"))
        return {"target": batch["target"], "instance_mask": batch["instance_mask"], "rgb": ret["comp_rgb"]}

    @torch.no_grad()
    def validation_step(self, batch, batch_idx):
        batch = self._unbatch(batch)
        return self.render_rays(batch, self.code_library(batch))

    @torch.no_grad()
    def test_step(self, batch, batch_idx):
        batch = self._unbatch(batch)
        return self.render_rays_test(batch, self.code_library(batch, is_test=True))

    def configure_optimizers(self):
        return build_adam([self.model, self.code_library], self.lr_init)   # (model_autodecoder.py:604-606; one arena, one launch: LitNeRF)
