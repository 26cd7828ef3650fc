"""Mirror of ``models/interface.py:LitModel``'s metric / gather helpers (interface.py:31-74) without the
pytorch-lightning base class (Lightning is the reference's harness, not part of the rendered path)."""
from __future__ import annotations

import math

import torch
import torch.distributed as dist


class LitModel(torch.nn.Module):
    def mse(self, image_pred, image_gt, valid_mask=None, reduction="mean"):
        """interface.py:64-70"""
        value = (image_pred - image_gt) ** 2
        if valid_mask is not None:
            value = value[valid_mask]
        return torch.mean(value) if reduction == "mean" else value

    @torch.no_grad()
    def psnr_legacy(self, image_pred, image_gt, valid_mask=None, reduction="mean"):
        """interface.py:72-74"""
        return -10 * torch.log10(self.mse(image_pred, image_gt, valid_mask, reduction))

    @torch.no_grad()
    def psnr_each(self, preds, gts):
        """interface.py:54-62"""
        out = []
        for pred, gt in zip(preds, gts):
            mse = torch.mean((torch.clip(pred, 0, 1) - torch.clip(gt, 0, 1)) ** 2)
            out.append(-10.0 * torch.log(mse) / math.log(10.0))
        return torch.stack(out)

    def all_gather(self, t: torch.Tensor) -> torch.Tensor:
        """(world, *t.shape) like LightningModule.all_gather; identity without a process group."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return t
        out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(out, t.contiguous())
        return torch.stack(out)

    def alter_gather_cat(self, outputs, key, image_sizes):
        """interface.py:31-51 with the world>1 layout fixed: ranks are concatenated rank-major (each rank's
        images stay contiguous) instead of the reference's per-pixel interleave."""
        each = torch.cat([output[key] for output in outputs])
        allv = self.all_gather(each).detach()
        if allv.dim() == each.dim() + 1:
            allv = allv.flatten(0, 1)
        if allv.shape[-1] == 1:
            allv = allv.squeeze(-1)
        ret, curr = [], 0
        for (h, w) in image_sizes:
            chunk = allv[curr: curr + h * w]
            if chunk.shape[0] == 0:
                continue
            ret.append(chunk.reshape(h, w, 3) if allv.dim() == 2 and allv.shape[-1] == 3 else chunk.reshape(h, w))
            curr += h * w
        return ret
