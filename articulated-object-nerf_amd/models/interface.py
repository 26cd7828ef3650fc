"""Mirror of ``models/interface.py:LitModel``'s metric / gather helpers (interface.py:31-74) without the
pytorch-lightning base class (Lightning is the reference's harness, not part of the rendered path)."""
from __future__ import annotations

import math
import os

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

    @torch.no_grad()
    def psnr(self, preds, gts, i_train=None, i_val=None, i_test=None):
        """interface.py:127-140: {"name": "PSNR", "mean": m, "test": m} over per-image PSNRs."""
        m = self.psnr_each(preds, gts).mean().item()
        return {"name": "PSNR", "mean": m, "test": m}


def get_obj_rgbs_from_segmap(all_segmap, all_pred_img, all_pred_target):
    """models/utils.py:102-109: per image, the predicted / target colours of the pixels inside the instance mask."""
    objs, tgts = [], []
    for seg, pred, target in zip(all_segmap, all_pred_img, all_pred_target):
        mask = seg.bool().unsqueeze(-1).expand(-1, -1, 3)
        objs.append(pred[mask])
        tgts.append(target[mask])
    return objs, tgts


class _LogSeries(list):
    """The values one name was logged with, as floats.  ``float(tensor)`` at log time is a host synchronisation, four per training step
    (the host then enqueues the backward only after the forward has finished), so device scalars wait in a SIDE buffer and are turned into
    floats when anybody looks at the list -- any read entry point, pickling and copying included (round 5 kept the pending
    tensors inside the list itself, and ``copy()``, ``+``, ``==``, ``repr``, ``np.asarray`` handed raw device tensors out) -- or 512 at a
    time with ONE device read.  The list proper only ever holds floats."""

    def __init__(self, *a):
        super().__init__(*a)
        self._wait = []          # device scalars logged since the last read, in order (they follow everything already in the list)

    def append(self, value):
        if isinstance(value, torch.Tensor):
            self._wait.append(value.detach())
            if len(self._wait) >= 512:
                self._settle()
        else:
            self._settle()
            list.append(self, float(value))

    def _settle(self):
        wait, self._wait = getattr(self, "_wait", []), []
        if wait:
            ts = [t.reshape(()).float() for t in wait]
            vals = torch.stack(ts).tolist() if len({t.device for t in ts}) == 1 else [float(t) for t in ts]
            list.extend(self, vals)

    def __reduce_ex__(self, protocol):          # pickle / copy.copy / copy.deepcopy: a plain list of floats
        self._settle()
        return (list, (list(self),))


def _settled(name):
    base = getattr(list, name)

    def method(self, *a, **k):
        self._settle()
        return base(self, *a, **k)

    method.__name__ = name
    return method


for _name in ("__getitem__", "__iter__", "__len__", "__repr__", "__eq__", "__ne__", "__lt__", "__le__", "__gt__", "__ge__", "__contains__", "__add__", "__mul__",
              "__rmul__", "__reversed__", "__setitem__", "__delitem__", "__iadd__", "__imul__", "copy", "count", "index", "extend", "insert", "pop", "remove",
              "reverse", "sort", "clear"):
    setattr(_LogSeries, _name, _settled(_name))


class Harness(LitModel):
    """What the two LightningModules of the reference share once Lightning is removed: the ``self.log`` sink, the
    learning-rate rule of ``optimizer_step`` (model.py:391-419 == model_autodecoder.py:607-636) and the PSNR half of
    ``test_epoch_end`` (model.py:450-485, model_autodecoder.py:665-701; SSIM / LPIPS need third-party networks and are
    out of scope)."""

    lr_init, lr_final, lr_delay_steps, lr_delay_mult = 5.0e-4, 5.0e-6, 2500, 0.01

    def _init_harness(self, hparams, defaults):
        from collections import defaultdict
        from types import SimpleNamespace

        hp = dict(defaults)
        hp.update(vars(hparams) if hparams is not None and not isinstance(hparams, dict) else (hparams or {}))
        self.hparams = SimpleNamespace(**hp)
        self.logged = defaultdict(_LogSeries)
        self.global_step = 0

    def log(self, name, value, **_):
        self.logged[name].append(value)

    def lr_at_step(self, step: int) -> float:
        """log-linear decay lr_init -> lr_final over run_max_steps, times a sine warm-up over lr_delay_steps."""
        if self.lr_delay_steps > 0:
            delay = self.lr_delay_mult + (1 - self.lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / self.lr_delay_steps, 0), 1))
        else:
            delay = 1.0
        t = min(max(step / self.hparams.run_max_steps, 0), 1)
        return delay * math.exp(math.log(self.lr_init) * (1 - t) + math.log(self.lr_final) * t)

    def optimizer_step(self, optimizer, closure=None):
        for pg in optimizer.param_groups:
            pg["lr"] = self.lr_at_step(self.global_step)
        optimizer.step(closure=closure)
        self.global_step += 1

    def fit_step(self, batch, batch_idx, optimizer, find_unused_parameters: bool = False):
        """One batch of Lightning's automatic optimisation under its DDP plugin (run.py:144-153): zero_grad, training_step,
        backward, the data-parallel gradient mean (ONE flat RCCL bucket, `parallel.allreduce_gradients`; a no-op without a process
        group or at world size 1), then the LR rule + optimizer step.  `find_unused_parameters` is DDPPlugin's argument (run.py:109,
        :129, :151): False -- the reference's value -- is DDP's strict contract (every rank produces gradients for the same parameters;
        a violation raises UnevenGradientsError on EVERY rank at the next exchange, or at `finish_fit()` / a checkpoint / test_epoch_end
        when there is no next one); True lets a rank whose batch did not touch a parameter adopt the others' mean."""
        from ..parallel import allreduce_gradients

        optimizer.zero_grad(set_to_none=True)
        loss = self.training_step(batch, batch_idx)
        loss.backward()
        allreduce_gradients(self, find_unused_parameters=find_unused_parameters)
        self.optimizer_step(optimizer)
        return loss.detach()

    def finish_fit(self) -> None:
        """End of a training loop (Lightning's on_train_end): the deferred check of the LAST gradient exchange, which no later
        fit_step will look at.  Also run by utils.save_checkpoint and test_epoch_end."""
        from ..parallel import check_gradient_exchange

        check_gradient_exchange()

    @torch.no_grad()
    def test_epoch_end(self, outputs, image_sizes, out_dir=None, name="image"):
        """Gather the per-image test outputs over ranks, PSNR over whole images and over object pixels, and (rank 0,
        when ``out_dir`` is given) the JPEG dump + results.json of the reference."""
        from ..utils import store_image, write_stats

        self.finish_fit()
        rgbs = self.alter_gather_cat(outputs, "rgb", image_sizes)
        masks = self.alter_gather_cat(outputs, "instance_mask", image_sizes)
        targets = self.alter_gather_cat(outputs, "target", image_sizes)
        psnr = self.psnr(rgbs, targets)
        objs, obj_targets = get_obj_rgbs_from_segmap(masks, rgbs, targets)
        psnr_obj = self.psnr(objs, obj_targets)
        psnr_obj["name"] = "PSNR_obj"
        self.log("test/psnr", psnr["test"])
        self.log("test/psnr_obj", psnr_obj["test"])
        if out_dir is not None and (not dist.is_initialized() or dist.get_rank() == 0):
            store_image(out_dir, rgbs, name)
            write_stats(os.path.join(out_dir, "results.json"), psnr, psnr_obj)
        return psnr, psnr_obj
