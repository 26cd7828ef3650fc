"""Checkpoint IO in the reference's Lightning key layout and result writers (SURVEY 5 "Checkpoint / resume",
8(f) ranks 3-4): a Lightning ``.ckpt`` is a ``torch.save`` dict whose ``state_dict`` keys are
``model.coarse_mlp.pts_linears.0.weight`` ... (+ ``code_library.embedding_instance_*.weight`` for the autodecoder)."""
from __future__ import annotations

import json
import os

import numpy as np
import torch
from PIL import Image


def save_checkpoint(path, lit_module, optimizer=None, epoch: int = 0):
    """Write ``{epoch, global_step, state_dict[, optimizer_states]}`` -- the subset of a Lightning 1.5 checkpoint that
    ``Trainer(resume_from_checkpoint=...)`` / ``load_from_checkpoint`` read for the weights.  A data-parallel run first settles the
    deferred check of its last gradient exchange (parallel.check_gradient_exchange): replicas that diverged are not checkpointed."""
    from ..parallel import check_gradient_exchange

    check_gradient_exchange()
    ckpt = {"epoch": epoch, "global_step": int(getattr(lit_module, "global_step", 0)),
            "pytorch-lightning_version": "1.5.2",
            "state_dict": {k: v.detach().cpu() for k, v in lit_module.state_dict().items()}}
    if optimizer is not None:
        ckpt["optimizer_states"] = [optimizer.state_dict()]
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(ckpt, path)
    return path


def load_checkpoint(path, lit_module, optimizer=None, strict: bool = True):
    """Load a reference (Lightning) checkpoint or one written by save_checkpoint into the drop-in module."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    lit_module.load_state_dict(ckpt["state_dict"], strict=strict)
    if hasattr(lit_module, "global_step"):
        lit_module.global_step = int(ckpt.get("global_step", 0))
    if optimizer is not None and ckpt.get("optimizer_states"):
        optimizer.load_state_dict(ckpt["optimizer_states"][0])
    return ckpt


def extract_model_state_dict(ckpt_path, model_name="model", prefixes_to_ignore=()):
    """utils/__init__.py:117-132 of the reference: strip the ``model.`` prefix of a Lightning state_dict."""
    ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=False)
    sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
    out = {}
    for k, v in sd.items():
        if not k.startswith(model_name + "."):
            continue
        k = k[len(model_name) + 1:]
        if any(k.startswith(p) for p in prefixes_to_ignore):
            continue
        out[k] = v
    return out


def store_image(dirpath, rgbs, name="image"):
    """models/utils.py:21-27: one JPEG per (h,w,3) float image, clipped to [0,1]."""
    os.makedirs(dirpath, exist_ok=True)
    paths = []
    for i, rgb in enumerate(rgbs):
        arr = (np.clip(rgb.detach().cpu().numpy(), 0, 1) * 255).astype(np.uint8)
        p = os.path.join(dirpath, f"{name}{str(i).zfill(3)}.jpg")
        Image.fromarray(arr).save(p)
        paths.append(p)
    return paths


def write_stats(fpath, *stats):
    """models/utils.py:62-73: merge metric dicts ({'name': ..., 'mean'/'test': ...}) into one results.json."""
    d = {}
    for stat in stats:
        d[stat["name"]] = {k: float(w) for k, w in stat.items() if k != "name" and k != "scene_wise"}
    os.makedirs(os.path.dirname(os.path.abspath(fpath)), exist_ok=True)
    with open(fpath, "w") as fp:
        json.dump(d, fp, indent=4, sort_keys=True)
    return d
