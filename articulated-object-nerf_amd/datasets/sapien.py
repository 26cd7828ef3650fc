"""GPU-side equivalent of the reference's single-scene dataset ``datasets/sapien.py:11-157``:
same on-disk format (``{split}/transforms.json`` = {"focal": fy | "camera_angle_x": a, "frames": {"r_i": 4x4 c2w}},
``{split}/rgb/r_i.png`` RGBA; datagen/data_utils.py:189-243), same per-item dict contract, but the rays
are generated on the GPU by ``aon_raygen`` per pose instead of being pre-computed on the CPU for every image (:83-113).

  * alpha blend onto white: img[:, :3]*a + (1-a)  (sapien.py:99,141), done on the device;
  * near = 2.0, far = 6.0 (:72-73); focal rule of :62-68;
  * ``rays_d`` and ``viewdirs`` are the same unit vectors (the reference's in-place normalisation aliasing, R0).
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch
from PIL import Image

from .ray_utils import get_frame_rays


def _frame_index(name: str) -> int:
    return int(name.split("_")[1].split(".")[0])


class SapienDataset(torch.utils.data.Dataset):
    def __init__(self, root_dir, split="train", img_wh=(320, 240), model_type=None, white_back=None, eval_inference=None,
                 device="cuda"):
        self.root_dir, self.split, self.img_wh, self.white_back = root_dir, split, tuple(img_wh), white_back
        self.device = torch.device(device)
        sub = {"train": "train", "val": "val"}.get(split, "test")   # the reference maps every other split to 'test'
        self.base_dir = os.path.join(root_dir, sub)
        with open(os.path.join(self.base_dir, "transforms.json")) as f:
            self.meta = json.load(f)
        files = [f for f in os.listdir(os.path.join(self.base_dir, "rgb")) if f.lower().endswith(".png")]
        self.img_files = sorted(files, key=_frame_index)
        w, h = self.img_wh
        if self.meta.get("camera_angle_x", False):
            self.focal = 0.5 * h / np.tan(0.5 * self.meta["camera_angle_x"]) * (w / 320)   # sapien.py:63-65
        else:
            self.focal = self.meta.get("focal", None)
            if self.focal is None:
                raise ValueError("focal length not found in transforms.json")
        self.near, self.far = 2.0, 6.0
        self.bounds = np.array([self.near, self.far])
        n_img = len(self.img_files) if eval_inference is not None else 1
        self.image_sizes = np.array([[h, w] for _ in range(n_img)])
        if split == "train":   # buffers of all rays / colours, like the reference, but resident on the GPU
            o, d, rgb = [], [], []
            for f in self.img_files:
                ro, vd = self.rays_of(f)
                img, _ = self.image_of(f)
                o.append(ro); d.append(vd); rgb.append(img)
            self.all_rays_o, self.all_rays_d, self.all_rgbs = torch.cat(o), torch.cat(d), torch.cat(rgb)

    # ---- pieces -------------------------------------------------------------------------------------------------
    def pose_of(self, img_file: str) -> torch.Tensor:
        return torch.tensor(self.meta["frames"][img_file.split(".")[0]], dtype=torch.float32)[:3, :4]

    def rays_of(self, img_file: str):
        w, h = self.img_wh
        return get_frame_rays(h, w, self.focal, self.pose_of(img_file), device=self.device)

    def image_of(self, img_file: str):
        """-> (rgb (h*w,3) blended onto white, valid_mask (h*w,) = alpha > 0), on the dataset's device."""
        img = Image.open(os.path.join(self.base_dir, "rgb", img_file))
        if img.size != self.img_wh:
            img = img.resize(self.img_wh, Image.LANCZOS)
        a = torch.from_numpy(np.asarray(img.convert("RGBA"), dtype=np.uint8).copy()).to(self.device)
        a = a.reshape(-1, 4).to(torch.float32) / 255.0      # torchvision ToTensor semantics
        return a[:, :3] * a[:, 3:] + (1 - a[:, 3:]), a[:, 3] > 0

    # ---- Dataset protocol ---------------------------------------------------------------------------------------
    def __len__(self):
        if self.split == "train":
            return self.all_rays_o.shape[0]
        return 1 if self.split == "val" else len(self.img_files)

    def __getitem__(self, idx):
        if self.split == "train":
            return {"rays_o": self.all_rays_o[idx], "rays_d": self.all_rays_d[idx], "viewdirs": self.all_rays_d[idx],
                    "target": self.all_rgbs[idx]}
        f = self.img_files[idx]
        ro, vd = self.rays_of(f)
        img, mask = self.image_of(f)
        return {"rays_o": ro, "rays_d": vd, "viewdirs": vd, "instance_mask": mask, "target": img}

    def train_batches(self, batch_size=2048, generator=None, drop_last=False):
        """Shuffled ray batches of the reference's hard-coded size (model.py:421-428), sampled on the device.  The
        reference's DataLoader does not set drop_last, so the shorter tail batch of an epoch is yielded too."""
        n = len(self)
        perm = torch.randperm(n, device=self.device, generator=generator)
        for i in range(0, n - batch_size + 1 if drop_last else n, batch_size):
            idx = perm[i: i + batch_size]
            yield {"rays_o": self.all_rays_o[idx], "rays_d": self.all_rays_d[idx], "viewdirs": self.all_rays_d[idx],
                   "target": self.all_rgbs[idx]}


def write_synthetic_scene(root_dir, n_train=3, n_val=1, img_wh=(32, 24), seed=0, use_camera_angle=False):
    """Write a tiny dataset in the reference's on-disk format (datagen/data_utils.py:189-243) for tests and demos:
    RGBA PNGs of a soft disc with transparent background, look-at-origin poses on the radius-4 sphere."""
    import math

    from .. import synthetic as syn

    w, h = img_wh
    rng = np.random.Generator(np.random.PCG64(seed))
    for split, n in (("train", n_train), ("val", n_val), ("test", n_val)):
        os.makedirs(os.path.join(root_dir, split, "rgb"), exist_ok=True)
        frames = {}
        for i in range(n):
            c2w = syn.look_at_pose(4.0, 360.0 * i / max(n, 1) + 10.0 * (split != "train"), 30.0)
            frames[f"r_{i}"] = torch.cat([c2w, torch.tensor([[0.0, 0.0, 0.0, 1.0]])]).tolist()
            yy, xx = np.mgrid[0:h, 0:w]
            r = np.hypot(xx - w / 2, yy - h / 2) / (0.35 * h)
            alpha = np.clip(1.5 - 1.5 * r, 0, 1)
            rgb = np.stack([0.5 + 0.5 * np.sin(xx / 3.0 + i), 0.5 + 0.5 * np.cos(yy / 4.0), np.full_like(r, 0.3 + 0.1 * i)], -1)
            rgba = np.concatenate([rgb, alpha[..., None]], -1) + rng.uniform(0, 1e-3, (h, w, 4))
            Image.fromarray((np.clip(rgba, 0, 1) * 255).astype(np.uint8), "RGBA").save(os.path.join(root_dir, split, "rgb", f"r_{i}.png"))
        meta = {"frames": frames}
        if use_camera_angle:
            meta["camera_angle_x"] = 2 * math.atan(0.5 * h / (syn.focal_from_fovy(h) * 320 / w))  # inverts sapien.py:63-65
        else:
            meta["focal"] = syn.focal_from_fovy(h)
        with open(os.path.join(root_dir, split, "transforms.json"), "w") as f:
            json.dump(meta, f)
    return root_dir
