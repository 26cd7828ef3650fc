"""GPU-side equivalent of the reference's articulated multi-instance dataset ``datasets/sapien_multi.py:11-479``
(SURVEY 8(a) R0, 8(f) rank 1-2).  On-disk tree (``sapien_multi.py:250-306``):

    root/{instance}/train/{deg}_degree/transforms.json   {"camera_angle_x": a, "frames": {"r_i": 4x4 c2w}}
    root/{instance}/train/{deg}_degree/rgb/r_i.png       RGB(A) render
    root/{instance}/train/{deg}_degree/seg/r_i.png       part/instance id map, > 0 = object

Same per-item dict as the reference (keys, shapes, dtypes), same random sources in the same order (``random.randint``
for instance and degree, ``np.random.randint(0, 59)`` for the view, ``torch.randint`` on the CPU generator for the
4096 pixel indices) so a seeded run picks the same rays; the work moves to the device: the frame's rays come from
``aon_raygen`` for the one pose that was drawn (the reference rebuilds directions + rays on the CPU for every item,
:281-304), masking / white background / gathers are device tensor ops on the uploaded uint8 image.

  * focal = 0.5 h / tan(0.5 camera_angle_x) * (w / 320)          (:277-280)
  * background outside the mask is 255 (white_back) or 0         (:186-197)
  * near = 2.0, far = 6.0                                        (:141-142)
  * train: 4096 rays of one random view; val: one whole random view; test_val: 19 views on the radius-4 spheric
    path over the 0-degree renders, ``articulation_id = idx`` (:395-479)
"""
from __future__ import annotations

import json
import math
import os
import random

import numpy as np
import torch
from PIL import Image

from .ray_utils import get_frame_rays

idx_to_deg = {
    "train": {i: 10 * i for i in range(10)},            # sapien_multi.py:11-14
    "val": {i: 10 * i + 5 for i in range(9)},
}


def _view_index(name: str) -> int:
    return int(name.split("_")[1].split(".")[0])


def _degree_index(name: str) -> int:
    return int(name.split("_")[0])


def create_spheric_poses(radius: float = 4.0) -> torch.Tensor:
    """sapien_multi.py:29-72 -> (40,4,4) float32: for theta in linspace(-180,180,41)[:-1] and phi = -30 deg,
    ``swap @ (R_theta @ (R_phi @ T_radius))`` with float32 factors multiplied in that order (bit-identical to the
    reference's poses; pinned by tests/golden/g14)."""
    f32 = lambda rows: torch.tensor(rows, dtype=torch.float64).to(torch.float32)
    swap = f32([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]])
    phi = -30.0 / 180.0 * np.pi
    cp, sp = np.cos(phi), np.sin(phi)
    poses = []
    for angle in np.linspace(-180, 180, 41)[:-1]:
        th = angle / 180.0 * np.pi
        ct, st = np.cos(th), np.sin(th)
        c2w = f32([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]])
        c2w = f32([[1, 0, 0, 0], [0, cp, -sp, 0], [0, sp, cp, 0], [0, 0, 0, 1]]) @ c2w
        c2w = f32([[ct, 0, -st, 0], [0, 1, 0, 0], [st, 0, ct, 0], [0, 0, 0, 1]]) @ c2w
        poses.append(swap @ c2w)
    return torch.stack(poses, 0)


class SapienDatasetMulti(torch.utils.data.Dataset):
    def __init__(self, root_dir, split="train", img_wh=(320, 240), model_type=None, white_back=None, eval_inference=None,
                 device="cuda", ray_batch_size: int = 4096):
        self.root_dir, self.split, self.img_wh, self.white_back = root_dir, split, tuple(img_wh), white_back
        self.device = torch.device(device)
        self.ids = np.sort([f.name for f in os.scandir(root_dir) if f.is_dir()])
        self.samples_per_epoch = 4000                                  # :137
        self.ray_batch_size = ray_batch_size                           # :381 (hard-coded 4096 in the reference)
        self.near, self.far = 2.0, 6.0
        w, h = self.img_wh
        if eval_inference is not None:
            self.image_sizes = np.array([[h, w] for _ in range(19)])   # :150-153
            self.poses_test = create_spheric_poses(radius=4.0)
        else:
            self.image_sizes = np.array([[h, w]])

    # ---- pieces -------------------------------------------------------------------------------------------------
    def degree_dirs(self, instance_id: str):
        names = [f.name for f in os.scandir(os.path.join(self.root_dir, instance_id, "train")) if f.is_dir()]
        return sorted(names, key=_degree_index)

    def view_dir(self, instance_id: str, degree_id: str) -> str:
        return os.path.join(self.root_dir, instance_id, "train", degree_id)   # every split reads the train tree (:255-275)

    def view_files(self, base_dir: str):
        return sorted(os.listdir(os.path.join(base_dir, "rgb")), key=_view_index)

    def focal_of(self, meta: dict) -> float:
        w, h = self.img_wh
        return 0.5 * h / np.tan(0.5 * meta["camera_angle_x"]) * (w / 320)

    def load_image_and_seg(self, img_path: str, seg_path: str):
        """-> (rgb (h*w,3) float in [0,1] with the background painted, mask (h*w,1) bool), on the device (:157-198, 209-213)."""
        w, h = self.img_wh
        img = Image.open(img_path).convert("RGB").resize((w, h), Image.LANCZOS)
        seg = np.array(Image.open(seg_path).resize((w, h), Image.LANCZOS)) > 0
        if seg.ndim == 3:
            seg = seg.any(-1)
        rgb = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).to(self.device).reshape(-1, 3)
        mask = torch.from_numpy(seg.reshape(-1, 1)).to(self.device)
        bg = 255 if self.white_back else 0
        rgb = torch.where(mask, rgb, torch.full_like(rgb, bg))
        return rgb.to(torch.float32) / 255.0, mask

    def read_data(self, instance_id: str, degree_id: str, image_id: int, c2w=None):
        """One view -> (rays_o, viewdirs, rgb, mask, src_img), all (h*w, .) on the device (:250-306)."""
        base = self.view_dir(instance_id, degree_id)
        with open(os.path.join(base, "transforms.json")) as f:
            meta = json.load(f)
        files = self.view_files(base) if self.split != "train" else os.listdir(os.path.join(base, "rgb"))
        img_file = files[image_id]
        if c2w is None:
            c2w = torch.tensor(meta["frames"][img_file.split(".")[0]], dtype=torch.float32)
        w, h = self.img_wh
        rays_o, viewdirs = get_frame_rays(h, w, self.focal_of(meta), c2w[:3, :4], device=self.device)
        rgb, mask = self.load_image_and_seg(os.path.join(base, "rgb", img_file), os.path.join(base, "seg", img_file))
        return rays_o, viewdirs, rgb, mask

    def _assemble(self, rays_o, viewdirs, rgb, mask, pix_inds=None):
        w, h = self.img_wh
        src = ((rgb.reshape(h, w, 3).permute(2, 0, 1) - 0.5) / 0.5).contiguous()      # T.Normalize(0.5, 0.5), :144
        if pix_inds is not None:
            rays_o, viewdirs, rgb, mask = rays_o[pix_inds], viewdirs[pix_inds], rgb[pix_inds], mask[pix_inds]
        return {"rays_o": rays_o, "rays_d": viewdirs, "viewdirs": viewdirs, "src_imgs": src, "target": rgb, "instance_mask": mask}

    # ---- Dataset protocol ---------------------------------------------------------------------------------------
    def __len__(self):
        return self.samples_per_epoch if self.split == "train" else (1 if self.split == "val" else 19)

    def __getitem__(self, idx):
        w, h = self.img_wh
        if self.split in ("train", "val"):
            inst = random.randint(0, len(self.ids) - 1)
            degs = self.degree_dirs(self.ids[inst])
            deg_idx = random.randint(0, len(degs) - 1)
            image_id = int(np.random.randint(0, 59))
            rays_o, viewdirs, rgb, mask = self.read_data(self.ids[inst], degs[deg_idx], image_id)
            pix = None
            if self.split == "train":
                pix = torch.randint(0, h * w, (self.ray_batch_size,)).to(self.device)     # CPU generator, like :235
            sample = self._assemble(rays_o, viewdirs, rgb, mask, pix)
            sample["deg"] = np.deg2rad(idx_to_deg["train"][deg_idx]).astype(np.float32)
            if self.split == "val":
                sample["img_wh"] = np.array((w, h))
        else:
            inst = random.randint(0, len(self.ids) - 1)
            deg_idx = idx                                                   # :451-457
            rays_o, viewdirs, rgb, mask = self.read_data(self.ids[inst], "0_degree", idx, c2w=self.poses_test[idx])
            sample = self._assemble(rays_o, viewdirs, rgb, mask)
            sample["img_wh"] = np.array((w, h))
        sample["instance_id"] = inst
        sample["articulation_id"] = deg_idx
        return sample


def write_synthetic_multi_scene(root_dir, n_instances=2, n_degrees=3, n_views=60, img_wh=(32, 24), seed=0):
    """A tiny tree in the reference's layout for tests and demos: per instance and joint state, ``n_views`` RGB renders
    of a coloured disc whose size depends on the state, the matching segmentation maps, and look-at poses."""
    from .. import synthetic as syn

    w, h = img_wh
    rng = np.random.Generator(np.random.PCG64(seed))
    angle = 2 * math.atan(0.5 * h / (syn.focal_from_fovy(h) * 320 / w))          # inverts the focal rule above
    yy, xx = np.mgrid[0:h, 0:w]
    for inst in range(n_instances):
        for d in range(n_degrees):
            base = os.path.join(root_dir, f"obj_{inst:03d}", "train", f"{10 * d}_degree")
            os.makedirs(os.path.join(base, "rgb"), exist_ok=True)
            os.makedirs(os.path.join(base, "seg"), exist_ok=True)
            frames = {}
            for i in range(n_views):
                c2w = syn.look_at_pose(4.0, 360.0 * i / n_views + 3.0 * inst, 30.0)
                frames[f"r_{i}"] = torch.cat([c2w, torch.tensor([[0.0, 0.0, 0.0, 1.0]])]).tolist()
                r = np.hypot(xx - w / 2 - 2 * math.cos(i), yy - h / 2) / ((0.25 + 0.05 * d) * h)
                seg = (r < 1.0)
                rgb = np.stack([0.5 + 0.5 * np.sin(xx / 3.0 + inst), 0.5 + 0.5 * np.cos(yy / 4.0 + d), np.full_like(r, 0.2 + 0.01 * i)], -1)
                rgb = np.clip(rgb + rng.uniform(0, 1e-2, rgb.shape), 0, 1)
                Image.fromarray((rgb * 255).astype(np.uint8), "RGB").save(os.path.join(base, "rgb", f"r_{i}.png"))
                Image.fromarray((seg * (1 + d)).astype(np.uint8), "L").save(os.path.join(base, "seg", f"r_{i}.png"))
            with open(os.path.join(base, "transforms.json"), "w") as f:
                json.dump({"camera_angle_x": angle, "frames": frames}, f)
    return root_dir
