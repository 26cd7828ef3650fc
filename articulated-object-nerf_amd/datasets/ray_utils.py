"""Mirror of the on-path part of the reference's ``datasets/ray_utils.py`` (get_ray_directions :71-90,
get_rays :118-159), computed on the GPU by HIP kernels instead of on the CPU inside the Dataset."""
from __future__ import annotations

import torch

from .. import ops


def get_ray_directions(H, W, focal, device=None):
    """(H,W,3) camera-space directions ((i-W/2)/focal, -(j-H/2)/focal, -1); no +0.5 pixel centre."""
    return ops.ray_directions(int(H), int(W), float(focal), device=device)


def get_rays(directions, c2w, output_view_dirs=False, output_radii=False):
    """ray_utils.py:118-159.  Returns (rays_o, rays_d) or, with output_view_dirs, (rays_o, viewdirs, rays_d)
    where rays_d is the same tensor as viewdirs (the reference normalises rays_d in place through its viewdirs
    alias, :146-147); with ``output_view_dirs=True, output_radii=True`` -- the only call form the reference's
    datasets use (sapien.py:102,145; sapien_multi.py:301,343) -- the 4-tuple (rays_o, viewdirs, rays_d, radii)
    with radii (H*W,) as ray_utils.py:138-143 computes them (needs (H,W,3) directions).  Like the reference,
    ``output_radii`` without ``output_view_dirs`` returns the 2-tuple."""
    rays_o, viewdirs = ops.get_rays(directions, c2w)
    if output_view_dirs:
        if output_radii:
            return rays_o, viewdirs, viewdirs, ops.ray_radii(directions, c2w)
        return rays_o, viewdirs, viewdirs
    return rays_o, viewdirs


def get_frame_rays(H, W, focal, c2w, pix_begin=0, pix_end=None, device=None):
    """Fused get_ray_directions + get_rays for a contiguous row-major pixel range (what a rank renders)."""
    return ops.raygen(c2w, int(H), int(W), float(focal), pix_begin, pix_end, device=device)
