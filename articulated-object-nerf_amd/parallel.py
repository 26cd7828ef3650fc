"""Multi-GPU form of the path: rays are independent units, so a frame (or a batch of frames) is sharded by
contiguous row-major ray ranges across ranks, weights are replicated, and the only exchange step is ONE all-gather
of the rendered pixels (rgb, acc, depth = 20 B/ray) -- RCCL over xGMI on GPUs (torch backend "nccl"), gloo in the
CPU tests.  No other collective exists on the forward path.

The reference's own gather (models/interface.py:31-51) interleaves ranks pixel by pixel when world > 1 (a latent
bug); this module keeps every rank's range contiguous instead.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced [begin, end) of n items for `rank` (first n % world ranks get one extra)."""
    base, rem = divmod(n, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def pack_pixels(rgb: torch.Tensor, acc: torch.Tensor, depth: torch.Tensor) -> torch.Tensor:
    """(n,5) fp32 = rgb(3) | acc | depth: one message per rank instead of three."""
    return torch.cat([rgb, acc[:, None], depth[:, None]], dim=1).contiguous()


def unpack_pixels(p: torch.Tensor):
    return p[:, :3], p[:, 3], p[:, 4]


def all_gather_pixels(level, group=None):
    """level = (rgb (n,3), acc (n,), depth (n,)) of this rank's ray range -> the same triple for ALL ranks' rays,
    rank-major (rank 0's range first).  Ranges may differ in length by one ray (see shard_range): shorter ranks are
    padded to the longest so a single all_gather_into_tensor moves everything."""
    rgb, acc, depth = level
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return rgb, acc, depth
    world = dist.get_world_size(group)
    mine = pack_pixels(rgb, acc, depth)
    n = torch.tensor([mine.shape[0]], dtype=torch.int64, device=mine.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    nmax = max(counts)
    if mine.shape[0] < nmax:
        mine = torch.cat([mine, mine.new_zeros(nmax - mine.shape[0], 5)], dim=0)
    out = mine.new_empty((world * nmax, 5))
    dist.all_gather_into_tensor(out, mine, group=group)
    if min(counts) != nmax:
        out = torch.cat([out[r * nmax: r * nmax + c] for r, c in enumerate(counts)], dim=0)
    return unpack_pixels(out)


def render_frame_sharded(model, H: int, W: int, focal: float, c2w, near: float, far: float, white_bkgd: bool,
                         raygen, group=None):
    """Config 3: one frame, ray ranges sharded over the ranks of `group`, pixels all-gathered.
    `raygen(H, W, focal, c2w, begin, end)` -> (rays_o, viewdirs) for the rank's row-major pixel range
    (aon_amd.datasets.ray_utils.get_frame_rays on GPUs).  Returns the full-frame fine-level (rgb, acc, depth)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    begin, end = shard_range(H * W, rank, world)
    rays_o, viewdirs = raygen(H, W, focal, c2w, begin, end)
    out = model({"rays_o": rays_o, "rays_d": viewdirs, "viewdirs": viewdirs}, False, white_bkgd, near, far)
    return all_gather_pixels(out[-1], group=group)


# --------------------------------------------------------------------------------------------------------------------
# training: data-parallel gradient exchange (the reference wraps its module in Lightning's DDPPlugin, run.py:151)
# --------------------------------------------------------------------------------------------------------------------
def broadcast_parameters(module, src: int = 0, group=None) -> None:
    """What DDP does at wrap time: every rank starts from rank `src`'s parameters."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        flat = torch.cat([p.detach().reshape(-1) for p in module.parameters()])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for p in module.parameters():
            p.copy_(flat[off: off + p.numel()].view_as(p))
            off += p.numel()


def allreduce_gradients(module, group=None) -> None:
    """Mean of the per-rank gradients in ONE flat bucket (vanilla: 1,191,688 fp32 = 4.77 MB; articulated 6.4 MB --
    a single message per step, which on 8 fully connected xGMI peers is latency- rather than bandwidth-bound, so one
    bucket beats DDP's default 25 MB bucketing logic trivially).  Call between loss.backward() and optimizer.step()."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    params = [p for p in module.parameters() if p.grad is not None]
    if not params:
        return
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= dist.get_world_size(group)
    off = 0
    for p in params:
        p.grad.copy_(flat[off: off + p.numel()].view_as(p.grad))
        off += p.numel()
