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


def _active(group) -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def all_gather_pixels(level, group=None, total: int | None = None, counts=None, force: bool = False, _pad_to: int = 0):
    """level = (rgb (n,3), acc (n,), depth (n,)) of this rank's ray range -> the same triple for ALL ranks' rays,
    rank-major (rank 0's range first), with ONE all_gather_into_tensor and no host synchronisation.

    Every rank must be able to name every rank's ray count without asking, so with more than one rank the layout is a
    REQUIRED argument: `total` = number of rays of the sharded frame (counts follow from `shard_range`, the layout
    `render_frame_sharded` uses) or `counts` = explicit per-rank list.  (Guessing "everyone holds what I hold" would pass
    the local check on every rank of an uneven layout and then hang or corrupt inside RCCL on mismatched buffers.)  Equal
    counts gather in place into the final (total,5) buffer; uneven ones (they differ by at most one ray under shard_range)
    pad to the longest and the padding rows are dropped with host-known offsets.  `force` runs the collective even at
    world size 1 (tests); `_pad_to` (tests) pads every rank's message to at least that many rows, which drives the
    uneven-layout branch -- padding, gather, drop -- through the real collective on a one-GPU box."""
    rgb, acc, depth = level
    if not (_active(group) or (force and dist.is_available() and dist.is_initialized())):
        return rgb, acc, depth
    world, n = dist.get_world_size(group), rgb.shape[0]
    if counts is None:
        if total is None:
            if world > 1:
                raise ValueError("all_gather_pixels: with more than one rank pass `total` (shard_range layout) or `counts`")
            total = n
        counts = [e - b for b, e in (shard_range(total, r, world) for r in range(world))]
    counts = [int(c) for c in counts]
    if len(counts) != world or counts[dist.get_rank(group)] != n:
        raise ValueError(f"all_gather_pixels: this rank holds {n} rays but the stated layout is {counts}")
    nmax = max(max(counts), int(_pad_to))
    mine = rgb.new_empty((nmax, 5))
    mine[:n, :3] = rgb
    mine[:n, 3] = acc
    mine[:n, 4] = depth
    out = mine.new_empty((world * nmax, 5))
    dist.all_gather_into_tensor(out, mine, group=group)
    if min(counts) != nmax:
        out = torch.cat([out[r * nmax: r * nmax + c] for r, c in enumerate(counts)], dim=0)
    return unpack_pixels(out)


def render_frame_sharded(model, H: int, W: int, focal: float, c2w, near: float, far: float, white_bkgd: bool,
                         raygen, group=None, force: bool = False):
    """Config 3: one frame, ray ranges sharded over the ranks of `group`, pixels all-gathered.
    `raygen(H, W, focal, c2w, begin, end)` -> (rays_o, viewdirs) for the rank's row-major pixel range
    (aon_amd.datasets.ray_utils.get_frame_rays on GPUs).  Returns the full-frame fine-level (rgb, acc, depth)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    begin, end = shard_range(H * W, rank, world)
    rays_o, viewdirs = raygen(H, W, focal, c2w, begin, end)
    out = model({"rays_o": rays_o, "rays_d": viewdirs, "viewdirs": viewdirs}, False, white_bkgd, near, far)
    return all_gather_pixels(out[-1], group=group, total=H * W, force=force)


# --------------------------------------------------------------------------------------------------------------------
# training: data-parallel gradient exchange (the reference wraps its module in Lightning's DDPPlugin, run.py:151)
# --------------------------------------------------------------------------------------------------------------------
def broadcast_parameters(module, src: int = 0, group=None, force: bool = False) -> None:
    """What DDP does at wrap time: every rank starts from rank `src`'s parameters (one flat broadcast)."""
    if not (_active(group) or (force and dist.is_available() and dist.is_initialized())):
        return
    with torch.no_grad():
        params = list(module.parameters())
        arena = _arena_bucket([p for p in params if p.requires_grad]) if all(p.requires_grad for p in params) else None
        if arena is not None:       # the parameters already ARE one flat buffer (aon_amd/arena.py): broadcast it in place
            dist.broadcast(arena.flat[: arena.total], src=src, group=group)
            return
        flat = torch.cat([p.detach().reshape(-1) for p in params])
        dist.broadcast(flat, src=src, group=group)
        torch._foreach_copy_([p.data for p in params], [c.view_as(p) for c, p in zip(flat.split([p.numel() for p in params]), params)])


class UnevenGradientsError(RuntimeError):
    """Ranks disagreed on which parameters produced a gradient while `find_unused_parameters=False` (see allreduce_gradients)."""


_pending_checks: list = []   # [(event or None, host flag tensor, message)] of earlier calls


def _raise_if_earlier_call_was_uneven() -> None:
    """Every rank must reach the SAME verdict about an earlier exchange before it enters the next one: a rank that raised while the
    others went into reduce_scatter would leave them hanging until the RCCL watchdog fires (ADVICE r4 -- rounds 3-4 looked at the
    pinned flag with a non-blocking event query, which can have completed on one rank and not on another).  So the event of the
    earlier call is WAITED for.  The flag itself is computed from the all-gathered counts, i.e. identical on every rank; with the wait
    the decision is too.  The wait costs nothing in steady state: the earlier exchange finished a whole forward + backward ago, and
    the host still runs up to one step ahead of the device."""
    pending, _pending_checks[:] = list(_pending_checks), []
    for ev, flag, msg in pending:
        if ev is not None:
            ev.synchronize()
        if bool(flag.item()):
            raise UnevenGradientsError(msg)


def allreduce_gradients(module, group=None, force: bool = False, shard_align: int = 64, find_unused_parameters: bool = False) -> None:
    """Mean of the per-rank gradients in ONE flat bucket (vanilla: 1,191,688 fp32 = 4.77 MB; articulated 6.4 MB), call
    between loss.backward() and optimizer.step().

    The bucket spans EVERY parameter that requires grad, in module order, with zeros where this rank produced no gradient
    -- ranks therefore always agree on the message size -- followed by one "had a gradient" flag per parameter, summed in the
    same exchange.  A parameter no rank touched (num_levels=1 leaves fine_mlp alone) keeps `.grad = None` exactly as under
    torch DDP, so the optimizer skips it and its Adam state does not advance.

    `find_unused_parameters=False` (the default, and what the reference wraps its module with: run.py:151) is DDP's contract:
    every rank produces gradients for the SAME set of parameters.  Then "`.grad is None` here" already means "None
    everywhere": the exchange needs NO host read and the call never synchronises with the device (round 3 read the flags
    back with `.tolist()` on every rank that lacked a gradient -- every step of a `num_levels=1` run).  The contract is still
    checked, the way DDP checks it -- late: a device-side comparison of the summed flags with {0, world} is copied to pinned
    memory behind the exchange, and the NEXT call (or `check_gradient_exchange()`) waits for that copy -- long finished by then --
    and raises UnevenGradientsError if it failed: on EVERY rank, before any of them enters the next collective.
    `find_unused_parameters=True` is the permissive mode (a rank may skip a code-library row another rank trained): ranks
    that lack a gradient read the summed flags (one host synchronisation, as torch DDP has in that mode) and adopt the mean
    of the ranks that had one.

    On RCCL the exchange is the direct reduce-scatter + all-gather pair (each of the 8 fully connected xGMI peers reduces one
    eighth of the bucket, scales it, and the eighths are gathered: 2 x 7/8 of the bucket per link instead of a ring's 2 x 7
    hops).  Every rank's shard is a multiple of `shard_align` elements (256 B): the bucket is zero-padded to world x that."""
    if not (_active(group) or (force and dist.is_available() and dist.is_initialized())):
        return
    _raise_if_earlier_call_was_uneven()
    params = [p for p in module.parameters() if p.requires_grad]
    if not params:
        return
    world = dist.get_world_size(group)
    ref = params[0]
    quantum = world * max(1, int(shard_align))
    had = [p.grad is not None for p in params]
    # Round 6: parameters that live in ONE arena (aon_amd/arena.py -- the harness's optimizer puts them there) are reduced IN PLACE: the
    # gradient arena already is the flat bucket (the HIP backward wrote every gradient into its slot), the per-parameter flags go into its
    # spare tail, and nothing is allocated or copied per step (rounds 3-5: a fresh torch.zeros bucket + 2 x 83-tensor copies).
    arena = _arena_bucket(params)
    if arena is not None:
        total = arena.total
        padded = (total + len(params) + quantum - 1) // quantum * quantum
        if padded > arena.capacity:
            arena = None
    with torch.no_grad():
        if arena is not None:
            flat = arena.grad[:padded]
            chunks = [arena.grad_view(i) for i in range(len(params))]
            for i, (p, h) in enumerate(zip(params, had)):
                if not h:
                    chunks[i].zero_()                      # this rank contributes nothing to that parameter
                elif not arena.grad_in_place(i):           # a gradient that arrived outside the arena (another autograd path): adopt it
                    chunks[i].copy_(p.grad)
                    p.grad = chunks[i]
            in_place = True
        else:
            sizes = [p.numel() for p in params]
            total = sum(sizes)
            padded = (total + len(params) + quantum - 1) // quantum * quantum
            flat = torch.zeros(padded, dtype=ref.dtype, device=ref.device)
            chunks = flat[:total].split(sizes)
            if any(had):
                torch._foreach_copy_([c for c, h in zip(chunks, had) if h], [p.grad.reshape(-1) for p, h in zip(params, had) if h])
            in_place = False
        flat[total: total + len(params)] = torch.tensor([float(h) for h in had], dtype=ref.dtype).to(ref.device, non_blocking=True)
        if in_place and padded > total + len(params):
            flat[total + len(params): padded].zero_()
        # one code path for both backends: reduce-scatter -> scale the shard -> all-gather.  gloo has no
        # reduce_scatter_tensor, so there the scatter is an all_reduce of which every rank keeps its own shard -- the shard
        # arithmetic (padding, offsets, rank order) is then exactly what RCCL runs and the CPU tests cover it.
        per = padded // world
        r = dist.get_rank(group)
        if dist.get_backend(group) == "nccl":
            shard = flat.new_empty(per)
            dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM, group=group)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            shard = flat[r * per: (r + 1) * per].clone()
            flat.fill_(float("nan"))   # nothing below may depend on what the all_reduce left outside this rank's shard
        # the flags are COUNTS and stay unscaled: only the gradient part of this rank's shard is divided by the world size
        lo, hi = r * per, (r + 1) * per
        ngrad = min(max(total - lo, 0), per)
        if ngrad:
            shard[:ngrad] /= world
        dist.all_gather_into_tensor(flat, shard, group=group)
        if not in_place and any(had):
            torch._foreach_copy_([p.grad for p, h in zip(params, had) if h], [c.view_as(p) for c, p, h in zip(chunks, params, had) if h])
        counts = flat[total: total + len(params)]
        if find_unused_parameters:
            if not all(had):   # the one host read of the exchange, and only on ranks that lack a gradient some other rank may have
                any_rank = (counts > 0).tolist()
                for c, p, h, a in zip(chunks, params, had, any_rank):
                    if not h and a:
                        p.grad = c.view_as(p) if in_place else c.view_as(p).clone()   # (in place: the slot itself becomes the gradient)
        elif world > 1 or force:
            # DDP's contract, checked without waiting: every count must be 0 or `world` (`force`: the world-1 GPU tests drive this path --
            # pinned flag, event, late look -- through the real backend)
            uneven = ((counts > 0.5) & (counts < world - 0.5)).any()
            msg = ("allreduce_gradients(find_unused_parameters=False): in an earlier step the ranks produced gradients for different sets "
                   "of parameters (torch DDP raises the same way, one iteration late: run.py:151); pass find_unused_parameters=True")
            if uneven.is_cuda:
                host = torch.empty((), dtype=torch.bool, pin_memory=True)
                host.copy_(uneven, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                _pending_checks.append((ev, host, msg))
            else:
                _pending_checks.append((None, uneven, msg))


def _arena_bucket(params):
    """The ParamArena whose layout IS this parameter list (same tensors, same order), or None."""
    from .arena import arena_of

    ao = arena_of(params)
    if ao is None:
        return None
    arena = ao[0]
    if len(arena.params) != len(params) or any(a is not b for a, b in zip(arena.params, params)):
        return None
    return arena


def check_gradient_exchange() -> None:
    """Wait for the deferred checks of earlier allreduce_gradients(find_unused_parameters=False) calls and raise
    UnevenGradientsError if one failed (end of an epoch, before a checkpoint)."""
    _raise_if_earlier_call_was_uneven()
