"""Mirror of ``models/code_library.py:12-71`` (CodeLibraryArticulated): three embedding tables and the test-time
articulation interpolation.  The latents it returns feed ``NeRF_AE_Art.forward`` where they are folded into the kernels' bias vectors.

The tables stay ``nn.Embedding`` modules (same ``state_dict`` keys).  On a GPU, for the reference's batch of one object in one state
(``instance_id`` / ``articulation_id`` of one element: sapien_multi.py:362-479), the three training lookups are ONE launch and their
dense table gradients ONE launch (``aon_code_library_fwd`` / ``_bwd``, round 6) where torch runs three index-selects and three
fill + scatter pairs per step; the gradients are written straight into the parameter arena's slots when the tables live in one
(``aon_amd/arena.py``).  Everything else -- CPU tensors ("stays torch"), batched ids, the test-time table -- is torch."""
from __future__ import annotations

import torch
import torch.nn.init as init
from torch import nn


class _Lookup3(torch.autograd.Function):
    """(shape table, appearance table, articulation table, instance_id, articulation_id) -> three (1, dim) rows; backward: the dense table
    gradients nn.Embedding's autograd produces (zeros except the looked-up row)."""

    @staticmethod
    def forward(ctx, w_shape, w_app, w_art, instance_id, articulation_id):
        from .. import ops
        from ..autograd import _arena_plan

        outs, ids = ops.code_library_fwd((w_shape, w_app, w_art), (instance_id, instance_id, articulation_id))
        ctx.ids = ids
        ctx.shapes = [tuple(w.shape) for w in (w_shape, w_app, w_art)]
        ctx.set_materialize_grads(False)
        _arena_plan(ctx, (w_shape, w_app, w_art))
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_shape, g_app, g_art):
        from .. import ops
        from ..autograd import _arena_done, _arena_grads

        slots = _arena_grads(ctx, [ctx.shapes])
        grads = ops.code_library_bwd((g_shape, g_app, g_art), ctx.ids, ctx.shapes, outs=None if slots is None else slots[0])
        _arena_done(ctx)
        return grads[0], grads[1], grads[2], None, None


class CodeLibraryArticulated(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        n_art, art_len = 10, 32
        self.embedding_instance_shape = nn.Embedding(hparams.N_max_objs, hparams.N_obj_code_length)
        self.embedding_instance_appearance = nn.Embedding(hparams.N_max_objs, hparams.N_obj_code_length)
        self.embedding_instance_articulation = nn.Embedding(n_art, art_len)
        init.xavier_uniform_(self.embedding_instance_shape.weight)
        init.xavier_uniform_(self.embedding_instance_appearance.weight)
        init.xavier_uniform_(self.embedding_instance_articulation.weight)

    def forward(self, batch, is_test=False):
        w = self.embedding_instance_shape.weight
        iid, aid = batch["instance_id"], batch["articulation_id"]
        if (not is_test and w.is_cuda and torch.is_tensor(iid) and torch.is_tensor(aid) and iid.numel() == 1 and aid.numel() == 1
                and iid.dim() == 1 and aid.dim() == 1 and iid.is_cuda and aid.is_cuda and iid.dtype == torch.int64 and aid.dtype == torch.int64):
            d, c, a = _Lookup3.apply(w, self.embedding_instance_appearance.weight, self.embedding_instance_articulation.weight, iid, aid)
            return {"density": d, "color": c, "articulation": a}
        ret = {"density": self.embedding_instance_shape(batch["instance_id"]),
               "color": self.embedding_instance_appearance(batch["instance_id"])}
        if is_test:
            table = self.get_interpolated_articulations(max_interpolations=2, device=batch["articulation_id"].device)
            ret["articulation"] = table[batch["articulation_id"]]
        else:
            ret["articulation"] = self.embedding_instance_articulation(batch["articulation_id"])
        return ret

    def get_interpolated_articulations(self, max_interpolations=2, device="cuda"):
        """code_library.py:55-71: 10 learned codes at even slots, mid-points of neighbours at odd slots (19 rows)."""
        w = self.embedding_instance_articulation.weight.to(device)
        n = w.shape[0]
        table = torch.zeros((n * max_interpolations) - 1, w.shape[1], device=device, dtype=w.dtype)
        table[0::2] = w
        table[1::2] = (w[:-1] + w[1:]) / 2
        return table
