"""Mirror of ``models/code_library.py:12-71`` (CodeLibraryArticulated): three embedding tables and the test-time
articulation interpolation.  Tiny (three lookups per call) and stays torch, as SURVEY 8(a) R12 prescribes; the
latents it returns feed ``NeRF_AE_Art.forward`` where they are folded into the kernels' bias vectors."""
from __future__ import annotations

import torch
import torch.nn.init as init
from torch import nn


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
