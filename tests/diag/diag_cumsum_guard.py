"""How often does exact_prefix_f64 (csrc/aon_render.hip) leave its tree-scan fast path?  CPU emulation of its two guards on
coarse weights of the synthetic scenes and on adversarial rows:  python tests/diag/diag_cumsum_guard.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import aon_amd.synthetic as syn  # noqa: E402
from oracle import nerf_oracle as orc  # noqa: E402


def guard_stats(w):
    w = torch.as_tensor(w, dtype=torch.float32)
    ws = w.sum(-1, keepdim=True)
    pad = torch.clamp(1e-5 - ws, min=0)
    w, ws = w + pad / 63, ws + pad
    pdf = (w / ws).numpy()[:, :62]
    P = np.cumsum(pdf.astype(np.float64), axis=1)
    e = (pdf.view(np.uint32) >> 23) & 255
    ulp = np.where(pdf == 0, 1 << 30, np.maximum(e, 1)).astype(np.int64)
    min_ulp = np.minimum.accumulate(ulp, axis=1)
    ed = ((P.view(np.uint64) >> 52) & 0x7FF).astype(np.int64)
    exact = (ed - min_ulp) <= 924
    far = (P * (1 - 2.0 ** -46)).astype(np.float32) == (P * (1 + 2.0 ** -46)).astype(np.float32)
    return {"rows": len(pdf), "rows_with_inexact_prefix": float((~exact).any(1).mean()), "rows_on_the_chain": float((~(exact | far)).any(1).mean())}


if __name__ == "__main__":
    for scale in (30.0, 2.0):
        sd = syn.make_nerf_state_dict(seed=0, density_scale=scale)
        _, aux = orc.nerf_forward(sd, syn.random_rays(3000, seed=1), False, True, 2.0, 6.0, return_aux=True)
        print(f"synthetic scene, density x{scale:g}:", guard_stats(aux[0]["weights"][:, 1:-1]))
    rng = np.random.default_rng(0)
    print("uniform weights:", guard_stats(rng.random((3000, 63), dtype=np.float32)))
    print("weights over 17 decades:", guard_stats(rng.random((3000, 63), dtype=np.float32) * np.exp(-rng.random((3000, 63)) * 40).astype(np.float32)))
