"""Decision-bit words written by the vanilla training forward against (stored post-ReLU plane > 0), bit by bit."""
import sys

import torch

sys.path.insert(0, ".")
import aon_amd.synthetic as syn  # noqa: E402
from aon_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
sd = {k: v.to(dev) for k, v in syn.make_nerf_state_dict(seed=0, density_scale=30.0).items()}
P = {k[len("fine_mlp."):]: v for k, v in sd.items() if k.startswith("fine_mlp.")}
pv = ops.pack_vanilla_mlp(P)
rays = {k: v.to(dev) for k, v in syn.random_rays(256, seed=3).items()}
t, _ = ops.sample_along_rays(rays["rays_o"], rays["rays_d"], 64, 2.0, 6.0, want_coords=False)
raw, planes, masks = ops.mlp_fwd_train(pv, rays["rays_o"], rays["rays_d"], rays["viewdirs"], t)
Np = ops.plane_samples(planes)
planes = ops.plane_rows_view(planes)
mk = masks.view(torch.int32).view(9, Np * 2, 4)          # [layer][pass*256 + tid][word]
for layer in range(9):
    rows = planes[64 + 256 * layer: 64 + 256 * (layer + 1)] if layer < 8 else planes[64 + 2048 + 256 + 32: 64 + 2048 + 256 + 32 + 128]
    nt = 8 if layer < 8 else 4
    bad = 0
    tot = 0
    for p_ in range(Np // 128):
        w = mk[layer, p_ * 256: p_ * 256 + 256]            # (256 threads, 4 words)
        tid = torch.arange(256, device=dev)
        lane, wave = tid & 63, tid >> 6
        m_, h = lane & 31, lane >> 5
        col = p_ * 128 + wave * 32 + m_
        for tl in range(nt):
            for r in range(16):
                feat = 32 * tl + (r & 3) + 8 * (r >> 2) + 4 * h
                want = rows[feat, col] > 0
                got = ((w[:, tl >> 1] >> ((tl & 1) * 16 + r)) & 1).bool()
                bad += int((want != got).sum())
                tot += 256
    print(f"layer {layer}: {bad} / {tot} decision bits differ from (plane > 0)")
