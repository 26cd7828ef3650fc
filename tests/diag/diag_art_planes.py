"""Which layer of the articulated training forward disagrees?  Every stored activation plane is recomputed with torch from
the stored plane of its input (fp32 matmul on the GPU) and compared; also raw(train) vs raw(inference)."""
import sys

import torch

sys.path.insert(0, ".")
import aon_amd.synthetic as syn  # noqa: E402
from aon_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
asd = {k: v.to(dev) for k, v in syn.make_art_state_dict(seed=0, density_scale=30.0).items()}
P = {k[len("fine_mlp."):]: v for k, v in asd.items() if k.startswith("fine_mlp.")}
lat = {"density": torch.randn(1, 128, device=dev) * 0.1, "color": torch.randn(1, 128, device=dev) * 0.1,
       "articulation": torch.randn(1, 32, device=dev) * 0.1}
pa, small = ops.pack_art_mlp(P), ops.art_prepare(P, lat)
rays = {k: v.to(dev) for k, v in syn.random_rays(300, seed=3).items()}
t, _ = ops.sample_along_rays(rays["rays_o"], rays["rays_d"], 64, 2.0, 6.0, want_coords=False)
a = ops.art_mlp_fwd(pa, small, rays["rays_o"], rays["rays_d"], rays["viewdirs"], t)
b, planes, masks = ops.art_mlp_fwd_train(pa, small, rays["rays_o"], rays["rays_d"], rays["viewdirs"], t)
n = 300 * 65
print("raw train vs inference: max abs diff per channel", (a - b).abs().amax(dim=(0, 1)).tolist())
pl = ops.plane_rows_view(planes)[:, :n]
D = lambda l: pl[32 + 128 * l: 32 + 128 * (l + 1)]
E = pl[544:544 + 63]
Hh = lambda l: pl[608 + 256 * l: 608 + 256 * (l + 1)]
BOT = pl[608 + 2048: 608 + 2048 + 256]
VE = pl[608 + 2048 + 256: 608 + 2048 + 256 + 27]
V = lambda l: pl[608 + 2048 + 256 + 32 + 128 * l: 608 + 2048 + 256 + 32 + 128 * (l + 1)]
sh, ap, ar = (lat[k].reshape(-1, 1).expand(-1, n) for k in ("density", "color", "articulation"))
relu = torch.relu


def lin(name, x):
    return P[name + ".weight"] @ x + P[name + ".bias"][:, None]


def rep(tag, got, want):
    print(f"{tag:<28} max|diff| {float((got - want).abs().max()):.3e}   (scale {float(want.abs().max()):.2e})")


pos, xd = pl[0:3], pl[3:6]
rep("D0", D(0), relu(lin("deformations_linear.0", torch.cat([pos, sh, ar]))))
for l in (1, 2, 3):
    rep(f"D{l}", D(l), relu(lin(f"deformations_linear.{l}", D(l - 1))))
rep("x'", xd, lin("deformation_layer", D(3)) + pos)
enc = torch.cat([xd] + [torch.sin(xd * 2.0 ** k) for k in range(10)] + [torch.sin(xd * 2.0 ** k + 0.5 * torch.pi) for k in range(10)])
# reference order: [x ; sin(2^l x) l-major ; sin(2^l x + pi/2)]
rep("enc", E, enc)
rep("H0", Hh(0), relu(lin("pts_linears.0", torch.cat([E, sh]))))
for l in (1, 2, 3, 4, 6, 7):
    rep(f"H{l}", Hh(l), relu(lin(f"pts_linears.{l}", Hh(l - 1))))
rep("H5", Hh(5), relu(lin("pts_linears.5", torch.cat([Hh(4), E, sh]))))
rep("bott", BOT, lin("bottleneck_layer", Hh(7)))
rep("V0", V(0), relu(lin("views_linear.0", torch.cat([BOT, VE, ap]))))
for l in (1, 2, 3):
    rep(f"V{l}", V(l), relu(lin(f"views_linear.{l}", V(l - 1))))
rep("rgb (raw train)", b.reshape(n, 4).T[:3], lin("rgb_layer", V(3)))
rep("rgb (raw inference)", a.reshape(n, 4).T[:3], lin("rgb_layer", V(3)))
rep("sigma", b.reshape(n, 4).T[3:4], lin("density_layer", Hh(7)))
vd = rays["viewdirs"].repeat_interleave(65, 0).T
venc = torch.cat([vd] + [torch.sin(vd * 2.0 ** k) for k in range(4)] + [torch.sin(vd * 2.0 ** k + 0.5 * torch.pi) for k in range(4)])
rep("view enc", VE, venc)

# where is V1 wrong?
got, want = V(1), relu(lin("views_linear.1", V(0)))
err = (got - want).abs()
print("V1 err by feature tile (4 x 32):", [f"{float(err[32 * t: 32 * t + 32].max()):.2e}" for t in range(4)])
print("V1 err by feature mod 8 :", [f"{float(err[k::8].max()):.2e}" for k in range(8)])
print("V1 err by sample wave slot (col mod 128 // 32):", [f"{float(err[:, [c for c in range(n) if (c % 128) // 32 == w]].max()):.2e}" for w in range(4)])
print("V1 err by pass (first 8):", [f"{float(err[:, 128 * p_: 128 * p_ + 128].max()):.2e}" for p_ in range(8)])
pre = lin("views_linear.1", V(0))
# hypothesis checks: missing bias? wrong input layer?
for tag, alt in (("no bias", relu(P["views_linear.1.weight"] @ V(0))), ("W2 instead", relu(lin("views_linear.2", V(0)))), ("W3 instead", relu(lin("views_linear.3", V(0)))),
                 ("bias of V2", relu(P["views_linear.1.weight"] @ V(0) + P["views_linear.2.bias"][:, None])),
                 ("input pre-relu Z0?", relu(lin("views_linear.1", lin("views_linear.0", torch.cat([BOT, VE, ap])))))):
    print(f"  alt {tag:<20} max|diff| {float((got - alt).abs().max()):.3e}")

W1, b1 = P["views_linear.1.weight"], P["views_linear.1.bias"]
g0 = got[:32]
base = (W1[:32] @ V(0))
print("tile0 hypotheses (pre-relu compare on positive entries):")
pos_mask = g0 > 0
def chk(tag, cand):
    c = relu(cand)
    print(f"  {tag:<40} max|diff| {float((g0 - c).abs().max()):.3e}")
chk("correct", base + b1[:32, None])
chk("no bias", base)
chk("2x bias", base + 2 * b1[:32, None])
for j in range(4):
    part = W1[:32, 32 * j: 32 * j + 32] @ V(0)[32 * j: 32 * j + 32]
    chk(f"missing input tile {j}", base - part + b1[:32, None])
    chk(f"input tile {j} doubled", base + part + b1[:32, None])
# per-k-slice (the MFMA step structure: q = 0..3 -> input regs 4q..4q+3 of a tile = features 8q + 4h + c)
for j in range(4):
    for q in range(4):
        cols = [32 * j + 8 * q + 4 * hh + c for hh in range(2) for c in range(4)]
        part = W1[:32, cols] @ V(0)[cols]
        chk(f"missing tile {j} q {q}", base - part + b1[:32, None])
for c0 in range(4):
    cols = [32 * j + 8 * q + 4 * hh + c0 for j in range(4) for q in range(4) for hh in range(2)]
    part = W1[:32, cols] @ V(0)[cols]
    chk(f"missing cc={c0} everywhere", base - part + b1[:32, None])
