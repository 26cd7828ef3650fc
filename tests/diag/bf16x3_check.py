"""Accuracy and speed of the opt-in bf16x3 engine against the exact-fp32 kernel and an fp64 evaluation of the network."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aon_amd.synthetic as syn
from aon_amd import ops
from oracle import nerf_oracle as orc

dev = torch.device("cuda:0")
sd = syn.make_nerf_state_dict(seed=0, density_scale=30.0)
params = {k[len("fine_mlp."):]: v.to(dev) for k, v in sd.items() if k.startswith("fine_mlp.")}
p32, pbf = ops.pack_vanilla_mlp(params), ops.pack_vanilla_mlp_bf16x3(params)
n, S = 700, 193
rays = syn.random_rays(n, seed=7)
t = torch.sort(torch.rand(n, S, generator=torch.Generator().manual_seed(7)) * 4 + 2, dim=-1).values
args = [rays[k].to(dev) for k in ("rays_o", "rays_d", "viewdirs")] + [t.to(dev)]
a = ops.mlp_fwd(p32, *args).cpu().double()
b = ops.mlp_fwd_bf16x3(pbf, *args).cpu().double()
# fp64 reference from the fp32 encodings (isolates the matrix arithmetic)
enc = orc.pos_enc(orc.cast_rays(t, rays["rays_o"], rays["rays_d"]), 0, 10).double()
venc = orc.pos_enc(rays["viewdirs"], 0, 4).double()
sd64 = {k: v.double() for k, v in sd.items()}
rgb, sig = orc.nerf_mlp(sd64, "fine_mlp.", enc, venc)
ref = torch.cat([rgb, sig], -1)
enc32 = orc.pos_enc(orc.cast_rays(t, rays["rays_o"], rays["rays_d"]), 0, 10); venc32 = orc.pos_enc(rays["viewdirs"], 0, 4)
rgb32, sig32 = orc.nerf_mlp(sd, "fine_mlp.", enc32, venc32)
cpu32 = torch.cat([rgb32, sig32], -1).double()
def stats(x, name):
    e = (x - ref).abs()
    print(f"{name:<22} rgb max {e[..., :3].max():.3e} mean {e[..., :3].mean():.3e} | sigma max {e[..., 3].max():.3e} mean {e[..., 3].mean():.3e}")
stats(cpu32, "torch CPU fp32"); stats(a, "HIP fp32 MFMA"); stats(b, "HIP bf16x3 (6 products)")
# speed: one fine-level launch over a full frame
H, W = 480, 640
ro, vd = ops.raygen(syn.look_at_pose(), H, W, syn.focal_from_fovy(H), device=dev)
tt, _ = ops.sample_along_rays(ro[:65536], vd[:65536], 192, 2.0, 6.0, want_coords=False)
for name, fn, pk in (("fp32", ops.mlp_fwd, p32), ("bf16x3", ops.mlp_fwd_bf16x3, pbf)):
    fn(pk, ro[:65536], vd[:65536], vd[:65536], tt); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): fn(pk, ro[:65536], vd[:65536], vd[:65536], tt)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(name, f"{dt*1e3:.2f} ms per 65536x193 launch -> {65536*193*1186816/dt/1e12:.1f} TFLOP/s algorithmic")
