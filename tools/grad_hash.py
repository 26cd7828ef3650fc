"""SHA-256 of every gradient of one training step (articulated config-5 step and the vanilla 4096-ray step, seeded inputs): an A/B aid for
changes that must not move a bit -- run it with two builds of the library (AON_HIP_LIB=... selects an alternative build) and diff.
    python tools/grad_hash.py > a.txt;  AON_HIP_LIB=articulated-object-nerf_amd/libaon_hip_prev.so python tools/grad_hash.py > b.txt;  diff a.txt b.txt"""
import hashlib
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def h(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]


def main():
    import aon_amd.synthetic as syn
    from aon_amd.models.code_library import CodeLibraryArticulated
    from aon_amd.models.vanilla_nerf.helper import train_loss
    from aon_amd.models.vanilla_nerf.model import NeRF
    from aon_amd.models.vanilla_nerf.model_autodecoder import NeRF_AE_Art

    dev = torch.device("cuda:0")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    rays = {k: v.to(dev) for k, v in syn.random_rays(n, seed=11).items()}
    target = syn.seeded_uniform(12, n, 3).to(dev)
    tr, u = syn.seeded_uniform(13, n, 65).to(dev), syn.seeded_uniform(14, n, 128).to(dev)
    model = NeRF_AE_Art().to(dev)
    model.load_state_dict(syn.make_art_state_dict(seed=0, density_scale=30.0))
    lib = CodeLibraryArticulated(types.SimpleNamespace(N_max_objs=1, N_obj_code_length=128)).to(dev)
    lib.load_state_dict(syn.make_code_library_state(seed=0, n_max_objs=1))
    lat = lib({"instance_id": torch.tensor([0], device=dev), "articulation_id": torch.tensor([3], device=dev)})
    out = model(rays, True, True, 2.0, 6.0, lat, t_rand=tr, u=u)
    loss, _ = train_loss(out, target, (lat["density"], lat["color"], lat["articulation"]), 1e-4)
    loss.backward()
    print("art loss", h(loss))
    for k, p in list(model.named_parameters()) + [("lib." + k, p) for k, p in lib.named_parameters()]:
        print("art", k, h(p.grad))
    van = NeRF().to(dev)
    van.load_state_dict(syn.make_nerf_state_dict(seed=0, density_scale=30.0))
    out = van(rays, True, True, 2.0, 6.0, t_rand=tr, u=u)
    loss, _ = train_loss(out, target)
    loss.backward()
    print("van loss", h(loss))
    for k, p in van.named_parameters():
        print("van", k, h(p.grad))


if __name__ == "__main__":
    main()
