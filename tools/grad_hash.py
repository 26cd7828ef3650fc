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


def hashes(n=4096):
    """-> ordered list of (label, 16-hex-digit hash) for the articulated and the vanilla step on n seeded rays."""
    out = []
    _run(n, lambda *a: out.append((" ".join(a[:-1]), a[-1])))
    return out


def main():
    if "--write" in sys.argv:     # tests/golden/g24_gradient_hashes.json (tests/test_hip_arena.py::test_gradient_bits_are_pinned)
        import json

        path = os.path.join(ROOT, "tests", "golden", "g24_gradient_hashes.json")
        json.dump({"n_rays": 4096, "note": "sha256[:16] of every gradient of the seeded articulated / vanilla 4096-ray steps on gfx950 (tools/grad_hash.py)",
                   "hashes": dict(hashes(4096))}, open(path, "w"), indent=0)
        print("wrote", path)
        return
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    _run(n, lambda *a: print(*a))


def _run(n, emit):
    import aon_amd.synthetic as syn
    from aon_amd.models.code_library import CodeLibraryArticulated
    from aon_amd.models.vanilla_nerf.helper import train_loss
    from aon_amd.models.vanilla_nerf.model import NeRF
    from aon_amd.models.vanilla_nerf.model_autodecoder import NeRF_AE_Art

    dev = torch.device("cuda:0")
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
    emit("art loss", h(loss))
    for k, p in list(model.named_parameters()) + [("lib." + k, p) for k, p in lib.named_parameters()]:
        emit("art", k, h(p.grad))
    van = NeRF().to(dev)
    van.load_state_dict(syn.make_nerf_state_dict(seed=0, density_scale=30.0))
    out = van(rays, True, True, 2.0, 6.0, t_rand=tr, u=u)
    loss, _ = train_loss(out, target)
    loss.backward()
    emit("van loss", h(loss))
    for k, p in van.named_parameters():
        emit("van", k, h(p.grad))


if __name__ == "__main__":
    main()
