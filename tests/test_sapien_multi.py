"""the articulated multi-instance dataset against the items the REAL reference
``SapienDatasetMulti`` produced on the same synthetic tree with the same seeds (tests/golden/g14, generator committed)."""
import os
import random

import numpy as np
import pytest
import torch


def _items(root, device, monkeypatch):
    from aon_amd.datasets.sapien_multi import SapienDatasetMulti

    real = os.listdir
    monkeypatch.setattr(os, "listdir", lambda p: sorted(real(p)))   # the golden fixed the train split's unsorted listdir
    out = {}
    for split, kw in (("train", {}), ("val", {}), ("test_val", {"eval_inference": "render"})):
        ds = SapienDatasetMulti(root, split=split, img_wh=(32, 24), white_back=True, device=device, **kw)
        random.seed(5); np.random.seed(6); torch.manual_seed(7)
        out[split] = (ds, ds[3])
    return out


def _compare(items, g, ray_atol):
    for split, (ds, item) in items.items():
        assert len(ds) == g[f"{split}_len"]
        want = {k[len(split) + 1:] for k in g if k.startswith(split + "_") and not k.endswith("_sum") and k != f"{split}_len"}
        assert set(item) == want, (split, set(item) ^ want)
        for k, v in item.items():
            ref = g[f"{split}_{k}"]
            v = torch.as_tensor(np.asarray(v)) if not isinstance(v, torch.Tensor) else v.cpu()
            if split == "train" and v.dim() >= 1 and v.shape[0] == 4096:
                full_sum = v.double().sum(0)
                v = v[:256]
                assert torch.allclose(full_sum, g[f"{split}_{k}_sum"].double(), rtol=0, atol=4096 * max(ray_atol, 1e-7)), (split, k)
            ref = torch.as_tensor(ref)
            assert tuple(v.shape) == tuple(ref.shape), (split, k, v.shape, ref.shape)
            if k in ("rays_d", "viewdirs"):
                assert (v - ref).abs().max().item() <= ray_atol, (split, k)
            elif v.dtype.is_floating_point:
                assert (v.double() - ref.double()).abs().max().item() <= 1e-6, (split, k)
            else:
                assert torch.equal(v.to(ref.dtype), ref), (split, k)


def test_spheric_poses_bit_exact(golden):
    from aon_amd.datasets.sapien_multi import create_spheric_poses, idx_to_deg

    assert torch.equal(create_spheric_poses(4.0), golden("g14_sapien_multi")["spheric_poses"])
    assert idx_to_deg["train"][9] == 90 and idx_to_deg["val"][8] == 85


def test_items_host_logic_cpu(tmp_path, golden, monkeypatch):
    """Everything but ray generation (file choice, RNG order, masking, gathers, normalisation) on CPU tensors; the GPU
    ray generator is replaced by the oracle's get_rays for this test only."""
    import aon_amd.datasets.sapien_multi as sm
    from oracle import nerf_oracle as orc

    def cpu_rays(h, w, focal, c2w, device=None):
        ro, vd, _ = orc.get_rays(orc.get_ray_directions(h, w, focal), c2w)
        return ro, vd

    monkeypatch.setattr(sm, "get_frame_rays", cpu_rays)
    root = sm.write_synthetic_multi_scene(str(tmp_path / "multi"), n_instances=2, n_degrees=3, n_views=60, img_wh=(32, 24), seed=0)
    _compare(_items(root, "cpu", monkeypatch), golden("g14_sapien_multi"), ray_atol=0.0)


@pytest.mark.gpu
def test_items_on_gpu_and_autodecoder_harness(tmp_path, golden, monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import aon_amd.datasets.sapien_multi as sm
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model_autodecoder import LitNeRF_AutoDecoder

    root = sm.write_synthetic_multi_scene(str(tmp_path / "multi"), n_instances=2, n_degrees=3, n_views=60, img_wh=(32, 24), seed=0)
    items = _items(root, "cuda", monkeypatch)
    _compare(items, golden("g14_sapien_multi"), ray_atol=2e-7)
    # the harness consumes DataLoader(batch_size=1)-shaped batches of these items
    lit = LitNeRF_AutoDecoder({"chunk": 300, "img_wh": (32, 24), "N_max_objs": 2, "N_obj_code_length": 128, "run_max_steps": 1000}).cuda()
    lit.model.load_state_dict(syn.make_art_state_dict(seed=0, density_scale=30.0))
    lit.code_library.load_state_dict(syn.make_code_library_state(seed=0, n_max_objs=2))

    def collate(item):
        out = {}
        for k, v in item.items():
            t = torch.as_tensor(np.asarray(v)) if not isinstance(v, torch.Tensor) else v
            out[k] = t.unsqueeze(0).cuda()
        return out

    opt = lit.configure_optimizers()
    losses = []
    for step in range(3):
        opt.zero_grad()
        loss = lit.training_step(collate(items["train"][1]), step)
        loss.backward()
        lit.optimizer_step(opt)
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert lit.code_library.embedding_instance_articulation.weight.grad is not None
    assert {"train/psnr1", "train/psnr0", "train/loss", "train/loss/reg"} <= set(lit.logged)
    ret = lit.validation_step(collate(items["val"][1]), 0)
    assert ret["comp_rgb"].shape == (32 * 24, 3) and "val/psnr" in lit.logged and "val/psnr_obj" in lit.logged
    out = lit.test_step(collate(items["test_val"][1]), 3)
    assert set(out) == {"target", "instance_mask", "rgb"} and out["rgb"].shape == (32 * 24, 3)
    stats = lit.test_epoch_end([out], image_sizes=[(24, 32)], out_dir=str(tmp_path / "ckpts"))
    assert np.isfinite(stats[0]["test"]) and os.path.exists(tmp_path / "ckpts" / "results.json")


@pytest.mark.gpu
def test_example_run_autodecoder(tmp_path, monkeypatch):
    """examples/run_autodecoder.py end to end on a synthetic tree: items -> training steps ->
    validation -> checkpoint (model + code library keys) -> 19 interpolated-articulation test renders + results.json."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import importlib.util
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("run_autodecoder", os.path.join(root, "examples", "run_autodecoder.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, "argv", ["run_autodecoder.py", "--synthetic", str(tmp_path / "multi"), "--img_wh", "32", "24", "--steps", "12",
                                      "--val_every", "6", "--exp_dir", str(tmp_path / "ck")])
    log, psnr = mod.main()
    assert len(log) == 2 and all(np.isfinite(r["val_psnr"]) for r in log) and np.isfinite(psnr["test"])
    ck = torch.load(tmp_path / "ck" / "last.ckpt", map_location="cpu", weights_only=False)
    assert "code_library.embedding_instance_articulation.weight" in ck["state_dict"] and "model.fine_mlp.deformation_layer.weight" in ck["state_dict"]
    assert os.path.exists(tmp_path / "ck" / "render" / "image018.jpg") and os.path.exists(tmp_path / "ck" / "render" / "results.json")
