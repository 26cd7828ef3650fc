"""dataset (reference on-disk format) -> ray tensors, checkpoint key layout, image / stats dump."""
import json
import os

import numpy as np
import pytest
import torch


def test_checkpoint_roundtrip_and_key_layout(tmp_path):
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model import LitNeRF
    from aon_amd.utils import extract_model_state_dict, load_checkpoint, save_checkpoint

    lit = LitNeRF()
    lit.load_state_dict({"model." + k: v for k, v in syn.make_nerf_state_dict(seed=5).items()})
    lit.global_step = 1234
    opt = lit.configure_optimizers()
    path = save_checkpoint(str(tmp_path / "last.ckpt"), lit, opt, epoch=3)
    raw = torch.load(path, map_location="cpu", weights_only=False)
    keys = set(raw["state_dict"])
    # the reference's names (model.py:77,85,87-89,144-145,218)
    for k in ("model.coarse_mlp.pts_linears.0.weight", "model.coarse_mlp.pts_linears.7.bias", "model.coarse_mlp.views_linear.0.weight",
              "model.coarse_mlp.bottleneck_layer.weight", "model.coarse_mlp.density_layer.bias", "model.fine_mlp.rgb_layer.weight"):
        assert k in keys
    assert len(keys) == 48 and raw["global_step"] == 1234 and raw["epoch"] == 3
    lit2 = LitNeRF()
    load_checkpoint(path, lit2, lit2.configure_optimizers())
    assert lit2.global_step == 1234
    for (k, a), (_, b) in zip(lit.state_dict().items(), lit2.state_dict().items()):
        assert torch.equal(a, b), k
    inner = extract_model_state_dict(path)
    assert set(inner) == {k[len("model."):] for k in keys}


def test_store_image_and_stats(tmp_path):
    from PIL import Image

    from aon_amd.utils import store_image, write_stats

    imgs = [torch.rand(6, 8, 3) * 1.4 - 0.2 for _ in range(2)]
    paths = store_image(str(tmp_path / "renders"), imgs, "image")
    assert [os.path.basename(p) for p in paths] == ["image000.jpg", "image001.jpg"]
    assert Image.open(paths[0]).size == (8, 6)
    d = write_stats(str(tmp_path / "results.json"), {"name": "PSNR", "test": 31.5}, {"name": "SSIM", "test": 0.9})
    assert json.load(open(tmp_path / "results.json")) == d == {"PSNR": {"test": 31.5}, "SSIM": {"test": 0.9}}


def test_dataset_image_io_cpu(tmp_path):
    """IO half of the dataset on CPU tensors (device='cpu' skips nothing but ray generation, which needs the GPU)."""
    from aon_amd.datasets.sapien import SapienDataset, write_synthetic_scene

    root = write_synthetic_scene(str(tmp_path / "scene"), n_train=2, n_val=1, img_wh=(16, 12))
    ds = SapienDataset.__new__(SapienDataset)   # construct without touching the GPU
    ds.root_dir, ds.split, ds.img_wh, ds.white_back, ds.device = root, "val", (16, 12), True, torch.device("cpu")
    ds.base_dir = os.path.join(root, "val")
    ds.meta = json.load(open(os.path.join(ds.base_dir, "transforms.json")))
    rgb, mask = ds.image_of("r_0.png")
    from PIL import Image
    ref = np.asarray(Image.open(os.path.join(ds.base_dir, "rgb", "r_0.png")), dtype=np.float32) / 255.0
    want = ref[..., :3] * ref[..., 3:] + (1 - ref[..., 3:])           # sapien.py:141
    np.testing.assert_allclose(rgb.numpy(), want.reshape(-1, 3), atol=1e-6)
    assert mask.dtype == torch.bool and mask.sum().item() == int((ref[..., 3] > 0).sum())
    assert ds.pose_of("r_0.png").shape == (3, 4)


@pytest.mark.gpu
def test_dataset_to_rays_and_render(tmp_path, nerf_sd):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from aon_amd.datasets.sapien import SapienDataset, write_synthetic_scene
    from aon_amd.models.vanilla_nerf.model import LitNeRF
    from oracle import nerf_oracle as orc

    W, H = 32, 24
    root = write_synthetic_scene(str(tmp_path / "scene"), n_train=3, n_val=1, img_wh=(W, H))
    train = SapienDataset(root, "train", (W, H), white_back=True)
    assert len(train) == 3 * W * H and train.near == 2.0 and train.far == 6.0
    item = train[5]
    assert set(item) == {"rays_o", "rays_d", "viewdirs", "target"} and torch.equal(item["rays_d"], item["viewdirs"])
    # rays equal the reference's CPU construction (get_ray_directions + get_rays) for every training pose
    dirs = orc.get_ray_directions(H, W, train.focal)
    for i, f in enumerate(train.img_files):
        ro, vd, _ = orc.get_rays(dirs, train.pose_of(f))
        sl = slice(i * W * H, (i + 1) * W * H)
        torch.testing.assert_close(train.all_rays_d[sl].cpu(), vd, rtol=0, atol=2e-7)
        assert torch.equal(train.all_rays_o[sl].cpu(), ro)
    val = SapienDataset(root, "val", (W, H), white_back=True)
    batch = val[0]
    assert set(batch) == {"rays_o", "rays_d", "viewdirs", "instance_mask", "target"} and len(val) == 1
    # camera_angle_x form of the focal rule gives the same focal (sapien.py:62-65)
    root2 = write_synthetic_scene(str(tmp_path / "scene2"), n_train=1, n_val=1, img_wh=(W, H), use_camera_angle=True)
    assert abs(SapienDataset(root2, "val", (W, H)).focal - val.focal) < 1e-3 * val.focal
    # end to end: validation_step of the harness on the dataset item (DataLoader batch_size=1 adds a leading dim)
    lit = LitNeRF({"chunk": 500, "img_wh": (W, H)}).cuda()
    lit.load_state_dict({"model." + k: v for k, v in nerf_sd.items()})
    out = lit.validation_step({k: v.unsqueeze(0) for k, v in batch.items()}, 0)
    assert out["comp_rgb"].shape == (W * H, 3) and "val/psnr" in lit.logged
    b = next(train.train_batches(batch_size=256, generator=torch.Generator(device="cuda").manual_seed(0)))
    loss = lit.training_step({k: v.unsqueeze(0) for k, v in b.items()}, 0)
    loss.backward()
    assert torch.isfinite(loss)


@pytest.mark.gpu
def test_example_run_single_scene(tmp_path, monkeypatch):
    """examples/run_single_scene.py end to end on a synthetic scene: dataset -> training steps through the harness ->
    validation -> checkpoint -> test renders + results.json (the slice of the reference's run.py that touches the path)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import importlib.util
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("run_single_scene", os.path.join(root, "examples", "run_single_scene.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, "argv", ["run_single_scene.py", "--synthetic", str(tmp_path / "scene"), "--img_wh", "32", "24", "--steps", "60",
                                      "--val_every", "20", "--batch", "512", "--exp_dir", str(tmp_path / "ck")])
    log, psnr = mod.main()
    assert len(log) == 3 and log[-1]["train_psnr_fine"] > log[0]["train_psnr_fine"] - 0.5 and np.isfinite(psnr["test"])
    assert os.path.exists(tmp_path / "ck" / "last.ckpt") and os.path.exists(tmp_path / "ck" / "render" / "results.json")
    assert os.path.exists(tmp_path / "ck" / "render" / "image000.jpg")
