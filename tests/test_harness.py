"""Harness-level equivalents: LitNeRF.render_rays / render_rays_test / validation_step /
training_step / learning-rate rule, mirroring models/vanilla_nerf/model.py:256-419 without pytorch-lightning."""
import numpy as np
import pytest
import torch


def test_lr_rule_matches_reference_formula():
    from aon_amd.models.vanilla_nerf.model import LitNeRF

    lit = LitNeRF()
    for step in (0, 1, 100, 2499, 2500, 50_000, 100_000, 150_000):
        # model.py:402-414 restated with numpy exactly as the reference writes it
        delay = 0.01 + (1 - 0.01) * np.sin(0.5 * np.pi * np.clip(step / 2500, 0, 1))
        t = np.clip(step / 100000, 0, 1)
        want = delay * np.exp(np.log(5e-4) * (1 - t) + np.log(5e-6) * t)
        assert abs(lit.lr_at_step(step) - want) <= 1e-12 * want
    opt = lit.configure_optimizers()
    assert isinstance(opt, torch.optim.Adam) and opt.defaults["betas"] == (0.9, 0.999) and opt.defaults["lr"] == 5e-4
    assert sorted(k for k, _ in lit.named_parameters())[0].startswith("model.coarse_mlp.")  # Lightning ckpt key layout


@pytest.mark.gpu
def test_render_rays_contract_and_training_step(nerf_sd):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model import LitNeRF

    dev = torch.device("cuda:0")
    lit = LitNeRF({"chunk": 1000}).to(dev)
    lit.load_state_dict({"model." + k: v for k, v in nerf_sd.items()})  # a Lightning checkpoint's state_dict keys
    H, W = 40, 60
    frame = {k: v.to(dev) for k, v in syn.make_rays(H, W).items()}
    batch = dict(frame, target=torch.rand(H * W, 3, device=dev), instance_mask=torch.ones(H * W, dtype=torch.bool, device=dev))
    ret = lit.render_rays(batch, 0)                      # 2400 rays in chunks of 1000
    assert set(ret) == {"comp_rgb", "acc", "depth"} and ret["comp_rgb"].shape == (H * W, 3) and ret["depth"].shape == (H * W,)
    with torch.no_grad():
        whole = lit.model(frame, False, True, 2.0, 6.0)
    assert torch.equal(ret["comp_rgb"], whole[1][0]) and torch.equal(ret["acc"], whole[1][1])   # chunking invariance
    mse = torch.mean((ret["comp_rgb"] - batch["target"]) ** 2)
    assert abs(lit.logged["val/psnr"][-1] - (-10 * torch.log10(mse)).item()) < 1e-4
    val = lit.validation_step({k: v.unsqueeze(0) for k, v in batch.items()}, 0)     # DataLoader batch_size=1 adds a dim
    assert torch.equal(val["comp_rgb"], ret["comp_rgb"])
    tst = lit.test_step({k: v.unsqueeze(0) for k, v in batch.items()}, 0)
    assert set(tst) == {"target", "instance_mask", "rgb"} and torch.equal(tst["rgb"], ret["comp_rgb"])
    # one optimisation step through the harness
    opt = lit.configure_optimizers()
    sel = torch.arange(0, H * W, 5, device=dev)
    train_batch = {k: v[sel].unsqueeze(0) for k, v in batch.items() if k != "instance_mask"}
    loss = lit.training_step(train_batch, 0)
    loss.backward()
    lit.optimizer_step(opt)
    assert opt.param_groups[0]["lr"] == pytest.approx(lit.lr_at_step(0)) and lit.global_step == 1
    assert {"train/psnr0", "train/psnr1", "train/loss"} <= set(lit.logged)
    loss2 = lit.training_step(train_batch, 1)
    assert torch.isfinite(loss2)


def test_harness_hands_constructor_arguments_through():
    """LitNeRF(model_kwargs=...) / LitNeRF_AutoDecoder(model_kwargs=...): the reference builds NeRF() / NeRF_AE_Art() (model.py:218,
    model_autodecoder.py:356); here the constructor arguments of the drop-in classes can be chosen at the harness level too."""
    from aon_amd.models.vanilla_nerf.model import LitNeRF
    from aon_amd.models.vanilla_nerf.model_autodecoder import LitNeRF_AutoDecoder

    lit = LitNeRF(model_kwargs=dict(num_coarse_samples=32, num_fine_samples=64, max_deg_point=6, deg_view=2, lindisp=True))
    assert lit.model.num_coarse_samples == 32 and lit.model._opts.Sf == 97 and lit.model._opts.lindisp
    assert lit.model.coarse_mlp.pts_linears[0].weight.shape == (256, 39) and lit.model._fused_inference and not lit.model._general
    assert LitNeRF().model.coarse_mlp.geometry.is_default
    art = LitNeRF_AutoDecoder(model_kwargs=dict(num_fine_samples=64, rgb_padding=0.01))
    assert art.model._opts.num_fine_samples == 64 and art.model._opts.rgb_padding == 0.01


def test_log_series_reads_device_scalars_lazily():
    """Harness.log keeps tensors as they are (no float() = host synchronisation per logged value) and hands out floats on access."""
    from aon_amd.models.interface import _LogSeries

    s = _LogSeries()
    s.append(torch.tensor(1.5))
    s.append(2.5)
    s.append(torch.tensor([3.5])[0])
    assert len(s) == 3 and s[-1] == 3.5 and list(s) == [1.5, 2.5, 3.5] and all(isinstance(v, float) for v in s)
    for i in range(600):
        s.append(torch.tensor(float(i)))
    assert len(s._wait) < 512 and s[3] == 0.0 and s[-1] == 599.0
    # round 6: the list proper only ever holds floats -- every other way of looking at it settles the pending scalars first
    import copy
    import pickle

    s.append(torch.tensor(7.0))
    assert s.copy()[-1] == 7.0 and all(isinstance(v, float) for v in s.copy())
    s.append(torch.tensor(8.0))
    assert (s + [1.0])[-2] == 8.0 and repr(s).endswith("8.0]") and 8.0 in s and s == list(s)
    s.append(torch.tensor(9.0))
    assert copy.deepcopy(s)[-1] == 9.0 and pickle.loads(pickle.dumps(s))[-1] == 9.0 and type(pickle.loads(pickle.dumps(s))) is list
