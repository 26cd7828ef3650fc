"""Gradient parity yardstick shared by the GPU training tests (see tests/test_hip_smooth.py): truth = the oracle's autograd in
fp64; yardstick = the oracle's own fp32 autograd (the reference's arithmetic) against that truth.  The HIP gradients must be as
close to the truth as the reference's fp32 is, up to a factor 5 (floor 1e-4 relative L2).

Parameters of at most four elements (the density / rgb / deformation head biases: signed sums over EVERY sample of a level) get the
lower floor `small_floor` = 2e-5: since round 4 their sums are accumulated in fp64 on the device (csrc/aon_wgrad.h head_wgrad_kernel,
csrc/aon_gmlp.hip colsum_kernel), so they carry no summation error of their own and need no allowance beyond the yardstick."""


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def assert_as_close_as_fp32(hip: dict, truth: dict, ref32: dict, what: str, factor: float = 5.0, floor: float = 1e-4, small_floor: float = 2e-5):
    bad, worst, worst_ratio = {}, (0.0, 0.0, ""), (0.0, 0.0, 0.0, "")
    for name, gh in hip.items():
        e_hip, e_ref = rel_l2(gh, truth[name]), rel_l2(ref32[name], truth[name])
        fl = small_floor if gh.numel() <= 4 else floor
        worst = max(worst, (e_hip, e_ref, name))
        if e_hip > fl:   # the ratio only means something above the floor
            worst_ratio = max(worst_ratio, (e_hip / max(e_ref, 1e-30), e_hip, e_ref, name))
        if e_hip > max(fl, factor * e_ref):
            bad[name] = f"hip {e_hip:.1e} vs reference-fp32 {e_ref:.1e}"
    print(f"{what}: worst gradient distance to the fp64 truth: hip {worst[0]:.2e} on {worst[2]} (reference fp32 there: {worst[1]:.2e}); "
          f"worst ratio above the floor: {worst_ratio[0]:.1f}x on {worst_ratio[3]} ({worst_ratio[1]:.1e} vs {worst_ratio[2]:.1e})")
    assert not bad, bad


def assert_as_close_as_fp32_fixture(hip: dict, g: dict, what: str, factor: float = 5.0, floor: float = 1e-4, small_floor: float = 2e-5):
    """The same yardstick with truth and yardstick read from a fixture produced by the REAL reference (tests/golden/make_golden_full.py,
    G21): `name|truth` = the reference's fp64 autograd (rounded to fp32 for storage: 6e-8 relative), `name|norm` = its L2 norm,
    `name|ref32_dist` = ||reference fp32 autograd - truth|| over the whole tensor.  Tensors above 65,536 elements are held by a fixed
    strided sample (`name|truth_sel`, step `name|sel_step`) with the sample's own norm and reference-fp32 distance, plus the full-tensor
    norm of the HIP gradient against the truth's (a scale or a column-block error outside the sample moves it)."""
    import torch

    bad, worst, worst_ratio = {}, (0.0, 0.0, ""), (0.0, 0.0, 0.0, "")
    names = sorted(k[: -len("|norm")] for k in g if k.endswith("|norm"))
    assert set(names) == set(hip), (set(names) ^ set(hip))
    for name in names:
        gh = hip[name].double().reshape(-1)
        nrm = max(float(g[f"{name}|norm"]), 1e-30)
        e_ref_full = float(g[f"{name}|ref32_dist"]) / nrm
        if f"{name}|truth" in g:
            truth = g[f"{name}|truth"].double().reshape(-1)
            assert truth.numel() == gh.numel(), name
            e_hip, e_ref = (gh - truth).norm().item() / nrm, e_ref_full
        else:
            assert list(hip[name].shape) == g[f"{name}|shape"].tolist(), name
            sel = torch.arange(g[f"{name}|truth_sel"].numel()) * int(g[f"{name}|sel_step"])
            nsel = max(float(g[f"{name}|norm_sel"]), 1e-30)
            e_hip = (gh[sel] - g[f"{name}|truth_sel"].double()).norm().item() / nsel
            e_ref = float(g[f"{name}|ref32_dist_sel"]) / nsel
            fl_n = max(floor, factor * e_ref_full)
            assert abs(gh.norm().item() - nrm) <= fl_n * nrm, (name, "full-tensor norm", gh.norm().item(), nrm)
        fl = small_floor if gh.numel() <= 4 else floor
        worst = max(worst, (e_hip, e_ref, name))
        if e_hip > fl:
            worst_ratio = max(worst_ratio, (e_hip / max(e_ref, 1e-30), e_hip, e_ref, name))
        if e_hip > max(fl, factor * e_ref):
            bad[name] = f"hip {e_hip:.1e} vs reference-fp32 {e_ref:.1e}"
    print(f"{what}: worst gradient distance to the reference's fp64 truth: hip {worst[0]:.2e} on {worst[2]} (reference fp32 there: {worst[1]:.2e}); "
          f"worst ratio above the floor: {worst_ratio[0]:.1f}x on {worst_ratio[3]} ({worst_ratio[1]:.1e} vs {worst_ratio[2]:.1e})")
    assert not bad, bad
