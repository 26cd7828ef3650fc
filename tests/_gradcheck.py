"""Gradient parity yardstick shared by the GPU training tests (see tests/test_hip_smooth.py): truth = the oracle's autograd in
fp64; yardstick = the oracle's own fp32 autograd (the reference's arithmetic) against that truth.  The HIP gradients must be as
close to the truth as the reference's fp32 is, up to a factor 5 (floor 1e-4 relative L2)."""


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def assert_as_close_as_fp32(hip: dict, truth: dict, ref32: dict, what: str, factor: float = 5.0, floor: float = 1e-4):
    bad, worst = {}, (0.0, 0.0, "")
    for name, gh in hip.items():
        e_hip, e_ref = rel_l2(gh, truth[name]), rel_l2(ref32[name], truth[name])
        worst = max(worst, (e_hip, e_ref, name))
        if e_hip > max(floor, factor * e_ref):
            bad[name] = f"hip {e_hip:.1e} vs reference-fp32 {e_ref:.1e}"
    print(f"{what}: worst gradient distance to the fp64 truth: hip {worst[0]:.2e} on {worst[2]} (reference fp32 there: {worst[1]:.2e})")
    assert not bad, bad
