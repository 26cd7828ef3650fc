"""Per-parameter report of chosen seeds of tests/test_hip_fuzz.py::test_training_sweep_constructor_arguments (GPU):
    python tests/diag/diag_fuzz_seeds.py 18 21 37
prints, per seed, the parameters whose distance to the fp64 truth exceeds 2 x the reference-fp32's, largest ratio first."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _gradcheck  # noqa: E402
import test_hip_fuzz as tf  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for seed in [int(a) for a in sys.argv[1:]]:
        rows = []

        def spy(hip, truth, ref32, what, factor=5.0, floor=1e-4, small_floor=2e-5):
            print(what)
            for name, gh in hip.items():
                e_hip, e_ref = _gradcheck.rel_l2(gh, truth[name]), _gradcheck.rel_l2(ref32[name], truth[name])
                rows.append((e_hip / max(e_ref, 1e-30), e_hip, e_ref, name, gh.numel(), truth[name].double().norm().item()))

        _gradcheck.assert_as_close_as_fp32, keep = spy, _gradcheck.assert_as_close_as_fp32
        try:
            tf.test_training_sweep_constructor_arguments(dev, seed)
        finally:
            _gradcheck.assert_as_close_as_fp32 = keep
        rows.sort(reverse=True)
        for ratio, e_hip, e_ref, name, numel, norm in rows[:6]:
            print(f"  seed {seed}: {ratio:7.1f}x  hip {e_hip:.2e}  ref32 {e_ref:.2e}  {name} ({numel} el, |truth| {norm:.2e})")


if __name__ == "__main__":
    main()
