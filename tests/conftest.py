import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    out = {}
    for k in z.files:
        a = z[k]
        out[k] = torch.from_numpy(a) if a.ndim > 0 else a.item()
    return out


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def nerf_sd():
    import aon_amd.synthetic as syn

    return syn.make_nerf_state_dict(seed=0, density_scale=30.0)


@pytest.fixture(params=["folded", "literal"])
def fold_form(request):
    """Round 5: both forms of the fused kernels' streams -- bottleneck_layer folded into views_linear[0] (the default) and the literal
    two layers (aon_set_bottleneck_fold(0)).  Tests that take this fixture run once per form; the switch is restored afterwards."""
    from aon_amd import ops

    before = ops.bottleneck_fold()
    ops.set_bottleneck_fold(request.param == "folded")
    yield request.param
    ops.set_bottleneck_fold(before)
