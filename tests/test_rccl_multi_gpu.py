"""GPU: RCCL with N > 1 ranks, the moment a box has them.  The one-GPU lease this repository is developed on cannot
run it -- there the multi-rank test SKIPS and the same worker runs at world size 1 (real collectives, `force=True`) so that its code
path is known to work; on the first multi-GPU box `test_rccl_with_every_gpu_of_the_box` spawns one process per GPU under
`python -m torch.distributed.run` and is the proof that RCCL saw N ranks: sharded frame bit-equal to rank 0's unsharded render,
gradient exchange equal to the mean of the per-rank gradients, DDP's contract violation raised on every rank (tests/_rccl_worker.py).
Reference: run.py:151 (DDPPlugin), models/interface.py:31-51 (all_gather of rendered pixels)."""
import json
import os
import socket
import subprocess
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(nproc: int):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "rccl.json")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "tests", "_rccl_worker.py"), out]
        p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
        return json.loads(open(out).read())


def test_rccl_worker_at_world_size_one():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    res = _launch(1)
    print(res)
    assert res["ok"] and res["world"] == 1 and res["ranks_seen"] == 1 and res["sharded_frame_bit_equal"] and res["grad_exchange_rel_err"] <= 1e-6
    assert res["arena_in_place"] and res["arena_same_parameters_after_adam"] and res["arena_exchange_rel_err"] <= 1e-6


@pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs >= 2 GPUs on the box (one-GPU lease: skipped)")
def test_rccl_with_every_gpu_of_the_box():
    n = min(torch.cuda.device_count(), 8)
    res = _launch(n)
    print(res)
    assert res["ok"] and res["world"] == n and res["ranks_seen"] == n
    assert res["sharded_frame_bit_equal"] and res["grad_exchange_rel_err"] <= 1e-6 and res["uneven_raised_everywhere"]
    assert res["arena_in_place"] and res["arena_same_parameters_after_adam"] and res["arena_exchange_rel_err"] <= 1e-6
