"""GPU: the launch path the driver uses for N > 1 -- `python -m torch.distributed.run ... bench.py` -- exercised at
--nproc-per-node 1 (one-GPU box): rank/env plumbing, RCCL initialisation, the weak-scaling all-gather inside the timed step and
BASELINE config 3's sharded-frame leg all run through the real collectives; the line it prints must agree with a plain
`python bench.py` run of the same flags.  No scaling number comes out of this: it pins the code path."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-alt", "--no-train-leg", "--no-extra-legs"]


def _run(cmd):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_under_torch_distributed_run_matches_the_plain_run():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    plain = _run([sys.executable, "bench.py", *FLAGS])
    launched = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                     "--master-port", str(port), "bench.py", *FLAGS])
    for res in (plain, launched):
        assert res["n_gpus"] == 1 and res["steps"] == 2 and res["scaling"] == "weak" and res["unit"] == "rays/s"
        sh = res["sharded_frame"]
        assert "error" not in sh, sh
        assert sh["frame_rows"] == 307200 and sh["world"] == 1 and sh["rays_per_s"] > 0
        assert res["roofline"]["frac"] > 0.5
    assert launched["config"]["exchange"] == "none"
    # same kernels, same work: the launched form (RCCL all-gather of 6 MB inside every step) within 3 % of the plain one
    assert abs(launched["value"] / plain["value"] - 1.0) < 0.03, (launched["value"], plain["value"])
    assert abs(launched["sharded_frame"]["rays_per_s"] / plain["sharded_frame"]["rays_per_s"] - 1.0) < 0.05
