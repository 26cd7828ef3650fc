"""Build libaon_hip.so (gfx950) in-tree with hipcc.  No torch extension machinery: the library is a plain
C-ABI shared object (include/aon_hip.h) loaded with ctypes.

    python articulated-object-nerf_amd/build.py [--force]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
# experiments: AON_BUILD_TAG=x builds build_x/ -> libaon_hip_x.so next to the product library (select it with AON_HIP_LIB)
_TAG = os.environ.get("AON_BUILD_TAG", "")
OBJ = os.path.join(PKG, "build" + ("_" + _TAG if _TAG else ""))
LIB = os.path.join(PKG, "libaon_hip" + ("_" + _TAG if _TAG else "") + ".so")
SOURCES = ["aon_mlp.hip", "aon_mlp_art.hip", "aon_train.hip", "aon_train_art.hip", "aon_render.hip", "aon_gmlp.hip", "aon_fold.hip", "aon_optim.hip", "aon_capi.hip"]
HEADERS = [os.path.join(CSRC, "aon_common.h"), os.path.join(CSRC, "aon_mlp_core.h"), os.path.join(CSRC, "aon_wgrad.h"), os.path.join(CSRC, "aon_art_common.h"), os.path.join(CSRC, "aon_ray_core.h"), os.path.join(CSRC, "aon_gmlp.h"), os.path.join(CSRC, "aon_fold.h"),
           os.path.join(os.path.dirname(PKG), "include", "aon_hip.h")]
# -ffp-contract=off: the stage kernels reproduce the reference's un-fused mul/add sequences; FMAs are explicit.
# Per-file code-generation choices, A/B-measured on MI355X (round 1, tools/ab_train.sh, 4096-ray vanilla training step):
# pinning the A-fragment read of step i+1 above the MFMAs of step i (AON_PIN_PREFETCH, aon_mlp_core.h) helps the vanilla
# backward chain (5.62 -> 4.95 ms per level pair) but costs the forward kernels 2.7 % (144.0 -> 140.2 TFLOP/s) and does
# nothing for the articulated chains, so it is applied to aon_train.hip only.  (-mllvm -amdgpu-mfma-vgpr-form on the same
# file: backward chain 5.26 ms but the weight-gradient kernel 0.525 -> 0.643 ms -- a net loss.)
# Round 4: with two segments per chain launch the pinned form of the vanilla chain spills 3.5 KB per lane (the sched_barrier per step
# leaves the allocator no room for the segment bookkeeping); unpinned it is at 0 scratch, so the flag is off for every file.
PER_FILE_FLAGS = {}
if "AON_PER_FILE_FLAGS" in os.environ:   # experiments: JSON {"file.hip": ["flag", ...]} replaces the table
    import json as _json
    PER_FILE_FLAGS = _json.loads(os.environ["AON_PER_FILE_FLAGS"])
FLAGS = (os.environ.get("AON_EXTRA_FLAGS", "").split()) + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-lambda-capture"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        cmd = [cc, *FLAGS, *PER_FILE_FLAGS.get(src, []), "-c", s, "-o", o]
        # an object is also stale when it was built with other flags (AON_EXTRA_FLAGS experiments must not reuse objects)
        stamp = o + ".cmd"
        same_cmd = os.path.exists(stamp) and open(stamp).read() == " ".join(cmd)
        if force or not same_cmd or _stale(o, [s] + HEADERS):
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, timeout=1800)
        with open(cmd[-1] + ".cmd", "w") as f:
            f.write(" ".join(cmd))

    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
