#!/bin/bash
# gpurun -- 'bash tools/ubench/run_invcdf_floor.sh'   -> gpurun_out/invcdf_floor.json
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I articulated-object-nerf_amd/csrc tools/ubench/invcdf_floor.hip -o /tmp/invcdf_floor && /tmp/invcdf_floor | tee gpurun_out/invcdf_floor.json
