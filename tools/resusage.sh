#!/bin/bash
# Register / scratch / occupancy table of the kernels of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage):
#   tools/resusage.sh aon_mlp.hip [extra flags]      -> one line per kernel
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
f=$1; shift
extra=""
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC $extra "$@" -c $ROOT/articulated-object-nerf_amd/csrc/$f -o /tmp/resusage_$$.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size" \
  | sed 's/.*remark: [^ ]* //; s/\[-Rpass.*//; s/Function Name: //' | paste - - - - - - | sed 's/  */ /g' | sort -u
rm -f /tmp/resusage_$$.o
