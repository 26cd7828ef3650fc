#!/bin/bash
# Experiment builds of the vanilla forward translation unit only (the other objects are the product's):
#   tools/exp_mlp.sh <tag> <flags...>   ->  articulated-object-nerf_amd/libaon_hip_<tag>.so   (select with AON_HIP_LIB)
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd); P=$ROOT/articulated-object-nerf_amd
tag=$1; shift
mkdir -p $P/build_$tag
/opt/rocm/bin/hipcc "$@" --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -Wno-unused-lambda-capture -c $P/csrc/aon_mlp.hip -o $P/build_$tag/aon_mlp.o
objs=""; for f in aon_mlp_art aon_train aon_train_art aon_render aon_gmlp aon_fold aon_capi; do objs="$objs $P/build/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libaon_hip_$tag.so $P/build_$tag/aon_mlp.o $objs
echo $P/libaon_hip_$tag.so
