#!/bin/bash
# Experiment build of ONE translation unit (the other objects are the product's):
#   tools/exp_tu.sh <tag> <file.hip> <flags...>   ->  articulated-object-nerf_amd/libaon_hip_<tag>.so   (select with AON_HIP_LIB)
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd); P=$ROOT/articulated-object-nerf_amd
tag=$1; tu=$2; shift 2
mkdir -p $P/build_$tag
/opt/rocm/bin/hipcc "$@" --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -Wno-unused-lambda-capture -c $P/csrc/$tu -o $P/build_$tag/${tu%.hip}.o
objs=""; for f in aon_mlp aon_mlp_art aon_train aon_train_art aon_render aon_gmlp aon_fold aon_capi; do
  if [ "$f.hip" = "$tu" ]; then objs="$objs $P/build_$tag/$f.o"; else objs="$objs $P/build/$f.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libaon_hip_$tag.so $objs
echo $P/libaon_hip_$tag.so
