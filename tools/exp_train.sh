#!/bin/bash
# A/B of the training kernels against experiment builds (built HERE before the gpurun call, they travel with the snapshot):
#   AON_BUILD_TAG=nostore AON_EXTRA_FLAGS="-DAON_EXPERIMENT_BUILD -DAON_EXP_NOSTORE" python articulated-object-nerf_amd/build.py
#   gpurun -- 'bash tools/exp_train.sh nostore nomask'
cd ${GRAFT_REPO_ROOT:-.}
ONLY=${ONLY:-art_fwd_train,art_bwd_chain,fwd_train,bwd_chain}
python tools/kernel_bench.py --only $ONLY $KB_ARGS 2>&1 | grep '^{' | cut -c1-230
for t in "$@"; do
  AON_HIP_LIB=articulated-object-nerf_amd/libaon_hip_$t.so python tools/kernel_bench.py --only $ONLY --tag $t $KB_ARGS 2>&1 | grep '^{' | cut -c1-230
done
