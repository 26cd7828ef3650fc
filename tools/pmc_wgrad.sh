#!/bin/bash
# PMC counters of the training step's kernels (separate passes; see MI355X_MICROARCH.md): gpurun -- 'bash tools/pmc_wgrad.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/pmc_wg; SUM=$REPO/gpurun_out/pmc_wg; mkdir -p $OUT $SUM
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/train_bench.py --rays 4096 --steps 2 --articulated --no-overlap --late-heads"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o t -- $CMD > $OUT/sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM -d $OUT/pmc_lds -o t -- $CMD > $OUT/lds.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o t -- $CMD > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o t -- $CMD > $OUT/write.log 2>&1
cd $REPO
python tools/summarize_rocprof.py $OUT $SUM/wg > $SUM/summary.log 2>&1
ls $SUM; tail -5 $SUM/summary.log
