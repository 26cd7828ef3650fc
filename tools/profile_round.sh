#!/bin/bash
# Round profile recipe (run on the GPU box through gpurun; outputs under gpurun_out/prof_<tag>/, summarised into profiles/
# by tools/summarize_rocprof.py + tools/kstats.py; raw databases stay in /tmp on the box):
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh r01'
# Kernel trace and PMC counters are collected in SEPARATE runs (never --pmc together with a trace domain other than
# --kernel-trace); FETCH_SIZE and WRITE_SIZE need a pass each (TCC counter budget, MI355X_MICROARCH.md HBM section).
set -u
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/prof_$TAG          # raw rocpd databases stay on the box (tens of MB); only the summaries travel back
SUM=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT $SUM
cd /tmp && export TMPDIR=/tmp
run() { timeout 600 "$@"; }
# 1. headline render bench: per-kernel durations
run rocprofv3 --kernel-trace --stats -d $OUT/stats -o $TAG -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-leg --no-extra-legs --no-alt > $OUT/stats_run.log 2>&1
# 2. counters, one pass per group
run rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o $TAG -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-train-leg --no-extra-legs --no-alt > $OUT/pmc_fetch.log 2>&1
run rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o $TAG -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-train-leg --no-extra-legs --no-alt > $OUT/pmc_write.log 2>&1
run rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d $OUT/pmc_sq -o $TAG -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-train-leg --no-extra-legs --no-alt > $OUT/pmc_sq.log 2>&1
run rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq2 -o $TAG -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-train-leg --no-extra-legs --no-alt > $OUT/pmc_sq2.log 2>&1
# 3. training steps (BASELINE config 5 per GPU, and the reference's vanilla batch).  Kernel statistics with the two levels'
#    backward SERIALISED on one stream (--no-overlap): on the product's two library streams kernels of the two levels share the
#    CUs and their individual durations stretch (likewise --late-heads: the early head reductions wait for compute units beside the chain and
#    would show the chain's duration as their own); the product's step time comes from the un-profiled runs below.
run rocprofv3 --kernel-trace --stats -d $OUT/train_art -o $TAG -- python $REPO/tools/train_bench.py --rays 4096 --steps 10 --articulated --no-overlap --late-heads > $OUT/train_art_serial.log 2>&1
run rocprofv3 --kernel-trace --stats -d $OUT/train_van -o $TAG -- python $REPO/tools/train_bench.py --rays 4096 --steps 10 --no-overlap --late-heads > $OUT/train_van_serial.log 2>&1
run python $REPO/tools/train_bench.py --rays 4096 --steps 20 --articulated > $OUT/train_art.log 2>&1
run python $REPO/tools/train_bench.py --rays 4096 --steps 20 > $OUT/train_van.log 2>&1
# 4. articulated render (BASELINE config 4)
run rocprofv3 --kernel-trace --stats -d $OUT/render_art -o $TAG -- python $REPO/tools/render_bench.py > $OUT/render_art.log 2>&1
# 5. per-ray kernels one by one at the round-2 chunk size and at a whole frame (HIP events, no profiler)
run python $REPO/tools/ray_kernel_bench.py > $OUT/ray_kernels.log 2>&1
grep -h '^{' $OUT/ray_kernels.log > $SUM/${TAG}_ray_kernels.jsonl
cd $REPO
python tools/summarize_rocprof.py $OUT $SUM/${TAG} > $SUM/summary.log 2>&1
f=$(ls $OUT/stats/*_results.db 2>/dev/null | head -1)
[ -n "$f" ] && python tools/roofline_table.py $f > $SUM/${TAG}_roofline_table.txt 2>&1
for d in train_art train_van render_art; do
  f=$(ls $OUT/$d/*_results.db 2>/dev/null | head -1)
  [ -n "$f" ] && python tools/kstats.py $f 16 > $SUM/${TAG}_${d}_kernel_stats.txt 2>&1
done
for l in stats_run pmc_fetch pmc_write pmc_sq pmc_sq2 train_art train_van train_art_serial train_van_serial render_art; do grep -h '^{' $OUT/$l.log | tail -1 > $SUM/$l.json; done
cat $SUM/train_art.json $SUM/train_van.json $SUM/train_art_serial.json $SUM/train_van_serial.json $SUM/render_art.json; cut -c1-300 $SUM/stats_run.json
