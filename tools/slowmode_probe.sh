#!/bin/bash
# Round 6: on some boxes of the pool whole processes ran the config-5 step at 33.6-42 ms instead of 30.4.  This probe runs the step four
# times; if a run is slow it prints, per process, the longest HOST step and the allocator's activity inside the timed loop -- which is
# what showed the cause (one hipMalloc of the 11-15 GB training buffers inside a step; fixed by ops._TRAIN_POOL: profiles/r06_slowmode.txt).
# Kept as the regression probe: with the pool no run should be slow and "device_mallocs_in_timed_loop" should only count 20 MB segments.
#   gpurun -- 'bash tools/slowmode_probe.sh'
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
run() { env "$@" python $REPO/tools/train_bench.py --articulated --rays 4096 --steps 30 $FL 2>/dev/null | grep "^{" | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))"; }
FL=""
a=$(run AON_SIDE_PRIORITY=0); b=$(run AON_SIDE_PRIORITY=0); c=$(FL="--torch-adam" run AON_SIDE_PRIORITY=0); d=$(FL="--torch-adam" run AON_SIDE_PRIORITY=0)
echo "probe (side streams at default priority): default $a $b  torch-adam $c $d"
slow=$(python -c "print(int(max($a,$b,$c,$d) > 31.5))")
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | head -6
if [ "$slow" = "0" ]; then echo "box is in the fast mode"; exit 0; fi
echo "SLOW MODE box: matrix"
for i in 1 2 3 4 5 6 7 8; do
  AON_SIDE_PRIORITY=0 python $REPO/tools/train_bench.py --articulated --rays 4096 --steps 30 $( [ $((i % 2)) = 0 ] && echo --torch-adam ) 2>/dev/null | grep "^{" | python -c "
import sys, json
r = json.loads(sys.stdin.read()); d = r['step_ms_device']; print(r['optimizer'], round(r['ms_per_step'], 2), 'enqueue', round(r['host_enqueue_ms_per_step'], 2), 'host max %.1f ms at step %d' % (d['host_ms_max'], d['host_argmax']), r['allocator'])"
done
