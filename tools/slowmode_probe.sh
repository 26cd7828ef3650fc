#!/bin/bash
# Round 6: some boxes of the pool run the config-5 step in a SLOW MODE (33.6-36.5 ms instead of 30.4) for whole processes at a time.  This probe
# runs the default step twice; on a box that shows the slow mode it runs a matrix of switches to find what it depends on.
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
