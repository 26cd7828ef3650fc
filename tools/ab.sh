#!/bin/bash
# Alternating A/B of the config-5 training step (or any tools/train_bench.py invocation) under two environments, N repetitions each, with the
# mean / min / max per variant -- the protocol round 6 settled on after three-run A/Bs had "measured" gains that were not there
# (profiles/LAB_NOTEBOOK.md, round 6): alternate the variants inside ONE gpurun call (boxes differ by up to 1 %, and by 14 % when one holds a lower
# clock), at least 6-8 repetitions, 30 steps per run, and look at the SPREAD as well as the mean (a one-off host stall shows as a single 33-40 ms run).
#   gpurun -- 'bash tools/ab.sh 8 "X=1" "AON_ART_AUX_HEADS=0" -- --articulated --rays 4096 --steps 30'
N=$1; A=$2; B=$3; shift 3; [ "$1" = "--" ] && shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
run() { env $1 python $REPO/tools/train_bench.py "${@:2}" 2>/dev/null | grep "^{" | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))"; }
ra=(); rb=()
for i in $(seq $N); do ra+=($(run "$A" "$@")); rb+=($(run "$B" "$@")); done
python - "$A" "${ra[*]}" "$B" "${rb[*]}" <<'PY'
import sys
for name, vals in ((sys.argv[1], sys.argv[2]), (sys.argv[3], sys.argv[4])):
    v = sorted(float(x) for x in vals.split())
    print(f"[{name}] n={len(v)} mean {sum(v)/len(v):.3f} min {v[0]:.3f} max {v[-1]:.3f} : {' '.join(f'{x:.3f}' for x in v)}")
PY
