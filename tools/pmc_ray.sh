#!/bin/bash
# SQ counters of one per-ray kernel (gpurun: bash tools/pmc_ray.sh "<label substring>" <rays>); one rocprofv3 pass per counter group.
ONLY=${1:-"composite S=65 (no"}
RAYS=${2:-307200}
cd /tmp && export TMPDIR=/tmp
i=0
rm -rf /tmp/pr
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_SALU SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/pr/pmc_$i -o x -- python $GRAFT_REPO_ROOT/tools/ray_kernel_bench.py --rays $RAYS --reps 3 --only "$ONLY" > /tmp/pr_$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import glob, sqlite3
for db in sorted(glob.glob('/tmp/pr/pmc_*/x_results.db')):
    cur = sqlite3.connect(db).cursor()
    for k, c, n, s, a, mn in cur.execute("select kernel_name, counter_name, count(*), sum(value), avg(value), min(value) from counters_collection where kernel_name like '%composite%' or kernel_name like '%sample_pdf%' group by kernel_name, counter_name"):
        print(f"{k[:52]:<52} {c:<24} n={n:<3} avg={a:.6g} min={mn:.6g}")
PY
