#!/bin/bash
# HBM traffic counters of one per-ray kernel (gpurun: bash tools/pmc_fetch.sh "<label substring>" <rays>)
ONLY=${1:-"composite S=65 (no"}
RAYS=${2:-307200}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/pf/pmc_$i -o x -- python $GRAFT_REPO_ROOT/tools/ray_kernel_bench.py --rays $RAYS --reps 3 --only "$ONLY" > /tmp/pf_$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import glob, sqlite3
for db in sorted(glob.glob('/tmp/pf/pmc_*/x_results.db')):
    cur = sqlite3.connect(db).cursor()
    for k, c, n, a, mn in cur.execute("select kernel_name, counter_name, count(*), avg(value), min(value) from counters_collection where kernel_name like '%composite%' or kernel_name like '%sample_pdf%' group by kernel_name, counter_name"):
        print(f"{k[:52]:<52} {c:<24} n={n:<3} avg={a:.6g} min={mn:.6g}")
PY
