cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-alt --no-train-leg"
i=0
for set in "SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" "SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/pp/pmc_$i -o x -- $B > /tmp/pp_$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import glob, sqlite3
for db in sorted(glob.glob('/tmp/pp/pmc_*/x_results.db')):
    cur = sqlite3.connect(db).cursor()
    for k, c, n, s, a in cur.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection where kernel_name like '%mlp_fwd_kernel%' group by kernel_name, counter_name"):
        print(f"{c:<32} n={n:<3} avg/dispatch={a:.6g}")
PY
