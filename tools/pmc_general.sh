# SQ counters of the layer-wise engine's kernels (gemm_tn_kernel, wgrad_nk_kernel) on a NeRFMLP.forward + training step:
#   gpurun --timeout 900 -- 'bash tools/pmc_general.sh'
cd /tmp && export TMPDIR=/tmp
cat > /tmp/gen_run.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import aon_amd.synthetic as syn
from aon_amd.models.vanilla_nerf.model import NeRF
dev = torch.device("cuda:0")
n = 4096
frame = syn.make_rays(64, n // 64, syn.look_at_pose(), syn.focal_from_fovy(64))
rays = {k: v[:n].to(dev) for k, v in frame.items()}
target = torch.rand(n, 3, device=dev)
gk = dict(min_deg_point=0, max_deg_point=6, deg_view=2)
model = NeRF(**gk).to(dev)
model._fused_inference = False   # (0, 6, 2) fits the fused kernels' slots: this script is about the layer-wise engine
model.load_state_dict(syn.make_general_nerf_state_dict(7, **gk))
out = model(rays, True, True, 2.0, 6.0)
(((out[0][0] - target) ** 2).mean() + ((out[1][0] - target) ** 2).mean()).backward()
torch.cuda.synchronize()
PY
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS" "SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d /tmp/pg/pmc_$i -o x -- python /tmp/gen_run.py > /tmp/pg_$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import glob, sqlite3
for db in sorted(glob.glob('/tmp/pg/pmc_*/x_results.db')):
    cur = sqlite3.connect(db).cursor()
    for k, c, n, s, a, d in cur.execute("select kernel_name, counter_name, count(*), sum(value), avg(value), avg(duration) from counters_collection where kernel_name like '%gemm_tn%' or kernel_name like '%wgrad_nk%' group by kernel_name, counter_name"):
        print(f"{k[:24]:<24} {c:<28} n={n:<4} avg/dispatch={a:.6g} avg_ns={d:.0f}")
PY
